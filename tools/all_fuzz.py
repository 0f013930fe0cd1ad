#!/usr/bin/env python3
"""Developer aid: random node shapes across ALL size classes and aspect ratios (n, m from 1 (m from 0) to 256, tiny n with large m
and the reverse), mixed bound kinds, a few equality rows -- node records through qpn_solve_nodes (every third shape also through a resident handle) and the same items as explicit M
through qpn_solve_avi_batch, both against the CPU oracle: statuses and pivot counts equal, active-set masks bit for bit, primals
within 1e-9 relative on solved items.  Usage: python tools/all_fuzz.py [trials] [seed]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
worst = 0.0; items = 0; t0 = time.time()


BIG = os.environ.get("BIG") == "1"                 # BIG=1: sides up to 500 (N <= 1000: the ABI's limit is 1024), few items


def dim(kind):
    if BIG:
        return int({0: rng.integers(1, 65), 1: rng.integers(65, 257), 2: rng.integers(257, 501), 3: rng.integers(257, 501),
                    4: rng.integers(1, 8)}[kind])
    return int({0: rng.integers(1, 33), 1: rng.integers(33, 65), 2: rng.integers(65, 129), 3: rng.integers(129, 257),
                4: rng.integers(1, 8)}[kind])


for t in range(trials):
    n, m = dim(int(rng.integers(0, 5))), dim(int(rng.integers(0, 5)))
    if rng.random() < 0.03:
        m = 0
    p = int(rng.integers(0, 9)); cnt = int(rng.integers(1, 3 if BIG else 7))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(30_000 + t, cnt, n, max(m, 1), max(p, 1))
    if m == 0:
        A = A[:, :0, :]; B = B[:, :0, :]; l = l[:, :0]; u = u[:, :0]
    if p == 0:
        Rm = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        Rm = Rm[:, :, :p]; B = rng.standard_normal((cnt, m, p)) * 0.1
    kind = rng.integers(0, 6, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    l = np.where(kind == 3, -np.inf, l); u = np.where(kind == 3, np.inf, u)
    eq = (kind == 4) & (rng.random(l.shape) < 0.01)
    u = np.where(eq, l, u)
    w = rng.standard_normal(p)
    # OPTS=1: a pivot budget that some items exhaust (status MAX_ITERS on both sides, the same items), parameters per node
    oo = og = None
    if os.environ.get("OPTS") == "1":
        if rng.random() < 0.5 and p > 0:
            w = rng.standard_normal((cnt, p))
        if rng.random() < 0.6:
            mp = n + int(rng.integers(0, 12))
            oo = binding.default_opts(); oo.max_pivots = mp
            og = eng.default_opts(); og.max_pivots = mp
    M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
    rc = binding.solve_avi_batch(M, q, lo, hi, kind=kd, opts=oo)
    routes = [("nodes", eng.solve_nodes(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u, w, opts=og)),
              ("explicit", eng.solve_avi_batch(colmajor(M), q, lo, hi, kind=kd, opts=og))]
    if t % 3 == 0:                                  # ... and the resident-records route (second sweep: the handle knows its records)
        h = eng.upload_nodes(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u)
        h.solve(w, opts=og)
        routes.append(("handle", {k: np.array(v) for k, v in h.solve(w, opts=og).items()}))
        h.close()
    for name, r in routes:
        tag = (name, t, n, m, p, cnt)
        assert np.array_equal(np.asarray(r["status"]), rc["status"]), (tag, np.asarray(r["status"]), rc["status"])
        ok = rc["status"] == 1
        assert np.array_equal(np.asarray(r["active"])[ok], rc["active"][ok]), tag
        # (resident symmetric records of the large class run Stage B by block principal pivoting unless a pivot budget is set: the
        #  same solution, its own count of basis changes)
        bpp = name == "handle" and n > 64 and n <= 256 and m <= 256 and max(n, m) > 128 and og is None
        if name != "explicit" and not bpp:
            assert np.array_equal(np.asarray(r["pivots"])[ok], rc["pivots"][ok]), (tag, np.asarray(r["pivots"]), rc["pivots"])
        if ok.any():
            d = np.max(np.abs(np.asarray(r["z"])[ok] - rc["z"][ok])) / max(1.0, np.max(np.abs(rc["z"][ok])))
            worst = max(worst, d)
            assert d <= 1e-9, (tag, d)
    items += cnt
    if t % 20 == 19:
        print(f"  {t + 1} shapes, {items} items, worst {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"{trials} random shapes ({items} items, both routes): statuses, pivot counts and masks equal to the oracle's, worst relative primal difference {worst:.2e}")
