#!/usr/bin/env python3
"""Developer aid: qpn_verify_nodes (row A8: verify_solution + solve_qp) on random node shapes and random points -- the node's AVI
solution, the solution shrunk or perturbed by a random amount, a point far outside -- against the oracle's verify_solution, node by
node: solution flag and path must agree; disagreements are printed with the margins involved (a point within rounding of an
accept threshold may legitimately fall either way) and counted.  Usage: python tools/verify_fuzz.py [trials] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
MAXDIM = int(os.environ.get("MAXDIM", "96"))       # MAXDIM=32: the one-wavefront class of verify_node32 only
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
nodes = flag_diff = path_diff = 0
for t in range(trials):
    n = int(rng.integers(1, MAXDIM + 1)); m = int(rng.integers(0, MAXDIM + 1)); cnt = int(rng.integers(1, 7)); p = int(rng.integers(1, 9))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(50_000 + t, cnt, n, max(m, 1), p)
    if m == 0:
        A = A[:, :0, :]; B = B[:, :0, :]; l = l[:, :0]; u = u[:, :0]
    kind = rng.integers(0, 5, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    if m and rng.integers(0, 3) == 0:
        # badly scaled rows: a solution graph's rows are normalised to a leading coefficient of 1 (src/sets.jl:76-89), which can
        # leave a row 1e7 times longer than its neighbours
        sc = 10.0 ** rng.uniform(-2, 7, size=(cnt, A.shape[1], 1)) * (rng.random((cnt, A.shape[1], 1)) < 0.3) + 1.0
        A = A * sc; B = B * sc; l = l * sc[:, :, 0]; u = u * sc[:, :, 0]
    w = rng.standard_normal(p)
    M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
    z = binding.solve_avi_batch(M, q, lo, hi, kind=kd)["z"]
    xs = z[:, :n]
    mode = int(rng.integers(0, 4))
    xd = [xs, xs * (1.0 - 10.0 ** rng.uniform(-9, -1)), xs + 10.0 ** rng.uniform(-9, -1) * rng.standard_normal(xs.shape), xs + 5.0][mode].copy()
    sol, lam, path = eng.verify_nodes(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u, xd, w)
    for i in range(cnt):
        sc, lc, pc = binding.verify_solution(Q[i], Rm[i], qd[i], A[i], B[i], l[i], u[i], xd[i], w)
        nodes += 1
        if bool(sol[i]) != sc or path[i] != pc:
            flag_diff += bool(sol[i]) != sc; path_diff += path[i] != pc
            print(f"  trial {t} node {i} (n={n} m={m} mode {mode}): HIP flag {int(sol[i])} path {path[i]}, oracle flag {int(sc)} path {pc}", flush=True)
print(f"{trials} shapes, {nodes} nodes: solution flag differs on {flag_diff}, path on {path_diff}")
