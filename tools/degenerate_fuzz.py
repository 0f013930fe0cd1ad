#!/usr/bin/env python3
"""Developer aid: DEGENERATE node records -- Q = G'G of rank r < n (r = 0: an LP), so the node's AVI is monotone but not strictly
and H is singular: the fused kernels decline, the general kernels run the whole pivoting method -- against the CPU oracle.
Statuses must agree; on solved items the masks are compared bit for bit and the primals to 1e-9 (both follow the same pivoting
rule, so they should land on the same vertex), and disagreements are COUNTED, not asserted: a tie broken differently by
rounding gives another solution of the same problem (certified by the post-check either way).
Usage: python tools/degenerate_fuzz.py [trials] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
items = solved = st_diff = mask_diff = z_diff = res_big = 0
for t in range(trials):
    n = int(rng.integers(1, 41)); m = int(rng.integers(1, 61)); cnt = int(rng.integers(1, 9)); p = 4
    Q, Rm, qd, A, B, l, u = P.synth_nodes(40_000 + t, cnt, n, m, p)
    r = int(rng.integers(0, n))                                # rank of Q
    G = rng.standard_normal((cnt, r, n))
    Q = np.einsum("bki,bkj->bij", G, G) / max(r, 1)
    # bounded feasible sets so that the LP-like nodes have solutions: two-sided bounds on most rows + a box on x through A
    k = min(n, m)
    A[:, :k, :] = 0.0
    A[:, np.arange(k), np.arange(k)] = 1.0                     # the first rows bound x_i directly
    w = rng.standard_normal(p)
    M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
    rc = binding.solve_avi_batch(M, q, lo, hi, kind=kd)
    rg = eng.solve_nodes(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u, w)
    sg = np.asarray(rg["status"]); items += cnt
    st_diff += int((sg != rc["status"]).sum())
    ok = (sg == 1) & (rc["status"] == 1); solved += int(ok.sum())
    if ok.any():
        mask_diff += int((np.asarray(rg["active"])[ok] != rc["active"][ok]).any(axis=1).sum())
        d = np.max(np.abs(np.asarray(rg["z"])[ok] - rc["z"][ok]), axis=1) / np.maximum(1.0, np.max(np.abs(rc["z"][ok]), axis=1))
        z_diff += int((d > 1e-9).sum())
        rs = np.asarray(rg["resid"])[ok]
        if np.max(rs) > 1e-8:
            k = int(np.argmax(rs)); res_big += 1
            print(f"  trial {t} (n={n} m={m} rank {r}): HIP residual {rs[k]:.2e}, oracle residual {rc['resid'][ok][k]:.2e}, "
                  f"primal difference {d[k]:.2e}, max |z| {np.max(np.abs(rc['z'][ok][k])):.2e}", flush=True)
print(f"{trials} degenerate shapes, {items} items ({solved} solved by both): status differs on {st_diff}, masks on {mask_diff}, primals (> 1e-9) on {z_diff}; natural-map residual above 1e-8 on {res_big} (large iterates: the residual is absolute)")
