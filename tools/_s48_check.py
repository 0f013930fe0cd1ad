import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
from oracle import binding
eng = qpn_amd.Engine(0)
eng.set_option(OPT_MID_ROUTE, 3)
rng = np.random.default_rng(1)
for (n, m, cnt) in [(48, 48, 40), (33, 33, 30), (40, 40, 30), (48, 20, 20), (10, 48, 20), (35, 47, 20), (1, 40, 10), (48, 1, 10), (41, 0, 5)]:
    Q, Rm, qd, A, B, l, u = P.synth_nodes(9000 + n + m, cnt, n, max(m, 1))
    if m == 0:
        A = A[:, :0, :]; B = B[:, :0, :]; l = l[:, :0]; u = u[:, :0]
    kind = rng.integers(0, 5, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    w = P.shared_params()
    M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
    rc = binding.solve_avi_batch(M, q, lo, hi, kind=kd)
    r = eng.solve_nodes(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u, w)
    st = np.asarray(r["status"]); ok = rc["status"] == 1
    same_st = np.array_equal(st, rc["status"])
    same_mask = np.array_equal(np.asarray(r["active"])[ok], rc["active"][ok])
    same_piv = np.array_equal(np.asarray(r["pivots"])[ok], rc["pivots"][ok])
    d = np.max(np.abs(np.asarray(r["z"])[ok] - rc["z"][ok])) if ok.any() else 0.0
    print(f"n={n} m={m}: status {same_st} masks {same_mask} pivots {same_piv} max|dz| {d:.2e} max resid {np.max(np.asarray(r['resid'])[ok]) if ok.any() else 0:.1e}", flush=True)
    if not same_st: print("   ", st, rc["status"])
