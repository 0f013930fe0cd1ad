import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
from oracle import binding
eng = qpn_amd.Engine(0)
rng = np.random.default_rng(4242)
for t in range(6):
    big = int(rng.integers(65, 129)); small = int(rng.integers(1, 129))
    n, m = (big, small) if rng.random() < 0.5 else (small, big)
    p = int(rng.integers(0, 9)); cnt = int(rng.integers(1, 16))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(20_000 + t, cnt, n, m, max(p, 1))
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
abi = [colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u]
M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
rc = binding.solve_avi_batch(M, q, lo, hi, kind=kd)
new = eng.solve_nodes(*abi, w)
eng.set_option(OPT_MID_ROUTE, 0)
old = eng.solve_nodes(*abi, w)
eng.set_option(OPT_MID_ROUTE, 1)
ex = eng.solve_avi_batch(M, q, lo, hi, kind=kd)
print("n m", n, m, "eq rows:", eq.sum(axis=1))
for name, r in (("oracle", rc), ("wg2", new), ("route0", old), ("explicit", ex)):
    print(name, "status", np.asarray(r["status"]), "pivots", np.asarray(r["pivots"]), "resid", np.asarray(r["resid"]) if "resid" in r and r["resid"] is not None else None)
z0o, z0r = rc["z"][0], np.asarray(old["z"])[0]
print("item 0: |z_oracle - z_route0| max", np.max(np.abs(z0o - z0r)), " |z_oracle - z_wg2|", np.max(np.abs(z0o - np.asarray(new["z"])[0])))
print("item 0 bounds kinds:", np.unique(kind[0], return_counts=True))
