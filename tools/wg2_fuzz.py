#!/usr/bin/env python3
"""Developer aid: random node shapes of the 65 .. 128 class (max(n, m) in 65 .. 128, the other side anything from 1), mixed bound
kinds, a few equality rows (-> the general kernel): the fused kernel of the class against the route of the large nodes
(QPN_OPT_MID_ROUTE = 0) on the same records -- statuses and active-set masks equal, primals within 1e-9 relative, residuals <= 1e-8.
Usage: python tools/wg2_fuzz.py [trials]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(4242)
worst = 0.0; declined = 0; items = 0
for t in range(trials):
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
    abi = [colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u]
    w = rng.standard_normal(p)
    new = eng.solve_nodes(*abi, w)
    eng.set_option(OPT_MID_ROUTE, 0)
    try:
        old = eng.solve_nodes(*abi, w)
    finally:
        eng.set_option(OPT_MID_ROUTE, 1)
    assert np.array_equal(new["status"], old["status"]), (t, n, m, p, new["status"], old["status"])
    ok = new["status"] == 1
    assert np.array_equal(new["active"][ok], old["active"][ok]), (t, n, m, p)
    if ok.any():
        d = np.max(np.abs(new["z"][ok] - old["z"][ok])) / max(1.0, np.max(np.abs(old["z"][ok])))
        worst = max(worst, d)
        assert d <= 1e-9, (t, n, m, p, d)
        assert np.max(new["resid"][ok]) <= 1e-8
    declined += int(eq.any(axis=1).sum()); items += cnt
print(f"{trials} random shapes ({items} nodes): statuses and masks equal, worst relative primal difference {worst:.2e}; "
      f"{declined} nodes with an equality row went through the general kernel")
