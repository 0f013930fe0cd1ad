#!/usr/bin/env python3
"""Developer aid: random node shapes of the mid-size class (n, m <= 64, one of them > 32), mixed bound kinds, the
fused one-workgroup-per-node kernel against round 2's three-kernel route (QPN_OPT_MID_ROUTE = 2) on the same records:
statuses and active-set masks equal, primals within 1e-9 relative.  Usage: python tools/mid_fuzz.py [trials]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
worst = 0.0; declined = 0
for t in range(trials):
    n = int(rng.integers(1, 65)); m = int(rng.integers(1, 65))
    if max(n, m) <= 32:
        if rng.random() < 0.5: n = int(rng.integers(33, 65))
        else: m = int(rng.integers(33, 65))
    p = int(rng.integers(0, 9)); cnt = int(rng.integers(1, 40))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(10_000 + t, cnt, n, m, max(p, 1))
    if p == 0:
        Rm = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        Rm = Rm[:, :, :p]; B = rng.standard_normal((cnt, m, p)) * 0.1
    kind = rng.integers(0, 6, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    l = np.where(kind == 3, -np.inf, l); u = np.where(kind == 3, np.inf, u)
    eq = (kind == 4) & (rng.random(l.shape) < 0.02)
    u = np.where(eq, l, u)
    abi = [colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u]
    w = rng.standard_normal(p)
    new = eng.solve_nodes(*abi, w)
    eng.set_option(OPT_MID_ROUTE, 2)
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
    declined += int(eq.any(axis=1).sum())
print(f"{trials} random shapes: statuses and masks equal, worst relative primal difference {worst:.2e}; {declined} nodes with an equality row went through the general kernel")
