#!/usr/bin/env python3
"""Developer aid: the REFERENCE form of a node's GAVI -- convert(::GAVI), src/avi.jl:113-128: z = [y; lambda; s], N_ref = n + 2 m, all
STD rows, the shape PATH is handed by solve_gavi (:101-111) -- on random node shapes (N_ref up to ~600) and mixed bound kinds: the
HIP solve of the converted AVI against the oracle's, and its primal block against the reduced form's (the same x, the same lambda).
Usage: python tools/refform_fuzz.py [trials] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
INF = np.inf
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 23)
worst = 0.0; items = 0
for t in range(trials):
    n = int(rng.integers(1, 120)); m = int(rng.integers(1, 200)); cnt = int(rng.integers(1, 5))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(80_000 + t, cnt, n, m)
    kind = rng.integers(0, 5, size=l.shape)
    l = np.where(kind == 1, -INF, l); u = np.where(kind == 2, INF, u)
    l = np.where(kind == 3, -INF, l); u = np.where(kind == 3, INF, u)
    w = P.shared_params()
    M, q, lo, hi, kd = P.reduced_blocks(Q, Rm, qd, A, B, l, u, w)
    rg = eng.solve_avi_batch(colmajor(M), q, lo, hi, kind=kd)
    Ms, qs, ls, us = [], [], [], []
    for i in range(cnt):
        Mg = np.hstack([Q[i], -A[i].T]); Ag = np.hstack([A[i], np.zeros((m, m))])
        Mr, qr, lr, ur = binding.convert_gavi(Mg, q[i, :n], np.full(n, -INF), np.full(n, INF), Ag, q[i, n:], l[i], u[i])
        Ms.append(Mr); qs.append(qr); ls.append(lr); us.append(ur)
    Ms, qs, ls, us = np.stack(Ms), np.stack(qs), np.stack(ls), np.stack(us)
    rr = eng.solve_avi_batch(colmajor(Ms), qs, ls, us)
    rc = binding.solve_avi_batch(Ms, qs, ls, us)
    assert np.array_equal(np.asarray(rr["status"]), rc["status"]), (t, n, m, np.asarray(rr["status"]), rc["status"])
    ok = rc["status"] == 1
    if ok.any():
        d = np.max(np.abs(np.asarray(rr["z"])[ok] - rc["z"][ok])) / max(1.0, np.max(np.abs(rc["z"][ok]))); worst = max(worst, d)
        assert d <= 1e-9, (t, n, m, d)
        both = ok & (np.asarray(rg["status"]) == 1)
        if both.any():
            d2 = np.max(np.abs(np.asarray(rr["z"])[both][:, :n + m] - np.asarray(rg["z"])[both])) / max(1.0, np.max(np.abs(np.asarray(rg["z"])[both])))
            assert d2 <= 1e-8, (t, n, m, "reference form vs reduced form", d2)
    items += cnt
print(f"{trials} shapes, {items} items in reference form (N_ref = n + 2 m up to ~500): HIP == oracle (worst {worst:.2e}), primal and dual blocks == the reduced form's")
