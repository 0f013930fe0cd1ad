#!/usr/bin/env python3
"""Developer aid: the large-node route with Stage B by block principal pivoting (resident symmetric records) against the CPU oracle
on random shapes of the class (64 < n <= 256, m <= 256), incl. nodes with many / few active rows and tight boxes (both bounds
finite).  Prints the worst relative |dz| and any status mismatch.  usage: python tools/bpp_fuzz.py [shapes]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
from oracle_engine import OracleEngine
shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(77)
eng = qpn_amd.Engine(0); orc = OracleEngine()
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
worst, bad = 0.0, 0
for k in range(shapes):
    n = int(rng.integers(65, 257)); m = int(rng.integers(1, 257)); cnt = int(rng.integers(2, 7))
    Q, R_, qd, A, B, l, u = synthetic.synth_nodes(9000 + 7 * k, cnt, n, m)
    mode = k % 3
    if mode == 1:                       # tight boxes around a feasible point: both bounds finite, many rows active
        x0 = rng.standard_normal((cnt, n))
        s0_ = np.einsum("bij,bj->bi", A, x0) + B @ synthetic.shared_params()
        l = s0_ - rng.uniform(0.0, 0.3, s0_.shape); u = s0_ + rng.uniform(0.0, 0.3, s0_.shape)
    elif mode == 2:                     # loose: few rows active
        l = l - 3.0; u = u + 3.0
    w = synthetic.shared_params()
    rec = [colmajor(Q), colmajor(R_), qd, colmajor(A), colmajor(B), l, u]
    ref = orc.solve_nodes(*rec, w)
    h = eng.upload_nodes(*[t(a) for a in rec])
    res = h.solve(t(w)); torch.cuda.synchronize()
    z = res["z"].cpu().numpy(); st = res["status"].cpu().numpy()
    z0 = np.asarray(ref["z"]); s0 = np.asarray(ref["status"])
    dz = np.max(np.abs(z - z0) / np.maximum(1.0, np.max(np.abs(z0), axis=1, keepdims=True)))
    ok = (st == 1) & (s0 == 1)
    dz = np.max((np.abs(z - z0) / np.maximum(1.0, np.max(np.abs(z0), axis=1, keepdims=True)))[ok], initial=0.0)
    worst = max(worst, dz); bad += int(np.any((st == 1) != (s0 == 1)))
    print(f"n={n:3d} m={m:3d} x{cnt} mode {mode}: status {st.tolist()} oracle {s0.tolist()} rel |dz| {dz:.2e} pivots {res['pivots'].cpu().numpy().tolist()}", flush=True)
    h.close()
print(f"{shapes} shapes: worst rel |dz| {worst:.2e}, status mismatches {bad}")
