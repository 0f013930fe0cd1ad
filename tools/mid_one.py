#!/usr/bin/env python3
"""Developer aid: ten sweeps of ONE mid-size node class (NN, CNT; HANDLE=1: resident records) -- the program tools/midsize_trace.sh
runs under rocprofv3."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(0)
n = m = int(os.environ.get("NN", "48")); cnt = int(os.environ.get("CNT", "4000"))
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
Q, R_, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
args = [t(colmajor(Q)), t(colmajor(R_)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
# HANDLE=1: resident records (the handle learns that no node declines: no assembly, no general-kernel launch)
nodes = eng.upload_nodes(*args[:-1]) if os.environ.get("HANDLE", "0") == "1" else None
sweep = (lambda: nodes.solve(args[-1])) if nodes is not None else (lambda: eng.solve_nodes(*args))
for _ in range(3):
    res = sweep(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): res = sweep()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"n=m={n} ({cnt} nodes): {dt*1e3:.3f} ms/batch = {cnt/dt/1e6:.2f} M solves/s, solved {(res['status']==1).float().mean().item()*100:.0f} %, mean pivots {res['pivots'].double().mean().item():.0f}")
