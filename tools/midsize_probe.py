#!/usr/bin/env python3
"""Developer aid: node-solve rate across node sizes (where the kernel families hand over), wall clock, two columns: the per-call
route (records passed every call, outputs allocated per call, the decline count read back every call) and the resident-records
route of the bench (qpn_nodes_upload once, then one launch per sweep in the handle's longest-first order; 48 untimed sweeps first,
so that the order has been re-sorted from the kernels' own pivot counts).  n = m = 64 twice: 2 000 nodes are 1.95 rounds of the 1 024 workgroups resident at once (the
second round nearly empty at the end), 4 096 are four full rounds."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for n, m, cnt in [(16, 16, 10000), (32, 32, 10000), (33, 33, 4000), (40, 40, 4000), (48, 48, 4000), (64, 64, 2000), (64, 64, 4096), (96, 96, 1000), (128, 128, 1000), (256, 256, 512)]:
    Q, R_, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
    args = [t(colmajor(Q)), t(colmajor(R_)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
    out = None
    for _ in range(3):
        res = eng.solve_nodes(*args)
    torch.cuda.synchronize()
    reps = 5
    dts = []
    for _ in range(3):                                  # three timed groups of five calls, the median group (a one-off stall -- a module
        t0 = time.perf_counter()                        # load, a clock ramp -- in one group does not decide the line)
        for _ in range(reps):
            res = eng.solve_nodes(*args)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) / reps)
    dt = sorted(dts)[1]
    st = res["status"].cpu().numpy()
    nodes = eng.upload_nodes(*args[:-1])
    for _ in range(48):
        rr = nodes.solve(args[-1])
    torch.cuda.synchronize()
    dth = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(4 * reps):
            rr = nodes.solve(args[-1])
        torch.cuda.synchronize()
        dth.append((time.perf_counter() - t0) / (4 * reps))
    dh = sorted(dth)[1]
    assert (rr["status"].cpu().numpy() == st).all()
    del nodes
    print(f"n=m={n:4d} ({cnt:5d} nodes): {dt*1e3:8.3f} ms/batch = {cnt/dt/1e3:10.1f} K solves/s per call; resident records: {dh*1e3:8.3f} ms = {cnt/dh/1e3:10.1f} K solves/s; solved {(st==1).mean()*100:.0f} %, mean pivots {res['pivots'].double().mean().item():.0f}", flush=True)
