#!/usr/bin/env python3
"""Developer aid: solve rate of the mid-size node classes on the resident-records route (the handle knows after its
first sweep that nothing declines: one launch per sweep on the fused route), fused kernels (QPN_OPT_MID_ROUTE = 1; 0 = the
general route).  HIP-event time over back-to-back sweeps.  Usage: python tools/mid_rate.py [sizes...]  (ROUTES=1  CNT=4000  REPS=20)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
sizes = [int(x) for x in sys.argv[1:]] or [33, 40, 48, 56, 64]
routes = [int(x) for x in os.environ.get("ROUTES", "1").split(",")]
cnt0 = int(os.environ.get("CNT", "4000")); reps = int(os.environ.get("REPS", "20"))
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for n in sizes:
    m = n
    cnt = cnt0 if n <= 64 else min(cnt0, 1024)
    Q, R_, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
    args = [t(colmajor(Q)), t(colmajor(R_)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
    for route in routes:
        eng.set_option(OPT_MID_ROUTE, route)
        nodes = eng.upload_nodes(*args[:-1])
        for _ in range(4):
            res = nodes.solve(args[-1]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): res = nodes.solve(args[-1])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ok = (res["status"] == 1).float().mean().item() * 100
        print(f"n=m={n:3d} route {route} ({cnt} nodes): {ms*1e3:8.1f} us/sweep = {cnt/ms/1e3:6.2f} M solves/s, solved {ok:.0f} %, "
              f"mean pivots {res['pivots'].double().mean().item():.1f}", flush=True)
        nodes.close()
eng.set_option(OPT_MID_ROUTE, 1)
