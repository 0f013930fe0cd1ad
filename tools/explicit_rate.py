#!/usr/bin/env python3
"""Developer probe: rate of qpn_solve_avi_batch on explicit node-shaped items (M assembled once, resident): the route
multi-node pools in reduced form and the Julia solve_avi_batch binding take.  CNT items of N = n + m."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for shape in (sys.argv[1:] or ["32x32"]):
    n, m = (int(v) for v in shape.split("x"))
    cnt = int(os.environ.get("CNT", "10000"))
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
    w = synthetic.shared_params()
    Mc, q, lo, hi, kind = eng.assemble_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w))
    for _ in range(20):
        res = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind)
    torch.cuda.synchronize()
    reps = int(os.environ.get("REPS", "200"))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        res = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    st = res["status"].cpu().numpy()
    print(f"n={n} m={m} (N={n+m}), {cnt} items: {ms*1e3:8.1f} us/call = {cnt/ms/1e3:6.2f} M items/s, solved {100*(st==1).mean():.1f} %, "
          f"mean pivots {res['pivots'].double().mean().item():.1f}", flush=True)
