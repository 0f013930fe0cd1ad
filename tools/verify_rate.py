#!/usr/bin/env python3
"""Developer aid: throughput of the batched verify (A8, qpn_verify_nodes) on the bench workload, at the solved
point (all nodes optimal: least-squares path) and at a perturbed point (fallback path for some nodes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
cnt, n, m = int(os.environ.get("CNT", "10000")), 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u)]
res = eng.solve_nodes(*args, t(w))
x = res["z"][:, :n].contiguous()
for tag, xd in (("at the solution", x), ("perturbed 1e-3", x + 1e-3 * torch.randn_like(x)),
                ("perturbed 1e-2 inwards", x * (1.0 - 1e-2))):
    for _ in range(3): sol, lam, path = eng.verify_nodes(*args, xd, t(w))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): sol, lam, path = eng.verify_nodes(*args, xd, t(w))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    pth = path.cpu().numpy()
    print(f"verify {tag}: {ms:.3f} ms per {cnt} nodes = {cnt / ms / 1e3:.2f} M nodes/s; optimal {int(sol.sum())}, paths {np.bincount(pth, minlength=6).tolist()}")
