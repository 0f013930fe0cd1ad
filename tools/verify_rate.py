#!/usr/bin/env python3
"""Developer aid: throughput of the batched verify (A8, qpn_verify_nodes) at the solved point (all nodes optimal: least-squares
path), at a perturbed point (infeasible / fallback) and at a shrunk point (bounded least-squares fallback on every node).
Env: N, M (node shape, default 32 x 32), CNT (nodes, default 10 000), MODE (0, 1, 2: one case only -- counter passes), REPS.
The per-call figure includes the Python call (three allocations and a ctypes call: ~25 us, it bounds the figure from above at the
32-class); the kernel's own duration comes from `rocprofv3 --kernel-trace` over the same script (tools/profile_r04.sh verify)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
n, m = int(os.environ.get("N", "32")), int(os.environ.get("M", "32"))
cnt = int(os.environ.get("CNT", "10000"))
reps = int(os.environ.get("REPS", "20"))
only = os.environ.get("MODE")
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u)]
res = eng.solve_nodes(*args, t(w))
x = res["z"][:, :n].contiguous()
wd = t(w)
cases = (("at the solution", x), ("perturbed 1e-3", x + 1e-3 * torch.randn_like(x)), ("shrunk by 1e-2", x * (1.0 - 1e-2)))
for i, (tag, xd) in enumerate(cases):
    if only is not None and int(only) != i:
        continue
    for _ in range(3): sol, lam, path = eng.verify_nodes(*args, xd, wd)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): sol, lam, path = eng.verify_nodes(*args, xd, wd)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pth = path.cpu().numpy()
    print(f"verify {n}x{m} {tag}: {ms:.3f} ms per {cnt} nodes = {cnt / ms / 1e3:.2f} M nodes/s; optimal {int(sol.sum())}, paths {np.bincount(pth, minlength=6).tolist()}")
