"""Developer aid: host time to enqueue one sweep (python + ctypes + HIP launch) on the resident-records route."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
cnt, n, m = int(os.environ.get("CNT", "10000")), 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
h = eng.upload_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u))
ring = t(np.random.default_rng(0).standard_normal((64, 8)))
x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
out = None
for k in range(50): out = h.solve(ring[k % 64], out=out, x_out=x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(500): out = h.solve(ring[k % 64], out=out, x_out=x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CNT {cnt}: host enqueue {1e6*(t1-t0)/500:.1f} us/sweep (the queue fills: includes back-pressure), total {1e6*(t2-t0)/500:.1f} us/sweep")
torch.cuda.synchronize()
ts = []
for k in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = h.solve(ring[k % 64], out=out, x_out=x); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"one call on an idle queue: median {1e6*ts[len(ts)//2]:.1f} us")
