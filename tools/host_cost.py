import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
cnt, n, m = int(os.environ.get("CNT", "1250")), 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w)]
x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
out = None
for _ in range(20): out = eng.solve_nodes(*args, out=out, x_out=x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): out = eng.solve_nodes(*args, out=out, x_out=x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CNT {cnt}: host enqueue {1e6*(t1-t0)/500:.1f} us/step, total {1e6*(t2-t0)/500:.1f} us/step")
