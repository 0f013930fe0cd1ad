import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
eng = qpn_amd.default_engine(0)
cnt, n, m = 512, 256, 256
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
Mc, q, lo, hi, kind = eng.assemble_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w))
res = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): res = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"config5: {cnt} x N=512: {dt*1e3:.2f} ms per batch = {cnt/dt:.0f} solves/s; status ok {int((res['status']==1).sum())}/{cnt}; mean pivots {res['pivots'].double().mean().item():.1f}; max resid {res['resid'].max().item():.2e}")
