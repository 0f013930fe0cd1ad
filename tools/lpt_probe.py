import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
cnt, n, m = 10000, 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
def run(order, tag):
    args = [t(colmajor(Q[order])), t(colmajor(R[order])), t(qd[order]), t(colmajor(A[order])), t(colmajor(B[order])), t(l[order]), t(u[order]), t(w)]
    x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
    out = None
    for _ in range(5): out = eng.solve_nodes(*args, out=out, x_out=x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): out = eng.solve_nodes(*args, out=out, x_out=x)
    e1.record(); torch.cuda.synchronize()
    print(tag, e0.elapsed_time(e1) / 50, "ms/step")
    return out["pivots"].cpu().numpy()
ident = np.arange(cnt)
piv = run(ident, "natural order      ")
run(np.argsort(-piv, kind="stable"), "longest first (LPT)")
run(np.argsort(piv, kind="stable"), "shortest first     ")
rng = np.random.default_rng(0)
run(rng.permutation(cnt), "random             ")
