import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
n = m = int(os.environ.get("NN", "40")); cnt = int(os.environ.get("CNT", "4000"))
Q, R_, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
args = [t(colmajor(Q)), t(colmajor(R_)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
for _ in range(3):
    res = eng.solve_nodes(*args)
torch.cuda.synchronize()
