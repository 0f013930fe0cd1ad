#!/usr/bin/env python3
"""Developer aid: run the fused node kernel on the bench workload with the constraint bounds scaled by
SCALE (wider bounds -> fewer active constraints -> fewer Lemke pivots).  Under `rocprofv3 --pmc
SQ_INSTS_VALU ...` two scales give the per-pivot instruction cost as a slope."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic, _lib
if os.environ.get('QPN_LIB'):
    _lib.LIB_PATH = os.environ['QPN_LIB']; _lib._lib = None
from qpn_amd.engine import colmajor
scale = float(os.environ.get("SCALE", "1.0"))
cnt, n, m = 10000, 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
l = l * scale; u = u * scale
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w)]
for _ in range(3):
    res = eng.solve_nodes(*args)
torch.cuda.synchronize()
print(f"SCALE {scale}: mean pivots {res['pivots'].double().mean().item():.2f} (Stage B {res['pivots'].double().mean().item() - n:.2f}), solved {(res['status'] == 1).sum().item()}")
