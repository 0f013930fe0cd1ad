#!/usr/bin/env python3
"""Diagnostic only: per-phase cycle shares of stage A of the mid-size node path (csrc/qpn_avi_schur_mid.hip), wave 0 of each
workgroup.  Builds a SEPARATE library with -DQPN_STAMPS; never the product build, never a timed number -- read the SHARES."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = "/tmp/libqpn_hip_stamps.so"
subprocess.check_call(["bash", os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc", "build.sh"), "-DQPN_STAMPS", "-DQPN_DEV_SWITCHES"], env=dict(os.environ, QPN_OUT=out, QPN_OBJ="/tmp/qpn_obj_stamps"))
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib
import problems as P
from qpn_amd.engine import colmajor
_lib.LIB_PATH = out
_lib._lib = None
eng = qpn_amd.Engine(0)
n = m = int(os.environ.get("NN", "48")); cnt = int(os.environ.get("CNT", "4000"))
Q, R, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
st = torch.zeros((cnt, 8), dtype=torch.int64, device="cuda:0")
eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
for _ in range(2):
    res = eng.solve_nodes(*args)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
if os.environ.get("QPN_MID_STAMP_REG"):
    names = ["setup+load", "loop control", "extract column", "pivot selection", "rank-1 update", "readback+check", "crash fast path"]
    s[:, 7] = 0
    print("Lemke kernel on the Schur problems (qpn_avi_reg.hip); mean pivots there:", float(res["pivots"].double().mean()) - n)
else:
    names = ["load + q", "steps: gather/publish/barrier", "steps: LU + U'", "steps: tile updates", "W~ out", "S product + stores"]
tot = s.sum(axis=1).mean()
print(f"n = m = {n}, {cnt} nodes: mean {tot:.0f} clocks (s_memtime) per workgroup")
for i, nm in enumerate(names):
    print(f"  {nm:32s} {s[:, i].mean():10.1f}  {100*s[:, i].mean()/tot:5.1f} %")
