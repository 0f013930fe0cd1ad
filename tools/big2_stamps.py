#!/usr/bin/env python3
"""Diagnostic only: per-phase clocks of schur_big2_stage_a (wave 0 of each workgroup), -DQPN_STAMPS library (see
tools/wg_stamps.py for how it is built).  CNT nodes of n = m = NN."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
out = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "libqpn_hip_stamps.so")
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib
import problems as P
from qpn_amd.engine import colmajor
_lib.LIB_PATH = out
_lib._lib = None
eng = qpn_amd.Engine(0)
n = m = int(os.environ.get("NN", "256"))
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
names = ["setup", "pivot block + first pivot rows -> LDS", "inversion: the three tile steps", "U'", "staging of later mega-chunks (2 barriers)", "inversion: diagonal tile (wave 0) + barrier", "update: this wave's tiles", "end-of-pass fence + barrier"]
for cnt in [int(x) for x in os.environ.get("CNT", "512,256").split(",")]:
    Q, R, qd, A, B, l, u = P.synth_nodes(7000 + n, cnt, n, m)
    args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
    st = torch.zeros((2 * cnt, 8), dtype=torch.int64, device="cuda:0")
    eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
    for _ in range(3):
        res = eng.solve_nodes(*args)
    torch.cuda.synchronize()
    sall = st.cpu().numpy().astype(np.float64)
    s = sall[:cnt]
    tot = s.sum(axis=1).mean()
    print(f"n = m = {n}, {cnt} nodes: mean {tot:.0f} clocks per workgroup, solved {(res['status'] == 1).float().mean().item() * 100:.0f} %")
    for i, nm in enumerate(names):
        print(f"  {nm:36s} {s[:, i].mean():12.0f}  {100 * s[:, i].mean() / tot:5.1f} %", flush=True)
    s = sall[cnt:]
    tot = s.sum(axis=1).mean()
    lp = float(res["pivots"].double().mean()) - n
    print(f"  Lemke kernel (thread 0): mean {tot:.0f} clocks per workgroup, {lp:.1f} pivots -> {tot / max(lp, 1):.0f} per pivot")
    for i, nm in enumerate(["setup", "entering column", "ratio test + two reductions", "pivot row -> pending pair", "bookkeeping + next column index", "folds", "exit"]):
        print(f"    {nm:36s} {s[:, i].mean():12.0f}  {100 * s[:, i].mean() / tot:5.1f} %", flush=True)
