#!/usr/bin/env python3
"""Diagnostic only (QPN_STAMPS build): per-phase clocks of the one-wavefront 33-48 kernel (csrc/qpn_avi_schur48.hip).  NN=48 CNT=256,4000"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib
import problems as P
from qpn_amd.engine import colmajor
_lib.LIB_PATH = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "libqpn_hip_stamps.so")
_lib._lib = None
eng = qpn_amd.Engine(0)
eng.set_option(_lib.OPT_MID_ROUTE, 1)
names = ["loads + tiles", "stage A", "S product, c, W~ parked", "Lemke: row, bookkeeping, exchange", "read-back", "post-check + stores",
         "Lemke: entering column through LDS", "Lemke: ratio test"]
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for n in [int(x) for x in os.environ.get("NN", "48").split(",")]:
    for cnt in [int(x) for x in os.environ.get("CNT", "256,4000").split(",")]:
        Q, R, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, n)
        args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
        st = torch.zeros((cnt, 8), dtype=torch.int64, device="cuda:0")
        eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
        for _ in range(2):
            res = eng.solve_nodes(*args)
        torch.cuda.synchronize()
        s = st.cpu().numpy().astype(np.float64).mean(axis=0)
        lp = float(res["pivots"].double().mean()) - n
        print(f"n = m = {n}, {cnt} nodes: {s[:8].sum():.0f} clocks per node; {lp:.1f} Lemke pivots ({(s[3] + s[6] + s[7]) / max(lp, 1):.0f} clocks each)")
        for k, nm in enumerate(names):
            print(f"   {nm:28s} {s[k]:9.0f}  {100 * s[k] / s[:8].sum():5.1f} %")
