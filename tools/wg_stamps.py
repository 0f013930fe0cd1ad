#!/usr/bin/env python3
"""Diagnostic only: per-phase cycle shares of the fused mid-size kernel (csrc/qpn_avi_schur_wg.hip), wave 0 of each
workgroup.  Builds a SEPARATE library with -DQPN_STAMPS; never the product build, never a timed number -- read the SHARES.
NN=48 CNT=4000 (CNT small = uncontended: one workgroup per CU or less)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# (built in-tree beforehand where there is no GPU -- `QPN_OUT=.../libqpn_hip_stamps.so QPN_OBJ=/tmp/qpn_obj_stamps build.sh
#  -DQPN_STAMPS`: git-ignored, travels with gpurun -- else built here)
out = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "libqpn_hip_stamps.so")
if not os.path.exists(out):
    out = "/tmp/libqpn_hip_stamps.so"
    env = dict(os.environ, QPN_OUT=out, QPN_OBJ="/tmp/qpn_obj_stamps")
    subprocess.check_call(["bash", os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc", "build.sh"), "-DQPN_STAMPS"], env=env,
                          stdout=subprocess.DEVNULL)
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib
import problems as P
from qpn_amd.engine import colmajor
_lib.LIB_PATH = out
_lib._lib = None
eng = qpn_amd.Engine(0)
names = ["load + read-back + post-check", "stage A: gather/publish + barrier", "stage A: LU + U'", "stage A: tile updates",
         "W~ hand-over, S, tile hand-over", "Lemke: column + barrier A", "Lemke: leader's turn + barrier B",
         "Lemke: exchange"]
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for n in [int(x) for x in os.environ.get("NN", "48").split(",")]:
    m = n
    for cnt in [min(int(x), 4000 if n <= 64 else 1024) for x in os.environ.get("CNT", "4000").split(",")]:
        Q, R, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
        args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
        st = torch.zeros((2 * cnt if n > 64 else cnt, 8), dtype=torch.int64, device="cuda:0")
        eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
        for _ in range(2):
            res = eng.solve_nodes(*args)
        torch.cuda.synchronize()
        sall = (st.cpu().numpy().astype(np.uint64) & np.uint64((1 << 48) - 1)).astype(np.float64)      # (bits 48.. of slot 0: SIMD ids, tools/wg_simd_probe.py)
        lp = float(res["pivots"].double().mean()) - n
        steps = n // 4 + (n % 4 > 0)
        if n > 64:      # csrc/qpn_avi_schur_wg2.hip: the LEADER's stamps (C wave 0), then H wave 0's
            for who, s, nm_ in (("leader (C wave 0)", sall[:cnt],
                                 ["load + stage A + read-back + post-check", "-", "-", "-", "W~ hand-over, waits for S, Ad staging",
                                  "Lemke: bookkeeping + barriers + next column derived", "Lemke: decision half of the turn", "-"]),
                                ("H wave 0", sall[cnt:],
                                 ["load + read-back + post-check", "stage A", "-", "-", "W~ hand-over, S, tile hand-over",
                                  "Lemke: decision read + publish + barrier A'", "Lemke: wait for the decision (barrier B)", "Lemke: exchange"])):
                tot = s.sum(axis=1).mean()
                print(f"n = m = {n}, {cnt} nodes, {who}: mean {tot:.0f} clocks (s_memtime) per workgroup; {steps} block pivots, {lp:.1f} Lemke pivots")
                for i, nm in enumerate(nm_):
                    if nm != "-":
                        print(f"  {nm:52s} {s[:, i].mean():10.1f}  {100*s[:, i].mean()/tot:5.1f} %")
                print(f"  per block pivot {s[:, 1].mean() / steps:8.0f}   per Lemke pivot {(s[:, 5:8].sum(axis=1).mean()) / max(lp, 1):8.0f}", flush=True)
            continue
        s = sall
        tot = s.sum(axis=1).mean()
        print(f"n = m = {n}, {cnt} nodes: mean {tot:.0f} clocks (s_memtime) per workgroup; {steps} block pivots, {lp:.1f} Lemke pivots")
        for i, nm in enumerate(names):
            print(f"  {nm:42s} {s[:, i].mean():10.1f}  {100*s[:, i].mean()/tot:5.1f} %")
        print(f"  per block pivot {(s[:, 1:4].sum(axis=1).mean()) / steps:8.0f}   per Lemke pivot {(s[:, 5:8].sum(axis=1).mean()) / max(lp, 1):8.0f}", flush=True)
