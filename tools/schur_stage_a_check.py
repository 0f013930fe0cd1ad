#!/usr/bin/env python3
"""Diagnostic: Stage A of the MFMA Schur kernel (qpn_avi_schur.hip) against numpy.
Builds a separate library with -DQPN_DIAG; never the product build."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
csrc = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc")
out = "/tmp/libqpn_hip_diag.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DQPN_DIAG", "-ffp-contract=off",
                       "-o", out] + [os.path.join(csrc, f) for f in
                       ("qpn_capi.hip", "qpn_avi_solve.hip", "qpn_avi_reg.hip", "qpn_avi_big.hip", "qpn_avi_schur.hip", "qpn_avi_schur_big.hip", "qpn_kkt.hip", "qpn_verify.hip")])
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib, synthetic
from qpn_amd.engine import colmajor
import problems as P
_lib.LIB_PATH = out; _lib._lib = None
eng = qpn_amd.Engine(0)
worst = 0.0
for (n, m, cnt) in [(32, 32, 64), (20, 31, 8), (5, 9, 8), (32, 1, 4), (1, 32, 4)]:
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(100, cnt, n, m)
    w = synthetic.shared_params()
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    dM, dq, dl, du, dk = t(colmajor(M)), t(q), t(lo), t(hi), t(kind, torch.uint8)
    st = torch.zeros(cnt, dtype=torch.int32, device="cuda:0")
    S = torch.zeros((cnt, 32, 32), dtype=torch.float64, device="cuda:0"); W = torch.zeros_like(S)
    c = torch.zeros((cnt, 32), dtype=torch.float64, device="cuda:0"); h = torch.zeros_like(c)
    eng.lib.qpn_debug_schur_stage_a.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 10
    rc = eng.lib.qpn_debug_schur_stage_a(eng.ctx, cnt, n + m, dM.data_ptr(), dq.data_ptr(), dl.data_ptr(), du.data_ptr(),
                                         dk.data_ptr(), st.data_ptr(), S.data_ptr(), c.data_ptr(), W.data_ptr(), h.data_ptr())
    torch.cuda.synchronize()
    assert rc == 0 and bool((st == -2).all()), (rc, st[:8])
    S, W, c, h = (x.cpu().numpy() for x in (S, W, c, h))
    for i in range(cnt):
        H = M[i][:n, :n]; Cb = M[i][:n, n:]; Ab = M[i][n:, :n]; D = M[i][n:, n:]
        Wr = np.linalg.solve(H, Cb); hr = np.linalg.solve(H, q[i][:n])
        Sr = D - Ab @ Wr; cr = q[i][n:] - Ab @ hr
        e = max(np.max(np.abs(S[i][:m, :m] - Sr)), np.max(np.abs(W[i][:n, :m] - Wr)), np.max(np.abs(c[i][:m] - cr)), np.max(np.abs(h[i][:n] - hr)))
        worst = max(worst, e)
    print(f"n={n} m={m}: max deviation so far {worst:.2e}")
assert worst < 1e-10, worst
print("Stage A (MFMA) matches numpy")
