#!/usr/bin/env python3
"""Diagnostic only: per-phase cycle shares of the register-tableau AVI kernel.

Builds a SEPARATE library with -DQPN_STAMPS (in-kernel s_memtime stamps; never the product
build, never a timed number -- read the SHARES), runs the config-4 batch once, prints the
mean cycles per phase per solve."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (built in-tree beforehand where there is no GPU -- `QPN_OUT=.../libqpn_hip_stamps.so QPN_OBJ=/tmp/qpn_obj_stamps build.sh
#  -DQPN_STAMPS`: git-ignored, travels with gpurun -- else built here)
out = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "libqpn_hip_stamps.so")
if not os.path.exists(out):
    out = "/tmp/libqpn_hip_stamps.so"
    subprocess.check_call(["bash", os.path.join(ROOT, "quadraticprogramnetworks.jl_amd", "csrc", "build.sh"), "-DQPN_STAMPS"],
                          env=dict(os.environ, QPN_OUT=out, QPN_OBJ="/tmp/qpn_obj_stamps"), stdout=subprocess.DEVNULL)
import numpy as np, torch
import qpn_amd
from qpn_amd import _lib, synthetic
from qpn_amd.engine import colmajor
_lib.LIB_PATH = out
_lib._lib = None
eng = qpn_amd.Engine(0)
cnt, n, m = int(os.environ.get("CNT", "10000")), 32, 32
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, cnt, n, m)
w = synthetic.shared_params()
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
Mc, q, lo, hi, kind = eng.assemble_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w))
st = torch.zeros((cnt, 8), dtype=torch.int64, device="cuda:0")
eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
fused = os.environ.get("FUSED", "0") == "1"
for _ in range(2):
    if fused:
        res = eng.solve_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w))
    else:
        res = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
piv = res["pivots"].cpu().numpy().mean()
names = ["setup+load", "loop control", "extract column", "pivot selection", "rank-1 update", "readback+check", "crash fast path"]
if fused or os.environ.get("SCHUR_NAMES"):
    names = ["setup+load", "B: column via LDS", "B: ratio test+row", "B: row via LDS+exchange", "B: bookkeeping/flips", "readback+check", "A: MFMA crash"]
endt = s[:, 7].copy(); s[:, 7] = 0
tot = s.sum(axis=1).mean()
print(f"mean pivots {piv:.1f} (Stage B: {piv - n:.1f}), mean cycles per solve {tot:.0f} ({tot/piv:.0f} per pivot)")
for i, nm in enumerate(names):
    print(f"  {nm:18s} {s[:, i].mean():10.0f} cycles  {100*s[:, i].mean()/tot:5.1f} %")

if fused:
    # timeline of the launch from the device-wide 100 MHz clock (10 ns ticks)
    raw = st.cpu().numpy()[:, 7].astype(np.uint64)
    start = (raw >> np.uint64(32)).astype(np.float64) * 0.01; endt = (raw & np.uint64(0xffffffff)).astype(np.float64) * 0.01
    t0 = start.min(); start -= t0; endt -= t0
    print(f"launch timeline (us from the first block's start): last start {start.max():.1f}, last end {endt.max():.1f}")
    order = np.argsort(start)
    for lo_, hi_ in ((0, 4096), (4096, 8192), (8192, cnt)):
        idx = order[lo_:hi_]
        if len(idx):
            d = endt[idx] - start[idx]
            print(f"  blocks {lo_}-{hi_} by start: starts {start[idx].min():6.1f}..{start[idx].max():6.1f} us, duration mean {d.mean():5.1f} (min {d.min():5.1f}, max {d.max():5.1f}) us, "
                  f"ends {endt[idx].min():6.1f}..{endt[idx].max():6.1f} us, load phase {s[idx, 0].mean():7.0f} cycles")
    for tq in (10, 20, 30, 40, 60, 80, 100, 120, 140, 160):
        print(f"    t = {tq:3d} us: running {int(((start <= tq) & (endt > tq)).sum()):5d}, finished {int((endt <= tq).sum()):5d}")
