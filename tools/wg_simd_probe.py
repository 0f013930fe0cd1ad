#!/usr/bin/env python3
"""Diagnostic only (QPN_STAMPS build): on which SIMD of its CU did each wavefront of the fused mid-size kernel run?
Wave 0 is the Lemke leader (ratio test + bookkeeping + its own tile): if the leaders of the workgroups resident on a CU share a
SIMD, that SIMD bounds the CU.  NN=48 CNT=4000"""
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
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
n = m = int(os.environ.get("NN", "48")); cnt = int(os.environ.get("CNT", "4000"))
Q, R, qd, A, B, l, u = P.synth_nodes(5000 + n, cnt, n, m)
args = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(P.shared_params())]
st = torch.zeros((cnt, 8), dtype=torch.int64, device="cuda:0")
eng.lib.qpn_debug_set_stamps(C.c_void_p(st.data_ptr()))
for _ in range(2):
    res = eng.solve_nodes(*args)
torch.cuda.synchronize()
s0 = st[:, 0].cpu().numpy().astype(np.uint64)
nw = (max(n, m) + 15) // 16
simd = np.stack([(s0 >> np.uint64(48 + 2 * v)) & np.uint64(3) for v in range(nw)], axis=1).astype(int)
print("leader (wave 0) SIMD histogram:", np.bincount(simd[:, 0], minlength=4))
for v in range(1, nw):
    print(f"wave {v}: SIMD relative to the leader's (mod 4):", np.bincount((simd[:, v] - simd[:, 0]) % 4, minlength=4))

# per CU: which SIMDs do the leaders of the workgroups that are resident TOGETHER sit on?  (first round: the first 4 (5) x 256 workgroups)
s1 = st[:, 1].cpu().numpy().astype(np.uint64)
hw = (s1 >> np.uint64(48)).astype(np.int64) & 0xFFFF
xcc = (s1 >> np.uint64(44)).astype(np.int64) & 0xF
cu_key = (xcc << 16) | (hw & 0xFF00)                 # XCC, SE, SH, CU
slot = hw & 0xF
first = np.arange(cnt) < (1280 if nw == 3 else 1024)
from collections import Counter, defaultdict
per_cu = defaultdict(list)
for b in np.nonzero(first)[0]:
    per_cu[int(cu_key[b])].append((int(simd[b, 0]), int(slot[b])))
pat = Counter(tuple(sorted(Counter(s for s, _ in v).values(), reverse=True)) for v in per_cu.values())
print(f"first round: {len(per_cu)} CUs; leaders per SIMD on a CU (sorted counts) -> number of CUs:", dict(pat))
same = sum(1 for v in per_cu.values() for (s, w) in v if s == (w & 3))
print("leader's SIMD id == its wave-slot id (mod 4) in", same, "of", int(first.sum()), "first-round workgroups; slot histogram:", np.bincount(slot[first], minlength=8)[:8])
