#!/usr/bin/env python3
"""Developer aid: BASELINE config 5 item shape (n = m = 256, N_red = 512) on the large-item path -- time per
batch and agreement between the blocked-MFMA-crash path and the general kernel (QPN_AVI_BIG_KERNEL=general: honoured by a
diagnostic build of the library only -- csrc/build.sh -DQPN_DEV_SWITCHES with QPN_OUT=..., loaded through QPN_HIP_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
cnt, n, m = int(os.environ.get("CNT", "64")), int(os.environ.get("NN", "256")), int(os.environ.get("MM", "256"))
Q, R, qd, A, B, l, u = P.synth_nodes(7000 + n, cnt, n, m)
M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, P.shared_params())
eng = qpn_amd.Engine(0)
t = lambda a, dt=torch.float64: torch.tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
args = (t(colmajor(M)), t(q), t(lo), t(hi))
kd = t(kind, torch.uint8)
res = eng.solve_avi_batch(*args, kind=kd)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = eng.solve_avi_batch(*args, kind=kd)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = res["status"].cpu().numpy()
print(f"{os.environ.get('QPN_AVI_BIG_KERNEL', 'schur-big')}: {cnt} items n={n} m={m}: {dt*1e3:.1f} ms, solved {(st == 1).sum()}, "
      f"max resid {res['resid'].max().item():.2e}, mean pivots {res['pivots'].double().mean().item():.1f}")
np.save(f"/tmp/c5_{os.environ.get('QPN_AVI_BIG_KERNEL', 'schur')}.npy", res["z"].cpu().numpy())
np.save(f"/tmp/c5a_{os.environ.get('QPN_AVI_BIG_KERNEL', 'schur')}.npy", res["active"].cpu().numpy())
if os.path.exists("/tmp/c5_general.npy") and os.path.exists("/tmp/c5_schur.npy"):
    za, zb = np.load("/tmp/c5_general.npy"), np.load("/tmp/c5_schur.npy")
    print("max |z_schur - z_general| =", np.max(np.abs(za - zb)), " masks equal:", np.array_equal(np.load("/tmp/c5a_general.npy"), np.load("/tmp/c5a_schur.npy")))
