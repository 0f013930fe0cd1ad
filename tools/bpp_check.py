#!/usr/bin/env python3
"""Developer aid: Stage B of the large-node route by block principal pivoting (symmetric resident records: QPN_OPT_SYM_ROUTE = 1)
against the delayed-update Lemke kernel alone (QPN_OPT_SYM_ROUTE = 0) on the same records: statuses, max |dz|, time per sweep.
usage: python tools/bpp_check.py [n m count]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_SYM_ROUTE
args = [int(v) for v in sys.argv[1:]] or [256, 256, 512]
eng = qpn_amd.Engine(0)
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
for i in range(0, len(args), 3):
    n, m, cnt = args[i:i + 3]
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(7000 + n + m, cnt, n, m)
    w = t(synthetic.shared_params())
    rec = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u)]
    out = {}
    for route in (1, 0):
        eng.set_option(OPT_SYM_ROUTE, route)
        h = eng.upload_nodes(*rec)
        for _ in range(3): res = h.solve(w); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): res = h.solve(w)
        e1.record(); torch.cuda.synchronize()
        out[route] = (res["z"].cpu().numpy(), res["status"].cpu().numpy(), res["pivots"].cpu().numpy(), e0.elapsed_time(e1) / 10,
                      res["resid"].cpu().numpy())
        h.close()
    z1, s1, p1, ms1, r1 = out[1]; z0, s0, p0, ms0, r0 = out[0]
    dz = np.max(np.abs(z1 - z0) / np.maximum(1.0, np.max(np.abs(z0), axis=1, keepdims=True)))
    print(f"n={n} m={m} x{cnt}: sym route {ms1:.3f} ms (solved {int((s1 == 1).sum())}, mean pivots {p1.mean():.1f}, max resid {r1.max():.2e}) | "
          f"Lemke only {ms0:.3f} ms (solved {int((s0 == 1).sum())}, mean pivots {p0.mean():.1f}); max rel |dz| {dz:.2e}", flush=True)
eng.set_option(OPT_SYM_ROUTE, 1)
