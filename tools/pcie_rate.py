#!/usr/bin/env python3
"""Developer aid: the host-pointer route (QPN_MEM_HOST) of qpn_solve_nodes on the bench workload -- node records and
results cross PCIe on every call.  Never the bench's `value`; DESIGN.md quotes it next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qpn_amd
from qpn_amd import synthetic
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(0)
Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, 10000, 32, 32, 8)
args = [np.ascontiguousarray(a) for a in (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, synthetic.shared_params(8))]
for _ in range(3): res = eng.solve_nodes(*args)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); res = eng.solve_nodes(*args); ts.append(time.perf_counter() - t0)
ts.sort()
mb = sum(a.nbytes for a in args) / 1e6
print(f"host-pointer route: {ts[len(ts)//2]*1e3:.2f} ms per 10 000 nodes (median of 10) = {10000/ts[len(ts)//2]/1e6:.2f} M solves/s; "
      f"{mb:.0f} MB of node records in, {res['z'].nbytes/1e6:.1f} MB of z out per call; solved {(res['status']==1).sum()}")
# the same sweep over RESIDENT records (qpn_nodes_upload once): per sweep only w goes up and the outputs come down
nodes = eng.upload_nodes(*args[:-1])
w = args[-1]
x = np.zeros((10000, 32))
for want, label in ((("z", "resid", "pivots", "active"), "all outputs (z, resid, pivots, active)"), ((), "status + primal blocks only")):
    for _ in range(3): out = nodes.solve(w, want=want, x_out=x)
    ts = []
    for k in range(20):
        t0 = time.perf_counter(); out = nodes.solve(w + 1e-3 * k, want=want, x_out=x); ts.append(time.perf_counter() - t0)
    ts.sort()
    down = sum(v.nbytes for v in out.values() if v is not None) + x.nbytes
    print(f"resident records, host outputs, {label}: {ts[len(ts)//2]*1e3:.3f} ms per 10 000 nodes (median of 20) = "
          f"{10000/ts[len(ts)//2]/1e6:.2f} M solves/s; {down/1e6:.2f} MB down per sweep; solved {(out['status']==1).sum()}")
