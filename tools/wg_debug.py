#!/usr/bin/env python3
"""Developer aid: the fused mid-size kernel (QPN_OPT_MID_ROUTE = 1) against round 2's three-kernel route (= 2) on a few
nodes, with the differences printed per block.  Usage: python tools/wg_debug.py n m [cnt] [p]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from qpn_amd._lib import OPT_MID_ROUTE
n, m = int(sys.argv[1]), int(sys.argv[2])
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 4
p = int(sys.argv[4]) if len(sys.argv) > 4 else 8
eng = qpn_amd.Engine(0)
Q, Rm, qd, A, B, l, u = P.synth_nodes(900 + n + m, cnt, n, m, max(p, 1))
Rm = Rm[:, :, :p]; B = B[:, :, :p]
abi = [colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u]
w = np.random.default_rng(n).standard_normal(p)
new = eng.solve_nodes(*abi, w)
eng.set_option(OPT_MID_ROUTE, 2)
old = eng.solve_nodes(*abi, w)
np.set_printoptions(linewidth=200, precision=4, suppress=False)
print("status new", new["status"], "old", old["status"])
print("pivots new", new["pivots"], "old", old["pivots"])
print("resid  new", new["resid"], "old", old["resid"])
dz = np.abs(new["z"] - old["z"])
print("max |dx| per node", dz[:, :n].max(axis=1), " max |dlambda| per node", dz[:, n:].max(axis=1))
k = 0
print("node 0 x   new", new["z"][k, :8], "\n         old", old["z"][k, :8])
print("node 0 lam new", new["z"][k, n:n + 8], "\n         old", old["z"][k, n:n + 8])
print("active equal:", np.array_equal(new["active"], old["active"]))
