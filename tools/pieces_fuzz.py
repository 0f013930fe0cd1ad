#!/usr/bin/env python3
"""Developer aid: qpn_local_pieces on random node shapes, bound kinds (one-sided, free, equal), recipes and piece -> node maps
against the oracle's restatement of local_piece (src/avi_solutions.jl:400-496), bit for bit (the piece is data movement).
Usage: python tools/pieces_fuzz.py [trials] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import qpn_amd
import problems as P
from qpn_amd.engine import colmajor
from oracle import binding
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
INF = np.inf
total = 0
for t in range(trials):
    n = int(rng.integers(1, 140)); m = int(rng.integers(0, 140)); p = int(rng.integers(0, 9)); cnt = int(rng.integers(1, 6))
    Q, Rm, qd, A, B, l, u = P.synth_nodes(60_000 + t, cnt, n, max(m, 1), max(p, 1))
    if m == 0:
        A = A[:, :0, :]; B = B[:, :0, :]; l = l[:, :0]; u = u[:, :0]
    if p == 0:
        Rm = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        Rm = Rm[:, :, :p]; B = rng.standard_normal((cnt, m, p))
    kind = rng.integers(0, 5, size=l.shape)
    l = np.where(kind == 1, -INF, l); u = np.where(kind == 2, INF, u)
    l = np.where(kind == 3, -INF, l); u = np.where(kind == 3, INF, u)
    u = np.where(kind == 4, l, u)
    pieces = int(rng.integers(1, 3 * cnt + 1))
    node_of = rng.integers(0, cnt, size=pieces).astype(np.int32)
    K = np.concatenate([rng.integers(1, 5, size=(pieces, n)), rng.integers(5, 9, size=(pieces, m))], axis=1).astype(np.uint8)
    Ap, lp, up, keep = eng.local_pieces(colmajor(Q), colmajor(Rm), qd, colmajor(A), colmajor(B), l, u, K, node_of=node_of)
    for k in range(pieces):
        b = node_of[k]
        Ao, lo, uo, ko = binding.local_piece(Q[b], Rm[b], qd[b], A[b], B[b], l[b], u[b], K[k])
        assert np.array_equal(Ap[k].T, Ao) and np.array_equal(lp[k], lo) and np.array_equal(up[k], uo) and np.array_equal(keep[k], ko), (t, k, n, m, p)
    total += pieces
print(f"{trials} shapes, {total} pieces: device == oracle restatement, bit for bit")
