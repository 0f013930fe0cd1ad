#!/usr/bin/env python3
"""Developer aid: qpn_solve_mcp_csc -- PATHSolver.solve_mcp's own argument list (src/avi.jl:64-70: SparseMatrixCSC{Float64,Int32},
1-based) -- on random box-MCPs (strictly monotone M = P + skew part, random sparsity, mixed bound kinds, random z0) and on
reference-form node AVIs (the converted GAVI of src/avi.jl:113-128) against the oracle's dense solve: status equal, z within 1e-9.
Usage: python tools/csc_fuzz.py [trials] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import scipy.sparse as sp
import qpn_amd
import problems as P
from oracle import binding
eng = qpn_amd.Engine(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
worst = 0.0; solved = 0
for t in range(trials):
    N = int(rng.integers(1, 200))
    M, q, l, u, z0 = P.random_box_mcp(rng, N)
    if rng.random() < 0.5:                                   # sparsify, keeping the diagonal dominant enough to stay a P-matrix
        mask = rng.random((N, N)) < rng.uniform(0.05, 0.6)
        np.fill_diagonal(mask, True)
        M = M * mask
        M[np.arange(N), np.arange(N)] = np.abs(M).sum(axis=1) + 0.1
    S = sp.csc_matrix(M)
    rc = binding.solve_avi(M, q, l, u, z0=z0)
    st, z, info = eng.solve_mcp_csc(N, S.indptr + 1, S.indices + 1, S.data, q, l, u, z0)
    assert st == rc["status"], (t, N, st, rc["status"])
    if st == 1:
        d = np.max(np.abs(z - rc["z"])) / max(1.0, np.max(np.abs(rc["z"]))); worst = max(worst, d); solved += 1
        assert d <= 1e-9, (t, N, d)
        assert info["resid"] <= 1e-8
print(f"{trials} random box-MCPs in CSC form: statuses equal to the oracle's ({solved} solved), worst relative difference {worst:.2e}")
