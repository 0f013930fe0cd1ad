"""Test double: the Engine interface (qpn_amd.engine.Engine) served by the CPU oracle.

TEST INFRASTRUCTURE ONLY -- lets `-m "not gpu"` tests drive the product's HOST logic (avi.py,
qp_processing.py, algorithm.py, sharding.py) without a GPU.  The product never imports this; the
`-m gpu` twins of those tests run the same host code on the real HIP engine."""
from __future__ import annotations

import numpy as np

from oracle import binding as ob


class OracleEngine:
    device = -1

    def solve_avi_batch(self, Mc, q, l, u, z0=None, kind=None, opts=None, want_active=True):
        Mc = np.asarray(Mc, dtype=np.float64)
        M = np.swapaxes(Mc, -1, -2)          # ABI layout (column-major per item) -> math layout
        return ob.solve_avi_batch(M, q, l, u, z0=z0, kind=kind)

    def check_avi_batch(self, Mc, q, l, u, z, kind=None, tol=1e-6, want_r=True):
        Mc = np.asarray(Mc, dtype=np.float64)
        q = np.asarray(q); batch, N = q.shape
        deg = np.zeros(batch, np.int32); r = np.zeros((batch, N))
        for b in range(batch):
            M = (Mc if Mc.ndim == 2 else Mc[b]).T
            k = None if kind is None else (kind if np.ndim(kind) == 1 else kind[b])
            _, deg[b], r[b] = ob.check_avi_solution(M, q[b], l[b], u[b], z[b], kind=k, tol=tol)
        return deg, r

    def comp_indices(self, zv, rv, l, u, tol=1e-2, shift=0):
        return ob.comp_indices(np.ravel(zv), np.ravel(rv), np.ravel(l), np.ravel(u), tol=tol, shift=shift).reshape(np.shape(zv))

    def assemble_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, w):
        batch, n = np.shape(qd); m = np.shape(l)[1]; N = n + m
        Mo = np.zeros((batch, N, N)); qo = np.zeros((batch, N)); lo = np.zeros((batch, N)); uo = np.zeros((batch, N))
        kind = np.zeros((batch, N), np.uint8)
        for b in range(batch):
            wb = w if np.ndim(w) == 1 else w[b]
            M, qo[b], lo[b], uo[b], kind[b] = ob.assemble_node(Qc[b].T, Rc[b].T, qd[b], Ac[b].T, Bc[b].T, l[b], u[b], wb)
            Mo[b] = M.T
        return Mo, qo, lo, uo, kind

    def verify_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, xd, w, tol=1e-4):
        batch, n = np.shape(qd); m = np.shape(l)[1]
        sol = np.zeros(batch, np.int32); path = np.zeros(batch, np.int32); lam = np.zeros((batch, max(m, 1)))
        for b in range(batch):
            wb = w if np.ndim(w) == 1 else w[b]
            s, lm, p = ob.verify_solution(Qc[b].T, Rc[b].T, qd[b], Ac[b].T, Bc[b].T, l[b], u[b], xd[b], wb, tol=tol)
            sol[b] = s; path[b] = p; lam[b, :m] = lm
        return sol, lam[:, :m], path
