"""Test double: the Engine interface (qpn_amd.engine.Engine) served by the CPU oracle.

TEST INFRASTRUCTURE ONLY -- lets `-m "not gpu"` tests drive the product's HOST logic (avi.py,
qp_processing.py, algorithm.py, sharding.py) without a GPU.  The product never imports this; the
`-m gpu` twins of those tests run the same host code on the real HIP engine."""
from __future__ import annotations

import numpy as np

from oracle import binding as ob


class OracleEngine:
    device = -1

    def __init__(self):
        import collections
        self.calls = collections.Counter()           # calls per entry point (the HIP engine counts its C-ABI calls the same way)

    def __getattribute__(self, name):
        attr = object.__getattribute__(self, name)
        if callable(attr) and not name.startswith("_") and name != "calls":
            object.__getattribute__(self, "calls")[name] += 1
        return attr

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

    # ---- the pieces of the solution graph and the pool assembly (rows F1 / A6), served by the oracle / by plain numpy ----
    def recipes_from_masks(self, mask, first=0, count=None):
        """all_Ks (src/avi_solutions.jl:200-215): recipes first .. first+count-1 of the Cartesian product of the rows' code
        sets, row 0 fastest (the device kernel's mixed-radix order)."""
        mask = np.asarray(mask, dtype=np.uint8)
        sets = [[c + 1 for c in range(8) if (int(mk) >> c) & 1] for mk in mask]
        total = int(np.prod([len(s) for s in sets])) if all(sets) else 0
        if count is None:
            count = total - first
        if first < 0 or first + count > total:
            raise ValueError("recipes_from_masks: range outside the product")
        K = np.zeros((count, len(sets)), np.uint8)
        for t in range(count):
            idx = first + t
            for i, s in enumerate(sets):
                K[t, i] = s[idx % len(s)]; idx //= len(s)
        return K, total

    def local_pieces(self, Qc, Rc, qd, Ac, Bc, l, u, K, node_of=None):
        K = np.atleast_2d(np.asarray(K, dtype=np.uint8))
        pieces = K.shape[0]
        n = np.shape(qd)[1]; m = np.shape(l)[1]; p = np.shape(Rc)[1]; N = n + m
        node_of = np.arange(pieces) if node_of is None else np.asarray(node_of)
        Ap = np.zeros((pieces, N + p, 2 * N)); lp = np.zeros((pieces, 2 * N)); up = np.zeros((pieces, 2 * N))
        keep = np.zeros((pieces, 2 * N), np.uint8)
        for t in range(pieces):
            b = int(node_of[t])
            A_, lp[t], up[t], keep[t] = ob.local_piece(np.asarray(Qc[b]).T, np.asarray(Rc[b]).T.reshape(n, p), qd[b],
                                                       np.asarray(Ac[b]).T.reshape(m, n), np.asarray(Bc[b]).T.reshape(m, p),
                                                       l[b], u[b], K[t])
            Ap[t] = A_.T                      # column-major per piece, as the ABI returns it
        return Ap, lp, up, keep

    def assemble_pools(self, n_i, m_i, dpos, nd, Qd, Qp, qd, Ad, Bp, l, u, w, form="reduced", share_M=None):
        """combine_gavis / its reduced form, restated with plain numpy on the stacked blocks (ABI layout in: column-major blocks;
        out: Mc column-major).  Inputs with a leading batch dimension give a batch of pool instances of the one shape."""
        dims = dict(Qd=2, Qp=2, qd=1, Ad=2, Bp=2, l=1, u=1, w=1)
        arrs = dict(Qd=Qd, Qp=Qp, qd=qd, Ad=Ad, Bp=Bp, l=l, u=u, w=w)
        batch = max([np.shape(a)[0] for k, a in arrs.items() if np.ndim(a) == dims[k] + 1], default=0)
        if batch:
            outs = [self._assemble_pool_one(n_i, m_i, dpos, nd, *[(np.asarray(a)[t] if np.ndim(a) == dims[k] + 1 else a)
                                                                 for k, a in arrs.items()], form=form) for t in range(batch)]
            return (np.stack([o[0] for o in outs]), np.concatenate([o[1] for o in outs]), np.concatenate([o[2] for o in outs]),
                    np.concatenate([o[3] for o in outs]), np.concatenate([o[4] for o in outs]))
        return self._assemble_pool_one(n_i, m_i, dpos, nd, Qd, Qp, qd, Ad, Bp, l, u, w, form=form)

    def _assemble_pool_one(self, n_i, m_i, dpos, nd, Qd, Qp, qd, Ad, Bp, l, u, w, form="reduced"):
        n_i = [int(v) for v in n_i]; m_i = [int(v) for v in m_i]; dpos = [int(v) for v in dpos]
        sn, sm = sum(n_i), sum(m_i)
        Qd = np.asarray(Qd, float).T.reshape(sn, nd); Ad = np.asarray(Ad, float).T.reshape(sm, nd)
        p = int(np.shape(w)[-1])
        Qp = np.asarray(Qp, float).T.reshape(sn, p); Bp = np.asarray(Bp, float).T.reshape(sm, p)
        qd = np.ravel(qd).astype(float); l = np.ravel(l).astype(float); u = np.ravel(u).astype(float); w = np.ravel(w).astype(float)
        owner_n = np.repeat(np.arange(len(n_i)), n_i); owner_m = np.repeat(np.arange(len(m_i)), m_i)
        # -A_i[:, dvars_i]' of player i: rows = the player's own variables, columns = its own constraint rows
        C = np.zeros((sn, sm))
        for r in range(sn):
            for c in range(sm):
                if owner_n[r] == owner_m[c]:
                    C[r, c] = -Ad[c, dpos[r]]
        inf = np.inf
        if form == "reduced":
            M = np.block([[Qd, C], [Ad, np.zeros((sm, sm))]])
            q = np.concatenate([qd + Qp @ w, Bp @ w])
            lo = np.concatenate([np.full(sn, -inf), l]); hi = np.concatenate([np.full(sn, inf), u])
            kind = np.concatenate([np.zeros(sn, np.uint8), np.ones(sm, np.uint8)])
        else:
            # z = [dvars (nd); xi (sn); lambda (sm); slack (sm)]: sum-of-xi rows, players' rows, A z - s = 0, s in [l, u]
            top = np.zeros((nd, nd + sn + 2 * sm))
            for r in range(sn):
                top[dpos[r], nd + r] = 1.0
            mid = np.hstack([Qd, np.zeros((sn, sn)), C, np.zeros((sn, sm))])
            con = np.hstack([Ad, np.zeros((sm, sn + sm)), -np.eye(sm)])
            bot = np.hstack([np.zeros((sm, nd + sn)), np.eye(sm), np.zeros((sm, sm))])
            M = np.vstack([top, mid, con, bot])
            q = np.concatenate([np.zeros(nd), qd + Qp @ w, Bp @ w, np.zeros(sm)])
            lo = np.concatenate([np.full(nd + sn + sm, -inf), l]); hi = np.concatenate([np.full(nd + sn + sm, inf), u])
            kind = np.zeros(nd + sn + 2 * sm, np.uint8)
        return np.ascontiguousarray(M.T), q[None], lo[None], hi[None], kind[None]

    # ---- node records: assemble + solve (the HIP engine fuses them: qpn_solve_nodes) ------------------------------------
    def solve_nodes(self, Qc, Rc, qd, Ac, Bc, l, u, w, z0=None, opts=None, want_active=True, out=None, x_out=None):
        batch, n = np.shape(qd); m = np.shape(l)[1]; N = n + m
        Ms = np.zeros((batch, N, N)); qs = np.zeros((batch, N)); lo = np.zeros((batch, N)); hi = np.zeros((batch, N))
        kind = np.zeros((batch, N), np.uint8)
        for b in range(batch):
            wb = w if np.ndim(w) == 1 else w[b]
            Ms[b], qs[b], lo[b], hi[b], kind[b] = ob.assemble_node(np.asarray(Qc[b]).T, np.asarray(Rc[b]).T.reshape(n, -1), qd[b],
                                                                   np.asarray(Ac[b]).T.reshape(m, n), np.asarray(Bc[b]).T.reshape(m, -1),
                                                                   l[b], u[b], wb)
        res = ob.solve_avi_batch(Ms, qs, lo, hi, z0=z0, kind=kind)
        if x_out is not None:
            x_out[:, :n] = res["z"][:, :n]
        return res

    # ---- all_Ks for many solutions at once (qpn_recipes_batch) ----------------------------------------------------------
    def recipes_batch(self, masks, offsets):
        masks = np.asarray(masks, dtype=np.uint8); offsets = np.asarray(offsets, dtype=np.int64)
        nodes, N = masks.shape
        K = np.zeros((int(offsets[-1]), N), np.uint8); node_of = np.zeros(int(offsets[-1]), np.int32)
        for b in range(nodes):
            cnt = int(offsets[b + 1] - offsets[b])
            if cnt:
                sets = [[c + 1 for c in range(8) if (int(mk) >> c) & 1] for mk in masks[b]]
                for t in range(cnt):
                    idx = t
                    for i, s_ in enumerate(sets):
                        if s_:
                            K[offsets[b] + t, i] = s_[idx % len(s_)]; idx //= len(s_)
                node_of[offsets[b]:offsets[b + 1]] = b
        return K, node_of

    # ---- local pieces with the multipliers eliminated (qpn_reduced_pieces) ----------------------------------------------
    def reduced_pieces(self, Qc, Rc, qd, Ac, Bc, l, u, K, node_of, tol=1e-9):
        """Per recipe: local_piece (src/avi_solutions.jl:400-496), then every multiplier column j = n .. n+m-1 eliminated through
        the alive equality row with the largest |A[i, j]| (first such row; the row then leaves); a column no equality pins while
        some alive row still holds it raises the piece's flag.  Output: the alive rows in order over the columns [x_d; x_p]."""
        K = np.atleast_2d(np.asarray(K, dtype=np.uint8)); node_of = np.asarray(node_of)
        pieces = K.shape[0]
        n = np.shape(qd)[1]; m = np.shape(l)[1]; p = np.shape(Rc)[1]; N = n + m
        cap = n + 2 * m
        Ar = np.zeros((pieces, n + p, cap)); lr = np.full((pieces, cap), -np.inf); ur = np.full((pieces, cap), np.inf)
        rows = np.zeros(pieces, np.int32); flags = np.zeros(pieces, np.int32)
        for t in range(pieces):
            b = int(node_of[t])
            A, lo, hi, keep = ob.local_piece(np.asarray(Qc[b]).T, np.asarray(Rc[b]).T.reshape(n, p), qd[b], np.asarray(Ac[b]).T.reshape(m, n),
                                             np.asarray(Bc[b]).T.reshape(m, p), l[b], u[b], K[t])
            A = np.array(A, dtype=np.float64); lo = np.array(lo); hi = np.array(hi)
            alive = np.asarray(keep).astype(bool)
            for j in range(n, N):
                cand = alive & (lo == hi) & np.isfinite(lo)
                col = np.where(cand, np.abs(A[:, j]), 0.0)
                i = int(np.argmax(col))
                if col[i] <= tol:
                    if np.any(alive & (np.abs(A[:, j]) > tol)):
                        flags[t] = 1
                    continue
                piv = A[i, j]
                for k in np.nonzero(alive & (A[:, j] != 0.0))[0]:
                    if k == i:
                        continue
                    f = A[k, j] / piv
                    A[k] = A[k] - f * A[i]; A[k, j] = 0.0
                    lo[k] = lo[k] - f * lo[i]; hi[k] = hi[k] - f * hi[i]
                alive[i] = False
            idx = np.nonzero(alive)[0]
            r = len(idx)
            assert r <= cap
            cols = list(range(n)) + list(range(N, N + p))
            Ar[t, :, :r] = A[np.ix_(idx, cols)].T
            lr[t, :r] = lo[idx]; ur[t, :r] = hi[idx]; rows[t] = r
        return Ar, lr, ur, rows, flags
