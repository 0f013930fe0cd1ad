"""Host-side mirror of the array part of src/avi_solutions.jl: comp_indices (the "active set")
and -- scope row F1 -- the local pieces of a node's solution map, made by the device kernels (solution_graph_pieces).

comp_indices masks: bit (c-1) set for code c.  Codes (src/avi_solutions.jl:390-399, :568-586):
  1  r >= 0, z = l      2  r = 0, l <= z <= u      3  r <= 0, z = u      4  l = z = u
  5..8 the same for the second GAVI condition (s = Az+Bw against l2,u2 with multiplier z2).
"""
from __future__ import annotations


import numpy as np

from .avi import GAVI, _eng
from .programs import Poly

INF = np.inf


def comp_indices(gavi: GAVI, z, w, tol=1e-2, engine=None):
    """src/avi_solutions.jl:587-612 -> uint8 mask per row of z = [z1; z2] (d1 + d2 entries).
    The request-matching loop (:522-541) is a no-op on the live path (request set always empty)."""
    eng = _eng(engine)
    d1, d2 = len(gavi.o), len(gavi.l2)
    z = np.asarray(z, dtype=np.float64)
    r1 = gavi.M @ z + gavi.N @ w + gavi.o
    m1 = eng.comp_indices(z[:d1], r1, gavi.l1, gavi.u1, tol=tol, shift=0) if d1 else np.zeros(0, np.uint8)
    s2 = gavi.A @ z + gavi.B @ w
    m2 = eng.comp_indices(s2, z[d1:], gavi.l2, gavi.u2, tol=tol, shift=4) if d2 else np.zeros(0, np.uint8)
    return np.concatenate([np.asarray(m1), np.asarray(m2)])


def masks_to_sets(mask):
    """uint8 masks -> the reference's Dict{Int,Set{Int}} (1-based rows and codes)."""
    return {i + 1: {c + 1 for c in range(8) if (int(m) >> c) & 1} for i, m in enumerate(mask)}


def node_records(Q, q, A, l, u, dec_inds):
    """The per-node GAVI of process_solution_graph (src/avi.jl:447-477) as node records (math layout):
    Qd = Q[dec, dec], R = Q[dec, param], qd = q[dec], Ad = A[:, dec], B = A[:, param]."""
    n = Q.shape[0]
    dec = list(dec_inds)
    par = [i for i in range(n) if i not in set(dec)]
    return dict(Qd=Q[np.ix_(dec, dec)], R=Q[np.ix_(dec, par)], qd=q[dec], Ad=A[:, dec], B=A[:, par], l=np.asarray(l, float),
                u=np.asarray(u, float), dec=dec, par=par)


def local_pieces(rec, K, engine=None, simplify=True):
    """local_piece (src/avi_solutions.jl:400-496) for every recipe in K [pieces, n+m] (codes 1..8 per row of z = [x_d; lambda])
    of ONE node (records from node_records): one qpn_local_pieces launch.  Returns a list of Poly over [x_d; lambda; x_p]
    (rows that pass find_non_trivial; `simplify` merges duplicate normals, the array part of src/sets.jl:255-311)."""
    from .engine import colmajor
    eng = _eng(engine)
    K = np.atleast_2d(np.asarray(K, dtype=np.uint8))
    one = lambda a: np.asarray(a, dtype=np.float64)[None]
    Ap, lp, up, keep = eng.local_pieces(colmajor(one(rec["Qd"])), colmajor(one(rec["R"])), one(rec["qd"]), colmajor(one(rec["Ad"])),
                                        colmajor(one(rec["B"])), one(rec["l"]), one(rec["u"]), K,
                                        node_of=np.zeros(K.shape[0], np.int32))
    out = []
    for t in range(K.shape[0]):
        rows = np.asarray(keep[t]).astype(bool)
        P = Poly(np.asarray(Ap[t]).T[rows], np.asarray(lp[t])[rows], np.asarray(up[t])[rows])
        out.append(_dedupe(P) if simplify else P)
    return out


def all_Ks(mask, engine=None, limit=4096):
    """all_Ks (src/avi_solutions.jl:200-215): every recipe compatible with the masks J of one solution (the Cartesian
    product of the rows' code sets), enumerated on the device.  Returns (K [count, N], total)."""
    eng = _eng(engine)
    _, total = eng.recipes_from_masks(np.asarray(mask, dtype=np.uint8), 0, 0)
    return eng.recipes_from_masks(np.asarray(mask, dtype=np.uint8), 0, min(total, limit))


def eliminate_multipliers(P: Poly, n: int, m: int, tol=1e-9, max_rows=4096) -> Poly:
    """A local piece over [x_d (n); lambda (m); x_p] brought down to [x_d; x_p]: the multipliers are eliminated through
    the piece's OWN equality rows (stationarity rows of free x_d, lambda_j = 0 of inactive rows, constraint rows of active
    ones) -- the substitution x2 = Ae_elim^+ (rhs - Ae_keep x1) of eliminate_variables, src/sets.jl:731-800, one column at a
    time with partial pivoting -- and, for a multiplier no equality pins (degenerate active sets), by one Fourier-Motzkin
    step.  The path's counterpart of project_and_permute (src/avi_solutions.jl:79-91), which the reference does by vertex
    enumeration (CDD: out of scope)."""
    A, l, u = (np.array(a, dtype=np.float64) for a in P.vectorize())
    alive = np.ones(A.shape[0], bool)
    left = []
    for j in range(n, n + m):
        col = np.where(alive & (l == u) & np.isfinite(l), np.abs(A[:, j]), 0.0)
        i = int(np.argmax(col)) if col.size else -1
        if i < 0 or col[i] <= tol:
            if np.any(alive & (np.abs(A[:, j]) > tol)):
                left.append(j)
            continue
        piv = A[i, j]
        for k in np.nonzero(alive & (np.abs(A[:, j]) > 0.0))[0]:
            if k == i:
                continue
            f = A[k, j] / piv
            A[k] -= f * A[i]; A[k, j] = 0.0
            l[k] -= f * l[i]; u[k] -= f * u[i]          # (row i is an equality: both bounds move by f b)
        alive[i] = False
    A, l, u = A[alive], l[alive], u[alive]
    for j in left:
        # Fourier-Motzkin on column j: rows as one-sided inequalities a.y <= b
        zero = np.abs(A[:, j]) <= tol
        Az, lz, uz = A[zero], l[zero], u[zero]
        ub, lb = [], []                                # a_j > 0 rows bound lambda_j from above, a_j < 0 from below
        for k in np.nonzero(~zero)[0]:
            for a_, b_ in ((A[k], u[k]), (-A[k], -l[k])):
                if not np.isfinite(b_):
                    continue
                (ub if a_[j] > 0 else lb).append((a_ / abs(a_[j]), b_ / abs(a_[j])))
        if len(ub) * len(lb) + Az.shape[0] > max_rows:
            raise RuntimeError("eliminate_multipliers: the piece needs a polyhedral projection (too many rows)")
        rows = [(au + al, bu + bl) for au, bu in ub for al, bl in lb]
        A = np.vstack([Az] + [r[0][None] for r in rows]) if rows else Az
        l = np.concatenate([lz, np.full(len(rows), -INF)]); u = np.concatenate([uz, np.array([r[1] for r in rows])])
        A[:, j] = 0.0
    keep_cols = [c for c in range(A.shape[1]) if c < n or c >= n + m]
    return Poly(A[:, keep_cols], l, u)


def solution_graph_pieces(Q, q, A, l, u, dec_inds, x, lam, engine=None, tol=1e-2, max_pieces=64):
    """The pieces of a node's solution map around (x, lam), in global x coordinates, from the DEVICE kernels:
    process_solution_graph (src/avi.jl:447-477: the node's own GAVI at z = [x_d; lam], w = x_p) -> comp_indices (the masks,
    qpn_comp_indices) -> all_Ks (src/avi_solutions.jl:200-215, qpn_recipes_from_masks) -> local_piece (:400-496,
    qpn_local_pieces) for every recipe in one launch -> the multipliers eliminated per piece (eliminate_multipliers) and the
    columns permuted back (permute!, src/avi_solutions.jl:86-87).  Covers what the first generation of LocalGAVISolutions
    (:118-130) yields; vertex exploration and remove_subsets are out of scope.  Unlike the host-only
    restatement for strictly convex leaves that preceded it (now a test-side checker, tests/strict_pieces.py) it needs
    neither Q_dd > 0 nor independent active rows."""
    x = np.asarray(x, dtype=np.float64)
    rec = node_records(np.asarray(Q, float), np.asarray(q, float), np.asarray(A, float), l, u, dec_inds)
    dec, par = rec["dec"], rec["par"]
    n, m = len(dec), len(rec["l"])
    z = np.concatenate([x[dec], np.asarray(lam, dtype=np.float64)])
    w = x[par]
    g = GAVI(np.hstack([rec["Qd"], -rec["Ad"].T]), rec["R"], rec["qd"], np.full(n, -INF), np.full(n, INF),
             np.hstack([rec["Ad"], np.zeros((m, m))]), rec["B"], rec["l"], rec["u"])
    mask = comp_indices(g, z, w, tol=tol, engine=engine)
    if np.any(np.asarray(mask) == 0):
        return []                                       # (x, lam) is not a solution of the node's GAVI at this tolerance
    K, total = all_Ks(mask, engine=engine, limit=max_pieces)
    lifted = local_pieces(rec, K, engine=engine, simplify=False)
    out = []
    for P in lifted:
        Pl = eliminate_multipliers(P, n, m)
        Al, ll, ul = Pl.vectorize()
        Ag = np.zeros((Al.shape[0], len(x)))
        Ag[:, dec] = Al[:, :n]; Ag[:, par] = Al[:, n:]
        Pg = _dedupe(Poly(Ag, ll, ul))
        if Pg.contains(x, tol=1e-5):                       # (level_batch.MEMBER_TOL)
            out.append(Pg)
    return out


_PROBES = {}


def _probe_vector(c):
    v = _PROBES.get(c)
    if v is None:
        v = _PROBES[c] = np.random.Generator(np.random.Philox(key=[97, c])).standard_normal(c)
    return v


def _dedupe(P: Poly, digits=6):
    """Merge rows with equal normals (intersection of their intervals), drop all-zero rows:
    the array part of simplify (src/sets.jl:255-311).  Works on the local form (the columns the rows touch)."""
    cols, A = P.local()
    if A.shape[0] == 0:
        return P
    # most pieces have neither: one random projection of the rows tells (equal normals project equally)
    h = A @ _probe_vector(A.shape[1])
    hs = np.sort(h)
    if A.any(axis=1).all() and not np.any(np.diff(hs) <= 1e-7 * (1.0 + np.abs(hs[1:]))):
        return P
    nz = np.nonzero(A.any(axis=1))[0]
    R = np.round(A[nz], digits) + 0.0                           # (+ 0.0: no negative zeros in the keys)
    _, first, inv = np.unique(R, axis=0, return_index=True, return_inverse=True)
    inv = np.asarray(inv).ravel()
    l = np.full(first.size, -INF); u = np.full(first.size, INF)
    np.maximum.at(l, inv, P.l[nz]); np.minimum.at(u, inv, P.u[nz])
    order = np.argsort(first)                                   # rows keep the order of their first appearance
    idx = nz[first[order]]
    return Poly.from_local(P.ncols, cols, A[idx], l[order], u[order], normalise=False)
