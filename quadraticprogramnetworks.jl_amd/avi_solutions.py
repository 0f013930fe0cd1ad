"""Host-side mirror of the array part of src/avi_solutions.jl: comp_indices (the "active set")
and -- scope row F1, first slice -- the local pieces of a strictly convex node's solution map.

comp_indices masks: bit (c-1) set for code c.  Codes (src/avi_solutions.jl:390-399, :568-586):
  1  r >= 0, z = l      2  r = 0, l <= z <= u      3  r <= 0, z = u      4  l = z = u
  5..8 the same for the second GAVI condition (s = Az+Bw against l2,u2 with multiplier z2).
"""
from __future__ import annotations

import itertools

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


def local_pieces_strict(Q, q, A, l, u, dec_inds, x, lam, tol=1e-2, max_pieces=64):
    """Local pieces of a node's solution map around (x, lam), in global x coordinates, for a
    node whose Q[dec,dec] is positive definite and whose active rows are linearly independent.

    Restates, for that regular case only, process_solution_graph (src/avi.jl:447-477) ->
    comp_indices -> all_Ks (src/avi_solutions.jl:200-215) -> local_piece (:400-496) -> project:
    for every recipe K (each weakly active row taken as active OR inactive) the piece is
        { x :  Q_dd x_d + Q_dp x_p + q_d = A_act,d' lam_act            (stationarity)
               A_act x = bound_act,  sign(lam_act) ok,  l <= A_inact x <= u }
    and lam_act is eliminated by substitution (no polyhedral projection / CDD needed).
    Returns a list of Poly over all variables."""
    n = len(x)
    dec = list(dec_inds)
    par = [i for i in range(n) if i not in set(dec)]
    Qdd = Q[np.ix_(dec, dec)]
    ax = A @ x
    m = A.shape[0]
    options = []
    for i in range(m):
        at_l = np.isfinite(l[i]) and abs(ax[i] - l[i]) <= tol
        at_u = np.isfinite(u[i]) and abs(ax[i] - u[i]) <= tol
        lam_zero = abs(lam[i]) <= tol
        opts = []
        if l[i] == u[i]:
            opts = ["eq"]
        else:
            if at_l and lam[i] >= -tol:
                opts.append("lo")
            if at_u and lam[i] <= tol:
                opts.append("up")
            if lam_zero:
                opts.append("in")
        if not opts:
            opts = ["in"]
        options.append(opts)
    pieces = []
    for rec in itertools.islice(itertools.product(*options), max_pieces):
        act = [i for i in range(m) if rec[i] != "in"]
        rows_A, rows_l, rows_u = [], [], []
        Ad = A[np.ix_(act, dec)] if act else np.zeros((0, len(dec)))
        # stationarity with lam eliminated: lam_act = (Ad Qdd^-1 Ad')^-1 (Ad Qdd^-1 g(x) + ...)
        # g(x) = Q_d,: x + q_d  (gradient rows);  Ad' lam = g  =>  project g on range(Ad'):
        Grow = Q[dec, :]                      # gradient is affine in the full x
        if act:
            try:
                W = np.linalg.solve(Qdd, Ad.T)            # Qdd^-1 Ad'
                S = Ad @ W
                Sinv = np.linalg.inv(S)
            except np.linalg.LinAlgError:
                continue
            # x_d is pinned by: A_act x = b_act and Qdd-stationarity in the null space of Ad.
            # lam(x) = Sinv (W' (Grow x + q_d))  evaluated with x_d free  -> affine in x
            Lx = Sinv @ (W.T @ Grow); Lc = Sinv @ (W.T @ q[dec])
            # stationarity residual: Grow x + q_d - Ad' lam(x) = 0   (len(dec) equalities)
            E = Grow - Ad.T @ Lx; ec = q[dec] - Ad.T @ Lc
        else:
            Lx = np.zeros((0, n)); Lc = np.zeros(0)
            E = Grow; ec = q[dec].copy()
        for k in range(E.shape[0]):
            if np.max(np.abs(E[k])) > 1e-12:
                rows_A.append(E[k]); rows_l.append(-ec[k]); rows_u.append(-ec[k])
        for j, i in enumerate(act):
            b = l[i] if rec[i] in ("lo", "eq") else u[i]
            rows_A.append(A[i]); rows_l.append(b); rows_u.append(b)
            if rec[i] == "lo":
                rows_A.append(Lx[j]); rows_l.append(-Lc[j]); rows_u.append(INF)
            elif rec[i] == "up":
                rows_A.append(Lx[j]); rows_l.append(-INF); rows_u.append(-Lc[j])
        for i in range(m):
            if rec[i] == "in":
                rows_A.append(A[i]); rows_l.append(l[i]); rows_u.append(u[i])
        P = Poly(np.array(rows_A).reshape(-1, n), np.array(rows_l), np.array(rows_u))
        P = _dedupe(P)
        if P.contains(x, tol=10 * tol):
            pieces.append(P)
    return pieces


def _dedupe(P: Poly, digits=6):
    """Merge rows with equal normals (intersection of their intervals), drop all-zero rows:
    the array part of simplify (src/sets.jl:255-311)."""
    A, l, u = P.vectorize()
    keep = {}
    for i in range(A.shape[0]):
        if not np.any(A[i]):
            continue
        key = tuple(np.round(A[i], digits))
        if key in keep:
            j = keep[key]
            l[j] = max(l[j], l[i]); u[j] = min(u[j], u[i])
        else:
            keep[key] = i
    idx = sorted(keep.values())
    return Poly(A[idx], l[idx], u[idx], normalise=False)
