"""TEST-SIDE CHECKER (not product code): the local pieces of a STRICTLY CONVEX leaf's solution map, restated on the host
with the multipliers eliminated by substitution.  It was the product's piece generator in rounds 1-2; the live loop now takes
its pieces from the device kernels (qpn_amd.avi_solutions.solution_graph_pieces) and this restatement checks them on the
regular cases it covers."""
from __future__ import annotations

import itertools

import numpy as np

from qpn_amd.avi_solutions import _dedupe
from qpn_amd.programs import Poly

INF = np.inf


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
