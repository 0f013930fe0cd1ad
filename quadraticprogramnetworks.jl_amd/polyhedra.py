"""Batched polyhedral primitives on the node-AVI path (first step).

The reference answers `isempty(poly)` / `exemplar(poly)` (src/sets.jl:591-655) with one OSQP LP per polyhedron --
thousands of tiny independent solves inside `remove_subsets` (:889-902) and the intersection tree
(src/intersection.jl:66-105).  Here a whole batch of closed polyhedra {x : l <= A x <= u} goes through ONE call of
the node solver: the projection of the origin,

        min 1/2 |x|^2   s.t.  l <= A x <= u,

is a strictly convex QP whose KKT system is exactly a node's reduced AVI with Q = I, q = 0 (src/avi.jl:205-251), so
    status SUCCESS   <=>  the polyhedron is non-empty, and x is its minimum-norm point (an exemplar),
    status RAY_TERM  <=>  it is empty (the feasibility LP has no solution: a secondary ray).
Polyhedra of different sizes share a batch: missing rows are padded with 0'x in (-inf, inf), missing variables with
unconstrained ones (their minimum-norm value is 0).

Differences from the reference, by design of this first step: bounds are closed (the open-bound flags rl / ru of
src/sets.jl:68-92 are not modelled), the exemplar is the minimum-norm point rather than OSQP's slack-maximising one
(both are members; `isempty` agrees wherever the answer does not hinge on the 1e-4 slack of :585 / :637).
"""
from __future__ import annotations

import numpy as np

INF = np.inf


def exemplar_batch(polys, engine):
    """-> (empty [B] bool, example [B] list of x or None, status [B] int32).

    `polys`: sequence of objects with `vectorize() -> (A, l, u)` (programs.Poly) or (A, l, u) triples.
    `engine`: anything with `solve_nodes` (the HIP engine: fused assembly + solve) or `solve_avi_batch`."""
    trip = [p.vectorize() if hasattr(p, "vectorize") else p for p in polys]
    B = len(trip)
    if B == 0:
        return np.zeros(0, bool), [], np.zeros(0, np.int32)
    dims = [np.atleast_2d(t[0]).shape for t in trip]
    d = max(1, max(s[1] for s in dims))
    m = max(1, max(s[0] for s in dims))
    A = np.zeros((B, m, d)); l = np.full((B, m), -INF); u = np.full((B, m), INF)
    for b, (Ab, lb, ub) in enumerate(trip):
        Ab = np.atleast_2d(np.asarray(Ab, dtype=np.float64))
        r, c = Ab.shape
        A[b, :r, :c] = Ab
        l[b, :r] = np.asarray(lb, dtype=np.float64); u[b, :r] = np.asarray(ub, dtype=np.float64)
    Q = np.broadcast_to(np.eye(d), (B, d, d)).copy()
    qd = np.zeros((B, d))
    if hasattr(engine, "solve_nodes"):
        from .engine import colmajor
        res = engine.solve_nodes(colmajor(Q), np.zeros((B, d, 1)), qd, colmajor(A), np.zeros((B, m, 1)), l, u, np.zeros(1))
    else:
        M = np.zeros((B, d + m, d + m))
        M[:, :d, :d] = Q; M[:, :d, d:] = -np.swapaxes(A, 1, 2); M[:, d:, :d] = A
        lo = np.concatenate([np.full((B, d), -INF), l], axis=1); hi = np.concatenate([np.full((B, d), INF), u], axis=1)
        kind = np.concatenate([np.zeros((B, d), np.uint8), np.ones((B, m), np.uint8)], axis=1)
        res = engine.solve_avi_batch(np.swapaxes(M, 1, 2), np.zeros((B, d + m)), lo, hi, kind=kind)
    status = np.asarray(res["status"]).astype(np.int32)
    z = np.asarray(res["z"])
    empty = status != 1
    example = [None if empty[b] else z[b, : dims[b][1]].copy() for b in range(B)]
    return empty, example, status


def isempty_batch(polys, engine):
    """`isempty(poly)` (src/sets.jl:649-655) for a whole batch: True where {x : l <= A x <= u} has no point.
    Raises if the solver ended in anything but SUCCESS / RAY_TERM for an item (its answer would be a guess)."""
    empty, _, status = exemplar_batch(polys, engine)
    bad = np.nonzero((status != 1) & (status != 2))[0]
    if bad.size:
        raise RuntimeError(f"isempty_batch: solver status {status[bad[0]]} on item {int(bad[0])}")
    return empty
