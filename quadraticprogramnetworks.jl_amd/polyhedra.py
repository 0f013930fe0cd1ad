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


def issubset_batch(pairs, engine, tol=1e-6):
    """`P1 ⊆ P2` (src/sets.jl:376-407) for a batch of pairs -> bool [len(pairs)].

    The reference minimises +-a'x over P1 for every finite bound of P2 (one OSQP LP each) and answers false when a
    minimum falls below the bound by more than `tol` or the LP is unbounded.  Equivalently: P1 ⊆ P2 iff, for every
    finite bound of P2, P1 intersected with the closed half-space beyond that bound (moved out by `tol`) is EMPTY --
    one emptiness query per bound, all pairs and bounds in one `isempty_batch` call.  An empty P1 is a subset of
    anything (the reference's LP is infeasible there and it answers false; noted, not mirrored)."""
    queries, owner = [], []
    for k, (P1, P2) in enumerate(pairs):
        A1, l1, u1 = (P1.vectorize() if hasattr(P1, "vectorize") else P1)
        A2, l2, u2 = (P2.vectorize() if hasattr(P2, "vectorize") else P2)
        A1 = np.atleast_2d(np.asarray(A1, dtype=np.float64)); A2 = np.atleast_2d(np.asarray(A2, dtype=np.float64))
        for i in range(A2.shape[0]):
            if np.isfinite(l2[i]):          # a violation is a point of P1 with a'x <= l2 - tol
                queries.append((np.vstack([A1, A2[i:i + 1]]), np.append(l1, -INF), np.append(u1, l2[i] - tol))); owner.append(k)
            if np.isfinite(u2[i]):          # ... or with a'x >= u2 + tol
                queries.append((np.vstack([A1, A2[i:i + 1]]), np.append(l1, u2[i] + tol), np.append(u1, INF))); owner.append(k)
    out = np.ones(len(pairs), bool)
    if queries:
        empty = isempty_batch(queries, engine)
        for e, k in zip(empty, owner):
            if not e:
                out[k] = False
    return out


def remove_subsets(polys, engine, tol=1e-6):
    """`remove_subsets(pu::PolyUnion)` (src/sets.jl:889-902): drop every polyhedron that is a subset of another one
    still kept, scanning in order like the reference; all k (k - 1) subset tests run as one batch first.
    -> (kept polys, is_subset mask)."""
    k = len(polys)
    idx = [(i, j) for i in range(k) for j in range(k) if i != j]
    sub = np.zeros((k, k), bool)
    if idx:
        res = issubset_batch([(polys[i], polys[j]) for i, j in idx], engine, tol=tol)
        for (i, j), r in zip(idx, res):
            sub[i, j] = r
    is_subset = np.zeros(k, bool)
    for i in range(k):
        if any(j != i and not is_subset[j] and sub[i, j] for j in range(k)):
            is_subset[i] = True
    return [p for p, s in zip(polys, is_subset) if not s], is_subset
