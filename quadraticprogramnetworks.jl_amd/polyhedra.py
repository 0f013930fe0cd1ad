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

exemplar_batch / isempty_batch (minimum-norm member, closed bounds) are the fast primitives behind issubset_batch and
remove_subsets.  The reference's own decision rule -- the slack-minimising LP of `exemplar` (src/sets.jl:591-642) with its
tolerance bands and open bounds (rl / ru, :68-92, :354-356) -- is exemplar_slack_batch / isempty_slack_batch, and
implicit_bounds_batch is `implicit_bounds` (:660-713): all of them LPs over the same rows, i.e. node-AVIs with Q = 0, batched
through the same node solver (the general kernels take them: an LP's H block has no pivots for the matrix-core path).
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
    keyed = {}                                   # per polyhedron (the same object shows up in many pairs): its rows as hashable keys

    def rows_of(P):
        got = keyed.get(id(P))
        if got is None:
            A, l, u = (P.vectorize() if hasattr(P, "vectorize") else P)
            A = np.atleast_2d(np.asarray(A, dtype=np.float64))
            Ar = np.round(A, 9) + 0.0
            got = keyed[id(P)] = (A, np.asarray(l, dtype=np.float64), np.asarray(u, dtype=np.float64), [Ar[r].tobytes() for r in range(A.shape[0])], P)
        return got

    for k, (P1, P2) in enumerate(pairs):
        A1, l1, u1, k1, _ = rows_of(P1)
        A2, l2, u2, k2, _ = rows_of(P2)
        # a bound of P2 that P1 carries itself -- the same normal (to 1e-9) with a bound at least as tight -- holds on all of P1: no LP
        own = {}
        for r, key in enumerate(k1):
            own.setdefault(key, []).append(r)
        for i in range(A2.shape[0]):
            mine = own.get(k2[i], ())
            lo1 = max((l1[r] for r in mine), default=-INF); hi1 = min((u1[r] for r in mine), default=INF)
            if np.isfinite(l2[i]) and not lo1 >= l2[i] - tol:          # a violation is a point of P1 with a'x <= l2 - tol
                queries.append((np.vstack([A1, A2[i:i + 1]]), np.append(l1, -INF), np.append(u1, l2[i] - tol))); owner.append(k)
            if np.isfinite(u2[i]) and not hi1 <= u2[i] + tol:          # ... or with a'x >= u2 + tol
                queries.append((np.vstack([A1, A2[i:i + 1]]), np.append(l1, u2[i] + tol), np.append(u1, INF))); owner.append(k)
    out = np.ones(len(pairs), bool)
    if queries:
        # a query the solver neither answers with a point (SUCCESS) nor with a ray (RAY_TERM: empty) counts against the subset
        # claim, as the reference's `ret.info.status_val != 1 -> return false` does (src/sets.jl:397-398): the piece is kept
        _, _, status = exemplar_batch(queries, engine)
        for st, k in zip(status, owner):
            if st != 2:
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


LP_CHUNK_BYTES = 1 << 30          # padded input of one batched LP call (host arrays; the device copy is as large again)


def issubset_batch_chunked(pairs, engine, tol=1e-6, chunk_bytes=None):
    """issubset_batch with the batch cut into calls whose padded host arrays stay below chunk_bytes (a level of a large net
    asks millions of subset questions; one call for all of them would need their padded copies all at once)."""
    chunk_bytes = chunk_bytes or LP_CHUNK_BYTES
    out = np.ones(len(pairs), bool)
    start = 0
    while start < len(pairs):
        cost, stop = 0, start
        while stop < len(pairs):
            (A1, l1, _), (A2, l2, u2) = [(_trip(P)) for P in pairs[stop]]
            rows1, d = np.atleast_2d(A1).shape
            nq = int(np.isfinite(l2).sum() + np.isfinite(u2).sum())
            c = nq * ((rows1 + 1) * d + d * d + 4 * (rows1 + 1 + d)) * 8
            if stop > start and cost + c > chunk_bytes:
                break
            cost += c; stop += 1
        out[start:stop] = issubset_batch(pairs[start:stop], engine, tol=tol)
        start = stop
    return out


def interior_members_batch(trips, engine, delta=1e-2, chunk=20000):
    """One member per polyhedron (A, l, u), well inside its INEQUALITY rows: the slack program of `exemplar` (src/sets.jl:608-619)
    with the equality rows (l == u) kept as equalities -- a lower-dimensional piece gets a point of its relative interior instead
    of eps = 0 at an arbitrary feasible point -- and a small proximal term,

        min  eps + delta/2 (|x|^2 + eps^2)   s.t.  a_i'x = l_i (equality rows),  a_i'x + eps >= l_i,  a_i'x - eps <= u_i (the others),

    so that every query is a strictly convex node: with the equality multipliers in the free block (level_batch.free_equalities)
    the fused node kernels take it, where the plain LP -- no pivots in its H block -- fell to the general kernel, the slowest
    call of a level's sweep.  The proximal term caps the slack at 1 / delta and picks the least-norm point among the deepest ones;
    the answer is used as ONE member of the polyhedron (remove_subsets_many), never as an optimum.
    -> list of x or None (empty / no answer)."""
    out = [None] * len(trips)
    # queries over polyhedra of one size (rows, columns) are packed together, straight into the ABI's column-major blocks: a level
    # asks tens of thousands of them, a Python loop per record costs more than their solve.  The counts of equality / lower /
    # upper rows differ from piece to piece: the free block is padded with idle multipliers (a unit diagonal entry, no coupling:
    # mu = 0) and the rows with inert ones (0'z in (-inf, inf)) up to the group's largest, so that a level is a handful of calls
    groups = {}
    prepared = []
    for i, (A, l, u) in enumerate(trips):
        A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        l = np.asarray(l, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
        prepared.append((A, l, u))
        groups.setdefault((A.shape[0], A.shape[1]), []).append(i)
    for (r, d), idx_all in sorted(groups.items()):
        for c0 in range(0, len(idx_all), chunk):
            idx = idx_all[c0:c0 + chunk]
            B = len(idx)
            A = np.stack([prepared[i][0] for i in idx]).reshape(B, r, d)
            l = np.stack([prepared[i][1] for i in idx]).reshape(B, r); u = np.stack([prepared[i][2] for i in idx]).reshape(B, r)
            eq = np.isfinite(l) & (l == u)
            lo = ~eq & np.isfinite(l); hi = ~eq & np.isfinite(u)
            ne, nlo, nhi = int(eq.sum(1).max(initial=0)), int(lo.sum(1).max(initial=0)), int(hi.sum(1).max(initial=0))

            def pick(mask, cnt):                                  # first `cnt` row indices with the mask set, ascending; valid flags
                order = np.argsort(~mask, axis=1, kind="stable")[:, :cnt]
                return order, np.take_along_axis(mask, order, axis=1)

            (E, Ev), (LO, LOv), (HI, HIv) = pick(eq, ne), pick(lo, nlo), pick(hi, nhi)
            bidx = np.arange(B)[:, None]
            rows_of = lambda sel, valid: A[bidx, sel] * valid[:, :, None]        # (whole rows by index pairs: no index broadcast over d)
            nf = d + 1 + ne                                       # free block: [x; eps; mu_E]
            mi = nlo + nhi
            mp = max(16, -(-mi // 16) * 16)
            Qc = np.zeros((B, nf, nf)); qd = np.zeros((B, nf)); Ac = np.zeros((B, nf, mp))
            ll = np.full((B, mp), -INF); uu = np.full((B, mp), INF)
            ar = np.arange(d + 1)
            Qc[:, ar, ar] = delta
            qd[:, d] = 1.0
            if ne:
                AE = rows_of(E, Ev)                               # [B, ne, d], zero rows in the idle slots
                # math layout: Qd' = [[delta I, -A_E'], [A_E, 0]] (eps column of A_E is 0); Qc is its transpose per item
                Qc[:, d + 1:, :d] = -AE
                Qc[:, :d, d + 1:] = np.swapaxes(AE, 1, 2)
                je = d + 1 + np.arange(ne)
                Qc[:, je, je] = np.where(Ev, 0.0, 1.0)            # idle multipliers: 1 * mu = 0
                qd[:, d + 1:] = np.where(Ev, -np.take_along_axis(l, E, axis=1), 0.0)
            if nlo:
                Ac[:, :d, :nlo] = np.swapaxes(rows_of(LO, LOv), 1, 2); Ac[:, d, :nlo] = np.where(LOv, 1.0, 0.0)
                ll[:, :nlo] = np.where(LOv, np.take_along_axis(l, LO, axis=1), -INF)
            if nhi:
                Ac[:, :d, nlo:mi] = np.swapaxes(rows_of(HI, HIv), 1, 2); Ac[:, d, nlo:mi] = np.where(HIv, -1.0, 0.0)
                uu[:, nlo:mi] = np.where(HIv, np.take_along_axis(u, HI, axis=1), INF)
            res = engine.solve_nodes(Qc, np.zeros((B, 1, nf)), qd, Ac, np.zeros((B, 1, mp)), ll, uu, np.zeros(1))
            st = np.asarray(res["status"]); z = np.asarray(res["z"])
            ok = (st == 1) & (z[:, d] <= 1e-6)
            for k in np.nonzero(ok)[0]:
                out[idx[k]] = z[k, :d].copy()
    return out


def remove_subsets_many(lists, engine, tol=1e-6, prefilter=True):
    """`remove_subsets` (src/sets.jl:889-902) for the solution graphs of ALL nodes of a level at once (src/algorithm.jl:84
    applies it to every node's S).  Every list is first brought down to the columns its pieces touch (a large net's pieces are
    local).  The reference asks k (k - 1) subset questions per list, each one LP per finite bound of the second polyhedron
    (src/sets.jl:376-407) -- 100 000 LPs for a node with 32 pieces.  Here a question is first put to ONE point: P1 ⊆ P2 needs
    every point of P1 in P2, so a member of P1 that violates a row of P2 by more than 10 tol settles "not a subset" without
    an LP (the reference's LP over P1 would come out below the bound by the same amount).  The member is a point of P1's
    relative interior (interior_members_batch: the slack LP of `exemplar`, src/sets.jl:591-642, over the inequality rows; one LP
    per piece, all pieces of the level in one batch) -- cells of a piecewise-affine solution map that merely touch are told
    apart by it -- and only the pairs it does not settle go to the LPs (issubset_batch, chunked).  -> list of kept lists."""
    comp, jobs, where = [], [], []
    flat, flat_of = [], []
    for a, polys in enumerate(lists):
        k = len(polys) if polys is not None else 0
        if k < 2:
            comp.append(None)
            continue
        cols = np.unique(np.concatenate([P.support() for P in polys]))
        trips = [(P.block(cols), P.l, P.u) for P in polys]
        comp.append(trips)
        if prefilter:
            flat += trips; flat_of += [(a, i) for i in range(k)]
    member = {}
    if flat:
        for key, pt in zip(flat_of, interior_members_batch(flat, engine)):      # (no answer for a piece: its pairs go to the LPs)
            member[key] = pt
    sub = {}
    for a, trips in enumerate(comp):
        if trips is None:
            continue
        k = len(trips)
        have = [i for i in range(k) if member.get((a, i)) is not None]
        refuted = np.zeros((k, k), bool)                    # refuted[i, j]: P1 = piece i has a member outside P2 = piece j
        if have:
            pts = np.stack([member[(a, i)] for i in have], axis=1)          # [d, members]
            for j in range(k):
                A2, l2, u2 = trips[j]
                ax = A2 @ pts[:A2.shape[1]]
                out = np.any(ax < (l2 - 10 * tol)[:, None], axis=0) | np.any(ax > (u2 + 10 * tol)[:, None], axis=0)
                refuted[have, j] = out
        sub[a] = np.zeros((k, k), bool)                     # (refuted pairs: not a subset)
        open_ = ~refuted
        np.fill_diagonal(open_, False)
        for i, j in np.argwhere(open_).tolist():
            jobs.append((trips[i], trips[j])); where.append((a, i, j))
    res = issubset_batch_chunked(jobs, engine, tol=tol) if jobs else []
    for (a, i, j), r in zip(where, res):
        sub[a][i, j] = bool(r)
    out = []
    for a, polys in enumerate(lists):
        if comp[a] is None:
            out.append(polys)
            continue
        k = len(polys)
        is_subset = np.zeros(k, bool)
        for i in range(k):
            if np.any(sub[a][i] & ~is_subset):              # (the diagonal is never set)
                is_subset[i] = True
        out.append([P for P, s_ in zip(polys, is_subset) if not s_])
    return out


# ---- the reference's own rules on the same solver: LPs as node-AVIs with Q = 0 --------------------------------------------
def _trip(p):
    return p.vectorize() if hasattr(p, "vectorize") else p


def _open(p, n):
    if hasattr(p, "open_bounds"):
        return p.open_bounds()
    return np.zeros(n, bool), np.zeros(n, bool)


def _solve_lps(cost, A, l, u, engine):
    """Batch of LPs  min cost_b' x  s.t.  l_b <= A_b x <= u_b  (B, m, d padded arrays) through the node solver.
    -> (status [B], x [B, d], lambda [B, m])."""
    B, m, d = A.shape
    if hasattr(engine, "solve_nodes"):
        from .engine import colmajor
        res = engine.solve_nodes(colmajor(np.zeros((B, d, d))), np.zeros((B, d, 1)), cost, colmajor(A), np.zeros((B, m, 1)), l, u,
                                 np.zeros(1))
    else:
        M = np.zeros((B, d + m, d + m))
        M[:, :d, d:] = -np.swapaxes(A, 1, 2); M[:, d:, :d] = A
        lo = np.concatenate([np.full((B, d), -INF), l], axis=1); hi = np.concatenate([np.full((B, d), INF), u], axis=1)
        kind = np.concatenate([np.zeros((B, d), np.uint8), np.ones((B, m), np.uint8)], axis=1)
        res = engine.solve_avi_batch(np.swapaxes(M, 1, 2), np.concatenate([cost, np.zeros((B, m))], axis=1), lo, hi, kind=kind)
    z = np.asarray(res["z"])
    return np.asarray(res["status"]).astype(np.int32), z[:, :d], z[:, d:]


def _isapprox(x, y, atol, rtol):
    """Julia's `isapprox(x, y; atol, rtol)` on vectors: norm(x - y) <= max(atol, rtol * max(norm(x), norm(y))) -- a condition on
    the 2-norm of the difference, not elementwise; when that norm is not finite (infinite entries), the component-wise scalar rule
    `a == b || (isfinite(a) && isfinite(b) && |a - b| <= max(atol, rtol * max(|a|, |b|)))` on every pair instead."""
    x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        d = np.linalg.norm(x - y)
        if np.isfinite(d):
            return bool(d <= max(atol, rtol * max(np.linalg.norm(x), np.linalg.norm(y))))
        fin = np.isfinite(x) & np.isfinite(y)
        close = np.abs(x - y) <= np.maximum(atol, rtol * np.maximum(np.abs(x), np.abs(y)))
        return bool(np.all((x == y) | (fin & close)))


def exemplar_slack_batch(polys, engine, tol=1e-2, slack_cap=1.0, strict=True):
    """`exemplar(poly; tol)` (src/sets.jl:591-642), the reference's own emptiness rule, for a batch:
        min eps  s.t.  A x + eps >= l,  -A x + eps >= -u                       (:608-619)
        eps > tol -> empty;  eps > -tol -> empty iff an OPEN bound is active (|dual| > tol), else a member;
        eps <= -tol -> a member with slack                                     (:625-641)
    plus the square-equality shortcut x = A \\ l (:599-606).  One LP per polyhedron in variables (x, eps), all in one call of
    the node solver.  eps is capped below at -slack_cap (an unbounded LP -- OSQP's status 4, which the reference does not
    handle -- means slack without end: a member either way).  -> (empty [B] bool, example list, eps [B]).
    Parity unpinned: the reference holds no fixture for this rule; checked against HiGHS on seeded polyhedra and the hand-checked
    edge cases of tests/test_polyhedra.py (the norm-based `isapprox` of :599, the slack cap)."""
    Bn = len(polys)
    if Bn == 0:
        return np.zeros(0, bool), [], np.zeros(0)
    trips = [tuple(np.asarray(a, dtype=np.float64) for a in _trip(p)) for p in polys]
    trips = [(np.atleast_2d(A), l, u) for A, l, u in trips]
    opens = [_open(p, len(t[1])) for p, t in zip(polys, trips)]
    empty = np.zeros(Bn, bool); eps_out = np.full(Bn, np.nan); example = [None] * Bn
    todo = []
    for b, ((A, l, u), (ol, oh)) in enumerate(zip(trips, opens)):
        n, d = A.shape
        if n == 0:
            empty[b] = False; continue                                                     # :594
        if n == d and not ol.any() and not oh.any() and _isapprox(l, u, tol, tol):             # :599-606
            try:
                x = np.linalg.solve(A, l)
                ok = _isapprox(A @ x, l, tol, tol)
            except np.linalg.LinAlgError:
                ok, x = False, None
            empty[b] = not ok; example[b] = x if ok else None
            continue
        todo.append(b)
    if todo:
        dmax = max(trips[b][0].shape[1] for b in todo) + 1
        mmax = max(2 * trips[b][0].shape[0] for b in todo) + 1
        A2 = np.zeros((len(todo), mmax, dmax)); l2 = np.full((len(todo), mmax), -INF); u2 = np.full((len(todo), mmax), INF)
        cost = np.zeros((len(todo), dmax)); cost[:, dmax - 1] = 1.0
        for k, b in enumerate(todo):
            A, l, u = trips[b]
            n, d = A.shape
            A2[k, :n, :d] = A; A2[k, :n, dmax - 1] = 1.0; l2[k, :n] = l                    # A x + eps >= l
            A2[k, n:2 * n, :d] = -A; A2[k, n:2 * n, dmax - 1] = 1.0; l2[k, n:2 * n] = -u   # -A x + eps >= -u
            A2[k, mmax - 1, dmax - 1] = 1.0; l2[k, mmax - 1] = -slack_cap                  # eps >= -cap
        st, x, lam = _solve_lps(cost, A2, l2, u2, engine)
        for k, b in enumerate(todo):
            if st[k] != 1:
                if strict:
                    raise RuntimeError(f"exemplar_slack_batch: solver status {st[k]} on item {b}")
                continue                                    # (not strict: no answer for this item -- empty False, no example, eps nan)
            n, d = trips[b][0].shape
            eps = x[k, dmax - 1]; eps_out[b] = eps
            ol, oh = opens[b]
            if eps > tol:
                empty[b] = True
            elif eps > -tol:
                act_l = np.abs(lam[k, :n]) > tol; act_u = np.abs(lam[k, n:2 * n]) > tol    # :629-631
                empty[b] = bool(np.any(act_l & ol) or np.any(act_u & oh))
            if not empty[b]:
                example[b] = x[k, :d].copy()
    return empty, example, eps_out


def isempty_slack_batch(polys, engine, tol=1e-4, x=None):
    """`isempty(poly; tol, x)` (src/sets.jl:647-655) for a batch: membership of the given point first, else the exemplar rule."""
    out = np.zeros(len(polys), bool)
    rest = []
    for b, p in enumerate(polys):
        if x is not None and hasattr(p, "contains") and p.contains(np.asarray(x, dtype=np.float64)):
            continue
        rest.append(b)
    if rest:
        e, _, _ = exemplar_slack_batch([polys[b] for b in rest], engine, tol=tol)
        out[rest] = e
    return out


def implicit_bounds_batch(polys, engine, tol=1e-4):
    """`implicit_bounds(poly; tol)` (src/sets.jl:660-713) for a batch: which rows have implicitly equal lower and upper
    bounds on the polyhedron, and their values.  Per row that is not an explicit equality, the two LPs min / max a_i' x over
    the polyhedron (:676-706) -- ALL rows of ALL polyhedra in one call; an unbounded LP (RAY_TERM) gives -+inf as OSQP's
    status 4 does there.  An empty polyhedron raises "Empty set" like the reference (:688-690).
    -> list of (implicitly_equality [n] bool, vals [n])."""
    trips = [tuple(np.asarray(a, dtype=np.float64) for a in _trip(p)) for p in polys]
    trips = [(np.atleast_2d(A), l, u) for A, l, u in trips]
    empt = isempty_batch(trips, engine) if trips else np.zeros(0, bool)
    if empt.any():
        raise RuntimeError(f"Empty set (polyhedron {int(np.nonzero(empt)[0][0])})")
    jobs = []                                                   # (poly, row, sign)
    out = []
    for b, (A, l, u) in enumerate(trips):
        n = A.shape[0]
        eq = np.zeros(n, bool); vals = np.full(n, INF)
        for i in range(n - 1, -1, -1):                          # (the reference walks the rows from the last, :670)
            if np.isclose(l[i], u[i], rtol=0, atol=tol) or (l[i] == u[i]):
                eq[i] = True; vals[i] = 0.5 * (l[i] + u[i])
            else:
                jobs.append((b, i, 1.0)); jobs.append((b, i, -1.0))
        out.append([eq, vals])
    if jobs:
        dmax = max(trips[b][0].shape[1] for b, _, _ in jobs); mmax = max(trips[b][0].shape[0] for b, _, _ in jobs)
        A2 = np.zeros((len(jobs), mmax, dmax)); l2 = np.full((len(jobs), mmax), -INF); u2 = np.full((len(jobs), mmax), INF)
        cost = np.zeros((len(jobs), dmax))
        big = np.zeros(len(jobs))
        for k, (b, i, sg) in enumerate(jobs):
            A, l, u = trips[b]
            n, d = A.shape
            A2[k, :n, :d] = A; l2[k, :n] = l; u2[k, :n] = u
            cost[k, :d] = sg * A[i]
            # The objective IS row i, so the LP is unbounded only through that row's own open side.  The pivoting method does
            # not always end an unbounded degenerate LP in a ray (it may stop at a point its own post-check then rejects), so
            # that side is closed far out and an optimum AT the far bound is read as "unbounded" (the reference reads OSQP's
            # dual-infeasible status the same way, :691-693, :704-706).
            fin = np.concatenate([l[np.isfinite(l)], u[np.isfinite(u)], [1.0]])
            big[k] = 1e6 * max(1.0, float(np.max(np.abs(fin))))
            if sg > 0 and l[i] == -INF:
                l2[k, i] = -big[k]
            if sg < 0 and u[i] == INF:
                u2[k, i] = big[k]
        st, x, _ = _solve_lps(cost, A2, l2, u2, engine)
        val = [float(trips[b][0][i] @ x[k, :trips[b][0].shape[1]]) if st[k] == 1 else np.nan for k, (b, i, sg) in enumerate(jobs)]
        # an optimum AT the far bound is read as "unbounded" only if it follows the bound: those LPs are solved once more with the
        # bound ten times as far out -- an unbounded row's optimum moves with it, a bounded row whose extreme merely lies beyond
        # 1e6 x the scale keeps its value (and is reported as the finite number it is)
        def closed_here(k):                                 # was this job's open side closed artificially?
            b, i, sg = jobs[k]
            return (sg > 0 and trips[b][1][i] == -INF) or (sg < 0 and trips[b][2][i] == INF)
        # (a row whose whole range lies beyond the far bound makes the closed LP infeasible -- RAY_TERM -- although the set is
        #  not empty: such jobs are looked at again too)
        again = [k for k in range(len(jobs)) if (st[k] == 1 and abs(val[k]) >= big[k] * (1.0 - 1e-6)) or (st[k] == 2 and closed_here(k))]
        if again:
            l3 = l2[again].copy(); u3 = u2[again].copy()
            for t, k in enumerate(again):
                b, i, sg = jobs[k]
                if sg > 0: l3[t, i] = -10.0 * big[k]
                else: u3[t, i] = 10.0 * big[k]
            st3, x3, _ = _solve_lps(cost[again], A2[again], l3, u3, engine)
            for t, k in enumerate(again):
                b, i, sg = jobs[k]
                if st3[t] == 1:
                    v3 = float(trips[b][0][i] @ x3[t, :trips[b][0].shape[1]])
                    if abs(v3) < 10.0 * big[k] * (1.0 - 1e-6):
                        val[k] = v3; big[k] = INF; st[k] = 1     # bounded after all
                    else:
                        big[k] = 0.0; val[k] = 0.0; st[k] = 1    # follows the bound: unbounded
                elif st3[t] == 2:
                    big[k] = 0.0; val[k] = 0.0; st[k] = 1        # still out of reach ten times further out: read as unbounded
        ext = {}
        for k, (b, i, sg) in enumerate(jobs):
            if st[k] == 1:
                v = val[k]
                if abs(v) >= big[k] * (1.0 - 1e-6):
                    v = -INF if sg > 0 else INF                 # at the far bound: unbounded in that direction
            elif st[k] == 2:
                v = -INF if sg > 0 else INF                     # unbounded in that direction (:691-693, :704-706)
            else:
                raise RuntimeError(f"implicit_bounds_batch: solver status {st[k]} on polyhedron {b}, row {i}")
            ext[(b, i, sg)] = v
        for b, (A, l, u) in enumerate(trips):
            eq, vals = out[b]
            for i in range(A.shape[0]):
                if (b, i, 1.0) in ext:
                    lo_, hi_ = ext[(b, i, 1.0)], ext[(b, i, -1.0)]
                    eq[i] = bool(np.isfinite(lo_) and np.isfinite(hi_) and abs(lo_ - hi_) <= tol)
                    if eq[i]:
                        vals[i] = 0.5 * (hi_ + lo_)
    return [(eq, vals) for eq, vals in out]
