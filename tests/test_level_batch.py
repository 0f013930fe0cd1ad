"""The level-wide batches of the outer loop (level_batch.py: src/algorithm.jl:44-52 and src/avi.jl:382-444 over a whole
level) on CPU, arithmetic served by the oracle test double; the `-m gpu` twin is tests/test_gpu_level_batch.py.

What is pinned here:
* the LOCAL forms of the data model (Quadratic.from_local, Poly.from_local) say the same as the dense ones;
* `components` finds the level's coupling graph (independent pairs, a Nash pool, two separate Nash pairs);
* a net of many leader-follower pairs through `solve()` ends where every pair ends on its own, with a number of engine calls
  that does not depend on the number of pairs;
* `combine_at` builds the reference's set  intersection_i (S_i union complement(R_i))  -- and that this differs from the union
  of (S_i intersected with R_i) at a kink, on an instance worked by hand;
* the engine double's batched recipes / reduced pieces agree with the product's host restatement."""
import numpy as np
import pytest

import qpn_amd  # noqa: F401
from qpn_amd import algorithm, examples, level_batch
from qpn_amd.programs import Poly, QPNet, Quadratic

INF = np.inf


@pytest.fixture()
def eng():
    from oracle_engine import OracleEngine
    return OracleEngine()


def test_local_forms_equal_dense_forms():
    g = np.random.default_rng(5)
    nv = 11
    idx = [7, 2, 9]
    Ql = g.standard_normal((3, 3)); ql = g.standard_normal(3)
    f = Quadratic.from_local(nv, idx, Ql, ql, 0.5)
    Q = np.zeros((nv, nv)); Q[np.ix_(idx, idx)] = Ql
    q = np.zeros(nv); q[idx] = ql
    fd = Quadratic(Q, q, 0.5)
    assert np.array_equal(f.Q, Q) and np.array_equal(f.q, q)
    x = g.standard_normal(nv)
    assert abs(f(x) - fd(x)) < 1e-13
    rows, cols = [2, 3, 9], [0, 7, 9, 10]
    assert np.array_equal(f.block(rows, cols), Q[np.ix_(rows, cols)]) and np.array_equal(fd.block(rows, cols), Q[np.ix_(rows, cols)])
    assert np.array_equal(f.q_at(rows), q[rows])
    assert f.row_support([2]).tolist() == sorted(np.nonzero(Q[2])[0].tolist()) == fd.row_support([2]).tolist()
    import scipy.sparse as sp
    fs = Quadratic(sp.csc_matrix(Q), q, 0.5)
    assert np.array_equal(fs.Q, Q) and np.array_equal(fs.block(rows, cols), Q[np.ix_(rows, cols)])
    # rows: the leading coefficient is the one of the SMALLEST variable index, whatever order the columns were given in
    A = np.array([[0.0, -2.0, 4.0], [3.0, 0.0, 0.0]])              # over the variables (9, 2, 7)
    P = Poly.from_local(nv, [9, 2, 7], A, [-1.0, 3.0], [INF, 6.0])
    Ad = np.zeros((2, nv)); Ad[:, [9, 2, 7]] = A
    Pd = Poly(Ad, [-1.0, 3.0], [INF, 6.0])
    assert np.array_equal(P.A, Pd.A) and np.array_equal(P.l, Pd.l) and np.array_equal(P.u, Pd.u)
    assert P.support().tolist() == [2, 7, 9] == Pd.support().tolist()
    assert np.array_equal(P.block([7, 0, 2]), Pd.A[:, [7, 0, 2]])
    assert P.contains(x) == Pd.contains(x)


def test_depth_map_and_reduction_without_matrix_powers():
    """add_edges by adjacency lists gives the sets create_minimal_adj_matrix / create_depth_map give (src/programs.jl:214-269)."""
    net = QPNet(6)
    for v in range(6):
        net.add_qp(np.eye(6), np.zeros(6), [], [v])
    net.add_edges([(1, 2), (2, 3), (1, 3), (4, 3), (5, 6), (1, 6)])
    assert net.network_edges[1] == {2, 6} and net.network_edges[4] == {3} and net.reachable_nodes[1] == {2, 3, 6}
    assert net.network_depth_map == {1: {1, 4, 5}, 2: {2, 6}, 3: {3}}
    assert net.decision_inds(1) == [0, 1, 2, 5]
    with pytest.raises(ValueError):
        net.add_edges([(1, 2), (2, 3), (3, 1)])


def test_components_of_a_level():
    pairs = examples.setup("synthetic_pairs", pairs=5, n=3, m=2)
    lv1, lv2 = sorted(pairs.network_depth_map[1]), sorted(pairs.network_depth_map[2])
    assert level_batch.components(pairs, lv2) == [[i] for i in lv2]
    assert level_batch.components(pairs, lv1) == [[i] for i in lv1]
    game = examples.setup("four_player_matrix_game", seed=3)
    assert level_batch.components(game, [1, 2, 3, 4]) == [[1, 2, 3, 4]]               # one Nash pool
    # two separate Nash pairs on one level: players (1, 2) read each other, players (3, 4) read each other
    net = QPNet(4)
    for a, b in ((0, 1), (1, 0), (2, 3), (3, 2)):
        Q = np.zeros((4, 4)); Q[a, a] = 2.0; Q[a, b] = 0.5
        net.add_qp(Q, np.ones(4), [], [a])
    net.add_edges([])
    assert level_batch.components(net, [1, 2, 3, 4]) == [[1, 2], [3, 4]]


@pytest.mark.parametrize("n,m", [(3, 3), (4, 6)])
def test_pairs_through_solve_equal_pairs_alone(eng, n, m):
    """BASELINE config 4's structure at small size: every pair of the large net ends where that pair ends as a net of its own,
    and the engine is called as often for 12 pairs as for 4."""
    from oracle_engine import OracleEngine
    counts = []
    for pairs in (4, 12):
        e = OracleEngine()
        net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
        ret = algorithm.solve(net, engine=e)
        assert ret["solved"], ret
        x = ret["x_opt"]
        for k in range(pairs):
            one = algorithm.solve(examples.setup("synthetic_pairs", pairs=1, n=n, m=m, first=k), engine=OracleEngine())
            assert one["solved"]
            assert np.max(np.abs(one["x_opt"] - x[2 * n * k:2 * n * (k + 1)])) <= 1e-9, k
        counts.append({k: v for k, v in e.calls.items() if k in ("verify_nodes", "recipes_batch", "reduced_pieces")})
    # the pairs differ in how many outer iterations they need, and a level sweeps until its last node is done: compare calls
    # per outer iteration instead of totals -- at most one verify per record shape and level, whatever the number of pairs
    for c in counts:
        assert c["recipes_batch"] == c["reduced_pieces"] and c["verify_nodes"] <= 3 * c["recipes_batch"] + 3


def test_two_nash_pairs_solve_as_one_batch_of_pools(eng):
    """Multi-node components of equal shape go through ONE assemble_pools + ONE solve_avi_batch call; the result equals the
    single AVI of the whole level (what the reference forms, src/avi.jl:399-400)."""
    from qpn_amd import avi
    g = np.random.default_rng(11)
    net = QPNet(8)
    for k in range(2):                                            # pair k: players on variables (4k, 4k+1) and (4k+2, 4k+3)
        for a, b in ((0, 2), (2, 0)):
            own = [4 * k + a, 4 * k + a + 1]; oth = [4 * k + b, 4 * k + b + 1]
            G = g.standard_normal((2, 2))
            Q = np.zeros((8, 8)); Q[np.ix_(own, own)] = G.T @ G + np.eye(2); Q[np.ix_(own, oth)] = 0.3 * g.standard_normal((2, 2))
            q = np.zeros(8); q[own] = g.standard_normal(2)
            A = np.zeros((2, 8)); A[0, own[0]] = 1; A[1, own[1]] = 1
            cid = net.add_constraint(A, [-0.3, -0.3], [0.3, 0.3])
            net.add_qp(Q, q, [cid], own)
    net.add_edges([])
    assert level_batch.components(net, [1, 2, 3, 4]) == [[1, 2], [3, 4]]
    x0 = np.zeros(8)
    x = avi.solve_qep(net, [1, 2, 3, 4], x0, {}, engine=eng)
    assert eng.calls["assemble_pools"] == 1 and eng.calls["solve_avi_batch"] == 1
    # the whole level as ONE pool, assembled by the host mirror of combine_gavis
    dec = list(range(8))
    lab = {i: avi.create_labeled_gavi_from_qp(net, i, {}) for i in (1, 2, 3, 4)}
    gv = avi.combine_gavis_reduced(8, dec, [], lab)
    z, st, _ = avi.solve_gavi(gv, np.zeros(gv.M.shape[1]), np.zeros(0), engine=eng)
    assert st == avi.StatusCode.SUCCESS and np.max(np.abs(z[:8] - x)) <= 1e-10
    xr = avi.solve_qep(net, [1, 2, 3, 4], x0, {}, engine=eng, reference_form=True)
    assert np.max(np.abs(xr - x)) <= 1e-9


def test_combine_at_is_the_intersection_of_unions_not_the_union_of_intersections(eng):
    """src/qp_processing.jl:260-291 on a hand-worked kink.  Variables (x, y); the child's graph is y = max(x, 0) with pieces
    R1 = {y = 0, x <= 0}, R2 = {y = x, y >= 0}; the point is the kink (0, 0).  Suppose the parent is optimal under both pieces with
    solution sets S1 = {x <= 0, y = 0} (all of R1) and S2 = {x = 0, y = 0} (only the kink of R2).
      reference:  (S1 u R1') n (S2 u R2')  at the kink  =  {S1 n S2, S1 n R2'-pieces, R1'-pieces n S2}  without R1' n R2'
      old union:  (S1 n R1) u (S2 n R2) = S1 u {kink}
    A point of S1 away from the kink, (-1, 0), lies in R2' (y != x there), so the reference keeps it through S1 n {y < x}'s
    sibling {y > x}; the point (1, 1) of R2 \\ S2 must NOT be in the result, and neither may any piece made of complements only."""
    from qpn_amd.qp_processing import combine_at
    R1 = Poly(np.array([[0.0, 1.0], [1.0, 0.0]]), [0.0, -INF], [0.0, 0.0])
    R2 = Poly(np.array([[1.0, -1.0], [0.0, 1.0]]), [0.0, 0.0], [0.0, INF])
    S1 = Poly(np.array([[0.0, 1.0], [1.0, 0.0]]), [0.0, -INF], [0.0, 0.0])
    S2 = Poly(np.array([[1.0, 0.0], [0.0, 1.0]]), [0.0, 0.0], [0.0, 0.0])
    x = np.zeros(2)
    out = combine_at([[R1], [R2]], [[S1], [S2]], x, eng)
    assert len(out) >= 2
    inside = lambda pt: any(P.contains(np.array(pt, float), tol=1e-9) for P in out)
    assert inside((0.0, 0.0))
    assert inside((-1.0, 0.0))                    # optimal under R1, outside R2: stays in the graph
    assert not inside((1.0, 1.0))                 # in R2 but not optimal under it: must go
    assert not inside((1.0, 0.5)) and not inside((-1.0, 1.0))        # outside both regions: complement-only products are skipped
    # every piece has at least one factor from a solution set: it satisfies S1's or S2's rows
    for P in out:
        A, l, u = P.vectorize()
        assert A.shape[0] >= 2
    # one combination: the solution set itself (src/qp_processing.jl:271-272) -- process_level does not call combine_at then
    # the size guard of :281-285
    many = [[R1]] * 4
    with pytest.raises(RuntimeError):
        combine_at(many, [[S1] * 6] * 4, x, eng)


def test_remove_subsets_many_keeps_what_the_scan_keeps(eng):
    from qpn_amd.polyhedra import remove_subsets, remove_subsets_many
    box = lambda lo, hi: Poly(np.eye(2), [lo, lo], [hi, hi])
    lists = [[box(-1, 1), box(-2, 2), box(0, 3)], [box(0, 1)], [box(0, 1), box(0, 1)]]
    got = remove_subsets_many(lists, eng)
    assert [len(g) for g in got] == [2, 1, 1]
    for polys, g in zip(lists, got):
        kept, _ = remove_subsets(polys, eng)
        assert len(kept) == len(g)


def test_engine_double_pieces_agree_with_the_host_restatement(eng):
    """recipes_batch / reduced_pieces of the CPU twin against avi_solutions (all_Ks per node, local_pieces +
    eliminate_multipliers per piece) on solved random nodes, some with weakly active rows."""
    from qpn_amd import avi_solutions as AS
    from qpn_amd.engine import colmajor
    import problems as P
    n, m, p, cnt = 4, 5, 2, 6
    Q, R, qd, A, B, l, u = P.synth_nodes(100, cnt, n, m, p)
    w = np.array([0.3, -0.2])
    res = eng.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
    assert np.all(res["status"] == 1)
    masks = np.asarray(res["active"]).astype(np.uint8)
    masks[2, n + 1] |= (1 << 4) | (1 << 5)                    # make one row weakly active: two codes -> two recipes
    tot = np.array([int(np.prod([max(1, bin(int(v)).count("1")) for v in row])) for row in masks])
    off = np.concatenate([[0], np.cumsum(tot)]).astype(np.int64)
    K, node_of = eng.recipes_batch(masks, off)
    assert K.shape == (off[-1], n + m) and np.array_equal(np.bincount(node_of, minlength=cnt), tot)
    for b in range(cnt):
        Kb, tb = eng.recipes_from_masks(masks[b])
        assert tb == tot[b] and np.array_equal(Kb, K[off[b]:off[b + 1]])
    Ar, lr, ur, rows, flags = eng.reduced_pieces(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, K, node_of)
    for t in range(K.shape[0]):
        b = int(node_of[t])
        Ap, lp, up, keep = eng.local_pieces(colmajor(Q[b:b + 1]), colmajor(R[b:b + 1]), qd[b:b + 1], colmajor(A[b:b + 1]),
                                            colmajor(B[b:b + 1]), l[b:b + 1], u[b:b + 1], K[t:t + 1], node_of=np.zeros(1, np.int32))
        kp = np.asarray(keep[0]).astype(bool)
        lifted = Poly(np.asarray(Ap[0]).T[kp], np.asarray(lp[0])[kp], np.asarray(up[0])[kp], normalise=False)   # rows as the kernel made them
        if flags[t]:
            continue
        want = AS.eliminate_multipliers(lifted, n, m)
        r = int(rows[t])
        got = Poly(Ar[t, :, :r].T, lr[t, :r], ur[t, :r])
        assert got.A.shape == want.A.shape
        assert np.allclose(got.A, want.A, atol=1e-12) and np.allclose(got.l, want.l) and np.allclose(got.u, want.u)


# ---- second half of round 4: what keeps a large net's sweeps cheap ------------------------------------------------------------
def test_free_equalities_is_the_same_complementarity_system(eng):
    """A record with equality rows, solved as it is and with the equality multipliers moved into the free block
    (level_batch.free_equalities): the same x, the multipliers of the equality rows equal to the extra free variables, those of the
    other rows equal -- src/avi.jl:113-128 pairs an equality row with an unbounded multiplier."""
    from qpn_amd.engine import colmajor
    import problems as P
    g = np.random.default_rng(11)
    n, m, p, cnt = 7, 9, 3, 5
    Q, R, qd, A, B, l, u = P.synth_nodes(40, cnt, n, m, p)
    w = g.standard_normal(p)
    for b in range(cnt):
        x0 = g.standard_normal(n)
        E = g.choice(m, 3, replace=False)
        l[b, E] = u[b, E] = A[b, E] @ x0 + B[b, E] @ w          # three equality rows through a common point: feasible
    for b in range(cnt):
        rec = dict(pid=b, dec=np.arange(n), par=np.arange(p), Qd=Q[b], R=R[b], qd=qd[b], Ad=A[b], B=B[b], l=l[b], u=u[b])
        fr = level_batch.free_equalities(rec)
        eq = l[b] == u[b]
        assert fr["nx"] == n and fr["Qd"].shape == (n + 3, n + 3) and fr["Ad"].shape == (m - 3, n + 3) and not np.any(fr["l"] == fr["u"])
        one = lambda r: eng.solve_nodes(colmajor(r["Qd"][None]), colmajor(r["R"][None]), r["qd"][None], colmajor(r["Ad"][None]),
                                        colmajor(r["B"][None]), r["l"][None], r["u"][None], w)
        a, c = one(rec), one(fr)
        assert int(np.asarray(a["status"])[0]) == 1 == int(np.asarray(c["status"])[0])
        za, zc = np.asarray(a["z"])[0], np.asarray(c["z"])[0]
        assert np.max(np.abs(za[:n] - zc[:n])) <= 1e-9
        assert np.max(np.abs(za[n:][eq] - zc[n:n + 3])) <= 1e-8 and np.max(np.abs(za[n:][~eq] - zc[n + 3:])) <= 1e-8
    plain = dict(pid=0, dec=np.arange(n), par=np.arange(p), Qd=Q[0], R=R[0], qd=qd[0], Ad=A[0], B=B[0], l=l[0] - 1.0, u=u[0] + 1.0)
    assert level_batch.free_equalities(plain) is plain          # nothing to move: the record itself


def test_results_kept_across_sweeps_change_nothing(eng):
    """process_level's memo (a node whose subtree variables and children's graphs are unchanged keeps its result) against the same
    run with the memo cleared before every sweep: the same iterate, bit for bit, and fewer nodes processed."""
    seen = {"memo": 0, "plain": 0}
    orig = level_batch.solution_pieces

    def run(clear):
        def counting(qpn, recs, batches, rets, x, engine, want, **kw):
            seen["plain" if clear else "memo"] += int(sum(want))
            return orig(qpn, recs, batches, rets, x, engine, want, **kw)
        level_batch.solution_pieces = counting
        orig_pl = level_batch.process_level

        def pl(qpn, players, x, S, engine=None, exploration_vertices=0):
            if clear:
                qpn.__dict__["_process_memo"] = {}
                qpn.__dict__["_subset_memo"] = {}
            return orig_pl(qpn, players, x, S, engine=engine, exploration_vertices=exploration_vertices)
        algorithm.process_level = pl
        try:
            return algorithm.solve(examples.setup("synthetic_pairs", pairs=10, n=4, m=6), engine=eng)
        finally:
            algorithm.process_level = orig_pl
            level_batch.solution_pieces = orig
    a, b = run(False), run(True)
    assert a["solved"] and b["solved"] and np.array_equal(a["x_opt"], b["x_opt"])
    assert seen["memo"] < seen["plain"]


def test_interior_members_are_members_well_inside(eng):
    """polyhedra.interior_members_batch: the point it returns satisfies the equality rows and sits strictly inside every
    inequality row that has room; an empty polyhedron gets None; polyhedra of one size with different row kinds share a call."""
    from qpn_amd import polyhedra
    g = np.random.default_rng(3)
    d = 5
    trips = []
    for t in range(8):
        A = g.standard_normal((7, d)); x0 = g.standard_normal(d)
        s = A @ x0
        l = s - g.uniform(0.2, 1.0, 7); u = s + g.uniform(0.2, 1.0, 7)
        l[:t % 3] = u[:t % 3] = s[:t % 3]                       # 0, 1 or 2 equality rows
        l[5] = -INF
        trips.append((A, l, u))
    trips.append((np.array([[1.0, 0, 0, 0, 0], [1.0, 0, 0, 0, 0]]), np.array([1.0, -INF]), np.array([INF, 0.0])))      # x1 >= 1 and x1 <= 0
    eng.calls.clear()
    pts = polyhedra.interior_members_batch(trips, eng)
    assert pts[-1] is None
    for (A, l, u), x in zip(trips[:-1], pts[:-1]):
        ax = A @ x
        eq = l == u
        assert np.max(np.abs(ax[eq] - l[eq]), initial=0.0) <= 1e-8
        assert np.all(ax[~eq] >= l[~eq] + 0.05) and np.all(ax[~eq] <= u[~eq] - 0.05)
    assert eng.calls["solve_nodes"] == 2                          # one call per polyhedron size (7 x 5 and 2 x 5)


def test_pair_4287_a_tiny_stationarity_row_does_not_reject_its_own_point(eng):
    """Pair 4287 of the 5 000-pair net of n = m = 32 (BASELINE configs[3]'s net): one row of the follower's reduced piece comes out
    of the elimination at size 1e-4 with a 1e-8 entry and a leading coefficient of 1e-6; dropped and normalised as the reference
    normalises rows (src/sets.jl:76-89) it misses the piece's own point by 7e-3, the leader calls the point infeasible, moves the
    follower back, and solve() ends in "Cycling detected".  The rows go to unit largest coefficient first (level_batch)."""
    ret = algorithm.solve(examples.setup("synthetic_pairs", pairs=1, n=32, m=32, first=4287), engine=eng)
    assert ret["solved"], ret.get("error")
