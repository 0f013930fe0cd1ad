"""Batched isempty / exemplar through the node-AVI path (first step) against an independent
LP feasibility answer (scipy HiGHS), on the CPU oracle engine here and on the HIP engine in the GPU suite."""
import numpy as np
import pytest
from scipy.optimize import linprog

import qpn_amd  # noqa: F401
from qpn_amd import polyhedra
from qpn_amd.programs import Poly


def _random_polys(seed, count, dmax=6, mmax=10):
    rng = np.random.default_rng(seed)
    out = []
    for t in range(count):
        d = int(rng.integers(1, dmax + 1)); m = int(rng.integers(1, mmax + 1))
        A = rng.standard_normal((m, d))
        x0 = rng.standard_normal(d)
        c = A @ x0
        l = c - np.abs(rng.standard_normal(m)) - 0.05; u = c + np.abs(rng.standard_normal(m)) + 0.05
        l = np.where(rng.random(m) < 0.3, -np.inf, l); u = np.where(rng.random(m) < 0.3, np.inf, u)
        if t % 3 == 1 and m >= 2:                      # contradictory pair: a'x <= -1 and a'x >= +1
            A[1] = A[0]; l[0], u[0] = -np.inf, c[0] - 1.0; l[1], u[1] = c[0] + 1.0, np.inf
        if t % 3 == 2 and m >= 2:                      # an equality row
            u[0] = l[0] = c[0]
        out.append((A, l, u))
    return out


def _lp_feasible(A, l, u):
    rows, rhs = [], []
    for i in range(A.shape[0]):
        if np.isfinite(u[i]): rows.append(A[i]); rhs.append(u[i])
        if np.isfinite(l[i]): rows.append(-A[i]); rhs.append(-l[i])
    if not rows:
        return True
    r = linprog(np.zeros(A.shape[1]), A_ub=np.array(rows), b_ub=np.array(rhs), bounds=[(None, None)] * A.shape[1], method="highs")
    return r.status == 0


def _check(engine, seed):
    polys = _random_polys(seed, 60)
    empty, example, status = polyhedra.exemplar_batch(polys, engine)
    assert set(np.unique(status)) <= {1, 2}
    want = np.array([not _lp_feasible(*p) for p in polys])
    assert np.array_equal(empty, want)
    assert want.any() and not want.all()
    for (A, l, u), e, x in zip(polys, empty, example):
        if not e:
            ax = A @ x
            assert np.all(ax >= l - 1e-8) and np.all(ax <= u + 1e-8) and x.shape == (A.shape[1],)
            # minimum norm: no feasible direction decreases |x| -- compare with the LP-feasible point scaled towards 0
    assert np.array_equal(polyhedra.isempty_batch(polys, engine), want)
    # Poly objects (normalised rows) give the same answers
    ep, _, _ = polyhedra.exemplar_batch([Poly(*p) for p in polys], engine)
    assert np.array_equal(ep, want)
    return empty, example


def test_isempty_and_exemplar_on_the_oracle_engine():
    from oracle_engine import OracleEngine
    _check(OracleEngine(), 5)
    e, x, s = polyhedra.exemplar_batch([], OracleEngine())
    assert e.shape == (0,) and x == []


def test_minimum_norm_point_known_answers():
    from oracle_engine import OracleEngine
    # {x in R^2 : x1 + x2 >= 2}: projection of the origin is (1, 1); the box [1, 3] x [-2, 5]: (1, 0); R^2 itself: 0
    polys = [(np.array([[1.0, 1.0]]), np.array([2.0]), np.array([np.inf])),
             (np.eye(2), np.array([1.0, -2.0]), np.array([3.0, 5.0])),
             (np.zeros((1, 2)), np.array([-np.inf]), np.array([np.inf]))]
    empty, ex, _ = polyhedra.exemplar_batch(polys, OracleEngine())
    assert not empty.any()
    assert np.allclose(ex[0], [1.0, 1.0], atol=1e-10) and np.allclose(ex[1], [1.0, 0.0], atol=1e-10) and np.allclose(ex[2], 0.0)


@pytest.mark.gpu
def test_isempty_and_exemplar_on_the_hip_engine(engine):
    from oracle_engine import OracleEngine
    e_g, x_g = _check(engine, 6)
    e_c, x_c = _check(OracleEngine(), 6)
    assert np.array_equal(e_g, e_c)
    for a, b in zip(x_g, x_c):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.max(np.abs(a - b)) <= 1e-9


def _subset_by_lp(P1, P2, tol=1e-6):
    """The reference's definition (src/sets.jl:376-407) with scipy-HiGHS as the LP solver; P1 assumed non-empty."""
    A1, l1, u1 = P1; A2, l2, u2 = P2
    rows, rhs = [], []
    for i in range(A1.shape[0]):
        if np.isfinite(u1[i]): rows.append(A1[i]); rhs.append(u1[i])
        if np.isfinite(l1[i]): rows.append(-A1[i]); rhs.append(-l1[i])
    kw = dict(A_ub=np.array(rows), b_ub=np.array(rhs)) if rows else {}
    for i in range(A2.shape[0]):
        for bound, dirn in ((l2[i], 1.0), (u2[i], -1.0)):
            if np.isfinite(bound):
                r = linprog(dirn * A2[i], bounds=[(None, None)] * A1.shape[1], method="highs", **kw)
                if r.status != 0 or r.fun < dirn * bound - tol:
                    return False
    return True


def _boxes_and_slabs(seed, count, d=3):
    rng = np.random.default_rng(seed)
    polys = []
    for t in range(count):
        c = rng.standard_normal(d); h = 0.2 + rng.random(d) * (2.0 if t % 2 else 0.6)
        G = np.eye(d) if t % 3 else np.linalg.qr(rng.standard_normal((d, d)))[0]
        l = G @ c - h; u = G @ c + h
        if t % 4 == 3: u[0] = np.inf                     # an unbounded slab
        polys.append((G, l, u))
    return polys


def test_issubset_and_remove_subsets_match_the_lp_definition():
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    polys = _boxes_and_slabs(3, 9)
    big = (np.eye(3), np.full(3, -50.0), np.full(3, 50.0))          # contains every bounded one
    polys.append(big)
    pairs = [(polys[i], polys[j]) for i in range(len(polys)) for j in range(len(polys)) if i != j]
    got = polyhedra.issubset_batch(pairs, eng)
    want = np.array([_subset_by_lp(a, b) for a, b in pairs])
    assert np.array_equal(got, want) and want.any() and not want.all()
    kept, mask = polyhedra.remove_subsets(polys, eng)
    # replay of the reference's loop on the LP answers
    k = len(polys); ref = np.zeros(k, bool)
    for i in range(k):
        if any(j != i and not ref[j] and _subset_by_lp(polys[i], polys[j]) for j in range(k)):
            ref[i] = True
    assert np.array_equal(mask, ref) and len(kept) == k - int(ref.sum()) and not mask[-1]


@pytest.mark.gpu
def test_issubset_on_the_hip_engine(engine):
    from oracle_engine import OracleEngine
    polys = _boxes_and_slabs(4, 12) + [(np.eye(3), np.full(3, -50.0), np.full(3, 50.0))]
    pairs = [(polys[i], polys[j]) for i in range(len(polys)) for j in range(len(polys)) if i != j]
    g = polyhedra.issubset_batch(pairs, engine)
    c = polyhedra.issubset_batch(pairs, OracleEngine())
    assert np.array_equal(g, c) and np.array_equal(g, np.array([_subset_by_lp(a, b) for a, b in pairs]))


# ---- F3 remainder: the reference's own exemplar rule, open bounds, implicit_bounds --------------------------------------
def _lp(c, A, l, u):
    rows, rhs = [], []
    for i in range(A.shape[0]):
        if np.isfinite(u[i]): rows.append(A[i]); rhs.append(u[i])
        if np.isfinite(l[i]): rows.append(-A[i]); rhs.append(-l[i])
    return linprog(c, A_ub=np.array(rows) if rows else None, b_ub=np.array(rhs) if rows else None,
                   bounds=[(None, None)] * A.shape[1], method="highs")


def _check_slack_and_implicit(engine, seed):
    polys = _random_polys(seed, 40, dmax=5, mmax=8)
    empty, example, eps = polyhedra.exemplar_slack_batch(polys, engine, tol=1e-4)
    for (A, l, u), e, x, ep in zip(polys, empty, example, eps):
        n, d = A.shape
        # the same LP with HiGHS: min eps s.t. A x + eps >= l, -A x + eps >= -u, eps >= -1
        A3 = np.vstack([np.hstack([A, np.ones((n, 1))]), np.hstack([-A, np.ones((n, 1))]), np.append(np.zeros(d), 1.0)[None]])
        r = _lp(np.append(np.zeros(d), 1.0), A3, np.concatenate([l, -u, [-1.0]]), np.full(2 * n + 1, np.inf))
        assert r.status == 0 and abs(r.x[-1] - ep) <= 1e-8
        assert e == (ep > 1e-4)
        if not e:
            assert np.all(A @ x >= l - 2e-4) and np.all(A @ x <= u + 2e-4)
    assert empty.any() and not empty.all()
    assert np.array_equal(polyhedra.isempty_slack_batch(polys, engine), empty)
    # implicit bounds on the non-empty ones: rows whose min and max over the set coincide
    keep = [polys[b] for b in range(len(polys)) if not empty[b] and eps[b] < -1e-3][:12]
    # add known structure: a row pinned by two others (x1 >= 1, x1 + x2 <= 1, x2 >= 0  =>  x1 = 1, x2 = 0, x1 + x2 = 1)
    pinned = (np.array([[1.0, 0.0], [1.0, 1.0], [0.0, 1.0]]), np.array([1.0, -np.inf, 0.0]), np.array([np.inf, 1.0, np.inf]))
    res = polyhedra.implicit_bounds_batch(keep + [pinned], engine)
    for (A, l, u), (eq, vals) in zip(keep + [pinned], res):
        for i in range(A.shape[0]):
            lo = _lp(A[i], A, l, u); hi = _lp(-A[i], A, l, u)
            if lo.status not in (0, 2, 3) or hi.status not in (0, 2, 3):
                continue                                    # (HiGHS gave no answer: model_status Unknown)
            # (the polyhedron is non-empty -- it has a point with margin -- so HiGHS' "infeasible" here is its presolve's
            #  "infeasible or unbounded": unbounded)
            vlo = -np.inf if lo.status in (2, 3) else lo.fun; vhi = np.inf if hi.status in (2, 3) else -hi.fun
            want = bool(abs(l[i] - u[i]) <= 1e-4) or (np.isfinite(vlo) and np.isfinite(vhi) and abs(vlo - vhi) <= 1e-4)
            assert eq[i] == want, (i, vlo, vhi)
            if eq[i] and not abs(l[i] - u[i]) <= 1e-4:
                assert abs(vals[i] - 0.5 * (vlo + vhi)) <= 1e-7
    eqp, valp = res[-1]
    assert list(eqp) == [True, True, True] and np.allclose(valp, [1.0, 1.0, 0.0], atol=1e-9)
    with pytest.raises(RuntimeError):
        polyhedra.implicit_bounds_batch([(np.array([[1.0]]), np.array([1.0]), np.array([0.0]))], engine)      # empty set


def test_exemplar_rule_open_bounds_and_implicit_bounds_on_the_oracle_engine():
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    _check_slack_and_implicit(eng, 11)
    # open bounds (src/sets.jl:68-92, :354-356): {0 < x <= 1} is non-empty, {0 < x <= 0} is empty, {0 <= x <= 0} is not;
    # x = 0 is a member of the closed set only
    P_open = Poly(np.array([[1.0]]), [0.0], [1.0], open_lo=[True])
    P_point_open = Poly(np.array([[1.0]]), [0.0], [0.0], open_lo=[True])
    P_point = Poly(np.array([[1.0]]), [0.0], [0.0])
    assert not P_open.contains(np.array([-1e-6])) and P_open.contains(np.array([0.5])) and Poly(np.array([[1.0]]), [0.0], [1.0]).contains(np.array([-1e-6]))
    e, ex, eps = polyhedra.exemplar_slack_batch([P_open, P_point_open, P_point,
                                                 Poly(np.array([[1.0, 1.0], [1.0, -1.0]]), [0.0, 0.0], [0.0, 0.0], open_hi=[True, False])],
                                                eng, tol=1e-4)
    assert list(e) == [False, True, False, True]
    assert abs(ex[0][0] - 0.5) <= 1e-9 and eps[0] == pytest.approx(-0.5)         # the slack-maximising member, not the minimum-norm one
    # a negative leading coefficient flips the row and swaps the relations with the bounds (:83-88)
    Pn = Poly(np.array([[-2.0]]), [-4.0], [2.0], open_lo=[True])
    assert np.array_equal(Pn.A, [[1.0]]) and Pn.l[0] == -1.0 and Pn.u[0] == 2.0 and not Pn.open_lo[0] and Pn.open_hi[0]


def test_implicit_bounds_far_optimum_is_not_read_as_unbounded():
    """A bounded row whose extreme lies beyond 1e6 x the scale of the bounds keeps its value (the far bound that closes an open side
    is moved out once more and the optimum has to follow it to count as unbounded); a row that is unbounded reads -inf / +inf."""
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    # x0 >= 0; x0 - 1e-7 x1 <= 0 (so x0 <= 1e-7 x1); 0 <= x1 <= 1: row "x0 + 0 x1" ... take the objective row a = (0, 1e8 ... )
    # simpler: rows  r0: x0 in [0, 1],  r1: 3e6 x0 + x1 in (-inf, inf) with x1 in [0, 1] (row r2)  -> r1 ranges over [0, 3e6 + 1]
    A = np.array([[1.0, 0.0], [3e6, 1.0], [0.0, 1.0]])
    INF = np.inf
    l = np.array([0.0, -INF, 0.0]); u = np.array([1.0, INF, 1.0])
    (eq, vals), = polyhedra.implicit_bounds_batch([(A, l, u)], eng)
    assert not eq.any()                                      # no row is an implicit equality: r1 spans [0, 3e6 + 1], finite both ways
    # pinned: with x0 = 1 fixed and x1 = 0 fixed, row r1 is the constant 3e6 -- beyond 1e6 x the bounds' scale, and finite
    l2 = np.array([1.0, -INF, 0.0]); u2 = np.array([1.0, INF, 0.0])
    (eq2, vals2), = polyhedra.implicit_bounds_batch([(A, l2, u2)], eng)
    assert eq2.tolist() == [True, True, True] and vals2[1] == pytest.approx(3e6, rel=1e-9)
    # truly unbounded: r1 free, x1 free
    A3 = np.array([[1.0, 0.0], [1.0, 1.0]]); l3 = np.array([0.0, -INF]); u3 = np.array([1.0, INF])
    (eq3, _), = polyhedra.implicit_bounds_batch([(A3, l3, u3)], eng)
    assert eq3.tolist() == [False, False]


def test_exemplar_rule_hand_checked_edges_on_the_oracle_engine():
    r"""Parity unpinned (the reference holds no fixture for `exemplar`): two edges of src/sets.jl:591-642 checked by hand.
    (1) `isapprox(l, u; atol, rtol)` at :599 is a condition on the NORM of l - u, not elementwise: four components 0.009
    apart pass an elementwise test at tol = 1e-2 but have norm 0.018 > max(atol, rtol * norm) = 0.01 -- the square-equality
    shortcut must NOT be taken (the slack LP runs: eps is reported); the same rows 0.004 apart (norm 0.008) take the shortcut
    (eps = NaN, x = A \ l).  (2) A polyhedron without a finite bound makes the slack LP unbounded (OSQP status 4, which the
    reference does not handle): eps stops at -slack_cap and the set is non-empty."""
    from oracle_engine import OracleEngine
    eng = OracleEngine()
    A = np.eye(4)
    l = np.full(4, 0.1)
    e, ex, eps = polyhedra.exemplar_slack_batch([(A, l, l + 0.009), (A, l, l + 0.004)], eng, tol=1e-2)
    assert list(e) == [False, False]
    assert np.isfinite(eps[0]) and eps[0] == pytest.approx(-0.0045, abs=1e-7)      # the LP ran: the centre of the box, slack 0.0045
    assert np.isnan(eps[1]) and np.allclose(ex[1], l)                              # the shortcut: x = A \ l
    assert polyhedra._isapprox(l, l + 0.004, 1e-2, 1e-2) and not polyhedra._isapprox(l, l + 0.009, 1e-2, 1e-2)
    # infinite entries: the norm is not finite -> Julia falls back to the component-wise rule (Inf == Inf, -Inf != Inf)
    assert polyhedra._isapprox([np.inf, 1.0], [np.inf, 1.005], 1e-2, 1e-2) and not polyhedra._isapprox([-np.inf], [np.inf], 1e-2, 1e-2)
    free = (np.array([[1.0, 2.0]]), np.array([-np.inf]), np.array([np.inf]))
    e, ex, eps = polyhedra.exemplar_slack_batch([free], eng, tol=1e-2, slack_cap=1.0)
    assert list(e) == [False] and eps[0] == pytest.approx(-1.0) and ex[0].shape == (2,)


@pytest.mark.gpu
def test_exemplar_rule_and_implicit_bounds_on_the_hip_engine(engine):
    _check_slack_and_implicit(engine, 11)
