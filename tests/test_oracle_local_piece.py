"""The oracle's local_piece (oracle/qpn_oracle.c::qpo_local_piece) against a line-by-line numpy statement of
src/avi_solutions.jl:400-441 + :491-496 (reducible_inds empty: the live caller `expand`, :246-247) on the per-node GAVI of
src/avi.jl:447-477, and a hand-checked instance: config 1's follower (min (y - x)^2 s.t. y >= 0)."""
import numpy as np

from oracle import binding as ob

INF = np.inf


def _numpy_local_piece(Qd, R, qd, Ad, B, l, u, K):
    n, m, p = Qd.shape[0], Ad.shape[0], R.shape[1]
    d1, d2 = n, m
    M = np.hstack([Qd, -Ad.T]); N = R; o = qd                               # src/avi.jl:466-468
    A = np.hstack([Ad, np.zeros((m, m))])                                   # :471
    I1 = np.hstack([np.eye(d1), np.zeros((d1, d2))]); I2 = np.hstack([np.zeros((d2, d1)), np.eye(d2)])
    Ap = np.vstack([np.hstack([M, N]), np.hstack([I2, np.zeros((d2, p))]), np.hstack([I1, np.zeros((d1, p))]), np.hstack([A, B])])
    bounds = np.zeros((d1 + d2, 4))
    for i in range(d1 + d2):                                                # :413-432
        c = K[i]
        if c == 1: bounds[i] = [-o[i], INF, -INF, -INF]
        elif c == 2: bounds[i] = [-o[i], -o[i], -INF, INF]
        elif c == 3: bounds[i] = [-INF, -o[i], INF, INF]
        elif c == 4: bounds[i] = [-INF, INF, -INF, INF]
        elif c == 5: bounds[i] = [0, INF, l[i - d1], l[i - d1]]
        elif c == 6: bounds[i] = [0, 0, l[i - d1], u[i - d1]]
        elif c == 7: bounds[i] = [-INF, 0, u[i - d1], u[i - d1]]
        else: bounds[i] = [-INF, INF, l[i - d1], u[i - d1]]
    lo = np.concatenate([bounds[:, 0], bounds[:, 2]]); hi = np.concatenate([bounds[:, 1], bounds[:, 3]])
    noisy = lo > hi
    lo[noisy] = hi[noisy]                                                   # :437-438
    Ap = np.where(np.abs(Ap) <= 1e-8, 0.0, Ap)                              # droptol!, :439
    keep = (np.isfinite(lo) | np.isfinite(hi)) & np.any(Ap != 0, axis=1)    # find_non_trivial, :384-388
    return Ap, lo, hi, keep.astype(np.uint8)


def test_oracle_local_piece_equals_the_numpy_statement():
    rng = np.random.default_rng(0)
    for n, m, p in [(3, 4, 2), (1, 0, 0), (6, 2, 0), (4, 7, 3), (9, 9, 1)]:
        for trial in range(4):
            Qd = rng.standard_normal((n, n)); R = rng.standard_normal((n, p)); qd = rng.standard_normal(n)
            Ad = rng.standard_normal((m, n)); B = rng.standard_normal((m, p))
            Ad[rng.random((m, n)) < 0.3] = 0.0                               # empty rows and tiny entries happen
            if m and n:
                Ad[0, :] = 0.0; B[0, :] = 0.0 if p else B[0, :]
            Qd[0, 0] = 5e-9
            l = -rng.random(m); u = rng.random(m)
            l[rng.random(m) < 0.3] = -INF; u[rng.random(m) < 0.3] = INF
            K = np.concatenate([rng.integers(1, 5, n), rng.integers(5, 9, m)]).astype(np.uint8)
            got = ob.local_piece(Qd, R, qd, Ad, B, l, u, K)
            want = _numpy_local_piece(Qd, R, qd, Ad, B, l, u, K)
            for a, b in zip(got, want):
                assert np.array_equal(a, b)


def test_follower_of_simple_bilevel_by_hand():
    """vars [w1, w2, x, y]; follower decides y: f = (y - x)^2 -> Qd = [2], R = Q[y, (w1,w2,x)] = [0, 0, -2], qd = [0];
    constraint y >= 0: Ad = [1], B = [0 0 0], l = 0, u = inf.  z = [y; lambda], w = [w1, w2, x].
    Recipe (2, 5) "constraint active": 2y - lambda - 2x = 0, lambda >= 0, y free (dropped), y = 0.
    Recipe (2, 6) "inactive":          2y - lambda - 2x = 0, lambda = 0,                     y >= 0."""
    Qd = np.array([[2.0]]); R = np.array([[0.0, 0.0, -2.0]]); qd = np.array([0.0])
    Ad = np.array([[1.0]]); B = np.zeros((1, 3)); l = np.array([0.0]); u = np.array([INF])
    Ap, lo, hi, keep = ob.local_piece(Qd, R, qd, Ad, B, l, u, np.array([2, 5], np.uint8))
    assert np.array_equal(Ap, np.array([[2, -1, 0, 0, -2], [0, 1, 0, 0, 0], [1, 0, 0, 0, 0], [1, 0, 0, 0, 0.0]]))
    assert np.array_equal(lo, [0, 0, -INF, 0]) and np.array_equal(hi, [0, INF, INF, 0]) and list(keep) == [1, 1, 0, 1]
    Ap, lo, hi, keep = ob.local_piece(Qd, R, qd, Ad, B, l, u, np.array([2, 6], np.uint8))
    assert np.array_equal(lo, [0, 0, -INF, 0]) and np.array_equal(hi, [0, 0, INF, INF]) and list(keep) == [1, 1, 0, 1]
