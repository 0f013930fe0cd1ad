"""Edge cases of the node-AVI path through the C-ABI, against the oracle: every ragged node shape of the fused
kernel, the pivot budget (MAX_ITERS) on every kernel family, the largest size the ABI takes, sizes beyond it."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _solve_both(engine, oracle, n, m, p, cnt, seed, max_pivots=0):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(seed, cnt, n, m, p)
    w = P.shared_params(p)
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    oo = oracle.default_opts(); eo = engine.default_opts()
    oo.max_pivots = max_pivots; eo.max_pivots = max_pivots
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind, opts=oo)
    rh = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w, opts=eo)
    return rc, rh


def test_every_ragged_shape_of_the_fused_kernel(engine, oracle):
    """n, m in 1..32: the fused kernel's non-FULL path (padding rows, partial tiles, short records)."""
    rng = np.random.default_rng(11)
    shapes = {(1, 1), (1, 32), (32, 1), (31, 31), (16, 16), (17, 15), (4, 29), (29, 4), (32, 31), (31, 32)}
    while len(shapes) < 40:
        shapes.add((int(rng.integers(1, 33)), int(rng.integers(1, 33))))
    for n, m in sorted(shapes):
        rc, rh = _solve_both(engine, oracle, n, m, 1 + (n + m) % 4, 12, 7000 + 40 * n + m)
        assert np.array_equal(rh["status"], rc["status"]), (n, m)
        assert np.array_equal(rh["active"], rc["active"]), (n, m)             # bit-exact masks
        assert np.array_equal(rh["pivots"], rc["pivots"]), (n, m)
        ok = rc["status"] == 1
        assert ok.any() and np.max(np.abs(rh["z"][ok] - rc["z"][ok])) <= 1e-9, (n, m)


@pytest.mark.parametrize("n,m,cnt", [(32, 32, 200), (7, 19, 60), (40, 50, 6), (100, 157, 3)])
def test_pivot_budget_is_honoured_like_the_oracle(engine, oracle, n, m, cnt):
    """max_pivots between the crash and the longest solve: some nodes finish, the others stop with MAX_ITERS (3)
    after exactly the budget -- the same nodes, the same counts, on the fused, the general and the large kernels."""
    full, _ = _solve_both(engine, oracle, n, m, 2, cnt, 8100 + n)
    budget = int(np.sort(full["pivots"])[len(full["pivots"]) // 2])          # the median solve's pivot count
    rc, rh = _solve_both(engine, oracle, n, m, 2, cnt, 8100 + n, max_pivots=budget)
    assert np.array_equal(rh["status"], rc["status"])
    assert set(np.unique(rc["status"])) == {1, 3}
    assert np.array_equal(rh["pivots"], rc["pivots"]) and rc["pivots"].max() == budget
    ok = rc["status"] == 1
    assert np.max(np.abs(rh["z"][ok] - rc["z"][ok])) <= 1e-9
    assert np.array_equal(rh["active"][ok], rc["active"][ok])


def test_largest_size_and_beyond(engine, oracle):
    """n + m = 1024 is the largest item of ABI v1 (general large-item kernel: m > 512); 1025 is QPN_ERR_SIZE, no fault."""
    from qpn_amd.engine import QpnError, colmajor
    n, m = 24, 1000
    rc, rh = _solve_both(engine, oracle, n, m, 1, 1, 9100)
    assert np.array_equal(rh["status"], rc["status"]) and rc["status"][0] == 1
    assert np.array_equal(rh["active"], rc["active"])
    assert np.max(np.abs(rh["z"] - rc["z"])) <= 1e-8 and rh["resid"][0] <= 1e-8
    Q, R, qd, A, B, l, u = P.synth_nodes(9101, 1, 25, 1000, 1)
    with pytest.raises(QpnError):
        engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, P.shared_params(1))
