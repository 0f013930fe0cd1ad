"""Seeded random node shapes across every size class of qpn_solve_nodes -- the one-wavefront kernel (n, m <= 32, incl. the
compile-time 16 and 32 shapes), the fused workgroup kernels (33 .. 64, 65 .. 128), the blocked crash from the records
(n > 64 with m or n beyond 128) and the general route behind them -- against the oracle: status identical, active-set masks
bit-exact, pivot counts identical, primals within 1e-9 relative, natural-map residual <= 1e-8 (the bar of DESIGN.md section 2).
Mixed bound kinds (one-sided and free rows), 0 .. 5 parameters, batches of 3 .. 5 nodes: the class boundaries (32 | 33, 64 | 65,
128 | 129) and sizes that are not multiples of 16 are what this is for."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = np.inf


def _shapes(seed, count, lo, hi):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(lo, hi + 1)); m = int(rng.integers(1, hi + 1))
        if rng.random() < 0.5:
            n, m = max(n, 1), int(rng.integers(lo, hi + 1))
        out.append((n, m, int(rng.integers(0, 6))))
    return out


EDGES = [(32, 33, 2), (33, 32, 0), (64, 65, 3), (65, 64, 1), (128, 129, 2), (129, 128, 2), (16, 16, 4), (32, 32, 5), (1, 40, 1), (40, 1, 1),
         (130, 7, 2), (97, 161, 1)]


@pytest.mark.parametrize("n,m,p", EDGES + _shapes(31, 10, 1, 40) + _shapes(32, 10, 33, 70) + _shapes(33, 8, 60, 130) + _shapes(34, 4, 120, 200))
def test_random_shapes_across_size_classes(engine, oracle, n, m, p):
    from qpn_amd.engine import colmajor
    cnt = 5 if n + m <= 140 else 3
    Q, R, qd, A, B, l, u = P.synth_nodes(7000 + 13 * n + m, cnt, n, m, max(p, 1))
    rng = np.random.default_rng(n * 1000 + m)
    if p == 0:
        R = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        R = R[:, :, :p]; B = 0.2 * rng.standard_normal((cnt, m, p))
    # one-sided and free rows among the two-sided ones
    kind = rng.random(l.shape)
    l = np.where(kind < 0.15, -INF, l); u = np.where((kind > 0.85) | ((kind > 0.05) & (kind < 0.1)), INF, u)
    w = rng.standard_normal(p)
    M, q, lo, hi, kd = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kd)
    rh = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
    assert np.array_equal(rh["status"], rc["status"]), (rh["status"], rc["status"])
    ok = rc["status"] == 1
    assert ok.all()
    assert np.array_equal(rh["active"], rc["active"]) and np.array_equal(rh["pivots"], rc["pivots"])
    scale = np.maximum(1.0, np.max(np.abs(rc["z"]), axis=1, keepdims=True))
    assert np.max(np.abs(rh["z"] - rc["z"]) / scale) <= 1e-9
    assert np.max(rh["resid"]) <= 1e-8
