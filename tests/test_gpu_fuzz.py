"""Differential fuzz of the batched AVI entry point against the oracle: seeded random problem classes over the whole
size range of each kernel family -- box-MCPs with mixed bound kinds and warm starts, GAVI rows anywhere (not
node-shaped), node-shaped items with equality / free / duplicated / zero constraint rows, singular and indefinite
blocks, infeasible and unbounded items.  The bar is test_gpu_avi_parity's (_cmp): status identical (whatever it is),
and on solved items masks bit-exact, |dz| <= 1e-9 (relative to max(1, |z|)), residual <= 1e-8, pivots identical."""
import numpy as np
import pytest

import problems as P
from test_gpu_avi_parity import _cmp

pytestmark = pytest.mark.gpu
INF = np.inf


def _batch(items):
    M = np.stack([i[0] for i in items]); q = np.stack([i[1] for i in items])
    lo = np.stack([i[2] for i in items]); hi = np.stack([i[3] for i in items])
    kind = np.stack([i[4] for i in items]); z0 = np.stack([i[5] for i in items])
    return M, q, lo, hi, kind, z0


def _run(engine, oracle, items, what):
    from qpn_amd.engine import colmajor
    M, q, lo, hi, kind, z0 = _batch(items)
    rc = oracle.solve_avi_batch(M, q, lo, hi, z0=z0, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, z0=z0, kind=kind)
    _cmp(rg, rc, what)
    return rc


@pytest.mark.parametrize("N", [1, 2, 3, 8, 9, 16, 17, 31, 32, 33, 48, 64, 65, 97, 130])
def test_box_mcp_all_bound_kinds(engine, oracle, N):
    rng = np.random.default_rng(100 + N)
    items = []
    for t in range(24 if N <= 64 else 4):
        M, q, l, u, z0 = P.random_box_mcp(rng, N, skew=rng.choice([0.0, 0.5, 2.0]), p_inf=rng.choice([0.0, 0.3, 1.0]),
                                          p_fix=rng.choice([0.0, 0.1, 0.5]))
        if t % 3 == 0:
            z0 = np.zeros(N)
        items.append((M, q, l, u, np.zeros(N, np.uint8), z0))
    rc = _run(engine, oracle, items, f"box-MCP N={N}")
    assert np.mean(rc["status"] == 1) > 0.5


@pytest.mark.parametrize("N", [4, 12, 24, 40, 64, 90])
def test_gavi_rows_anywhere(engine, oracle, N):
    """GAVI rows interleaved with STD rows (the reference's combined pools, src/avi.jl:305-377): the general kernels."""
    rng = np.random.default_rng(200 + N)
    items = []
    for t in range(16 if N <= 64 else 3):
        n = max(1, N // 2); m = N - n
        G = rng.standard_normal((n, n)); Q = G @ G.T / n + 0.3 * np.eye(n)
        A = rng.standard_normal((m, n))
        M = np.block([[Q, -A.T], [A, np.zeros((m, m))]])
        q = np.concatenate([rng.standard_normal(n), 0.3 * rng.standard_normal(m)])
        lo = np.concatenate([np.full(n, -INF), -np.abs(rng.standard_normal(m)) - 0.1])
        hi = np.concatenate([np.full(n, INF), np.abs(rng.standard_normal(m)) + 0.1])
        kind = np.concatenate([np.zeros(n, np.uint8), np.ones(m, np.uint8)])
        z0 = np.concatenate([rng.standard_normal(n), np.zeros(m)])
        perm = rng.permutation(N)                               # symmetric permutation: same problem, rows anywhere
        items.append((M[np.ix_(perm, perm)], q[perm], lo[perm], hi[perm], kind[perm], z0[perm]))
    rc = _run(engine, oracle, items, f"interleaved GAVI N={N}")
    assert np.all(rc["status"] == 1)


@pytest.mark.parametrize("n,m", [(5, 9), (16, 16), (32, 32), (20, 44), (70, 60)])
def test_node_shaped_items_with_awkward_rows(engine, oracle, n, m):
    """Node-shaped items (the matrix-core kernels' domain) whose rows make them decline or work hard: equality rows,
    free rows, duplicated and zero constraint rows, one-sided bounds, singular and indefinite Q."""
    rng = np.random.default_rng(300 + n + m)
    items = []
    for t in range(24 if n + m <= 64 else 4):
        G = rng.standard_normal((n, n))
        Q = G @ G.T / n + 0.2 * np.eye(n)
        mode = t % 6
        if mode == 3:
            Q = G[:, : n // 2] @ G[:, : n // 2].T / n              # singular (rank n/2)
        if mode == 4:
            Q = Q - 1.5 * np.eye(n) * np.mean(np.diag(Q))          # indefinite
        A = rng.standard_normal((m, n))
        l = -np.abs(rng.standard_normal(m)) - 0.2; u = np.abs(rng.standard_normal(m)) + 0.2
        if mode == 0:
            eq = rng.random(m) < 0.15; u = np.where(eq, l, u)      # equality rows
        if mode == 1:
            fr = rng.random(m) < 0.3; l = np.where(fr, -INF, l); u = np.where(fr, INF, u)   # free rows
            os_ = rng.random(m) < 0.3; u = np.where(os_, INF, u)                           # one-sided
        if mode == 2 and m >= 4:
            A[1] = A[0]; l[1] = l[0]; u[1] = u[0]; A[2] = 0.0      # duplicate and zero rows
        if mode == 5:
            l = l + 3.0; u = u + 3.0; A[: m // 2] = -A[m // 2: 2 * (m // 2)]   # shifted, opposing rows: often infeasible
        qd = 2 * rng.standard_normal(n)
        Mx, q, lo, hi, kind = P.reduced_blocks(Q[None], np.zeros((1, n, 0)), qd[None], A[None], np.zeros((1, m, 0)),
                                               l[None], u[None], np.zeros(0))
        items.append((Mx[0], q[0], lo[0], hi[0], kind[0], np.zeros(n + m)))
    rc = _run(engine, oracle, items, f"awkward nodes n={n} m={m}")
    assert np.any(rc["status"] == 1)


def test_large_path_random_shapes(engine, oracle):
    """Node shapes across the large-item path's internal boundaries: one / two panels per pass (2 n_pad against the
    LDS rows), odd and even panel counts, m on either side of the Lemke launches' split at (N + 1) / 2, padding in
    both blocks, 8 / 16 pending pairs."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(77)
    shapes = [(33, 33), (48, 17), (17, 48), (64, 64), (65, 64), (80, 81), (81, 80), (97, 40), (31, 130), (130, 31),
              (144, 145), (160, 100), (100, 160), (200, 90), (257, 40), (40, 257), (272, 273)]
    while len(shapes) < 24:
        n = int(rng.integers(20, 220)); m = int(rng.integers(20, 220))
        if n + m > 64:
            shapes.append((n, m))
    for n, m in shapes:
        cnt = 3 if n + m <= 300 else 2
        Q, R, qd, A, B, l, u = P.synth_nodes(9000 + 7 * n + m, cnt, n, m, 2)
        w = P.shared_params(2)
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
        rg = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
        _cmp(rg, rc, f"large path n={n} m={m}")
        assert np.all(rc["status"] == 1), (n, m)
