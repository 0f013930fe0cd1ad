"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): status identical, active-set masks bit-exact, primals within the
stated fp64 tolerance |dz| <= 1e-9 * max(1, |z|_inf); natural-map residual <= 1e-8 on every
solved item.  The oracle is the checker only (oracle/qpn_oracle.h)."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu

INF = np.inf
ZTOL = 1e-9


def _cmp(res_gpu, res_cpu, what):
    st_g = np.asarray(res_gpu["status"]); st_c = np.asarray(res_cpu["status"])
    assert np.array_equal(st_g, st_c), f"{what}: status differs at {np.nonzero(st_g != st_c)[0][:10]}"
    ok = st_c == 1
    zg = np.asarray(res_gpu["z"])[ok]; zc = np.asarray(res_cpu["z"])[ok]
    scale = np.maximum(1.0, np.max(np.abs(zc), axis=1, keepdims=True)) if zc.size else 1.0
    err = np.max(np.abs(zg - zc) / scale) if zc.size else 0.0
    assert err <= ZTOL, f"{what}: primal deviation {err:g}"
    ag = np.asarray(res_gpu["active"])[ok]; ac = np.asarray(res_cpu["active"])[ok]
    assert np.array_equal(ag, ac), f"{what}: {np.sum(ag != ac)} active-set codes differ"
    assert np.all(np.asarray(res_gpu["resid"])[ok] <= 1e-8), f"{what}: residual above 1e-8"
    assert np.array_equal(np.asarray(res_gpu["pivots"]), np.asarray(res_cpu["pivots"])), f"{what}: pivot counts differ"


def test_kat_simple_bilevel_level2(engine, oracle):
    """Hand-derived AVI known answers, SURVEY.md section 8(c)(2): config-1 level-2 AVI in the
    reference form z=[y,xi,lam,s] built per src/avi.jl:113-128, :244, :356-367."""
    from qpn_amd.engine import colmajor
    M = np.array([[0, 1, 0, 0], [2, 0, -1, 0], [1, 0, 0, -1], [0, 0, 1, 0]], float)
    l = np.array([-INF, -INF, -INF, 0.0]); u = np.full(4, INF)
    xs = [-2.0, 0.5, 0.0]
    exp = [[0, 0, 4, 0], [.5, 0, 0, .5], [0, 0, 0, 0]]
    q = np.array([[0, -2 * x, 0, 0] for x in xs], float)
    r = engine.solve_avi_batch(colmajor(M), q, np.tile(l, (3, 1)), np.tile(u, (3, 1)))
    assert list(r["status"]) == [1, 1, 1]
    assert np.allclose(r["z"], exp, atol=1e-12)
    # weakly active constraint at x = 0: codes {1,2} on the slack row (src/avi_solutions.jl:543-547)
    assert r["active"][2, 3] == 0b0011
    assert r["active"][0, 3] == 0b0001 and r["active"][1, 3] == 0b0010


@pytest.mark.parametrize("seed", [0, 1])
def test_reduced_nodes_host_path(engine, oracle, seed):
    """Ragged sizes 1 <= n <= 32, 0 <= m <= 32 (N <= 64), one launch per size class."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(seed)
    for n, m in [(1, 0), (1, 1), (3, 7), (8, 8), (16, 5), (5, 30), (32, 32), (32, 0), (20, 31)]:
        cnt = 24
        Q, R, qd, A, B, l, u = P.synth_nodes(int(rng.integers(0, 10**6)), cnt, n, m)
        w = P.shared_params()
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
        rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
        _cmp(rg, rc, f"reduced n={n} m={m}")
        assert np.all(rc["status"] == 1)


def test_box_mcp_with_warm_start(engine, oracle):
    """All-STD rows: exactly PATHSolver.solve_mcp's problem class (src/avi.jl:64), mixed
    -Inf/+Inf/fixed bounds, non-symmetric M, z0 warm start of the bounded variables."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(7)
    for N in [1, 2, 5, 17, 33, 64]:
        probs = [P.random_box_mcp(rng, N) for _ in range(16)]
        M = np.stack([p[0] for p in probs]); q = np.stack([p[1] for p in probs])
        l = np.stack([p[2] for p in probs]); u = np.stack([p[3] for p in probs]); z0 = np.stack([p[4] for p in probs])
        rc = oracle.solve_avi_batch(M, q, l, u, z0=z0)
        rg = engine.solve_avi_batch(colmajor(M), q, l, u, z0=z0)
        _cmp(rg, rc, f"box N={N}")


def test_reference_form_equals_reduced_form(engine, oracle):
    """convert(::GAVI) (src/avi.jl:113-128) of the per-node GAVI gives the same primal as the
    reduced blocks; N_ref = n + 2m <= 64 here."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(3)
    for n, m in [(4, 6), (10, 20), (16, 24), (20, 22)]:
        Q, R, qd, A, B, l, u = P.synth_nodes(int(rng.integers(0, 10**6)), 8, n, m)
        w = P.shared_params()
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
        Ms, qs, ls, us = [], [], [], []
        for i in range(8):
            Mg = np.hstack([Q[i], -A[i].T]); Ag = np.hstack([A[i], np.zeros((m, m))])
            Mr, qr, lr, ur = oracle.convert_gavi(Mg, q[i, :n], np.full(n, -INF), np.full(n, INF), Ag,
                                                 np.zeros(m), l[i], u[i])
            Ms.append(Mr); qs.append(qr); ls.append(lr); us.append(ur)
        rr = engine.solve_avi_batch(colmajor(np.stack(Ms)), np.stack(qs), np.stack(ls), np.stack(us))
        assert np.all(rr["status"] == 1) and np.all(rg["status"] == 1)
        assert np.max(np.abs(rr["z"][:, :n + m] - rg["z"])) < 1e-9
        rco = oracle.solve_avi_batch(np.stack(Ms), np.stack(qs), np.stack(ls), np.stack(us))
        _cmp(rr, rco, "reference form")


def test_infeasible_and_lp_like_nodes(engine, oracle):
    """Ray termination on infeasible nodes (status 2, as the reference's FAILURE path
    src/avi.jl:72-76) and Q = 0 nodes (robust_avoid_simple's regime)."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(11)
    Ms, qs, ls, us, ks = [], [], [], [], []
    n, m = 6, 14
    for t in range(40):
        if t % 2 == 0:   # equality rows + boxes: often infeasible
            G = rng.standard_normal((n, n)); Q = G.T @ G / n + 0.1 * np.eye(n)
            A = rng.standard_normal((m, n)); l = -1 - np.abs(rng.standard_normal(m)); u = 1 + np.abs(rng.standard_normal(m))
            eq = rng.random(m) < 0.2; u = np.where(eq, l, u)
        else:            # LP-like: Q = 0, box rows first
            Q = np.zeros((n, n))
            A = np.vstack([np.eye(n), rng.standard_normal((m - n, n))])
            l = np.concatenate([-2 * np.ones(n), -1 - np.abs(rng.standard_normal(m - n))])
            u = np.concatenate([2 * np.ones(n), 1 + np.abs(rng.standard_normal(m - n))])
        c = rng.standard_normal(n)
        M, q, lo, hi, kind = P.reduced_blocks(Q[None], np.zeros((1, n, 0)), c[None], A[None], np.zeros((1, m, 0)),
                                              l[None], u[None], np.zeros(0))
        Ms.append(M[0]); qs.append(q[0]); ls.append(lo[0]); us.append(hi[0]); ks.append(kind[0])
    M = np.stack(Ms); q = np.stack(qs); lo = np.stack(ls); hi = np.stack(us); kind = np.stack(ks)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    _cmp(rg, rc, "infeasible/LP")
    assert np.any(rc["status"] == 2) and np.any(rc["status"] == 1)
    assert np.all(rc["status"][1::2] == 1)


def test_shared_M_four_player_game(engine, oracle):
    """Config 3 shape: one Nash pool of 4 players, M identical across draws, only q differs
    (strideM = 0); reduced form N = 8 + 8 (examples/four_player_matrix_game.jl:123-157)."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(5)
    draws = 200
    M = None; qs = []
    for _ in range(draws):
        Qs, qv, _c = P.four_player_game(rng)
        H = np.zeros((8, 8)); g = np.zeros(8)
        for i in range(4):
            H[2 * i:2 * i + 2, :] = Qs[i][2 * i:2 * i + 2, :]
            g[2 * i:2 * i + 2] = qv[i][2 * i:2 * i + 2]
        Mi = np.zeros((16, 16)); Mi[:8, :8] = H; Mi[:8, 8:] = -np.eye(8); Mi[8:, :8] = np.eye(8)
        M = Mi; qs.append(np.concatenate([g, np.zeros(8)]))
    q = np.stack(qs)
    lo = np.tile(np.concatenate([np.full(8, -INF), np.full(8, -5.0)]), (draws, 1))
    hi = np.tile(np.concatenate([np.full(8, INF), np.full(8, 5.0)]), (draws, 1))
    kind = np.concatenate([np.zeros(8, np.uint8), np.ones(8, np.uint8)])
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    _cmp(rg, rc, "four-player shared M")
    assert np.all(rc["status"] == 1)


def test_device_path_config4_shape(engine, oracle):
    """Device-resident buffers (torch tensors, zero-copy), config-4 item shape n = m = 32."""
    import torch
    from qpn_amd.engine import colmajor
    cnt = 512
    Q, R, qd, A, B, l, u = P.synth_nodes(0, cnt, 32, 32)
    w = P.shared_params()
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    dev = "cuda:0"
    t = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device=dev)
    rg = engine.solve_avi_batch(t(colmajor(M)), t(q), t(lo), t(hi), kind=t(kind, torch.uint8))
    torch.cuda.synchronize()
    rg = {k: v.cpu().numpy() for k, v in rg.items()}
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    _cmp(rg, rc, "device path 32x32")
    assert np.all(rc["status"] == 1)


def _node_shaped_items(rng, cnt, n, m, dscale=1.0, general=True):
    """Explicit node-shaped items M = [[H, C], [A, D]], x rows free, the others GAVI rows with bounds: monotone (M + M' PSD),
    all four blocks full.  general: C is not -A' (a PSD part couples the two halves); dscale scales D."""
    N = n + m
    Ms = np.zeros((cnt, N, N)); q = rng.standard_normal((cnt, N))
    for i in range(cnt):
        if general:
            L = rng.standard_normal((N, N)) / np.sqrt(N)
            K = rng.standard_normal((N, N)); K = 0.3 * (K - K.T)
            M = L @ L.T + 0.5 * np.eye(N) + K
        else:
            Lh = rng.standard_normal((n, n)) / np.sqrt(n); Ld = rng.standard_normal((m, m)) / np.sqrt(m)
            A = rng.standard_normal((m, n))
            M = np.block([[Lh @ Lh.T + 0.5 * np.eye(n), -A.T], [A, dscale * (Ld @ Ld.T + 0.1 * np.eye(m))]])
        Ms[i] = M
    lo = np.concatenate([np.full((cnt, n), -INF), -rng.uniform(0.1, 1.0, (cnt, m))], axis=1)
    hi = np.concatenate([np.full((cnt, n), INF), rng.uniform(0.1, 1.0, (cnt, m))], axis=1)
    half = rng.random((cnt, m)) < 0.3
    hi[:, n:][half] = INF                                   # one-sided rows too
    kind = np.concatenate([np.zeros(n, np.uint8), np.ones(m, np.uint8)])
    return Ms, q, lo, hi, kind


@pytest.mark.parametrize("n,m", [(32, 32), (20, 28), (16, 16), (32, 9)])
def test_explicit_node_shaped_items_full_blocks(engine, oracle, n, m):
    """qpn_solve_avi_batch on explicit node-shaped items whose four blocks are all full (C != -A', D != 0): the blocks go
    through LDS once (csrc/qpn_avi_schur.hip, explicit-M load); N = 64 takes the compile-time-size instantiation."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(77 + n + m)
    M, q, lo, hi, kind = _node_shaped_items(rng, 150, n, m)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    _cmp(rg, rc, f"explicit node-shaped items {n}x{m}")
    assert np.all(rc["status"] == 1)


def test_explicit_node_shaped_large_D_is_declined_late(engine, oracle):
    """max |M| sits in the D block, which the matrix-core kernel reads only after the crash: a pivot below 1e-4 max |D| must
    still send the item to the general kernel (same result as the checker either way)."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(5)
    for dscale in (1.0, 1e5, 1e7):
        M, q, lo, hi, kind = _node_shaped_items(rng, 60, 32, 32, dscale=dscale, general=False)
        rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
        rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
        st_g = np.asarray(rg["status"]); st_c = np.asarray(rc["status"])
        # (at the largest scale a few items fail the post-check's absolute tolerance -- in the checker as well)
        assert np.array_equal(st_g, st_c) and np.mean(st_c == 1) > 0.9
        ok = st_c == 1
        zc = np.asarray(rc["z"])[ok]; zg = np.asarray(rg["z"])[ok]
        assert np.max(np.abs(zg - zc) / np.maximum(1.0, np.max(np.abs(zc), axis=1, keepdims=True))) <= ZTOL
        assert np.array_equal(np.asarray(rg["active"])[ok], np.asarray(rc["active"])[ok])


def test_mcp_csc_entry_point(engine, oracle):
    """qpn_solve_mcp_csc takes Julia's SparseMatrixCSC{Float64,Int32} arrays as-is (1-based)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2)
    M, q, l, u, z0 = P.random_box_mcp(rng, 12)
    M[np.abs(M) < 0.3] = 0.0
    M += 0.5 * np.eye(12)
    csc = sp.csc_matrix(M)
    st, z, info = engine.solve_mcp_csc(12, csc.indptr + 1, csc.indices + 1, csc.data, q, l, u, z0)
    rc = oracle.solve_avi(M, q, l, u, z0=z0)
    assert st == rc["status"] == 1
    assert np.max(np.abs(z - rc["z"])) < 1e-10
