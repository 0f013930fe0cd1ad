"""GPU parity for the kernels around the solve: check_avi_solution (src/avi.jl:148-156),
comp_indices (src/avi_solutions.jl:511-562), per-node assembly (src/avi.jl:205-251 + :305-377)
and verify_solution (src/qp_processing.jl:57-149), each against the CPU oracle."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = np.inf


def test_assemble_nodes_bit_exact(engine, oracle):
    from qpn_amd.engine import colmajor
    # the last three shapes take the one-workgroup-per-column-strip kernel of large nodes (N > 64)
    for n, m, p in [(32, 32, 8), (5, 9, 3), (7, 0, 2), (3, 4, 0), (40, 33, 3), (97, 130, 2), (130, 31, 1)]:
        cnt = 16 if n + m <= 64 else 3
        Q, R, qd, A, B, l, u = P.synth_nodes(100, cnt, n, m, p if p else 1)
        if p == 0:
            R = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0)); w = np.zeros(0)
        else:
            B = np.random.default_rng(0).standard_normal((cnt, m, p)); w = P.shared_params(p)
        Mo, qo, lo, uo, kind = engine.assemble_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
        for i in range(cnt):
            Mr, qr, lr, ur, kr = oracle.assemble_node(Q[i], R[i], qd[i], A[i], B[i], l[i], u[i], w)
            assert np.array_equal(Mo[i].T, Mr) and np.array_equal(qo[i], qr)
            assert np.array_equal(lo[i], lr) and np.array_equal(uo[i], ur) and np.array_equal(kind[i], kr)
        # and against the plain numpy statement of the blocks
        Mn, qn, ln, un, kn = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        assert np.array_equal(np.swapaxes(Mo, 1, 2), Mn) and np.allclose(qo, qn, atol=1e-14)


def test_check_avi_solution(engine, oracle):
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(4)
    cnt, n, m = 32, 10, 14
    Q, R, qd, A, B, l, u = P.synth_nodes(7, cnt, n, m)
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, P.shared_params())
    z = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"]
    z[::2] += 1e-3 * rng.standard_normal(z[::2].shape)      # break every other item
    deg, r = engine.check_avi_batch(colmajor(M), q, lo, hi, z, kind=kind, tol=1e-6)
    for i in range(cnt):
        bad, dc, rc = oracle.check_avi_solution(M[i], q[i], lo[i], hi[i], z[i], kind=kind[i], tol=1e-6)
        assert deg[i] == dc and np.array_equal(r[i], rc)
    assert np.all(deg[1::2] == 0) and np.all(deg[::2] > 0)


def test_comp_indices_all_codes(engine, oracle):
    """Every code 1..4 incl. weakly active rows (several codes) and l == u (code 4 only)."""
    tol = 1e-2
    l = np.array([0, 0, 0, 0, 0, -INF, -INF, 1.0, 1.0, 0, 0, -INF])
    u = np.array([1, 1, 1, 1, 1, INF, 2.0, 1.0, 1.005, INF, 1, INF])
    z = np.array([0, 0, 0.5, 1, 1, 3.0, 2.0, 1.0, 1.0, 0.004, 0.5, -7.0])
    r = np.array([1, 0, 0, 0, -2, 0.0, -1.0, 5.0, -3.0, 0.005, 0.2, 0.0])
    for shift in (0, 4):
        mg = engine.comp_indices(z, r, l, u, tol=tol, shift=shift)
        mc = oracle.comp_indices(z, r, l, u, tol=tol, shift=shift)
        assert np.array_equal(mg, mc)
    m = engine.comp_indices(z, r, l, u)
    assert list(m[:5]) == [0b001, 0b011, 0b010, 0b110, 0b100]
    assert m[5] == 0b010 and m[6] == 0b100 and m[7] == 0b1000 and m[8] == 0b1000
    assert m[9] == 0b011 and m[10] == 0 and m[11] == 0b010
    rng = np.random.default_rng(9)
    cnt = 10000
    l = -np.abs(rng.standard_normal(cnt)); u = np.abs(rng.standard_normal(cnt))
    z = np.clip(rng.standard_normal(cnt), l, u); r = 0.02 * rng.standard_normal(cnt)
    assert np.array_equal(engine.comp_indices(z, r, l, u), oracle.comp_indices(z, r, l, u))


def _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd, w, what):
    from qpn_amd.engine import colmajor
    sol, lam, path = engine.verify_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, xd, w)
    cnt, n = qd.shape
    m = l.shape[1]
    for i in range(cnt):
        sc, lc, pc = oracle.verify_solution(Q[i], R[i], qd[i], A[i], B[i], l[i], u[i], xd[i], w)
        assert bool(sol[i]) == sc, f"{what}[{i}]: solution flag {sol[i]} vs {sc} (paths {path[i]} / {pc})"
        assert path[i] == pc, f"{what}[{i}]: path {path[i]} vs {pc}"
        if sc and m:
            qt = Q[i] @ xd[i] + R[i] @ w + qd[i]
            assert np.linalg.norm(A[i].T @ lam[i] - qt) <= 2e-4       # the accept test of :119 / :138
            nz = np.abs(lc) > 0
            if pc == 2 and nz.any() and np.linalg.matrix_rank(A[i][nz]) == np.sum(nz):
                assert np.max(np.abs(lam[i] - lc)) <= 1e-7, f"{what}[{i}]: duals differ"
    return sol, lam, path


def test_verify_nodes_at_solutions_and_off(engine, oracle):
    """At the node's AVI solution verify accepts (path 2); at a perturbed point it rejects."""
    cnt, n, m = 48, 12, 16
    Q, R, qd, A, B, l, u = P.synth_nodes(1000, cnt, n, m)
    w = P.shared_params()
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    z = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"]
    xd = z[:, :n].copy()
    sol, lam, path = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd, w, "at solution")
    assert np.all(sol == 1)
    assert np.max(np.abs(lam - z[:, n:])) < 1e-6        # duals agree with the AVI's multipliers
    xd2 = 0.995 * xd        # stays feasible (0 is interior), active rows stay within 1e-2, gradient off
    sol2, _, path2 = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd2, w, "shrunk")
    assert np.sum(sol2) < cnt // 2 and np.any(path2 >= 3)
    xd3 = xd + 5.0                                          # far outside: infeasible, path 0
    sol3, _, path3 = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd3, w, "infeasible")
    assert np.all(sol3 == 0) and np.all(path3 == 0)


def test_verify_worked_trace_simple_bilevel(engine, oracle):
    """leader of test/simple_bilevel.jl case w = [-2,-3] at (x,y) = (0,0).
    Piece 1 {y = 0, x <= 0}: LSQ duals [-4, 6] have a wrong sign -> bounded-LSQ fallback -> suboptimal
    (path 4).  Piece 2 {y = x, y >= 0}: duals (10, 4) -> optimal (path 2)."""
    w = np.array([-2.0, -3.0])
    Q = 2 * np.eye(2); R = -2 * np.eye(2); qd = np.zeros(2); xd = np.zeros((1, 2))
    # piece 1 rows over (x, y):  y in [0,0] (both),  x in (-inf, 0]
    A1 = np.array([[0.0, 1.0], [1.0, 0.0]]); l1 = np.array([0.0, -INF]); u1 = np.array([0.0, 0.0])
    # piece 2 rows:  y - x in [0,0]... normalised as x - y (src/sets.jl:76-89), y >= 0
    A2 = np.array([[1.0, -1.0], [0.0, 1.0]]); l2 = np.array([0.0, 0.0]); u2 = np.array([0.0, INF])
    B = np.zeros((1, 2, 2))
    s1, lam1, p1 = _verify_case(engine, oracle, Q[None], R[None], qd[None], A1[None], B, l1[None], u1[None], xd, w, "piece1")
    s2, lam2, p2 = _verify_case(engine, oracle, Q[None], R[None], qd[None], A2[None], B, l2[None], u2[None], xd, w, "piece2")
    assert s1[0] == 0 and p1[0] == 4
    assert s2[0] == 1 and p2[0] == 2
    assert np.allclose(np.abs(lam2[0]), [4.0, 10.0], atol=1e-9)


def test_verify_unconstrained_and_degenerate(engine, oracle):
    """m == 0 shortcut (:91-96) and duplicated active rows (rank-deficient A_bar)."""
    rng = np.random.default_rng(8)
    cnt, n = 8, 6
    Q, R, qd, A, B, l, u = P.synth_nodes(50, cnt, n, 0)
    w = P.shared_params()
    xs = np.stack([np.linalg.solve(Q[i], -(qd[i] + R[i] @ w)) for i in range(cnt)])
    xs[::2] += 1e-2
    sol, _, path = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xs, w, "m=0")
    assert list(sol) == [0, 1] * (cnt // 2) and np.all(path == 1)
    # duplicated rows active at the bound
    m = 4
    A = np.zeros((cnt, m, n)); A[:, 0, 0] = 1; A[:, 1, 0] = 1; A[:, 2, 1] = 1; A[:, 3, :] = rng.standard_normal((cnt, n))
    l = np.tile(np.array([0.0, 0.0, 0.0, -50.0]), (cnt, 1)); u = np.tile(np.array([INF, INF, INF, 50.0]), (cnt, 1))
    B = np.zeros((cnt, m, 8))
    xd = np.zeros((cnt, n))
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    z = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"]
    _verify_case(engine, oracle, Q, R, qd, A, B, l, u, z[:, :n].copy(), w, "duplicate rows")


def test_solve_nodes_fused_equals_assemble_then_solve(engine, oracle):
    """qpn_solve_nodes (assembly fused into the solve, M never materialised) against the oracle and
    against the two-call path, host and device buffers, matrix-core sizes and general sizes."""
    import torch
    from qpn_amd.engine import colmajor
    # ((16, 16): the compile-time n = m = 16 instantiation; 4 400 nodes: more than the resident set, the staggered launch)
    for n, m, p, cnt in [(32, 32, 8, 300), (16, 16, 8, 300), (16, 16, 4, 4400), (7, 19, 3, 40), (32, 5, 8, 20), (3, 32, 1, 20), (40, 50, 4, 3),
                         (6, 0, 2, 4)]:
        Q, R, qd, A, B, l, u = P.synth_nodes(4000 + n, cnt, n, m, p)
        B = 0.3 * np.random.default_rng(n).standard_normal((cnt, m, p))
        w = P.shared_params(p)
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
        args = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
        rh = engine.solve_nodes(*args)
        assert np.array_equal(rh["status"], rc["status"]) and np.all(rc["status"] == 1)
        assert np.array_equal(rh["active"], rc["active"]) and np.array_equal(rh["pivots"], rc["pivots"])
        assert np.max(np.abs(rh["z"] - rc["z"])) <= 1e-9 and np.max(rh["resid"]) <= 1e-8
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
        rd = engine.solve_nodes(*[t(x) for x in args])
        torch.cuda.synchronize()
        assert np.array_equal(rd["z"].cpu().numpy(), rh["z"]) and np.array_equal(rd["active"].cpu().numpy(), rh["active"])
    # an LP-like batch (Q = 0) is declined by the matrix-core kernel and solved by the gated fallback
    n, m, cnt = 6, 14, 10
    rng = np.random.default_rng(3)
    Q = np.zeros((cnt, n, n)); R = np.zeros((cnt, n, 1)); B = np.zeros((cnt, m, 1)); qd = rng.standard_normal((cnt, n))
    A = np.stack([np.vstack([np.eye(n), rng.standard_normal((m - n, n))]) for _ in range(cnt)])
    l = np.tile(np.concatenate([-2 * np.ones(n), -1.5 * np.ones(m - n)]), (cnt, 1)); u = -l
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, np.zeros(1))
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rh = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, np.zeros(1))
    assert np.array_equal(rh["status"], rc["status"]) and np.all(rc["status"] == 1)
    assert np.array_equal(rh["active"], rc["active"]) and np.max(np.abs(rh["z"] - rc["z"])) <= 1e-9


def test_solve_nodes_into_scatters_primal_blocks(engine, oracle):
    """qpn_solve_nodes_into: the primal block of every node lands in the caller's iterate (row-strided),
    for the matrix-core shape, the declined-item fallback, general sizes and host buffers; the rest of
    the iterate is left untouched (src/algorithm.jl:97-101 write-back)."""
    import torch
    from qpn_amd.engine import colmajor
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    for n, m, p, cnt in [(32, 32, 8, 200), (7, 19, 3, 40), (40, 50, 4, 3)]:
        Q, R, qd, A, B, l, u = P.synth_nodes(4100 + n, cnt, n, m, p)
        w = P.shared_params(p)
        args = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
        ref = engine.solve_nodes(*args)
        # device, rows strided (x lives inside a wider iterate)
        xbig = torch.full((cnt, n + 5), -7.0, dtype=torch.float64, device="cuda:0")
        rd = engine.solve_nodes(*[t(x) for x in args], x_out=xbig[:, 2:2 + n + 1])
        torch.cuda.synchronize()
        xb = xbig.cpu().numpy()
        assert np.array_equal(rd["z"].cpu().numpy(), ref["z"])
        assert np.array_equal(xb[:, 2:2 + n], ref["z"][:, :n])
        assert np.all(xb[:, :2] == -7.0) and np.all(xb[:, 2 + n:] == -7.0)
        # host buffers
        xh = np.full((cnt, n), np.nan)
        rh = engine.solve_nodes(*args, x_out=xh)
        assert np.array_equal(xh, rh["z"][:, :n])
    # declined items (Q = 0) go through the scan-mode fallback, which writes x too
    n, m, cnt = 6, 14, 10
    rng = np.random.default_rng(3)
    Q = np.zeros((cnt, n, n)); R = np.zeros((cnt, n, 1)); B = np.zeros((cnt, m, 1)); qd = rng.standard_normal((cnt, n))
    A = np.stack([np.vstack([np.eye(n), rng.standard_normal((m - n, n))]) for _ in range(cnt)])
    l = np.tile(np.concatenate([-2 * np.ones(n), -1.5 * np.ones(m - n)]), (cnt, 1)); u = -l
    args = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, np.zeros(1))
    x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
    rd = engine.solve_nodes(*[t(a) for a in args], x_out=x)
    torch.cuda.synchronize()
    assert np.all(rd["status"].cpu().numpy() == 1)
    assert np.array_equal(x.cpu().numpy(), rd["z"].cpu().numpy()[:, :n])


def test_abi_argument_errors_are_codes_not_crashes(engine):
    """Error behaviour at the boundary: API misuse comes back as a negative code with
    a message, never as an exception from native code or a fault; empty batches are a no-op."""
    import ctypes as C
    from qpn_amd import _lib
    lib, ctx = engine.lib, engine.ctx
    n, m, p, cnt = 4, 3, 1, 2
    Q, R, qd, A, B, l, u = P.synth_nodes(1, cnt, n, m, p)
    from qpn_amd.engine import colmajor
    arrs = [np.ascontiguousarray(a) for a in (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, P.shared_params(p))]
    ptr = lambda a: C.c_void_p(a.ctypes.data)
    z = np.zeros((cnt, n + m)); st = np.zeros(cnt, np.int32); x = np.zeros((cnt, n))
    o = _lib.AviOpts(); lib.qpn_avi_default_opts(C.byref(o))
    base = [ctx, cnt, n, m, p] + [ptr(a) for a in arrs] + [0, ptr(z), ptr(st), None, None, None, C.byref(o), 0]
    assert lib.qpn_solve_nodes_into(*base, ptr(x), n) == 0 and np.all(st == 1)
    assert lib.qpn_solve_nodes_into(*base, ptr(x), n - 1) < 0                      # stride_x < n
    assert b"stride_x" in lib.qpn_ctx_last_error(ctx)
    bad = list(base); bad[1] = -1
    assert lib.qpn_solve_nodes_into(*bad, None, 0) < 0                              # negative batch
    bad = list(base); bad[5] = None
    assert lib.qpn_solve_nodes_into(*bad, None, 0) < 0                              # null Qd
    bad = list(base); bad[1] = 0
    assert lib.qpn_solve_nodes_into(*bad, None, 0) == 0                             # empty batch: no-op
    assert lib.qpn_order_nodes_by_pivots(ctx, None, 5, 0) < 0
    assert lib.qpn_order_nodes_by_pivots(ctx, ptr(st), 0, 0) < 0
    assert lib.qpn_set_node_order(ctx, ptr(st), -3, 0) < 0
    assert lib.qpn_set_node_order(ctx, None, 0, 0) == 0                             # clearing is always fine
    assert lib.qpn_solve_nodes_into(None, *base[1:], None, 0) < 0                   # no context
    # route options: unknown option / value out of range are argument errors, valid settings are accepted and idempotent
    assert lib.qpn_ctx_set_option(ctx, 99, 0) < 0 and b"unknown option" in lib.qpn_ctx_last_error(ctx)
    assert lib.qpn_ctx_set_option(ctx, _lib.OPT_MID_ROUTE, 2) < 0 and lib.qpn_ctx_set_option(ctx, _lib.OPT_BIG_ROUTE, 0) < 0
    assert lib.qpn_ctx_set_option(ctx, _lib.OPT_BIG_ROUTE, -1) < 0 and lib.qpn_ctx_set_option(None, _lib.OPT_BIG_ROUTE, 1) < 0
    for opt, val in ((_lib.OPT_MID_ROUTE, 1), (_lib.OPT_BIG_ROUTE, 1)):
        assert lib.qpn_ctx_set_option(ctx, opt, val) == 0 and lib.qpn_ctx_set_option(ctx, opt, val) == 0


@pytest.mark.parametrize("n,m,cnt", [(70, 40, 6), (40, 90, 6), (100, 130, 4), (256, 256, 3)])
def test_verify_wide_nodes(engine, oracle, n, m, cnt):
    """Row A8 beyond n, m <= 64 (one workgroup per node, csrc/qpn_verify.hip::verify_wide_stage1/2; config 5's nodes are
    n = m = 256): at the AVI solution (least-squares duals accepted, path 2), at a shrunk point (bounded-LSQ fallback on the
    large-item AVI kernel, paths 3/4) and far outside (infeasible, path 0) -- flags and paths equal to the oracle's, duals
    within 1e-7 where the active rows are independent."""
    Q, R, qd, A, B, l, u = P.synth_nodes(3000 + n, cnt, n, m)
    w = P.shared_params()
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    z = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"]
    xd = z[:, :n].copy()
    sol, lam, path = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd, w, f"wide {n}x{m} at solution")
    assert np.all(sol == 1) and np.all(path == 2)
    assert np.max(np.abs(lam - z[:, n:])) < 1e-6
    xd2 = 0.995 * xd
    sol2, _, path2 = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd2, w, f"wide {n}x{m} shrunk")
    assert np.any(path2 >= 3)
    sol3, _, path3 = _verify_case(engine, oracle, Q, R, qd, A, B, l, u, xd + 5.0, w, f"wide {n}x{m} infeasible")
    assert np.all(sol3 == 0) and np.all(path3 == 0)
    # resident-records route gives the same
    from qpn_amd.engine import colmajor
    nodes = engine.upload_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u)
    s4, l4, p4 = nodes.verify(xd2, w)
    assert np.array_equal(s4, sol2) and np.array_equal(p4, path2)
    nodes.close()
