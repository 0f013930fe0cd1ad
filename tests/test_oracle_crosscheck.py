"""The oracle against INDEPENDENT solvers (different algorithms): semismooth Newton on the natural
map, brute-force active-set enumeration, scipy's LP solver; plus structural identities of the
restated assembly (convert, reduced vs reference form)."""
import numpy as np
import pytest

import problems as P
import pyref

INF = np.inf


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reduced_nodes_vs_newton(oracle, seed):
    rng = np.random.default_rng(seed)
    converged = 0
    for _ in range(60):
        n, m = int(rng.integers(1, 33)), int(rng.integers(0, 33))
        Q, R, qd, A, B, l, u = P.synth_node(int(rng.integers(0, 10**6)), n, m)
        M, q, lo, hi, kind = (a[0] for a in P.reduced_blocks(Q[None], R[None], qd[None], A[None], B[None], l[None], u[None], P.shared_params()))
        r = oracle.solve_avi(M, q, lo, hi, kind=kind)
        zn, resn = pyref.solve_newton(M, q, lo, hi, kind)
        assert r["status"] == 1 and r["resid"] <= 1e-8
        if resn < 1e-9:          # the independent Newton iteration converged: answers must agree
            converged += 1
            assert np.max(np.abs(r["z"] - zn)) < 1e-8
    assert converged >= 50


def test_box_mcp_vs_newton_and_enumeration(oracle):
    rng = np.random.default_rng(5)
    for _ in range(80):
        N = int(rng.integers(1, 40))
        M, q, l, u, z0 = P.random_box_mcp(rng, N)
        r = oracle.solve_avi(M, q, l, u, z0=z0)
        zn, resn = pyref.solve_newton(M, q, l, u)
        assert r["status"] == 1 and r["resid"] <= 1e-8
        if resn < 1e-9:
            assert np.max(np.abs(r["z"] - zn)) < 1e-8
    for _ in range(25):
        N = int(rng.integers(1, 7))
        M, q, l, u, z0 = P.random_box_mcp(rng, N)
        sols = pyref.solve_enumerate(M, q, l, u)
        r = oracle.solve_avi(M, q, l, u, z0=z0)
        assert len(sols) == 1 and np.allclose(r["z"], sols[0], atol=1e-8)     # strongly monotone: unique


def test_lp_like_nodes_vs_linprog(oracle):
    from scipy.optimize import linprog
    rng = np.random.default_rng(1)
    for _ in range(40):
        n = int(rng.integers(1, 8)); m = n + int(rng.integers(1, 10))
        A = np.vstack([np.eye(n), rng.standard_normal((m - n, n))])
        l = np.concatenate([-2 * np.ones(n), -1 - np.abs(rng.standard_normal(m - n))])
        u = np.concatenate([2 * np.ones(n), 1 + np.abs(rng.standard_normal(m - n))])
        c = rng.standard_normal(n)
        M, q, lo, hi, kind = oracle.assemble_node(np.zeros((n, n)), np.zeros((n, 0)), c, A, np.zeros((m, 0)), l, u, np.zeros(0))
        r = oracle.solve_avi(M, q, lo, hi, kind=kind)
        lp = linprog(c, A_ub=np.vstack([A, -A]), b_ub=np.concatenate([u, -l]), bounds=[(None, None)] * n, method="highs")
        assert r["status"] == 1 and abs(c @ r["z"][:n] - lp.fun) < 1e-7


def test_ray_termination_iff_infeasible(oracle):
    from scipy.optimize import linprog
    rng = np.random.default_rng(2)
    seen = set()
    for _ in range(150):
        n, m = int(rng.integers(1, 10)), int(rng.integers(1, 18))
        G = rng.standard_normal((n, n)); Q = G.T @ G / n + 0.1 * np.eye(n)
        A = rng.standard_normal((m, n)); l = -1 - np.abs(rng.standard_normal(m)); u = 1 + np.abs(rng.standard_normal(m))
        eq = rng.random(m) < 0.15
        if eq.sum() > n:
            eq[:] = False
        u = np.where(eq, l, u)
        M, q, lo, hi, kind = oracle.assemble_node(Q, np.zeros((n, 0)), rng.standard_normal(n), A, np.zeros((m, 0)), l, u, np.zeros(0))
        r = oracle.solve_avi(M, q, lo, hi, kind=kind)
        lp = linprog(np.zeros(n), A_ub=np.vstack([A, -A]), b_ub=np.concatenate([u, -l]), bounds=[(None, None)] * n, method="highs")
        assert (lp.status == 0) == (r["status"] == 1)
        seen.add(r["status"])
    assert seen == {1, 2}


def test_convert_gavi_matches_reference_layout(oracle):
    """src/avi.jl:113-128 block layout, and reference form == reduced form on the primal."""
    rng = np.random.default_rng(3)
    n, m = 3, 4
    Q, R, qd, A, B, l, u = P.synth_node(9, n, m)
    Mg = np.hstack([Q, -A.T]); Ag = np.hstack([A, np.zeros((m, m))]); bw = rng.standard_normal(m)
    Mr, qr, lr, ur = oracle.convert_gavi(Mg, qd, np.full(n, -INF), np.full(n, INF), Ag, bw, l, u)
    exp = np.block([[Mg, np.zeros((n, m))], [Ag, -np.eye(m)], [np.zeros((m, n)), np.eye(m), np.zeros((m, m))]])
    assert np.array_equal(Mr, exp)
    assert np.array_equal(qr, np.concatenate([qd, bw, np.zeros(m)]))
    assert np.array_equal(lr, np.concatenate([np.full(n + m, -INF), l])) and np.array_equal(ur, np.concatenate([np.full(n + m, INF), u]))
    M, q, lo, hi, kind = oracle.assemble_node(Q, np.zeros((n, 0)), qd, A, np.zeros((m, 0)), l - bw, u - bw, np.zeros(0))
    z_red = oracle.solve_avi(M, q, lo, hi, kind=kind)["z"]
    z_ref = oracle.solve_avi(Mr, qr, lr, ur)["z"]
    assert np.max(np.abs(z_red[:n] - z_ref[:n])) < 1e-10


def test_verify_solution_vs_numpy(oracle):
    """verify accepts exactly at KKT points; duals equal the AVI multipliers."""
    for node in range(30):
        n, m = 6, 9
        Q, R, qd, A, B, l, u = P.synth_node(300 + node, n, m)
        w = P.shared_params()
        M, q, lo, hi, kind = (a[0] for a in P.reduced_blocks(Q[None], R[None], qd[None], A[None], B[None], l[None], u[None], w))
        z = oracle.solve_avi(M, q, lo, hi, kind=kind)["z"]
        s, lam, path = oracle.verify_solution(Q, R, qd, A, B, l, u, z[:n], w)
        assert s and path == 2 and np.max(np.abs(lam - z[n:])) < 1e-8
        s2, _, p2 = oracle.verify_solution(Q, R, qd, A, B, l, u, 0.99 * z[:n], w)
        if np.any(np.abs(z[n:]) > 1e-3) or np.linalg.norm(Q @ (0.99 * z[:n]) + R @ w + qd) > 1e-4:
            assert not s2
