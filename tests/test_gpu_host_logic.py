"""GPU twins of the host-logic tests: the same reference-named host code (solve, solve_base!,
process_qp, verify_solution, solve_qep, comp_indices) with the arithmetic on the HIP engine, checked
against the reference's own end-to-end data and against the oracle-backed run."""
import numpy as np
import pytest

import goldenio as G

pytestmark = pytest.mark.gpu


def test_reference_end_to_end_cases_on_gpu(engine):
    """test/simple_bilevel.jl:17-21 through the C-ABI (BASELINE config 1 data, on the HIP path)."""
    from qpn_amd import algorithm, examples
    c = G.load("simple_bilevel_cases.json")
    from qpn_amd.qp_processing import local_recipe_count
    for w, xs, min_pieces in zip(c["w"], c["accepted_xy"], c["min_pieces_root_graph"]):
        net = examples.setup("simple_bilevel", gen_solution_map=True)
        ret = algorithm.solve(net, np.array(list(w) + c["x0"], float), engine=engine)
        assert ret["solved"], ret
        assert any(np.allclose(ret["x_opt"], list(w) + list(xy), atol=c["atol"]) for xy in xs), (w, ret["x_opt"])
        # test/simple_bilevel.jl:20 itself: the root's solution graph (combine + remove_subsets over the device-made pieces)
        assert len(ret["Sol"][2]) >= min_pieces, (w, len(ret["Sol"][2]))
        # and the path's own count (see tests/test_oracle_golden.py): local pieces of the root at the equilibrium
        assert local_recipe_count(net, 2, ret["x_opt"], ret["Sol"], engine=engine) >= min_pieces, w


def test_outer_loops_match_oracle_backed_run(engine):
    from oracle_engine import OracleEngine
    from qpn_amd import algorithm, examples
    for name, kw in [("four_player_matrix_game", dict(seed=1)), ("four_player_matrix_game", dict(seed=4)),
                     ("synthetic_pairs", dict(pairs=3, n=3, m=3)), ("synthetic_pairs", dict(pairs=2, n=5, m=7))]:
        rg = algorithm.solve(examples.setup(name, **kw), engine=engine)
        rc = algorithm.solve(examples.setup(name, **kw), engine=OracleEngine())
        assert rg["solved"] and rc["solved"]
        assert np.max(np.abs(rg["x_opt"] - rc["x_opt"])) <= 1e-9


def test_four_player_draws_host_mirror_batched(engine, oracle):
    """Config 3 through the HOST mirror of combine_gavis (per-draw setup + combine_gavis_reduced, 100 draws), shared M
    (strideM = 0), one Nash pool per draw (N_red = 16); equilibrium certified per draw by check_avi_solution."""
    from qpn_amd import avi, examples
    from qpn_amd.engine import colmajor
    draws = 100          # (the full 1 000-draw batch, assembled by ONE qpn_assemble_pools call: tests/test_gpu_pools.py)
    rng = np.random.Generator(np.random.Philox(key=[20240422, 3]))
    cs = rng.standard_normal((draws, 4, 4, 2))
    qs = []
    M = None
    for d in range(draws):
        net = examples.setup("four_player_matrix_game", constellations=cs[d])
        dec = list(range(8))
        lab = {i: avi.create_labeled_gavi_from_qp(net, i, {}) for i in (1, 2, 3, 4)}
        g = avi.combine_gavis_reduced(8, dec, [], lab)
        Md = np.vstack([g.M, g.A])
        if M is None:
            M = Md
            l = np.concatenate([g.l1, g.l2]); u = np.concatenate([g.u1, g.u2])
        else:
            assert np.array_equal(M, Md)                # M identical across draws, only q differs
        qs.append(np.concatenate([g.o, np.zeros(8)]))
    q = np.stack(qs)
    kind = np.concatenate([np.zeros(8, np.uint8), np.ones(8, np.uint8)])
    L = np.tile(l, (draws, 1)); U = np.tile(u, (draws, 1))
    rg = engine.solve_avi_batch(colmajor(M), q, L, U, kind=kind)
    rc = oracle.solve_avi_batch(M, q, L, U, kind=kind)
    assert np.all(rg["status"] == 1) and np.array_equal(rg["active"], rc["active"])
    assert np.max(np.abs(rg["z"] - rc["z"])) <= 1e-9 and np.max(rg["resid"]) <= 1e-8
    deg, _ = engine.check_avi_batch(colmajor(M), q, L, U, rg["z"], kind=kind)
    assert np.all(deg == 0)


def test_batched_subpiece_verification_on_gpu(engine):
    """Row F2 on the HIP path: one qpn_verify_nodes call for all sub-piece combinations == one call per
    combination == the oracle-backed run (verdicts, paths, duals)."""
    from oracle_engine import OracleEngine
    from qpn_amd import algorithm, examples
    from qpn_amd.programs import Poly
    from qpn_amd.qp_processing import verify_solution, verify_solutions_batched
    net = examples.setup("synthetic_pairs", pairs=3, n=5, m=7)
    ret = algorithm.solve(net, engine=engine)
    assert ret["solved"]
    x = ret["x_opt"]
    rng = np.random.default_rng(7)
    for pid in sorted(net.network_depth_map[1]):
        qp = net.qps[pid]
        child = next(iter(net.network_edges[pid]))
        base = [net.constraints[c].poly for c in qp.constraint_indices]
        pieces = list(ret["Sol"][child])
        a = rng.standard_normal((3, len(x)))
        stacks = [base + [p] for p in pieces] + [base, base + [Poly(a, a @ x + 1.0, a @ x + 2.0)],
                                                 base + [Poly(a, a @ x - 1.0, a @ x + 1.0)]]
        many = verify_solutions_batched(qp, pid, stacks, net.decision_inds(pid), x, engine=engine)
        one = [verify_solution(qp, pid, cons, net.decision_inds(pid), x, engine=engine) for cons in stacks]
        ref = verify_solutions_batched(qp, pid, stacks, net.decision_inds(pid), x, engine=OracleEngine())
        for rb, r1, rr in zip(many, one, ref):
            assert rb["solution"] == r1["solution"] == rr["solution"] and rb["path"] == r1["path"] == rr["path"]
            if rr["lam"] is not None:
                assert np.allclose(rb["lam"], r1["lam"], atol=1e-9) and np.allclose(rb["lam"], rr["lam"], atol=1e-7)


def test_robust_avoid_level3_through_process_qp_with_device_made_pieces(engine):
    """BASELINE config 2, level 3 (the separating-hyperplane players: LP-like, Q = 0 -- outside what the strictly convex
    host restatement of rounds 1-2 could do): solve_qep forms the level's pool on the device (qpn_assemble_pools) and solves
    it; process_qp then verifies each node at the equilibrium and returns its solution-graph pieces, made by the device
    kernels (masks -> qpn_recipes_from_masks -> qpn_local_pieces) and brought down to x-coordinates.  Every piece contains
    the equilibrium, lives in the 18 variables of the net, and constrains only what the node's own KKT system touches."""
    from qpn_amd import avi, examples
    from qpn_amd.qp_processing import process_qp, verify_solution
    net = examples.setup("robust_avoid_simple")
    x = np.array(net.default_initialization, float)
    level3 = sorted(net.network_depth_map[3])
    x = avi.solve_qep(net, level3, x, {}, engine=engine)               # pool {s1, s2}: N_ref = 52, assembled on the device
    x_ref = avi.solve_qep(net, level3, np.array(net.default_initialization, float), {}, engine=engine, reference_form=True)
    assert np.max(np.abs(x - x_ref)) <= 1e-8
    for pid in level3:
        ret = process_qp(net, pid, x, {}, engine=engine)
        assert ret["solution"] and ret["S"] is not None and len(ret["S"]) >= 1, (pid, ret)
        dec = set(net.decision_inds(pid))
        base = [net.constraints[c].poly for c in net.qps[pid].constraint_indices]
        touched = set(np.nonzero(np.any(np.vstack([b.vectorize()[0] for b in base]) != 0, axis=0))[0]) | dec
        for P in ret["S"]:
            A, l, u = P.vectorize()
            assert A.shape[1] == net.num_vars and P.contains(x, tol=1e-6)
            assert set(np.nonzero(np.any(np.abs(A) > 1e-12, axis=0))[0]) <= touched
        # moving along a piece keeps the node optimal: a point of the piece close to x still verifies
        P = ret["S"][0]
        A, l, u = P.vectorize()
        eq = np.isfinite(l) & (l == u)
        Z = np.linalg.svd(A[eq])[2][int(np.linalg.matrix_rank(A[eq])):].T if eq.any() else np.eye(net.num_vars)
        if Z.shape[1]:
            y = x + 1e-3 * Z @ np.random.default_rng(pid).standard_normal(Z.shape[1])
            if P.contains(y, tol=1e-9):
                assert verify_solution(net.qps[pid], pid, base, net.decision_inds(pid), y, engine=engine)["solution"]
