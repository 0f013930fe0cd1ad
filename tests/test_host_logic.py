"""Host logic of the product package on CPU (arithmetic served by the oracle test double):
data model (src/programs.jl), pool assembly (src/avi.jl:205-377), outer loop (src/algorithm.jl)."""
import numpy as np
import pytest

import pyref

INF = np.inf


@pytest.fixture()
def eng():
    from oracle_engine import OracleEngine
    return OracleEngine()


def test_slice_normalisation_flips_bounds():
    """src/sets.jl:76-89: leading coefficient +1; a negative one flips the row and swaps l/u."""
    import qpn_amd  # noqa: F401
    from qpn_amd.programs import Poly
    P = Poly(np.array([[0, -2.0, 4.0], [3.0, 0, 0], [0, 0, 0]]), [-1.0, 3.0, -1.0], [INF, 6.0, 1.0])
    assert np.allclose(P.A, [[0, 1, -2], [1, 0, 0], [0, 0, 0]])
    assert np.allclose(P.l, [-INF, 1.0, -1.0]) and np.allclose(P.u, [0.5, 2.0, 1.0])
    assert P.contains(np.array([1.5, 0.0, 0.0])) and not P.contains(np.array([1.5, 1.0, 0.0]))


def test_network_structure():
    """add_edges!: transitive reduction, reachability, depth map, cycle detection (src/programs.jl:214-285)."""
    import qpn_amd  # noqa: F401
    from qpn_amd import examples
    from qpn_amd.programs import QPNet
    net = examples.setup("robust_avoid_simple")
    assert net.num_levels() == 3 and net.num_vars == 18 and len(net.qps) == 5
    assert net.network_depth_map == {1: {5}, 2: {3, 4}, 3: {1, 2}}
    assert net.network_edges[5] == {3, 4} and net.reachable_nodes[5] == {1, 2, 3, 4}
    ix = net.problem_data["index"]
    assert net.decision_inds(3) == sorted(ix["uo"][0] + ix["s"][0] + [ix["eps"][0]])
    assert len(net.decision_inds(5)) == 12
    net2 = QPNet(3)
    for v in range(3):
        net2.add_qp(np.eye(3), np.zeros(3), [], [v])
    net2.add_edges([(1, 2), (2, 3), (1, 3)])                    # 1->3 is redundant
    assert net2.network_edges[1] == {2} and net2.reachable_nodes[1] == {2, 3}
    assert net2.network_depth_map == {1: {1}, 2: {2}, 3: {3}}
    net3 = QPNet(2)
    net3.add_qp(np.eye(2), np.zeros(2), [], [0]); net3.add_qp(np.eye(2), np.zeros(2), [], [1])
    with pytest.raises(ValueError):
        net3.add_edges([(1, 2), (2, 1)])
    with pytest.warns(UserWarning):
        net3.set_options(not_an_option=1)                       # :312-320 only warns


def test_pool_assembly_reference_vs_reduced(eng):
    """combine_gavis (reference form with xi rows, src/avi.jl:305-377) and the reduced pool give the
    same primal; sizes match SURVEY.md section 8 (config 3: N_ref = 32, reduced 16)."""
    import qpn_amd  # noqa: F401
    from qpn_amd import avi, examples
    net = examples.setup("four_player_matrix_game", seed=3)
    x = np.zeros(8)
    pool = [1, 2, 3, 4]
    dec = sorted(set().union(*[set(net.decision_inds(i)) for i in pool]))
    lab = {i: avi.create_labeled_gavi_from_qp(net, i, {}) for i in pool}
    g_ref = avi.combine_gavis(8, dec, [], lab)
    g_red = avi.combine_gavis_reduced(8, dec, [], lab)
    assert len(g_ref.l1) + 2 * len(g_ref.l2) == 32 and len(g_red.l1) + len(g_red.l2) == 16
    H = g_red.M[:, :8]
    assert np.allclose(np.diag(H), 8.0) and np.allclose(np.sort(np.linalg.eigvals(H).real)[[0, -1]], [2, 10])
    z_ref, st_ref, _ = avi.solve_gavi(g_ref, np.zeros(g_ref.M.shape[1]), np.zeros(0), engine=eng, reference_form=True)
    z_red, st_red, _ = avi.solve_gavi(g_red, np.zeros(16), np.zeros(0), engine=eng)
    assert st_ref == st_red == avi.StatusCode.SUCCESS
    assert np.max(np.abs(z_ref[:8] - z_red[:8])) < 1e-9
    x1 = avi.solve_qep(net, pool, x, {}, engine=eng)
    x2 = avi.solve_qep(net, pool, x, {}, engine=eng, reference_form=True)
    assert np.max(np.abs(x1 - x2)) < 1e-9 and np.max(np.abs(x1 - z_red[:8])) < 1e-12
    # independent check of the Nash point: box-MCP in x with the stacked pseudo-gradient
    g = g_red.o
    zn, resn = pyref.solve_newton(H, g, np.full(8, -5.0), np.full(8, 5.0))
    assert resn < 1e-9 and np.max(np.abs(zn - x1)) < 1e-8


def test_four_player_nash_outer_loop(eng):
    """Flat Nash net (edge_list = []): one sweep of verify, one solve_qep, converged on re-verify."""
    import qpn_amd  # noqa: F401
    from qpn_amd import algorithm, examples
    from qpn_amd.qp_processing import verify_solution
    for seed in range(6):
        net = examples.setup("four_player_matrix_game", seed=seed)
        ret = algorithm.solve(net, engine=eng)
        assert ret["solved"], ret
        for pid, qp in net.qps.items():
            cons = [net.constraints[c].poly for c in qp.constraint_indices]
            assert verify_solution(qp, pid, cons, net.decision_inds(pid), ret["x_opt"], engine=eng)["solution"]


def test_two_level_pairs_outer_loop(eng):
    """Config-4 structure at small size: leader-follower pairs, followers' solution maps generated
    as local pieces, leaders verified on every piece, equilibrium certified by verify_solution."""
    import qpn_amd  # noqa: F401
    from qpn_amd import algorithm, examples
    from qpn_amd.qp_processing import verify_solution
    net = examples.setup("synthetic_pairs", pairs=3, n=3, m=3)
    ret = algorithm.solve(net, engine=eng)
    assert ret["solved"], ret
    x = ret["x_opt"]
    for pid in sorted(net.network_depth_map[2]):               # followers optimal w.r.t. own constraints
        qp = net.qps[pid]
        cons = [net.constraints[c].poly for c in qp.constraint_indices]
        assert verify_solution(qp, pid, cons, net.decision_inds(pid), x, engine=eng)["solution"]
    for pid in sorted(net.network_depth_map[1]):               # leaders optimal on every local piece
        qp = net.qps[pid]
        child = next(iter(net.network_edges[pid]))
        for piece in ret["Sol"][child]:
            cons = [net.constraints[c].poly for c in qp.constraint_indices] + [piece]
            assert verify_solution(qp, pid, cons, net.decision_inds(pid), x, engine=eng)["solution"]


def test_comp_indices_gavi_wrapper(eng):
    """src/avi_solutions.jl:587-612: codes 1..4 on z1 rows, 5..8 on the constraint rows."""
    import qpn_amd  # noqa: F401
    from qpn_amd import avi
    from qpn_amd.avi_solutions import comp_indices, masks_to_sets
    # follower of simple_bilevel: z = [y, lam], w = [w1, w2, x]
    g = avi.GAVI(M=np.array([[2.0, -1.0]]), N=np.array([[0, 0, -2.0]]), o=np.zeros(1), l1=np.array([-INF]),
                 u1=np.array([INF]), A=np.array([[1.0, 0.0]]), B=np.zeros((1, 3)), l2=np.zeros(1), u2=np.array([INF]))
    J = masks_to_sets(comp_indices(g, np.array([0.0, 4.0]), np.array([0, 0, -2.0]), engine=eng))
    assert J == {1: {2}, 2: {5}}                      # strongly active
    J = masks_to_sets(comp_indices(g, np.array([0.5, 0.0]), np.array([0, 0, 0.5]), engine=eng))
    assert J == {1: {2}, 2: {6}}                      # inactive
    J = masks_to_sets(comp_indices(g, np.array([0.0, 0.0]), np.array([0, 0, 0.0]), engine=eng))
    assert J == {1: {2}, 2: {5, 6}}                   # weakly active: SURVEY.md section 8(c)(2)


def test_solve_qp_path_branch(eng):
    """solve_qp(...; solver=:PATH), src/qp_processing.jl:12-33, against scipy."""
    import qpn_amd  # noqa: F401
    from qpn_amd.qp_processing import solve_qp
    from scipy.optimize import minimize
    rng = np.random.default_rng(0)
    G = rng.standard_normal((4, 4)); Q = G @ G.T + np.eye(4); q = rng.standard_normal(4)
    A = rng.standard_normal((3, 4)); l = -0.2 * np.ones(3); u = 0.3 * np.ones(3)
    x = solve_qp(Q, q, A, l, u, engine=eng)
    ref = minimize(lambda v: 0.5 * v @ Q @ v + q @ v, np.zeros(4), jac=lambda v: Q @ v + q,
                   constraints=[{"type": "ineq", "fun": lambda v: A @ v - l, "jac": lambda v: A},
                                {"type": "ineq", "fun": lambda v: u - A @ v, "jac": lambda v: -A}], tol=1e-12)
    assert np.max(np.abs(x - ref.x)) < 1e-6
    with pytest.raises(ValueError):
        solve_qp(Q, q, A, l, u, solver="OSQP", engine=eng)


def test_batched_subpiece_verification_equals_one_by_one(eng):
    """SURVEY section 8(f) row F2 (src/qp_processing.jl:162-205): all sub-piece combinations of a node go
    through ONE batched verify; ragged stacks are padded with inert rows.  Same verdicts, paths and duals
    as one verify_solution per combination, and process_qp reports the first failing combination."""
    import qpn_amd  # noqa: F401
    from qpn_amd import algorithm, examples
    from qpn_amd.programs import Poly
    from qpn_amd.qp_processing import process_qp, verify_solution, verify_solutions_batched
    net = examples.setup("synthetic_pairs", pairs=2, n=4, m=5)
    ret = algorithm.solve(net, engine=eng)
    assert ret["solved"]
    x = ret["x_opt"]
    rng = np.random.default_rng(2)
    for pid in sorted(net.network_depth_map[1]):
        qp = net.qps[pid]
        child = next(iter(net.network_edges[pid]))
        base = [net.constraints[c].poly for c in qp.constraint_indices]
        pieces = list(ret["Sol"][child])
        # extra stacks of other lengths: an empty appendix, a doubled piece, a piece the point violates
        a = rng.standard_normal((2, len(x)))
        bad = Poly(a, a @ x + 1.0, a @ x + 2.0)
        stacks = [base + [p] for p in pieces] + [base, base + [pieces[0], pieces[0]], base + [bad]]
        one = [verify_solution(qp, pid, cons, net.decision_inds(pid), x, engine=eng) for cons in stacks]
        many = verify_solutions_batched(qp, pid, stacks, net.decision_inds(pid), x, engine=eng)
        assert len(many) == len(one)
        for r1, rb in zip(one, many):
            assert r1["solution"] == rb["solution"] and r1["path"] == rb["path"] and r1["e"] == rb["e"]
            if r1["lam"] is not None:
                assert np.allclose(r1["lam"], rb["lam"], atol=1e-9)
        assert not many[-1]["solution"] and many[-1]["path"] == 0         # the violated stack: infeasible
        assert all(r["solution"] for r in many[:len(pieces)])
        # process_qp on a solution-graph dict whose second piece is violated: first failure in product order
        S = {child: [pieces[0], bad] + pieces[1:]}
        out = process_qp(net, pid, x, S, engine=eng)
        assert out["solution"] is False and out["subpiece_assignments"] == {child: 1}


def test_robust_avoid_pool_avis_all_levels_with_fixture_pieces(eng):
    """BASELINE config 2 on the host logic (oracle engine): the pool AVIs of the three levels of robust_avoid_simple with
    the committed child pieces have the sizes SURVEY section 8 derives (N_ref 52 / 60-100 / 60-120) and solve; the
    HIP twin (device assembly + HIP solve) is tests/test_gpu_pools.py."""
    import json, os
    from qpn_amd import avi, examples
    from qpn_amd.programs import Poly
    net = examples.setup("robust_avoid_simple")
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "robust_avoid_pieces.json")))
    x = np.asarray(fx["x"])
    S = {int(k): Poly(np.asarray(v["A"]), np.asarray(v["l"], dtype=float), np.asarray(v["u"], dtype=float), normalise=False)
         for k, v in fx["pieces"].items()}
    sizes = {}
    for level in (3, 2, 1):
        pool = sorted(net.network_depth_map[level])
        dec = sorted(set().union(*[set(net.decision_inds(i)) for i in pool]))
        par = [i for i in range(net.num_vars) if i not in set(dec)]
        lab = {i: avi.create_labeled_gavi_from_qp(net, i, S) for i in pool}
        g = avi.combine_gavis(net.num_vars, dec, par, lab)
        sizes[level] = len(g.l1) + 2 * len(g.l2)
        z0 = np.concatenate([x[dec], np.zeros(g.M.shape[1] - len(dec))])
        z, st, info = avi.solve_gavi(g, z0, x[par], engine=eng, reference_form=True)
        assert st == avi.StatusCode.SUCCESS and info["resid"] <= 1e-8
        b = avi.pool_blocks(net.num_vars, dec, par, lab)
        assert sum(b["n_i"]) == len(dec) and b["Qd"].shape == (len(dec), len(dec)) and b["Ad"].shape[0] == len(g.l2)
    assert sizes == {3: 52, 2: 68, 1: 80}
