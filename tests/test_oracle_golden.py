"""The oracle against every golden vector available for this path:
the reference's own end-to-end test data (test/simple_bilevel.jl:4-21), the hand-derived AVI known
answers, the worked verify_solution trace, and the committed regression vectors."""
import numpy as np
import pytest

import goldenio as G

INF = np.inf


def test_avi_known_answers(oracle):
    k = G.load("avi_kats.json")
    M = np.array(k["M"], float); l = G.dec(k["l"]); u = G.dec(k["u"])
    for c in k["cases"]:
        r = oracle.solve_avi(M, np.array(c["q"], float), l, u)
        assert r["status"] == 1 and r["resid"] <= 1e-12
        assert np.allclose(r["z"], c["z"], atol=1e-12)
        codes = {i + 1 for i in range(4) if (r["active"][3] >> i) & 1}
        assert codes == set(c["slack_row_codes"])
        bad, deg, _ = oracle.check_avi_solution(M, np.array(c["q"], float), l, u, r["z"])
        assert not bad and deg == 0


def test_reference_end_to_end_cases(oracle):
    """test/simple_bilevel.jl:17-21: x_opt ~ [w; x*] (atol 1e-4) for one of the listed equilibria.
    Drives the product's host loop (solve -> solve_base! -> process_qp -> verify_solution ->
    solve_qep) with the oracle as the arithmetic back-end."""
    import qpn_amd  # noqa: F401
    from qpn_amd import algorithm, examples
    from oracle_engine import OracleEngine
    c = G.load("simple_bilevel_cases.json")
    from qpn_amd.qp_processing import local_recipe_count
    for w, xs, min_pieces in zip(c["w"], c["accepted_xy"], c["min_pieces_root_graph"]):
        net = examples.setup("simple_bilevel", gen_solution_map=True)
        ret = algorithm.solve(net, np.array(list(w) + c["x0"], float), engine=OracleEngine())
        assert ret["solved"], ret
        assert any(np.allclose(ret["x_opt"], list(w) + list(xy), atol=c["atol"]) for xy in xs), (w, ret["x_opt"])
        # follower's solution graph has the local pieces the reference derives
        assert 1 <= len(ret["Sol"][1]) <= 2
        # The reference's second assertion, test/simple_bilevel.jl:20: length(collect(ret.Sol[2])) >= s -- the ROOT's solution
        # graph as process_qp leaves it: each sub-piece combination's local pieces (device kernels, multipliers eliminated),
        # put together by combine (src/qp_processing.jl:260-291 -> qp_processing.combine_at) and remove_subsets
        # (src/algorithm.jl:84).  The reference's graph additionally grows by vertex exploration, so its count can only be larger.
        assert len(ret["Sol"][2]) >= min_pieces, (w, len(ret["Sol"][2]))
        # ... and the count the hot path alone yields at the equilibrium (one local piece per recipe compatible with the
        # root's active-set masks, over the child pieces for which it is optimal) reaches the same lower bound
        assert local_recipe_count(net, 2, ret["x_opt"], ret["Sol"], engine=OracleEngine()) >= min_pieces, w


def test_worked_trace_w_minus2_minus3(oracle):
    """call order and values for w = [-2,-3]."""
    w = np.array([-2.0, -3.0]); Q = 2 * np.eye(2); R = -2 * np.eye(2); z2 = np.zeros(2)
    B = np.zeros((2, 2))
    s, lam, path = oracle.verify_solution(Q, R, z2, np.array([[0., 1], [1, 0]]), B, [0, -INF], [0, 0], z2, w)
    assert (s, path) == (False, 4)          # LSQ gives [-4, 6] -> wrong sign -> PATH-QP fallback -> suboptimal
    s, lam, path = oracle.verify_solution(Q, R, z2, np.array([[1., -1], [0, 1]]), B, [0, 0], [0, INF], z2, w)
    assert (s, path) == (True, 2) and np.allclose(lam, [4, 10])
    # after solve_qep on piece 1: (x,y) = (-2,0); follower multiplier 4 (strongly active), leader [0, 6]
    s, lam, path = oracle.verify_solution(np.array([[2.0]]), np.array([[0, 0, -2.0]]), [0.0], np.array([[1.0]]),
                                          np.zeros((1, 3)), [0.0], [INF], [0.0], np.array([-2, -3, -2.0]))
    assert s and np.allclose(lam, [4.0])


def test_regression_vectors(oracle):
    v = G.load("oracle_vectors.json")["vectors"]
    assert len(v) == 24
    for t in v:
        M = G.dec(t["M"]); kind = np.array(t["kind"], np.uint8)
        r = oracle.solve_avi(M, G.dec(t["q"]), G.dec(t["l"]), G.dec(t["u"]), z0=G.dec(t["z0"]), kind=kind)
        assert r["status"] == t["status"] and r["pivots"] == t["pivots"]
        assert np.array_equal(r["active"], np.array(t["active"], np.uint8))
        assert np.allclose(r["z"], G.dec(t["z"]), atol=1e-12, rtol=0)


def test_comp_indices_codes(oracle):
    """src/avi_solutions.jl:511-562 incl. infinite bounds and l ~ u within tol."""
    l = np.array([0, 0, 0, 0, 0, -INF, -INF, 1.0, 1.0, 0, 0, -INF])
    u = np.array([1, 1, 1, 1, 1, INF, 2.0, 1.0, 1.005, INF, 1, INF])
    z = np.array([0, 0, 0.5, 1, 1, 3.0, 2.0, 1.0, 1.0, 0.004, 0.5, -7.0])
    r = np.array([1, 0, 0, 0, -2, 0.0, -1.0, 5.0, -3.0, 0.005, 0.2, 0.0])
    m = oracle.comp_indices(z, r, l, u)
    assert list(m) == [1, 3, 2, 6, 4, 2, 4, 8, 8, 3, 0, 2]
    assert list(oracle.comp_indices(z, r, l, u, shift=4)) == [x << 4 for x in [1, 3, 2, 6, 4, 2, 4, 8, 8, 3, 0, 2]]
