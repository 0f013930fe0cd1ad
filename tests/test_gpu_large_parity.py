"""GPU parity for large items (64 < N <= 1024; qpn_avi_big.hip): BASELINE config 5's item shape
(n = m = 256, N_red = 512) and the reference-form pools of config 2 (N_ref 52..~120)."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
INF = np.inf


def _cmp(rg, rc, what, ztol=1e-9):
    assert np.array_equal(np.asarray(rg["status"]), np.asarray(rc["status"])), what
    ok = np.asarray(rc["status"]) == 1
    zg = np.asarray(rg["z"])[ok]; zc = np.asarray(rc["z"])[ok]
    scale = np.maximum(1.0, np.max(np.abs(zc), axis=1, keepdims=True))
    assert np.max(np.abs(zg - zc) / scale) <= ztol, what
    assert np.array_equal(np.asarray(rg["active"])[ok], np.asarray(rc["active"])[ok]), what
    assert np.all(np.asarray(rg["resid"])[ok] <= 1e-8), what
    assert np.array_equal(np.asarray(rg["pivots"]), np.asarray(rc["pivots"])), what


@pytest.mark.parametrize("n,m,cnt", [(40, 40, 6), (48, 80, 4), (100, 157, 3), (256, 256, 2), (300, 330, 2), (17, 500, 2)])
def test_large_reduced_nodes(engine, oracle, n, m, cnt):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(7000 + n, cnt, n, m)
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, P.shared_params())
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    assert np.all(rc["status"] == 1)
    _cmp(rg, rc, f"large n={n} m={m}")


@pytest.mark.parametrize("n,m,cnt", [(5, 99, 7), (1, 100, 5), (3, 70, 6), (7, 120, 4), (1, 300, 3)])
def test_large_items_with_very_few_free_rows(engine, oracle, n, m, cnt):
    """n (2 m + n + 1) < 15 m: the padded row stride of the delayed-update Lemke's dictionary (m + 1 rounded up to 16) would
    not fit an item's N (N + 1) slot of the workspace -- the rows of one item ran into the next item's slot and, behind the
    last item, into the crash's outputs (found by tools/wg2_fuzz.py at n = 5, m = 99: item 0 of 7 came back FAILURE with the
    oracle's pivot count).  Such items take the plain stride.  Explicit M and node records (the large-node route)."""
    from qpn_amd import _lib
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(7300 + m, cnt, n, m)
    w = P.shared_params()
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    _cmp(rg, rc, f"few free rows, explicit M, n={n} m={m}")
    engine.set_option(_lib.OPT_MID_ROUTE, 0)
    try:
        rn = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
    finally:
        engine.set_option(_lib.OPT_MID_ROUTE, 1)
    _cmp(rn, rc, f"few free rows, node records on the large-node route, n={n} m={m}")


def test_large_box_mcp_and_device_path(engine, oracle):
    import torch
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(12)
    N = 130
    probs = [P.random_box_mcp(rng, N) for _ in range(5)]
    M = np.stack([p[0] for p in probs]); q = np.stack([p[1] for p in probs])
    l = np.stack([p[2] for p in probs]); u = np.stack([p[3] for p in probs]); z0 = np.stack([p[4] for p in probs])
    rc = oracle.solve_avi_batch(M, q, l, u, z0=z0)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda:0")
    rg = engine.solve_avi_batch(t(colmajor(M)), t(q), t(l), t(u), z0=t(z0))
    torch.cuda.synchronize()
    _cmp({k: v.cpu().numpy() for k, v in rg.items()}, rc, "large box device")


def test_robust_avoid_deepest_pool_reference_form(engine, oracle):
    """BASELINE config 2: level-3 pool {s1, s2} of robust_avoid_simple (LP-like, Q = 0): nd = 6,
    d2 = 20 -> N_ref = 52 (reference form, src/avi.jl:113-128), reduced form N = 26.  Degenerate
    regime: pinned by the A3 check + residual, and GPU == oracle; both forms give the same primal
    objective (epsilon values)."""
    from qpn_amd import avi, examples
    net = examples.setup("robust_avoid_simple")
    x = net.default_initialization.copy()
    pool = sorted(net.network_depth_map[3])
    dec = sorted(set().union(*[set(net.decision_inds(i)) for i in pool]))
    par = [i for i in range(net.num_vars) if i not in set(dec)]
    lab = {i: avi.create_labeled_gavi_from_qp(net, i, {}) for i in pool}
    g_ref = avi.combine_gavis(net.num_vars, dec, par, lab)
    assert len(g_ref.l1) + 2 * len(g_ref.l2) == 52
    w = x[par]
    outs = {}
    for name, eng in (("gpu", engine), ("cpu", __import__("oracle_engine").OracleEngine())):
        z_ref, st_ref, info_ref = avi.solve_gavi(g_ref, np.concatenate([x[dec], np.zeros(g_ref.M.shape[1] - len(dec))]), w,
                                                 engine=eng, reference_form=True)
        g_red = avi.combine_gavis_reduced(net.num_vars, dec, par, lab)
        z_red, st_red, info_red = avi.solve_gavi(g_red, np.concatenate([x[dec], np.zeros(len(g_red.l2))]), w, engine=eng)
        assert st_ref == st_red == avi.StatusCode.SUCCESS
        assert info_ref["resid"] <= 1e-8 and info_red["resid"] <= 1e-8
        outs[name] = (z_ref, z_red)
    ix = net.problem_data["index"]
    eps_pos = [dec.index(e) for e in ix["eps"]]
    for a_, b_ in zip(outs["gpu"], outs["cpu"]):
        assert np.max(np.abs(a_ - b_)) <= 1e-9                       # GPU == oracle
    assert np.allclose(outs["gpu"][0][eps_pos], outs["gpu"][1][eps_pos], atol=1e-8)   # same LP optimum


def test_large_blocked_crash_declines_fall_back(engine, oracle):
    """Large node-shaped items go through the blocked MFMA crash (qpn_avi_schur_big.hip); items it must
    decline -- an equality GAVI row, a singular H block (LP-like node, Q = 0), a non-node shape -- are solved
    by the general large-item kernel in the same call.  Mixed batch, every item against the oracle."""
    from qpn_amd.engine import colmajor
    n, m, cnt = 40, 44, 6
    Q, R, qd, A, B, l, u = P.synth_nodes(7300, cnt, n, m)
    l = l.copy(); u = u.copy()
    l[1, 3] = u[1, 3] = 0.25                                  # item 1: equality GAVI row -> declined
    Q[2] = 0.0                                                # item 2: LP-like (H = 0) -> pivot test fails -> declined
    A[2, :n, :] = np.eye(n)                                   #         (box rows make it bounded)
    l[2, :n] = -2.0; u[2, :n] = 2.0
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, P.shared_params())
    kind = kind.copy()
    kind[3, 5] = 1; lo[3, 5] = -3.0; hi[3, 5] = 3.0           # item 3: a GAVI row inside the x block -> other shape
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rg = engine.solve_avi_batch(colmajor(M), q, lo, hi, kind=kind)
    assert np.all(rc["status"] == 1)
    _cmp(rg, rc, "large mixed batch")
