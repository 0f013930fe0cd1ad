"""Node records with n, m <= 128 and one of them > 64: the fused one-workgroup-per-node kernel of csrc/qpn_avi_schur_wg2.hip
(two wavefronts per row tile: H waves and C waves; Lemke with a leader wave; one launch, no workspace) against
* the oracle (status, active-set masks bit-exact, primals within 1e-9 relative: the bar of DESIGN.md section 2),
* the route these sizes took before (qpn_ctx_set_option QPN_OPT_MID_ROUTE = 0: assembled blocks + the workgroup crash of the
  large nodes): same statuses, masks and pivot counts, primals within 1e-9,
* the independent check kernel on stand-alone assembled blocks (A3, src/avi.jl:148-156),
and its decline handling (a block pivot below the threshold, an equality row), the resident-records route, device buffers."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _records(seed, cnt, n, m, p):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(seed, cnt, n, m, max(p, 1))
    rng = np.random.default_rng(seed)
    if p == 0:
        R = np.zeros((cnt, n, 0)); B = np.zeros((cnt, m, 0))
    else:
        R = R[:, :, :p]; B = rng.standard_normal((cnt, m, p)) * 0.1
    return (Q, R, qd, A, B, l, u), [colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u]


def _oracle(oracle, rec, w):
    M, q, lo, hi, kind = P.reduced_blocks(*rec, w)
    return oracle.solve_avi_batch(M, q, lo, hi, kind=kind)


def _same(res, ref, tol=1e-9):
    assert np.array_equal(np.asarray(res["status"]), ref["status"])
    assert np.array_equal(np.asarray(res["active"]), ref["active"])
    z, zr = np.asarray(res["z"]), ref["z"]
    assert np.max(np.abs(z - zr)) <= tol * max(1.0, np.max(np.abs(zr)))


# every size class (NR = 5 .. 8), square and ragged, the VERDICT's (65, 65) (96, 96) (128, 100), one dimension small
SHAPES = [(65, 65, 8), (80, 80, 3), (96, 96, 8), (112, 97, 2), (128, 100, 8), (128, 128, 8), (70, 128, 4), (128, 5, 2), (3, 90, 0),
          (81, 33, 1)]


@pytest.mark.parametrize("n,m,p", SHAPES)
def test_wg2_nodes_against_oracle_and_previous_route(engine, oracle, n, m, p):
    cnt = 6
    rec, abi = _records(1900 + n + m, cnt, n, m, p)
    w = np.random.default_rng(n).standard_normal(p)
    res = engine.solve_nodes(*abi, w)
    assert np.all(res["status"] == 1)
    ref = _oracle(oracle, rec, w)
    _same(res, ref)
    from qpn_amd._lib import OPT_MID_ROUTE
    engine.set_option(OPT_MID_ROUTE, 0)
    try:
        old = engine.solve_nodes(*abi, w)
    finally:
        engine.set_option(OPT_MID_ROUTE, 1)
    assert np.array_equal(res["status"], old["status"]) and np.array_equal(res["active"], old["active"])
    assert np.max(np.abs(res["z"] - old["z"])) <= 1e-9 * max(1.0, np.max(np.abs(old["z"])))
    assert np.array_equal(res["pivots"], old["pivots"])
    assert np.max(res["resid"]) <= 1e-8
    # independent certificate: check kernel on blocks from the stand-alone assembly kernel
    Mc, q, lo, hi, kind = engine.assemble_nodes(*abi, w)
    degree, _ = engine.check_avi_batch(Mc, q, lo, hi, res["z"], kind=kind, tol=1e-6)
    assert int(np.asarray(degree).sum()) == 0


def test_wg2_nodes_per_node_parameters_handle_and_primal_blocks(engine, oracle):
    n, m, p, cnt = 96, 72, 6, 24
    rec, abi = _records(177, cnt, n, m, p)
    W = np.random.default_rng(3).standard_normal((cnt, p))
    ref = _oracle(oracle, rec, W)
    nodes = engine.upload_nodes(*abi)
    x = np.zeros((cnt, n + 3))
    for sweep in range(3):                       # the handle learns after the first sweep that nothing declines
        out = nodes.solve(W, x_out=x)
        _same(out, ref)
        assert np.array_equal(x[:, :n], out["z"][:, :n]) and np.all(x[:, n:] == 0)
    info = nodes.info()
    assert info["decline_state"] == 2 and info["declined"] == 0
    per_call = engine.solve_nodes(*abi, W)
    for k in ("z", "status", "resid", "pivots", "active"):
        assert np.array_equal(out[k], per_call[k]), k
    nodes.close()


def test_wg2_nodes_declines_go_to_the_general_kernel(engine, oracle):
    n, m, p, cnt = 72, 66, 4, 8
    rec, abi = _records(131, cnt, n, m, p)
    Q, R, qd, A, B, l, u = [a.copy() for a in rec]
    # node 2: a tiny pivot in a later 4 x 4 block (the no-pivoting test fails there)
    Qs = Q[2].copy(); Qs[41, 41] = 1e-9; Qs[41, :41] *= 1e-3; Qs[41, 42:] *= 1e-3; Qs[:41, 41] *= 1e-3; Qs[42:, 41] *= 1e-3
    Q[2] = 0.5 * (Qs + Qs.T)
    # node 5: an equality row
    u[5, 60] = l[5, 60]
    from qpn_amd.engine import colmajor
    abi2 = [colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u]
    w = np.random.default_rng(9).standard_normal(p)
    res = engine.solve_nodes(*abi2, w)
    ref = _oracle(oracle, (Q, R, qd, A, B, l, u), w)
    _same(res, ref)
    nodes = engine.upload_nodes(*abi2)
    for sweep in range(3):
        out = nodes.solve(w)
        _same(out, ref)
    info = nodes.info()
    assert info["decline_state"] == 3 and info["declined"] >= 1
    nodes.close()


def test_wg2_nodes_device_buffers_shards_and_batch_properties(engine, oracle):
    """600 nodes of n = m = 96 on device buffers: every node solved and certified by the independent check kernel,
    multipliers only on rows at a bound, node ranges solved alone give identical rows, a seeded subset against the oracle."""
    import torch
    n, m, p, cnt = 96, 96, 8, 600
    rec, abi = _records(9696, cnt, n, m, p)
    w = P.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    dev = [t(a) for a in abi] + [t(w)]
    x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
    res = engine.solve_nodes(*dev, x_out=x)
    torch.cuda.synchronize()
    host = {k: v.cpu().numpy() for k, v in res.items()}
    assert np.all(host["status"] == 1) and np.max(host["resid"]) <= 1e-8
    assert np.array_equal(x.cpu().numpy(), host["z"][:, :n])
    Mc, q, lo, hi, kind = engine.assemble_nodes(*dev)
    degree, r = engine.check_avi_batch(Mc, q, lo, hi, res["z"], kind=kind, tol=1e-6)
    torch.cuda.synchronize()
    assert int(degree.sum().item()) == 0
    lam = host["z"][:, n:]; s = r.cpu().numpy()[:, n:]
    l, u = rec[5], rec[6]
    at_bound = (np.abs(s - l) <= 1e-6) | (np.abs(s - u) <= 1e-6)
    assert np.all(at_bound[np.abs(lam) > 1e-9])
    for lo_, hi_ in [(0, 5), (123, 300), (599, 600)]:
        part = [a[lo_:hi_] for a in dev[:-1]] + [dev[-1]]
        r2 = engine.solve_nodes(*part)
        torch.cuda.synchronize()
        assert np.array_equal(r2["z"].cpu().numpy(), host["z"][lo_:hi_])
        assert np.array_equal(r2["active"].cpu().numpy(), host["active"][lo_:hi_])
    idx = np.sort(np.random.default_rng(1).choice(cnt, 24, replace=False))
    ref = _oracle(oracle, tuple(a[idx] for a in rec), w)
    assert np.array_equal(host["status"][idx], ref["status"]) and np.array_equal(host["active"][idx], ref["active"])
    assert np.max(np.abs(host["z"][idx] - ref["z"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["z"])))


@pytest.mark.parametrize("seed", range(5))
def test_wg2_nodes_mixed_bound_kinds_random_shapes(engine, oracle, seed):
    """Random shapes of the class with one-sided, free and equal bounds mixed in (equal bounds send a node to the general
    kernel inside the same call): everything against the oracle."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(1, 129)); m = int(rng.integers(1, 129))
    if max(n, m) <= 64:
        n = 65 + 13 * seed
    p = int(rng.integers(0, 7)); cnt = 5
    rec, _ = _records(seed, cnt, n, m, p)
    Q, R, qd, A, B, l, u = [a.copy() for a in rec]
    kind = rng.integers(0, 6, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    l = np.where(kind == 3, -np.inf, l); u = np.where(kind == 3, np.inf, u)
    eq = (kind == 4) & (rng.random(l.shape) < 0.03)
    u = np.where(eq, l, u)
    abi = [colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u]
    w = rng.standard_normal(p)
    res = engine.solve_nodes(*abi, w)
    ref = _oracle(oracle, (Q, R, qd, A, B, l, u), w)
    _same(res, ref)
    assert np.max(res["resid"][res["status"] == 1], initial=0.0) <= 1e-8
