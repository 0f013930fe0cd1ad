"""Mid-size node records (n, m <= 64, one of them > 32): the fused kernels -- ONE wavefront per node up to max(n, m) = 48
(csrc/qpn_avi_schur48.hip: the 32-class kernel with three tiles a side), one workgroup per node beyond
(csrc/qpn_avi_schur_wg.hip: crash on the matrix cores straight from the records, Lemke spread over the workgroup's
wavefronts) -- read-back and post-check from the records, ONE launch, no workspace, against
* the oracle (status, active-set masks bit-exact, primals within 1e-9 relative: the bar of DESIGN.md section 2),
* the general route (qpn_ctx_set_option QPN_OPT_MID_ROUTE = 0: assembled blocks + the workgroup crash of the large nodes /
  the general kernels): same statuses and masks, primals within 1e-9 (summation order differs) -- round 2's three-kernel route
  and the workgroup kernel for 33 .. 48 (routes 2 and 3 of round 3) are gone, their A/B figures are in profiles/r03_mid*,
* the independent check kernel on stand-alone assembled blocks (A3, src/avi.jl:148-156),
and its decline handling: nodes whose leading 4 x 4 block of Qd fails the no-pivoting test, and nodes with an equality
row, go to the general kernel inside the same call."""
import os

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


SHAPES = [(33, 33, 8), (40, 50, 3), (48, 48, 8), (64, 64, 8), (64, 5, 2), (5, 64, 2), (40, 20, 4), (33, 1, 0), (17, 64, 8),
          (64, 33, 1), (49, 31, 5), (35, 47, 2), (10, 48, 3), (1, 40, 2), (48, 1, 1), (44, 44, 8)]


@pytest.mark.parametrize("n,m,p", SHAPES)
def test_mid_nodes_against_oracle_and_previous_route(engine, oracle, n, m, p):
    cnt = 10
    rec, abi = _records(900 + n + m, cnt, n, m, p)
    w = np.random.default_rng(n).standard_normal(p)
    res = engine.solve_nodes(*abi, w)
    assert np.all(res["status"] == 1)
    ref = _oracle(oracle, rec, w)
    _same(res, ref)
    from qpn_amd._lib import OPT_MID_ROUTE
    for route in (0,):               # the general route: assembled blocks -> the large-node / general kernels
        engine.set_option(OPT_MID_ROUTE, route)
        try:
            old = engine.solve_nodes(*abi, w)
        finally:
            engine.set_option(OPT_MID_ROUTE, 1)
        assert np.array_equal(res["status"], old["status"]) and np.array_equal(res["active"], old["active"]), route
        assert np.max(np.abs(res["z"] - old["z"])) <= 1e-9 * max(1.0, np.max(np.abs(old["z"]))), route
        assert np.array_equal(res["pivots"], old["pivots"]), route
    assert np.max(res["resid"]) <= 1e-8
    # independent certificate: check kernel on blocks from the stand-alone assembly kernel
    Mc, q, lo, hi, kind = engine.assemble_nodes(*abi, w)
    degree, _ = engine.check_avi_batch(Mc, q, lo, hi, res["z"], kind=kind, tol=1e-6)
    assert int(np.asarray(degree).sum()) == 0


def test_mid_nodes_per_node_parameters_handle_and_primal_blocks(engine, oracle):
    n, m, p, cnt = 48, 40, 6, 40
    rec, abi = _records(77, cnt, n, m, p)
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


def test_mid_nodes_declines_go_to_the_general_kernel(engine, oracle):
    n, m, p, cnt = 40, 36, 4, 12
    rec, abi = _records(31, cnt, n, m, p)
    Q, R, qd, A, B, l, u = [a.copy() for a in rec]
    # node 2: a tiny leading pivot (the 4 x 4 block fails the no-pivoting test)
    Qs = Q[2].copy(); Qs[0, 0] = 1e-9; Qs[0, 1:] *= 1e-3; Qs[1:, 0] *= 1e-3
    Q[2] = 0.5 * (Qs + Qs.T) + np.diag([0.0] + [0.5] * (n - 1))
    # node 7: an equality row
    u[7, 4] = l[7, 4]
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
    assert info["decline_state"] == 3 and info["declined"] == 2
    nodes.close()


def test_mid_nodes_device_buffers_and_shards(engine):
    import torch
    n, m, p, cnt = 64, 64, 8, 300
    rec, abi = _records(5, cnt, n, m, p)
    w = P.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    dev = [t(a) for a in abi] + [t(w)]
    x = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
    res = engine.solve_nodes(*dev, x_out=x)
    torch.cuda.synchronize()
    host = engine.solve_nodes(*abi, w)
    for k in ("z", "status", "active", "pivots"):
        assert np.array_equal(res[k].cpu().numpy(), host[k]), k
    assert np.all(host["status"] == 1) and np.array_equal(x.cpu().numpy(), host["z"][:, :n])
    for lo_, hi_ in [(0, 7), (123, 300), (299, 300)]:          # node ranges solved alone give identical rows
        part = [a[lo_:hi_] for a in dev[:-1]] + [dev[-1]]
        r = engine.solve_nodes(*part)
        torch.cuda.synchronize()
        assert np.array_equal(r["z"].cpu().numpy(), host["z"][lo_:hi_])
        assert np.array_equal(r["active"].cpu().numpy(), host["active"][lo_:hi_])


@pytest.mark.parametrize("seed", range(6))
def test_mid_nodes_mixed_bound_kinds_random_shapes(engine, oracle, seed):
    """Random shapes of the class with one-sided, free and equal bounds mixed in (equal bounds send a node to the general
    kernel inside the same call): everything against the oracle."""
    from qpn_amd.engine import colmajor
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 65)); m = int(rng.integers(1, 65))
    if max(n, m) <= 32:
        n = 33 + seed
    p = int(rng.integers(0, 7)); cnt = 8
    rec, _ = _records(seed, cnt, n, m, p)
    Q, R, qd, A, B, l, u = [a.copy() for a in rec]
    kind = rng.integers(0, 6, size=l.shape)
    l = np.where(kind == 1, -np.inf, l); u = np.where(kind == 2, np.inf, u)
    l = np.where(kind == 3, -np.inf, l); u = np.where(kind == 3, np.inf, u)
    eq = (kind == 4) & (rng.random(l.shape) < 0.1)
    u = np.where(eq, l, u)
    abi = [colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u]
    w = rng.standard_normal(p)
    res = engine.solve_nodes(*abi, w)
    ref = _oracle(oracle, (Q, R, qd, A, B, l, u), w)
    _same(res, ref)
    assert np.max(res["resid"][res["status"] == 1], initial=0.0) <= 1e-8


def test_mid_nodes_full_batch_properties(engine, oracle):
    """4 000 nodes of n = m = 48 (the size the mid-size probe quotes): every node solved and certified by the independent
    check kernel on stand-alone assembled blocks, multipliers only on rows at a bound, resident-records route == per-call
    route bit for bit, a seeded subset against the oracle."""
    import torch
    n, m, p, cnt = 48, 48, 8, 4000
    rec, abi = _records(4848, cnt, n, m, p)
    w = P.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    dev = [t(a) for a in abi] + [t(w)]
    res = engine.solve_nodes(*dev)
    torch.cuda.synchronize()
    host = {k: v.cpu().numpy() for k, v in res.items()}
    assert np.all(host["status"] == 1) and np.max(host["resid"]) <= 1e-8
    Mc, q, lo, hi, kind = engine.assemble_nodes(*dev)
    degree, r = engine.check_avi_batch(Mc, q, lo, hi, res["z"], kind=kind, tol=1e-6)
    torch.cuda.synchronize()
    assert int(degree.sum().item()) == 0
    lam = host["z"][:, n:]; s = r.cpu().numpy()[:, n:]
    l, u = rec[5], rec[6]
    at_bound = (np.abs(s - l) <= 1e-6) | (np.abs(s - u) <= 1e-6)
    assert np.all(at_bound[np.abs(lam) > 1e-9])
    nodes = engine.upload_nodes(*dev[:-1])
    for sweep in range(2):
        out = nodes.solve(dev[-1])
        torch.cuda.synchronize()
    for k in ("z", "status", "active", "pivots"):
        assert np.array_equal(out[k].cpu().numpy(), host[k]), k
    assert nodes.info()["decline_state"] == 2
    nodes.close()
    idx = np.sort(np.random.default_rng(1).choice(cnt, 40, replace=False))
    ref = _oracle(oracle, tuple(a[idx] for a in rec), w)
    assert np.array_equal(host["status"][idx], ref["status"]) and np.array_equal(host["active"][idx], ref["active"])
    assert np.max(np.abs(host["z"][idx] - ref["z"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["z"])))


@pytest.mark.parametrize("n,m,cnt", [(40, 37, 2600), (60, 52, 1150), (80, 70, 300)])
def test_handle_schedule_changes_nothing_in_the_results(engine, oracle, n, m, cnt):
    """The size classes above 32 through a resident handle beyond their resident sets (2 048 wavefronts of the one-wavefront kernel,
    1 024 workgroups of the 49-64 class, 256 of the 65-128 class): the handle installs its longest-first schedule from the sweeps'
    own pivot counts (as in the 32-class); the rows it produces are bit for bit the per-call route's (which has no schedule),
    sweep after sweep, with new parameters every sweep; a subset against the oracle."""
    p = 5
    rec, abi = _records(41, cnt, n, m, p)
    nodes = engine.upload_nodes(*abi)
    rng = np.random.default_rng(8)
    for sweep in range(4):
        w = rng.standard_normal(p)
        a = nodes.solve(w)
        b = engine.solve_nodes(*abi, w)
        for k in ("z", "status", "resid", "pivots", "active"):
            assert np.array_equal(a[k], b[k]), (sweep, k)
    assert nodes.info()["scheduled"]
    idx = np.sort(rng.choice(cnt, 60, replace=False))
    ref = _oracle(oracle, tuple(x[idx] for x in rec), w)
    _same({k: a[k][idx] for k in ("status", "active", "z")}, ref)
    nodes.close()
