"""BASELINE config 5 at FULL size: 512 nodes x 256 variables (n = m = 256, N_red = 512) through the C-ABI, the batch the
"large per-node KKT" path (blocked MFMA crash + delayed-update Lemke, csrc/qpn_avi_schur_big.hip) is sized for --
1 GB of assembled blocks, stage-A workspaces for 512 items, the Lemke launches split by reduced size.

The oracle needs ~1 s per item of this size, so (as for config 4, tests/test_gpu_fullsize.py) the full batch is held to
size-independent properties and a seeded subset is compared with the oracle directly:
* every item solved, natural-map residual <= 1e-8, and the INDEPENDENT check kernel (A3, src/avi.jl:148-156) on blocks from
  the stand-alone assembly kernel finds no violation on any of the 512 items;
* a GAVI row with a non-zero multiplier sits at a bound; masks consistent with that;
* resident-records route (qpn_solve_nodes_h) == per-call route, bit for bit; primal write-back == z[:, :n];
* shard invariance (node ranges solved alone give identical rows: what multi-GPU sharding relies on);
* seeded subset vs the oracle: status, active-set masks bit-exact, primals within 1e-9 relative.
"""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu

NODES, N_, M_, P_ = 512, 256, 256, 8


@pytest.fixture(scope="module")
def full5(engine):
    import torch
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(0, NODES, N_, M_, P_)
    w = P.shared_params(P_)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    host = dict(Q=Q, R=R, qd=qd, A=A, B=B, l=l, u=u, w=w)
    dev = [t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u), t(w)]
    x = torch.zeros((NODES, N_), dtype=torch.float64, device="cuda:0")
    res = engine.solve_nodes(*dev, x_out=x)
    torch.cuda.synchronize()
    return host, dev, {k: v.cpu().numpy() for k, v in res.items()}, x.cpu().numpy()


def test_config5_full_batch_solved_and_certified(engine, full5):
    import torch
    host, dev, res, x = full5
    assert np.all(res["status"] == 1)
    assert np.max(res["resid"]) <= 1e-8
    assert np.array_equal(x, res["z"][:, :N_])
    Mc, q, lo, hi, kind = engine.assemble_nodes(*dev)                 # 1 GB of blocks: the stand-alone assembly kernel
    z = torch.tensor(res["z"], dtype=torch.float64, device="cuda:0")
    degree, r = engine.check_avi_batch(Mc, q, lo, hi, z, kind=kind, tol=1e-6)
    torch.cuda.synchronize()
    assert int(degree.sum().item()) == 0
    lam = res["z"][:, N_:]
    s = r.cpu().numpy()[:, N_:]
    del Mc, r
    at_bound = (np.abs(s - host["l"]) <= 1e-6) | (np.abs(s - host["u"]) <= 1e-6)
    assert np.all(at_bound[np.abs(lam) > 1e-9])
    assert np.all((res["active"][:, N_:] >> 4) > 0) and np.all(res["active"][:, :N_] == 2)
    # the Lemke phase really ran on every item (pivots = n crash pivots + Lemke pivots)
    assert np.all(res["pivots"] > N_) and res["pivots"].mean() > N_ + 50


def test_config5_handle_route_and_shards(engine, full5):
    import torch
    host, dev, res, _ = full5
    from qpn_amd import _lib
    nodes = engine.upload_nodes(*dev[:-1])
    assert nodes.info()["symmetric"]
    # resident records with symmetric Qd blocks (QPN_OPT_SYM_ROUTE, the default): at this size Stage B of every node is block
    # principal pivoting (schur_big_bpp) with the delayed-update Lemke kernel behind it for what it leaves -- another method on the
    # same Schur problem: the same solution to rounding, the same active sets, its own count of basis changes (in `pivots`: the n
    # crash pivots + the complementarity pairs it switched); with the option off the handle runs the per-call route's kernels --
    # bit for bit its answer
    sym = {k: v.cpu().numpy() for k, v in nodes.solve(dev[-1]).items()}
    for k in ("status", "active"):
        assert np.array_equal(sym[k], res[k]), k
    assert np.all(sym["pivots"] > N_) and np.max(sym["resid"]) <= 1e-8
    assert np.max(np.abs(sym["z"] - res["z"])) <= 1e-10 * max(1.0, np.max(np.abs(res["z"])))
    engine.set_option(_lib.OPT_SYM_ROUTE, 0)
    try:
        out = nodes.solve(dev[-1])
        torch.cuda.synchronize()
    finally:
        engine.set_option(_lib.OPT_SYM_ROUTE, 1)
    for k in ("z", "status", "active", "pivots"):
        assert np.array_equal(out[k].cpu().numpy(), res[k]), k
    # row A8 at this size: every node's verify_solution accepts its AVI solution, duals = the AVI's multipliers
    xd = out["z"][:, :N_].contiguous()
    sol, lam, path = nodes.verify(xd, dev[-1])
    torch.cuda.synchronize()
    assert int(sol.sum().item()) == NODES and bool((path == 2).all())
    assert float((lam - out["z"][:, N_:]).abs().max()) < 1e-6
    nodes.close()
    for lo_, hi_ in [(0, 64), (448, 512), (255, 258)]:
        part = [a[lo_:hi_] for a in dev[:-1]] + [dev[-1]]
        r = engine.solve_nodes(*part)
        torch.cuda.synchronize()
        assert np.array_equal(r["z"].cpu().numpy(), res["z"][lo_:hi_])
        assert np.array_equal(r["active"].cpu().numpy(), res["active"][lo_:hi_])


def test_config5_subset_against_oracle(oracle, full5):
    host, _, res, _ = full5
    idx = np.sort(np.random.default_rng(5).choice(NODES, 24, replace=False))
    M, q, lo, hi, kind = P.reduced_blocks(host["Q"][idx], host["R"][idx], host["qd"][idx], host["A"][idx], host["B"][idx],
                                          host["l"][idx], host["u"][idx], host["w"])
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    assert np.array_equal(res["status"][idx], rc["status"])
    assert np.array_equal(res["active"][idx], rc["active"])          # bit-exact active sets
    zr = rc["z"]
    assert np.max(np.abs(res["z"][idx] - zr)) <= 1e-9 * max(1.0, np.max(np.abs(zr)))


@pytest.mark.parametrize("n,m,cnt", [(256, 256, 3), (100, 130, 4), (70, 40, 5), (256, 100, 3), (200, 256, 3), (129, 1, 4), (250, 255, 3)])
def test_large_nodes_from_records_vs_oracle_and_previous_route(engine, oracle, n, m, cnt):
    """The blocked crash straight from the records (csrc/qpn_avi_schur_big2.hip: rank-64 block pivots on a tile-format
    workspace, sizes that are not multiples of 16 padded in the conversion pass) against the oracle -- status, masks bit-exact,
    primals within 1e-9 relative, pivot counts -- and against the explicit-M route over the assembled blocks: same masks and
    pivot counts, primals within 1e-9."""
    from qpn_amd import _lib
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(9000 + n + m, cnt, n, m, 4)
    w = P.shared_params(4)
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    args = (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
    rh = engine.solve_nodes(*args)
    assert np.array_equal(rh["status"], rc["status"]) and np.all(rc["status"] == 1)
    assert np.array_equal(rh["active"], rc["active"]) and np.array_equal(rh["pivots"], rc["pivots"])
    scale = np.maximum(1.0, np.max(np.abs(rc["z"]), axis=1, keepdims=True))
    assert np.max(np.abs(rh["z"] - rc["z"]) / scale) <= 1e-9 and np.max(rh["resid"]) <= 1e-8
    # the same nodes as an assembled M through the explicit-M route of large node-shaped items (qpn_assemble_nodes +
    # qpn_solve_avi_batch: the kernels round 2 ran config 5 on -- its node-record switch QPN_OPT_BIG_ROUTE = 0 is gone)
    Mc, qq, ll, uu, kk = engine.assemble_nodes(*args)
    r0 = engine.solve_avi_batch(Mc, qq, ll, uu, kind=kk)
    assert np.array_equal(r0["status"], rh["status"]) and np.array_equal(r0["active"], rh["active"])
    assert np.array_equal(r0["pivots"], rh["pivots"]) and np.max(np.abs(r0["z"] - rh["z"]) / scale) <= 1e-9


def test_large_nodes_declines_take_the_general_path(engine, oracle):
    """A node whose Qd has a block pivot below the threshold (here: Q = 0 in the first 3 variables) and a node with an equality
    row are declined by the blocked crash and solved by the general kernel behind it: same results as the checker."""
    from qpn_amd.engine import colmajor
    n, m, cnt = 80, 70, 4
    Q, R, qd, A, B, l, u = P.synth_nodes(9500, cnt, n, m, 4)
    w = P.shared_params(4)
    Q[1, :3, :] = 0.0; Q[1, :, :3] = 0.0                  # a singular leading block: the crash declines node 1
    A[1, :3, :] = 0.0; A[1, 0, 0] = 1.0; A[1, 1, 1] = 1.0; A[1, 2, 2] = 1.0      # ... which the bounds keep well-posed
    l[1, :3] = -0.5; u[1, :3] = 0.5
    u[2, 5] = l[2, 5]                                      # an equality row: node 2 is declined too
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    rc = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)
    rh = engine.solve_nodes(colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, w)
    assert np.array_equal(rh["status"], rc["status"])
    ok = rc["status"] == 1
    assert ok[0] and ok[3]
    scale = np.maximum(1.0, np.max(np.abs(rc["z"]), axis=1, keepdims=True))
    assert np.max((np.abs(rh["z"] - rc["z"]) / scale)[ok]) <= 1e-9
    assert np.array_equal(rh["active"][ok], rc["active"][ok])
