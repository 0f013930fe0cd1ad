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
    nodes = engine.upload_nodes(*dev[:-1])
    out = nodes.solve(dev[-1])
    torch.cuda.synchronize()
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
