"""Multi-GPU exchange without a collective: the solve kernel stores every primal block
into all replicas of the iterate (qpn_set_primal_mirrors) and qpn_sweep_status combines the stop/raise pair over
the ranks through mailboxes and is the barrier.

The GPU box has ONE GPU, so the cross-process part runs two ranks on the same device (gloo as the host channel
for the 64-byte IPC handles): same API calls, same kernels, same memory-ordering code as across xGMI; what it
cannot show is the link itself."""
import os
import socket

import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def _nodes(lo, cnt, n, m, p=3):
    from qpn_amd.engine import colmajor
    Q, R, qd, A, B, l, u = P.synth_nodes(lo, cnt, n, m, p)
    return (colmajor(Q), colmajor(R), qd, colmajor(A), colmajor(B), l, u, P.shared_params(p))


@pytest.mark.parametrize("n,m", [(32, 32), (7, 11), (40, 30)])      # fused kernel (full / ragged) and the general path
def test_mirror_stores_in_one_process(engine, n, m):
    """Two buffers of this process stand in for two GPUs' replicas: after one solve both hold the primal blocks,
    rows outside the written range stay untouched, and clearing the mirrors stops the replication."""
    import torch
    from qpn_amd.sharding import _device_tensor
    total, lo, cnt = 300, 40, 200
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    args = [t(a) for a in _nodes(lo, cnt, n, m)]
    a0, _ = engine.shared_alloc(total * n * 8)
    a1, _ = engine.shared_alloc(total * n * 8)
    a2, _ = engine.shared_alloc(total * n * 8)
    try:
        x0, x1, x2 = (_device_tensor(a, (total, n), "cuda:0") for a in (a0, a1, a2))
        assert float(x0.abs().sum()) == 0.0
        engine.set_primal_mirrors(a0, total * n * 8, [a1, a2])
        res = engine.solve_nodes(*args, x_out=x0[lo:lo + cnt])
        torch.cuda.synchronize()
        assert np.all(res["status"].cpu().numpy() == 1)
        want = np.zeros((total, n)); want[lo:lo + cnt] = res["z"].cpu().numpy()[:, :n]
        for x in (x0, x1, x2):
            assert np.array_equal(x.cpu().numpy(), want)
        # an x_out outside the registered buffer is not mirrored; neither is anything after clearing
        other = torch.zeros((cnt, n), dtype=torch.float64, device="cuda:0")
        x1.zero_()
        engine.solve_nodes(*args, x_out=other)
        engine.set_primal_mirrors()
        x2.zero_()
        engine.solve_nodes(*args, x_out=x0[lo:lo + cnt])
        torch.cuda.synchronize()
        assert np.array_equal(other.cpu().numpy(), want[lo:lo + cnt])
        assert float(x1.abs().sum()) == 0.0 and float(x2.abs().sum()) == 0.0
        with pytest.raises(Exception):
            engine.set_primal_mirrors(a0, total * n * 8, [a1] * 8)          # more than QPN_MAX_MIRRORS
        with pytest.raises(Exception):
            engine.set_primal_mirrors(a0, total * n * 8, [a0])              # a buffer cannot mirror itself
    finally:
        engine.set_primal_mirrors()
        del x0, x1, x2
        for a in (a0, a1, a2):
            engine.shared_free(a)


def test_sweep_status_local_and_timeout(engine):
    import torch
    from qpn_amd._lib import SWEEP_BOX_BYTES
    st = torch.ones(10_000, dtype=torch.int32, device="cuda:0")
    rs = torch.rand(10_000, dtype=torch.float64, device="cuda:0") * 1e-9
    out = torch.full((4,), -1.0, dtype=torch.float64, device="cuda:0"); out[3] = 0.0
    engine.sweep_status(st, rs, out)
    assert out.tolist() == [0.0, float(rs.max()), 1.0, 0.0]
    st[17] = 4; st[9_999] = 3; rs[5] = float("nan")
    engine.sweep_status(st, rs, out)
    o = out.tolist()
    assert o[0] == 2.0 and np.isnan(o[1]) and o[2:] == [1.0, 0.0]
    # two "ranks" whose peer never posts: the wait is bounded and reported
    b0, _ = engine.shared_alloc(SWEEP_BOX_BYTES, fine_grained=True)
    b1, _ = engine.shared_alloc(SWEEP_BOX_BYTES, fine_grained=True)
    try:
        rs[5] = 0.0
        engine.sweep_status(st, rs, out, rank=0, world=2, boxes=[b0, b1], epoch=1, timeout_ms=20)
        torch.cuda.synchronize()
        assert out.tolist()[2:] == [0.0, 1.0] and out.tolist()[0] == 2.0
        # the peer posts (same process, its own call): both sides now complete and agree
        st1 = torch.ones(50, dtype=torch.int32, device="cuda:0"); st1[3] = 2
        rs1 = torch.full((50,), 0.5, dtype=torch.float64, device="cuda:0")
        out1 = torch.zeros(4, dtype=torch.float64, device="cuda:0")
        engine.sweep_status(st1, rs1, out1, rank=1, world=2, boxes=[b0, b1], epoch=1, timeout_ms=2000)
        engine.sweep_status(st, rs, out, rank=0, world=2, boxes=[b0, b1], epoch=1, timeout_ms=2000)
        torch.cuda.synchronize()
        assert out.tolist() == [3.0, 0.5, 1.0, 1.0] and out1.tolist() == [3.0, 0.5, 1.0, 0.0]
    finally:
        engine.shared_free(b0); engine.shared_free(b1)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, total, n, m, sweeps, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qpn_amd
    from qpn_amd import sharding
    eng = qpn_amd.Engine(0)
    sh = sharding.SharedIterate(eng, dist, total, n, "cuda:0", timeout_ms=20_000)
    lo, hi = sharding.node_range(total, world, rank)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    args = [t(a) for a in _nodes(lo, hi - lo, n, m)]
    outs = []
    for s in range(sweeps):
        if s == 1 and rank == 1:
            args[2] = args[2] * 1.5                      # sweep 2 differs from sweep 1 on rank 1 only
        res = eng.solve_nodes(*args, x_out=sh.x[lo:hi])
        o = sh.finish_sweep(res["status"], res["resid"])
        # in stream order after finish_sweep the replica is complete: snapshot it WITHOUT a host barrier (the peer
        # may already be storing its next sweep -- into the other half)
        outs.append((sh.x_done.clone(), o.clone(), res["resid"].max().clone()))
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{f"x{s}": v[0].cpu().numpy() for s, v in enumerate(outs)},
             **{f"o{s}": v[1].cpu().numpy() for s, v in enumerate(outs)},
             **{f"r{s}": v[2].cpu().numpy() for s, v in enumerate(outs)})
    sh.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,n,m", [(2, 2000, 32, 32), (2, 333, 9, 6), (4, 4001, 32, 32)])
def test_ranks_share_the_iterate(engine, tmp_path, world, total, n, m):
    import torch
    import torch.multiprocessing as mp
    sweeps = 3
    mp.spawn(_rank_main, args=(world, _free_port(), total, n, m, sweeps, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # expectation: the same sweeps in this process, all nodes at once
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
    from qpn_amd import sharding
    lo1, hi1 = sharding.node_range(total, world, 1)
    base = [t(a) for a in _nodes(0, total, n, m)]
    for s in range(sweeps):
        args = list(base)
        if s >= 1:
            qd = base[2].clone(); qd[lo1:hi1] *= 1.5; args[2] = qd
        res = engine.solve_nodes(*args)
        torch.cuda.synchronize()
        x = res["z"].cpu().numpy()[:, :n]
        for r in range(world):
            assert np.array_equal(got[r][f"x{s}"], x), (s, r)                 # every replica, bit for bit
            o = got[r][f"o{s}"]
            assert o[0] == 0.0 and o[2] == 1.0 and o[3] == 0.0 and o[1] == float(res["resid"].max())
        assert max(float(got[r][f"r{s}"]) for r in range(world)) == got[0][f"o{s}"][1]
