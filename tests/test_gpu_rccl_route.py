"""The exchange route north_star names -- an RCCL all-gather of the primal blocks -- end to end on device tensors:
sharding.GatheredIterate (ONE in-place all_gather_into_tensor per sweep carrying [primal blocks | sweep status]) under the
"nccl" backend (= RCCL on ROCm), with the solve kernel writing straight into the send row.

The round's GPU box has one GPU: the one-rank case runs there (it is still RCCL's in-place all-gather on device memory and
the whole bench path around it); the two-rank case needs two GPUs (RCCL refuses two ranks on one device) and is skipped
otherwise -- it is the test a multi-GPU lease runs.  CPU counterpart with two ranks: tests/test_sharding_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
import numpy as np
import torch, torch.distributed as dist
sys.path[:0] = [os.environ["QPN_ROOT"], os.path.join(os.environ["QPN_ROOT"], "tests")]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{rank}"))
import qpn_amd
from qpn_amd import sharding, synthetic
from qpn_amd.engine import colmajor
eng = qpn_amd.Engine(rank)
total, n, m, sweeps = int(os.environ["QPN_TOTAL"]), 32, 32, 3
lo, hi = sharding.node_range(total, world, rank)
Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo, hi - lo, n, m)
dev = f"cuda:{rank}"
t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
nodes = eng.upload_nodes(t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u))
it = sharding.GatheredIterate(eng, dist, total, n, dev)
w0 = t(synthetic.shared_params())
xs = []
for k in range(sweeps):
    # the next sweep's parameters are read from the WHOLE gathered iterate, on the device
    w = w0 if k == 0 else w0 + 0.1 * it.x_all()[:: total // 8, 0][:8]
    res = nodes.solve(w.contiguous(), x_out=it.x_local)
    it.finish_sweep(res["status"], res["resid"])
    nfail, maxres = it.sweep_result()
    assert nfail == 0 and maxres <= 1e-8, (nfail, maxres)
    xs.append(it.x_all().cpu().numpy())
if rank == world - 1:
    np.save(os.environ["QPN_OUT"], np.stack(xs))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_ranks(world, total, tmp_path):
    out = str(tmp_path / "xs.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), QPN_ROOT=ROOT, QPN_TOTAL=str(total), QPN_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def _single_process_reference(oracle, total, sweeps=3):
    import problems as P
    n, m = 32, 32
    Q, R, qd, A, B, l, u = P.synth_nodes(0, total, n, m)
    w0 = P.shared_params()
    xs, x = [], None
    for k in range(sweeps):
        w = w0 if k == 0 else w0 + 0.1 * x[:: total // 8, 0][:8]
        M, q, lo, hi, kind = P.reduced_blocks(Q, R, qd, A, B, l, u, w)
        x = oracle.solve_avi_batch(M, q, lo, hi, kind=kind)["z"][:, :n]
        xs.append(x)
    return np.stack(xs)


def test_rccl_all_gather_route_one_rank(oracle, tmp_path):
    total = 64
    xs = _run_ranks(1, total, tmp_path)
    ref = _single_process_reference(oracle, total)
    assert np.max(np.abs(xs - ref)) <= 1e-9


def test_rccl_all_gather_route_two_ranks(oracle, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    total = 128
    xs = _run_ranks(2, total, tmp_path)
    ref = _single_process_reference(oracle, total)
    assert np.max(np.abs(xs - ref)) <= 1e-9


def test_bench_distributed_path_prints_the_contract_line(tmp_path):
    """bench.py through its N > 1 code path (one rank, --force-dist): RCCL route, explicit scaling / exchange fields,
    pre-warm disclosed."""
    env = dict(os.environ, MASTER_PORT=str(_free_port()), MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "3", "--warmup", "1",
                        "--nodes", "512", "--no-prewarm", "--no-cpu-baseline"], env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["config"]["exchange"] == "rccl" and d["config"]["scaling"] == "strong" and d["scaling"] == "strong"
    assert d["prewarm_steps"] == 0 and d["config"]["solved"] == 512 and d["config"]["max_resid"] <= 1e-8
    assert "ONE in-place RCCL all-gather" in d["config"]["workload"]
