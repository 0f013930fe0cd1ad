#!/usr/bin/env python3
"""bench.py -- node-AVI solves/sec (fp64) on the synthetic 10 000-node x 32-var QPNet.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1 under torch.distributed.run,
one rank per GPU, RCCL).  One STEP = one pass of the hot path over the whole net:
    per rank, for its contiguous node range:
        (A5+A6) qpn_assemble_nodes   per-node KKT blocks from the node records and the parameters w
        (A2+A3+A9) qpn_solve_avi_batch   every node-AVI from cold duals, post-check + active sets
    all ranks: RCCL all-gather of the primal blocks (the iterate x the outer loop needs), N > 1 only.
Inputs (node records) are resident in HBM before the timed region.  Total work is fixed at 10 000
nodes (BASELINE.json configs[3]) => "scaling": "strong".  value = solved node-AVIs / second over
all ranks (failed items do not count; there are none on this workload).

Extra objects on the JSON line: "roofline" (HBM bound; achieved = algorithmic bytes of SURVEY.md
section 8(d) per solve-kernel launch / mean launch duration, measured with HIP events on the launch
stream) and "cpu_baseline" (the CPU oracle -- a port, not PATH -- on the host cores, rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

NODES, NVAR, NCON, NPAR = 10_000, 32, 32, 8
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes", type=int, default=NODES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="assemble M in HBM, then solve (two kernels) instead of the fused pass")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL path even with one rank (testing)")
    ap.add_argument("--no-schedule", action="store_true",
                    help="natural node order (default: longest-first schedule hint, refreshed from the pivot counts every 16 steps)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        torch.cuda.set_device(0)
    dev = torch.device(f"cuda:{local_rank}")

    import qpn_amd
    from qpn_amd import sharding, synthetic
    from qpn_amd.engine import colmajor

    eng = qpn_amd.Engine(local_rank)
    n, m, p = NVAR, NCON, NPAR
    N = n + m
    lo_id, hi_id = sharding.node_range(args.nodes, world, rank)
    cnt = hi_id - lo_id

    # ---- this rank's node records, generated from the per-node Philox streams, then made resident
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo_id, cnt, n, m, p)
    w_host = synthetic.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    dQ, dR, dqd, dA, dB, dl, du, dw = (t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)),
                                       t(colmajor(B)), t(l), t(u), t(w_host))
    x_all = torch.zeros((args.nodes, n), dtype=torch.float64, device=dev)
    counts = [sharding.node_range(args.nodes, world, r) for r in range(world)]
    ev_pairs = []
    bufs = {"asm": None, "sol": None}      # output buffers are allocated once and reused

    def step(record):
        if args.unfused:
            bufs["asm"] = eng.assemble_nodes(dQ, dR, dqd, dA, dB, dl, du, dw, out=bufs["asm"])
            Mc, q, lo, hi, kind = bufs["asm"]
            res = bufs["sol"] = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind, out=bufs["sol"])   # cold start
            xloc = res["z"][:, :n].contiguous()
            if not use_dist:
                x_all[lo_id:hi_id].copy_(xloc)
        else:
            # (A5+A6+A2+A3+A9) one fused pass: KKT blocks assembled on the fly inside the solve kernel,
            # primal blocks written straight into this rank's rows of the iterate x
            xloc = x_all[lo_id:hi_id]
            res = bufs["sol"] = eng.solve_nodes(dQ, dR, dqd, dA, dB, dl, du, dw, out=bufs["sol"], x_out=xloc)
        if use_dist:
            sharding.all_gather_primal(x_all, xloc, counts, dist)
        return res

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Schedule hint (qpn_order_nodes_by_pivots): the outer loop sweeps the same nodes again and again, so the
    # pivot counts of one sweep order the next ones longest-first (shorter launch tail).  It is refreshed on the
    # device every REFRESH steps, INSIDE the timed region; it changes which wavefront solves which node, nothing else.
    REFRESH = 16
    use_sched = not (args.no_schedule or args.unfused)

    def maybe_refresh(i, res):
        if use_sched and res is not None and (i == 1 or i % REFRESH == 0):      # first sweep's counts, then periodically
            eng.order_nodes_by_pivots(res["pivots"])

    res = None
    for i in range(args.warmup):
        maybe_refresh(i, res)
        res = step(False)
    barrier()
    # ONE pair of HIP events brackets the whole timed region on the launch stream (a pair per step costs two
    # barrier packets per step, ~10 us of pipeline bubbles -- measured); the per-launch duration reported in
    # "roofline" is elapsed / steps, i.e. it also carries the near-empty fallback launch and the launch gaps
    # (and, for N > 1, the all-gather): an upper bound of the solve kernel's own duration.
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        maybe_refresh(i + args.warmup, res)
        res = step(True)
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    ev_pairs.append((ev0, ev1))

    solved_local = int((res["status"] == 1).sum().item())
    max_resid = float(res["resid"].max().item())
    tt = torch.tensor([dt, float(solved_local), max_resid], dtype=torch.float64, device=dev)
    if use_dist:
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, solved, max_resid = float(mx[0]), int(sm[1]), float(mx[2])
    else:
        solved = solved_local
    kern_ms = float(np.sum([a.elapsed_time(b) for a, b in ev_pairs])) / max(args.steps, 1) if ev_pairs else float("nan")

    if rank == 0:
        per_solve = synthetic.algorithmic_bytes(n, m)
        achieved = per_solve * cnt / (kern_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("avi_solve_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "node-AVI solves/sec (fp64) on synthetic N-node QPNet",
            "value": solved * args.steps / dt / 1.0,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.nodes}-node x {n}-var two-level QPNet "
                                   f"(n=m={n}, N_red={N}, p={p}; BASELINE.json configs[3]); step = "
                                   "KKT assembly + cold-start AVI solve + check + active sets ("
                                   + ("two kernels" if args.unfused else "one fused kernel") + ")"
                                   + (" + RCCL all-gather of primals" if use_dist else "")
                                   + ("; longest-first node schedule refreshed from the previous sweep's pivot counts every 16 steps" if use_sched else ""),
                       "nodes": args.nodes, "n": n, "m": m, "params": p,
                       "sharding": f"node ranges over {world} GPU(s)",
                       "max_resid": max_resid, "solved": solved},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "avi_solve_schur<nodes> (+ gated fallback launches)" if not args.unfused else "assemble + avi_solve", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_solve": per_solve, "solves_per_launch": cnt},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Q, R, qd, A, B, l, u, w_host)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(Q, R, qd, A, B, l, u, w):
    """The CPU oracle (a port -- NOT Julia+PATH, which cannot run here) on the same node records,
    all host cores, OpenMP over nodes.  Bounded: the 10 000-node batch, best of 3 passes."""
    from oracle import binding as ob
    ob.build()
    cnt, n = qd.shape
    m = l.shape[1]
    N = n + m
    Mc = np.zeros((cnt, N, N))           # column-major per item: Mc[b, j, i] = M[i, j]
    Mc[:, :n, :n] = np.swapaxes(Q, 1, 2)
    Mc[:, :n, n:] = np.swapaxes(A, 1, 2)     # M[n+r, j] = A[r, j]
    Mc[:, n:, :n] = -A                        # M[i, n+r] = -A[r, i]
    q = np.concatenate([qd + R @ w, B @ w], axis=1)
    lo = np.concatenate([np.full((cnt, n), -np.inf), l], axis=1)
    hi = np.concatenate([np.full((cnt, n), np.inf), u], axis=1)
    kind = np.concatenate([np.zeros((cnt, n), np.uint8), np.ones((cnt, m), np.uint8)], axis=1)
    z0 = np.zeros((cnt, N))
    Mc = np.ascontiguousarray(Mc.reshape(cnt, N * N))
    best = None
    cores = ob.num_threads()
    for _ in range(3):
        t0 = time.perf_counter()
        r = ob.solve_avi_batch_colmajor(Mc, N * N, q, lo, hi, z0, np.ascontiguousarray(kind), N, nthreads=cores)
        dt = time.perf_counter() - t0
        rate = (cnt - r["nfail"]) / dt
        best = rate if best is None else max(best, rate)
    return {"value": best, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"the same {cnt}-node batch (solve only, blocks pre-assembled), best of 3 passes, "
                      "CPU restatement oracle/qpn_oracle.c with OpenMP -- not Julia+PATH"}


if __name__ == "__main__":
    main()
