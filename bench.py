#!/usr/bin/env python3
"""bench.py -- node-AVI solves/sec (fp64) on the synthetic 10 000-node x 32-var QPNet.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1 under torch.distributed.run,
one rank per GPU, RCCL).  One STEP = one pass of the hot path over the whole net:
    per rank, for its contiguous node range:
        (A5+A6) qpn_assemble_nodes   per-node KKT blocks from the node records and the parameters w
        (A2+A3+A9) qpn_solve_avi_batch   every node-AVI from cold duals, post-check + active sets
    N > 1 only: every rank needs the whole iterate x for the next sweep.  Default exchange "p2p": the solve
    kernel itself stores each primal block into every rank's replica of x over xGMI (IPC-shared buffers,
    qpn_set_primal_mirrors) and the sweep ends with qpn_sweep_status -- a 24-byte mailbox exchange that is the
    stop/raise decision and the barrier; no collective on the data path.  "--exchange rccl" (and the automatic
    fallback when buffers cannot be shared or the warm-up self-check against an RCCL all-gather fails):
    all-gather of the primal blocks + a 2-double all-reduce.
Before the W warm-up steps the bench runs 800 untimed steps (~0.1 s; `--no-prewarm` skips them): the GPU leaves its idle
power state only after ~0.1 s of work, and a timed region that starts earlier reads 3-10 % low whatever the code does.
Inputs (node records) are resident in HBM before the timed region.  Independent node-AVIs partition over the
ranks, so per-GPU work is fixed at 10 000 nodes (BASELINE.json configs[3] on every GPU; the N-GPU net has
N x 10 000 nodes) => "scaling": "weak"; "--scaling strong" shards ONE 10 000-node net instead (latency-bound:
a node-AVI is a ~40 us dependent pivot chain whatever the batch).  value = solved
node-AVIs / second over all ranks (failed items do not count; there are none on this workload).

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
    # defaults: 0.12 s of timed region -- runs of a few ms read 3 % low (the first tens of ms after an idle period)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--nodes", type=int, default=NODES, help="nodes per GPU (weak) / in the whole net (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--exchange", choices=("p2p", "rccl"), default="p2p", help="N > 1: how the iterate is replicated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="assemble M in HBM, then solve (two kernels) instead of the fused pass")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL path even with one rank (testing)")
    ap.add_argument("--no-prewarm", action="store_true",
                    help="skip the 800 untimed steps (~0.1 s) that bring the GPU out of its idle power state before the W warm-up steps")
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
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)            # --force-dist without a launcher
        backend = os.environ.get("QPN_BENCH_BACKEND", "nccl")      # "gloo": rehearsal with several ranks on ONE GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device(f"cuda:{local_rank}")

    import qpn_amd
    from qpn_amd import sharding, synthetic
    from qpn_amd.engine import colmajor

    eng = qpn_amd.Engine(local_rank)
    n, m, p = NVAR, NCON, NPAR
    N = n + m
    total = args.nodes * world if args.scaling == "weak" else args.nodes
    lo_id, hi_id = sharding.node_range(total, world, rank)
    cnt = hi_id - lo_id

    # ---- this rank's node records, generated from the per-node Philox streams, then made resident
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo_id, cnt, n, m, p)
    w_host = synthetic.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    dQ, dR, dqd, dA, dB, dl, du, dw = (t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)),
                                       t(colmajor(B)), t(l), t(u), t(w_host))
    counts = [sharding.node_range(total, world, r) for r in range(world)]
    shared = None
    exchange = "none"
    if use_dist:
        exchange = "rccl" if args.unfused else args.exchange      # the replica stores belong to the fused kernel
        if exchange == "p2p":
            try:
                shared = sharding.SharedIterate(eng, dist, total, n, dev, timeout_ms=10_000)
            except RuntimeError as e:              # raised on ALL ranks together
                if rank == 0:
                    print(f"[bench] p2p exchange unavailable ({e}); using RCCL collectives", file=sys.stderr, flush=True)
                exchange = "rccl (p2p setup failed)"
    x_all = torch.zeros((total, n), dtype=torch.float64, device=dev) if shared is None else None
    sweep_out = torch.zeros(4, dtype=torch.float64, device=dev)
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
            xloc = (shared.x if shared is not None else x_all)[lo_id:hi_id]
            res = bufs["sol"] = eng.solve_nodes(dQ, dR, dqd, dA, dB, dl, du, dw, out=bufs["sol"], x_out=xloc)
        if shared is not None:
            shared.finish_sweep(res["status"], res["resid"])       # mailbox exchange: status pair + barrier
        elif use_dist:
            sharding.all_gather_primal(x_all, xloc, counts, dist)
            eng.sweep_status(res["status"], res["resid"], sweep_out)
            dist.all_reduce(sweep_out, op=dist.ReduceOp.MAX)      # any failure anywhere / worst residual
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
    if not use_sched:
        eng.set_auto_schedule(0)               # natural node order: also switch the context's own refresh off

    def maybe_refresh(i, res):
        if use_sched and res is not None and (i == 1 or i % REFRESH == 0):      # first sweep's counts, then periodically
            eng.order_nodes_by_pivots(res["pivots"])

    def p2p_ok():
        """Self-check of the p2p route (outside the timed region): every replica equals what a collective all-gather of
        the same blocks gives, and the last mailbox wait was served -- on every rank."""
        xd = shared.x_done.clone()
        chk = torch.zeros_like(xd)
        sharding.all_gather_primal(chk, xd[lo_id:hi_id], counts, dist)
        okv = torch.tensor([float(torch.equal(chk, xd) and float(shared.out[2]) == 1.0)], dtype=torch.float64, device=dev)
        dist.all_reduce(okv, op=dist.ReduceOp.MIN)
        return float(okv) == 1.0 and not os.environ.get("QPN_BENCH_FORCE_FALLBACK")      # (the env: fallback rehearsal)

    def drop_p2p(why):
        nonlocal shared, exchange, x_all
        if rank == 0:
            print(f"[bench] p2p exchange failed its self-check ({why}); using RCCL collectives", file=sys.stderr, flush=True)
        shared.close(); shared = None
        exchange = f"rccl (p2p self-check failed: {why})"
        x_all = torch.zeros((total, n), dtype=torch.float64, device=dev)

    res = None
    if shared is not None:
        # probe first, with a short mailbox timeout: a route that does not work must cost seconds, not the warm-up
        barrier()
        shared.timeout_ms = 2_000
        res = step(False); res = step(False)
        barrier()
        if p2p_ok():
            shared.timeout_ms = 10_000
        else:
            drop_p2p("probe")
    if not args.no_prewarm:
        # Power state, not warm-up of the code: a timed region that starts less than ~0.1 s after the GPU's first launch
        # reads 3-4 % low whatever W is (200 steps after 20: 81.5 M, after 1 000: 84.9 M -- same binary).  These
        # steps are not counted in W or K and nothing of them is kept.
        for _ in range(800):                   # ~0.1 s; a fixed count: every rank runs the same number of sweeps (mailbox epochs)
            res = step(False)
        torch.cuda.synchronize()
        res = None
    for i in range(args.warmup):
        maybe_refresh(i, res)
        res = step(False)
    barrier()
    if shared is not None:
        if args.warmup > 0 and not p2p_ok():
            drop_p2p("after warm-up")
            for i in range(min(args.warmup, 5)):
                res = step(False)
            barrier()
        else:
            shared.out[3] = 0.0                # start-up skew may have cost a wait before; the timed region may not miss one
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

    sweep_ok = shared is None or float(shared.out[3]) == 0.0        # no barrier of the timed region was missed
    solved_local = int((res["status"] == 1).sum().item()) if sweep_ok else 0      # a missed barrier voids the run
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
        # fp64 pipe utilisation of the dominant kernel from the committed PMC pass (rocprofv3 --pmc, own run):
        # (VALU-active + MFMA-busy cycles) / wall cycles per SIMD; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count in
        # units of 4 cycles, 4 waves share a SIMD
        pipe = None
        pj = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pj):
            try:
                sq = json.load(open(pj)).get("avi_solve_schur_sq_per_launch", {})
                pipe = (4.0 * sq["SQ_ACTIVE_INST_VALU"] + sq["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * sq["SQ_WAVE_CYCLES"] / 4.0)
            except Exception:
                pipe = None
        out = {
            "metric": baseline_metric(),
            "value": solved * args.steps / dt / 1.0,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {total}-node x {n}-var two-level QPNet "
                                   f"(n=m={n}, N_red={N}, p={p}; BASELINE.json configs[3]); step = "
                                   "KKT assembly + cold-start AVI solve + check + active sets ("
                                   + ("two kernels" if args.unfused else "one fused kernel") + ")"
                                   + ((" + primal blocks stored into every rank's replica of x by the solve kernel (xGMI p2p) + mailbox status/barrier kernel"
                                       if shared is not None else " + RCCL all-gather of primals + 2-double all-reduce") if use_dist else "")
                                   + ("; longest-first node schedule refreshed from the previous sweep's pivot counts every 16 steps" if use_sched else ""),
                       "nodes": total, "nodes_per_gpu": cnt, "n": n, "m": m, "params": p,
                       "sharding": f"node ranges over {world} GPU(s)", "exchange": exchange,
                       "max_resid": max_resid, "solved": solved},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "avi_solve_schur<nodes> (+ gated fallback launches)" if not args.unfused else "assemble + avi_solve", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_solve": per_solve, "solves_per_launch": cnt,
                         "fp64_pipe_busy_frac": pipe},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Q, R, qd, A, B, l, u, w_host)
        print(json.dumps(out), flush=True)
    if use_dist:
        if shared is not None:
            shared.close()
        dist.destroy_process_group()


def baseline_metric():
    """The metric string of BASELINE.json (the driver compares against it)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "node-AVI solves/sec (fp64) on synthetic N-node QPNet, 1/2/4/8 GPUs"


def host_cores(omp_threads):
    """Threads the CPU baseline may really use: OpenMP's count, capped by the affinity mask and by the cgroup CPU quota
    (on the GPU box the job owns 16 of the host's 256 hardware threads: more threads only burst and get throttled)."""
    c = omp_threads
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            c = min(c, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, int(c))


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
    cores = host_cores(ob.num_threads())
    # one untimed pass (thread pool, page faults), then whole passes until ~1 s of wall time = tens of core-seconds
    ob.solve_avi_batch_colmajor(Mc, N * N, q, lo, hi, z0, np.ascontiguousarray(kind), N, nthreads=cores)
    solved, spent, passes = 0, 0.0, 0
    while passes < 5 or (spent < 1.0 and passes < 200):
        t0 = time.perf_counter()
        r = ob.solve_avi_batch_colmajor(Mc, N * N, q, lo, hi, z0, np.ascontiguousarray(kind), N, nthreads=cores)
        spent += time.perf_counter() - t0
        solved += cnt - r["nfail"]
        passes += 1
    best = solved / spent
    return {"value": best, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"the same {cnt}-node batch (solve only, blocks pre-assembled), {passes} passes = {spent:.2f} s "
                      f"wall on {cores} threads after one untimed pass, CPU restatement oracle/qpn_oracle.c with OpenMP "
                      "-- not Julia+PATH"}


if __name__ == "__main__":
    main()
