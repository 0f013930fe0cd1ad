#!/usr/bin/env python3
"""bench.py -- node-AVI solves/sec (fp64) on the synthetic 10 000-node x 32-var QPNet (BASELINE.json configs[3]).

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1: one rank per GPU under torch.distributed.run, RCCL;
started without a launcher, ``--gpus N`` starts the N ranks itself as child processes before anything touches the GPU).

One STEP = one sweep of the hot path over the whole net.  Per rank, for its contiguous node range, ONE launch:
    (A5+A6) per-node KKT blocks from the resident node records and this sweep's parameters w   } fused kernel
    (A2+A3+A9) every node-AVI from cold duals, post-check, active sets, primal write-back      } avi_solve_schur<nodes>
The node records are resident (qpn_nodes_upload, include/qpn_hip.h): between two sweeps of the outer loop
(src/algorithm.jl:13-117) only the parameters change.  Every step takes the NEXT parameter vector of a ring of 64
(w_k = w + 0.25 N(0,1), fixed seed): consecutive sweeps solve different problems (other right-hand sides, active sets and
pivot counts), as consecutive outer iterations do, so that nothing -- the longest-first schedule hint in particular, which is
refreshed from a sweep's own pivot counts every 16 sweeps inside the timed region -- is tuned to one repeated problem.
N > 1: every rank needs the whole iterate x for the next sweep.  Default exchange "rccl": ONE in-place RCCL all-gather per
sweep of [primal blocks | sweep status] (the stop/raise pair rides in the same message).  "--exchange p2p": the solve kernel
stores each primal block into every rank's replica over xGMI and a mailbox kernel ends the sweep (no collective; checked
against an all-gather after warm-up, and the run aborts on the first missed barrier).
Default N > 1 mode is STRONG scaling: the ONE 10 000-node net of BASELINE configs[3] sharded over the ranks; a second,
shorter measurement with 10 000 nodes per GPU (weak scaling) is reported in the "weak_scaling" object of the same line.
Before the W warm-up steps the bench runs `prewarm_steps` untimed steps (~0.1 s, reported in the JSON; `--no-prewarm` skips
them): the GPU leaves its idle power state only after ~0.1 s of work.
value = solved node-AVIs / second over all ranks (failed items do not count; there are none on this workload).

Extra objects on the JSON line: "roofline" (HBM bound; achieved = algorithmic bytes of SURVEY.md section 8(d) per launch /
mean launch duration, measured live with HIP events on the launch stream) and "cpu_baseline" (the CPU oracle -- a port, not
PATH -- on the host cores, rank 0, N = 1).  ``--config 5`` measures BASELINE configs[4] (512 nodes x 256 variables) instead.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NODES, NVAR, NCON, NPAR = 10_000, 32, 32, 8
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md:36
FP64_PEAK_TFLOPS = 78.6   # vector = matrix fp64 (datasheet; confirmed here: 64 cycles per v_mfma_f64_16x16x4, tools/mfma_rate)
PREWARM_STEPS = 800
RING = 64


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.12 s of timed region -- runs of a few ms read 3 % low (the first tens of ms after an idle period)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, choices=(4, 5), default=4,
                    help="4: 10 000 nodes x 32 variables (BASELINE configs[3], the metric's config); 5: 512 nodes x 256 variables")
    ap.add_argument("--sym-route", type=int, choices=(0, 1), default=1,
                    help="A/B: 1 (default) = resident records whose Qd blocks are all symmetric take the kernel variants that use it, 0 = never")
    ap.add_argument("--nodes", type=int, default=None, help="nodes in the whole net (strong) / per GPU (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: shard ONE net (strong, BASELINE configs[3]) or give every GPU its own (weak)")
    ap.add_argument("--exchange", choices=("rccl", "p2p"), default="rccl", help="N > 1: how the iterate is replicated")
    ap.add_argument("--no-second-scaling", action="store_true", help="N > 1: skip the second (other-mode) measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="assemble M in HBM, then solve (two kernels) instead of the fused pass")
    ap.add_argument("--per-call-records", action="store_true",
                    help="pass the node records with every call (qpn_solve_nodes_into) instead of the resident handle")
    ap.add_argument("--fixed-w", action="store_true", help="the same parameter vector every step (default: ring of 64)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the distributed path even with one rank (testing)")
    ap.add_argument("--no-prewarm", action="store_true",
                    help=f"skip the {PREWARM_STEPS} untimed steps (~0.1 s) that bring the GPU out of its idle power state")
    ap.add_argument("--no-scaling-proxy", action="store_true",
                    help="N = 1: skip the modelled_strong_scaling object (per-sweep time at 10 000 / 5 000 / 2 500 / 1 250 nodes on ONE GPU)")
    ap.add_argument("--outer-loop-pairs", type=int, default=250,
                    help="N = 1: leader-follower pairs (n = m = 32) of the net the outer_loop object runs through solve(); 0 skips it")
    ap.add_argument("--no-schedule", action="store_true",
                    help="natural node order (default: longest-first schedule refreshed from the pivot counts every 16 steps)")
    return ap.parse_args()


def maybe_spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as fresh child processes (nothing here has touched the GPU yet)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None and args.gpus > 1:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    if ws is not None and int(ws) != args.gpus and not args.force_dist:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)


def main():
    args = parse_args()
    maybe_spawn_ranks(args)
    import numpy as np
    import torch

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

    import qpn_amd
    eng = qpn_amd.Engine(local_rank)
    from qpn_amd import _lib as _qlib
    eng.set_option(_qlib.OPT_SYM_ROUTE, args.sym_route)
    env = dict(eng=eng, dist=dist, use_dist=use_dist, world=world, rank=rank, dev=torch.device(f"cuda:{local_rank}"),
               np=np, torch=torch)
    if args.config == 5:
        out = run_config5(env, args)
    else:
        nodes = args.nodes if args.nodes is not None else NODES
        first = run_case(env, args, nodes, args.scaling, args.steps, args.warmup, prewarm=not args.no_prewarm,
                         with_cpu=(world == 1 and not args.no_cpu_baseline))
        out = first
        if world > 1 and not args.no_second_scaling:
            other = "weak" if args.scaling == "strong" else "strong"
            second = run_case(env, args, nodes, other, min(args.steps, 200), min(args.warmup, 20), prewarm=False, with_cpu=False)
            if rank == 0:
                out[f"{other}_scaling"] = {k: second[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")} | {
                    "nodes": second["config"]["nodes"], "nodes_per_gpu": second["config"]["nodes_per_gpu"],
                    "exchange": second["config"]["exchange"], "frac": second["roofline"]["frac"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def run_case(env, args, nodes, scaling, steps, warmup, prewarm, with_cpu):
    """One measurement of the config-4 sweep; returns the JSON object (on every rank; rank 0's is printed)."""
    np, torch = env["np"], env["torch"]
    eng, dist, use_dist, world, rank, dev = env["eng"], env["dist"], env["use_dist"], env["world"], env["rank"], env["dev"]
    from qpn_amd import sharding, synthetic
    from qpn_amd.engine import colmajor

    n, m, p = NVAR, NCON, NPAR
    N = n + m
    total = nodes * world if scaling == "weak" else nodes
    lo_id, hi_id = sharding.node_range(total, world, rank)
    cnt = hi_id - lo_id

    # ---- this rank's node records, generated from the per-node Philox streams, then made resident
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo_id, cnt, n, m, p)
    w_host = synthetic.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    drec = (t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u))
    ring_host = np.repeat(w_host[None, :], RING, axis=0)
    if not args.fixed_w:
        ring_host = ring_host + 0.25 * np.random.Generator(np.random.Philox(key=[synthetic.SEED, 2 ** 41])).standard_normal((RING, p))
    ring = t(ring_host)
    use_handle = not (args.unfused or args.per_call_records)
    handle = eng.upload_nodes(*drec) if use_handle else None
    use_sched = not (args.no_schedule or args.unfused)
    REFRESH = 16
    if handle is not None:
        handle.set_schedule(REFRESH if use_sched else 0)     # the handle refreshes its own longest-first schedule
    elif not use_sched:
        eng.set_auto_schedule(0)
    else:
        eng.set_auto_schedule(REFRESH)

    counts = [sharding.node_range(total, world, r) for r in range(world)]
    shared = gathered = None
    exchange = "none"
    x_all = None
    if use_dist:
        exchange = "rccl" if args.unfused else args.exchange      # the replica stores belong to the fused kernel
        if exchange == "p2p":
            try:
                shared = sharding.SharedIterate(eng, dist, total, n, dev, timeout_ms=10_000)
            except RuntimeError as e:              # raised on ALL ranks together
                if rank == 0:
                    print(f"[bench] p2p exchange unavailable ({e}); using the RCCL all-gather", file=sys.stderr, flush=True)
                exchange = "rccl (p2p setup failed)"
        if shared is None:
            if len({hi - lo for lo, hi in counts}) == 1:
                gathered = sharding.GatheredIterate(eng, dist, total, n, dev)
            else:
                x_all = torch.zeros((total, n), dtype=torch.float64, device=dev)
    else:
        x_all = torch.zeros((total, n), dtype=torch.float64, device=dev)
    sweep_out = torch.zeros(4, dtype=torch.float64, device=dev)
    bufs = {"asm": None, "sol": None}      # output buffers are allocated once and reused
    kstep = [0]

    def step():
        w = ring[kstep[0] % RING]
        kstep[0] += 1
        if shared is not None:
            xloc = shared.x[lo_id:hi_id]
        elif gathered is not None:
            xloc = gathered.x_local
        else:
            xloc = x_all[lo_id:hi_id]
        if args.unfused:
            bufs["asm"] = eng.assemble_nodes(*drec, w, out=bufs["asm"])
            Mc, q, lo, hi, kind = bufs["asm"]
            res = bufs["sol"] = eng.solve_avi_batch(Mc, q, lo, hi, kind=kind, out=bufs["sol"])   # cold start
            xloc.copy_(res["z"][:, :n])
        elif handle is not None:
            # (A5+A6+A2+A3+A9) one fused pass over the resident records: KKT blocks assembled on the fly inside the solve
            # kernel, primal blocks written straight into this rank's rows of the iterate x
            res = bufs["sol"] = handle.solve(w, out=bufs["sol"], x_out=xloc)
        else:
            res = bufs["sol"] = eng.solve_nodes(*drec, w, out=bufs["sol"], x_out=xloc)
        if shared is not None:
            shared.finish_sweep(res["status"], res["resid"])       # mailbox exchange: status pair + barrier
        elif gathered is not None:
            gathered.finish_sweep(res["status"], res["resid"])     # ONE all-gather: primal blocks + status pair
        elif use_dist:
            sharding.all_gather_primal(x_all, xloc, counts, dist)
            eng.sweep_status(res["status"], res["resid"], sweep_out)
            dist.all_reduce(sweep_out, op=dist.ReduceOp.MAX)      # any failure anywhere / worst residual
        return res

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

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
        nonlocal shared, exchange, gathered, x_all
        if rank == 0:
            print(f"[bench] p2p exchange failed its self-check ({why}); using the RCCL all-gather", file=sys.stderr, flush=True)
        shared.close(); shared = None
        exchange = f"rccl (p2p self-check failed: {why})"
        if len({hi - lo for lo, hi in counts}) == 1:
            gathered = sharding.GatheredIterate(eng, dist, total, n, dev)
        else:
            x_all = torch.zeros((total, n), dtype=torch.float64, device=dev)

    def p2p_poll(where):
        """The mailbox waits are bounded (10 s each): a lost peer must end the run at once, not after steps x 10 s."""
        if shared is not None and float(shared.out[3]) != 0.0:
            print(f"[bench] rank {rank}: a sweep barrier of the p2p exchange was missed ({where}); aborting", file=sys.stderr, flush=True)
            os._exit(3)

    res = None
    if shared is not None:
        # probe first, with a short mailbox timeout: a route that does not work must cost seconds, not the warm-up
        barrier()
        shared.timeout_ms = 2_000
        res = step(); res = step()
        barrier()
        if p2p_ok():
            shared.timeout_ms = 10_000
            shared.out[3] = 0.0
        else:
            drop_p2p("probe")
    prewarm_steps = 0
    if prewarm:
        # Power state, not warm-up of the code: a timed region that starts less than ~0.1 s after the GPU's first launch
        # reads 3-4 % low whatever W is.  These steps are not counted in W or K and nothing of them is kept.
        for i in range(PREWARM_STEPS):            # a fixed count: every rank runs the same number of sweeps (mailbox epochs)
            res = step()
            if shared is not None and i % 100 == 99:
                p2p_poll("pre-warm")
        torch.cuda.synchronize()
        prewarm_steps = PREWARM_STEPS
    for i in range(warmup):
        res = step()
    barrier()
    if shared is not None:
        p2p_poll("warm-up")
        if warmup > 0 and not p2p_ok():
            drop_p2p("after warm-up")
            for i in range(min(warmup, 5)):
                res = step()
            barrier()
    # HIP events on the launch stream bracket the timed region (a pair per step costs two barrier packets per step, ~10 us of
    # pipeline bubbles -- measured): ev0 before the first step, evm behind the first step, ev1 behind the last.  `value` is the
    # wall clock over all K steps.  The per-launch duration reported in "roofline" is (ev1 - evm) / (K - 1): the launches that
    # run back to back -- the first launch of the region starts on a GPU the synchronize() before it left idle, behind the
    # host's launch latency, and that gap is not the kernel's.  It still carries the launch gaps, the schedule refreshes (and,
    # for N > 1, the exchange): an upper bound of the solve kernel's own duration.  ("kernel_ms_all" = (ev1 - ev0) / K.)
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    evm = torch.cuda.Event(enable_timing=True)
    # (torch creates the HIP event behind an Event object on its FIRST record(): each of the three is recorded once here, so that
    #  no event creation sits between the start of the clock and the first launch)
    for ev in (ev0, evm, ev1):
        ev.record()
    barrier()
    sweeps_before = handle.info()["sweeps"] if handle is not None else None      # (the window is NOT placed: a schedule
    t0 = time.perf_counter()                                                      #  re-sort falls where it falls, see below)
    ev0.record()
    for i in range(steps):
        res = step()
        if i == 0:
            evm.record()
        if shared is not None and i % REFRESH == REFRESH - 1:
            p2p_poll("timed region")
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms_all = ev0.elapsed_time(ev1) / max(steps, 1)
    kern_ms = evm.elapsed_time(ev1) / (steps - 1) if steps >= 2 else kern_ms_all

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
    info = handle.info() if handle is not None else None

    per_solve = synthetic.algorithmic_bytes(n, m)
    value = solved * steps / dt
    # The roofline fraction is the one that follows from `value` (per GPU): algorithmic bytes x solves/s / peak.  The
    # HIP-event time per launch is reported next to it ("kernel_ms", "frac_from_kernel_events"), never instead of it.
    achieved = per_solve * (value / world) / 1e9
    achieved_ev = per_solve * cnt / (kern_ms * 1e-3) / 1e9
    # schedule re-sorts of the resident-records handle inside the timed region (one 9-14 us single-workgroup launch each,
    # every 64 sweeps once settled, every 16 before): counted, not avoided
    resorts = None
    if sweeps_before is not None and use_sched:
        resorts = sum(1 for sw in range(sweeps_before, sweeps_before + steps) if sw % (16 if sw < 128 else 64) == 0)
    route = ("two kernels (assemble, solve)" if args.unfused else
             "one fused kernel over resident node records (qpn_solve_nodes_h)" if handle is not None else
             "one fused kernel + one gated general-kernel launch, records passed per call (qpn_solve_nodes_into)")
    if use_dist:
        xch = (" + primal blocks stored into every rank's replica of x by the solve kernel (xGMI p2p) + mailbox status/barrier kernel"
               if shared is not None else
               " + ONE in-place RCCL all-gather of [primal blocks | sweep status]" if gathered is not None else
               " + RCCL all-gather of primals + 4-double all-reduce")
    else:
        xch = ""
    out = {
        "metric": baseline_metric(),
        "value": value,
        "unit": "solves/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "prewarm_steps": prewarm_steps,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {total}-node x {n}-var two-level QPNet (n=m={n}, N_red={N}, p={p}; BASELINE.json "
                               f"configs[3]); step = KKT assembly + cold-start AVI solve + check + active sets + primal "
                               f"write-back, {route}{xch}; parameters: "
                               + ("the same vector every step" if args.fixed_w else f"ring of {RING} vectors, the next one every step")
                               + ("; longest-first node schedule refreshed from a sweep's own pivot counts every 16 steps" if use_sched else "; natural node order"),
                   "nodes": total, "nodes_per_gpu": cnt, "n": n, "m": m, "params": p,
                   "sharding": f"node ranges over {world} GPU(s)", "scaling": scaling, "exchange": exchange,
                   "w_ring": 1 if args.fixed_w else RING, "schedule": "longest-first/16" if use_sched else "natural",
                   "nodes_needing_general_kernel": (info["declined"] if info and info["decline_state"] >= 2 else None),
                   "qd_blocks_bitwise_symmetric": (bool(info.get("symmetric")) if info else None),
                   "schedule_resorts_in_timed_region": resorts,
                   "max_resid": max_resid, "solved": solved},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "frac_definition": "value / n_gpus x algorithmic_bytes_per_solve / peak (the driver's clock over all K steps)",
                     "kernel": ("avi_solve_schur<nodes" + (", symmetric Qd>" if info and info.get("symmetric") and n == 32 and m == 32 else ">"))
                               if not args.unfused else "assemble + avi_solve", "kernel_ms": kern_ms,
                     "frac_from_kernel_events": achieved_ev / HBM_PEAK_GBS,
                     "kernel_ms_all": kern_ms_all, "launches_averaged": max(steps - 1, 1),
                     "algorithmic_bytes_per_solve": per_solve, "solves_per_launch": cnt} | committed_counters(),
    }
    if with_cpu and rank == 0:
        out["cpu_baseline"] = cpu_baseline(np, Q, R, qd, A, B, l, u, w_host)
    if world == 1 and not use_dist and handle is not None and not args.no_scaling_proxy and total == NODES:
        out["modelled_strong_scaling"] = strong_scaling_proxy(env, drec, ring, n, m, kern_ms_all)
    if world == 1 and not use_dist and args.outer_loop_pairs > 0 and total == NODES:
        try:
            out["outer_loop"] = outer_loop(env, args.outer_loop_pairs, n, m)
        except Exception as e:      # the extra object must never cost the run its line
            out["outer_loop"] = {"error": f"{type(e).__name__}: {e}"}
    if handle is not None:
        handle.close()
    if shared is not None:
        shared.close()
    return out


def outer_loop(env, pairs, n, m):
    """BASELINE configs[3]'s NET through the kept API: `pairs` independent leader-follower pairs (n = m per node: 2 x pairs nodes
    on two levels, 2 n pairs variables) through algorithm.solve itself -- solve_base!'s sweeps (src/algorithm.jl:13-117) with the
    per-node map of process_qp (:44-52) and the level's AVI step (:95) served as level-wide batches (level_batch.py).  Reported:
    wall time per outer iteration (one process_level or solve_level sweep over a level), split into the C-ABI calls' own time
    (verify / solve / solution-graph pieces / the LP batches of remove_subsets and combine: staging + kernels + read-back, the
    calls are synchronous) and the host's (record assembly, polyhedral bookkeeping: Python), and the calls per sweep."""
    import warnings
    from qpn_amd import algorithm, examples, level_batch
    eng = env["eng"]
    sweeps = {"process": 0, "solve": 0}
    t_sweep = {"process": 0.0, "solve": 0.0}
    orig_p, orig_s = level_batch.process_level, level_batch.solve_level

    def proc(*a, **k):
        t = time.perf_counter(); r = orig_p(*a, **k); t_sweep["process"] += time.perf_counter() - t; sweeps["process"] += 1
        return r

    def solv(*a, **k):
        t = time.perf_counter(); r = orig_s(*a, **k); t_sweep["solve"] += time.perf_counter() - t; sweeps["solve"] += 1
        return r

    t0 = time.perf_counter()
    net = examples.setup("synthetic_pairs", pairs=pairs, n=n, m=m)
    t_setup = time.perf_counter() - t0
    eng.calls.clear(); eng.seconds.clear()
    algorithm.process_level = proc; level_batch.solve_level = solv
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            ret = algorithm.solve(net, engine=eng)
            wall = time.perf_counter() - t0
    finally:
        algorithm.process_level = orig_p; level_batch.solve_level = orig_s
    sec, calls = dict(eng.seconds), dict(eng.calls)
    grp = {"verify": ("qpn_verify_nodes", "qpn_verify_nodes_h"),
           "solve": ("qpn_solve_nodes_into", "qpn_solve_nodes_h", "qpn_solve_avi_batch", "qpn_assemble_pools"),
           "pieces": ("qpn_comp_indices", "qpn_recipes_batch", "qpn_reduced_pieces", "qpn_local_pieces"),
           "upload": ("qpn_nodes_upload",)}
    n_sweeps = max(1, sweeps["process"] + sweeps["solve"])
    dev_s = sum(sec.values())
    return {"net": f"synthetic_pairs: {pairs} leader-follower pairs, n = m = {n} per node ({2 * pairs} nodes on 2 levels, "
                   f"{2 * n * pairs} variables), through algorithm.solve on the HIP engine",
            "solved": bool(ret["solved"]), "error": ret.get("error"),
            "wall_s": wall, "setup_s": t_setup,
            "sweeps": sweeps, "ms_per_outer_iteration": wall / n_sweeps * 1e3,
            "ms_per_process_level_sweep": t_sweep["process"] / max(1, sweeps["process"]) * 1e3,
            "ms_per_solve_level_sweep": t_sweep["solve"] / max(1, sweeps["solve"]) * 1e3,
            "abi_ms_per_outer_iteration": {k: sum(sec.get(f, 0.0) for f in fs) / n_sweeps * 1e3 for k, fs in grp.items()},
            "abi_calls_per_outer_iteration": {k: sum(calls.get(f, 0) for f in fs) / n_sweeps for k, fs in grp.items()},
            "abi_share_of_wall": dev_s / wall if wall > 0 else None,
            "note": "solve also counts the node-solver calls behind the LP batches of remove_subsets / combine; the host share is "
                    "Python (record assembly, polyhedral bookkeeping)"}


def strong_scaling_proxy(env, drec, ring, n, m, full_ms):
    """What strong scaling of the ONE 10 000-node net can reach, measured on this GPU (no 8-GPU node needed to check the
    claim): the per-sweep time of the same resident-records route over the first 10 000 / G nodes, G = 1, 2, 4, 8 (what one
    rank of G does per sweep), plus the cost of the exchange that ends a sweep -- the status kernel every route launches and
    an in-place RCCL all-gather of [primal blocks | status] on a ONE-rank group (a lower bound of the G-rank collective: no
    link is crossed).  speedup_G = t(10 000) / (t(10 000 / G) + exchange)."""
    np, torch = env["np"], env["torch"]
    eng, dev = env["eng"], env["dev"]
    sizes = (10_000, 5_000, 2_500, 1_250)
    per = {}
    for cnt in sizes:
        h = eng.upload_nodes(*[a[:cnt] for a in drec])
        h.set_schedule(16)
        x = torch.zeros((cnt, n), dtype=torch.float64, device=dev)
        o = None
        for i in range(160):
            o = h.solve(ring[i % RING], out=o, x_out=x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 400
        e0.record()
        for i in range(K):
            o = h.solve(ring[i % RING], out=o, x_out=x)
        e1.record(); torch.cuda.synchronize()
        per[cnt] = e0.elapsed_time(e1) / K
        h.close()
    # the exchange's own cost
    st = torch.ones(1_250, dtype=torch.int32, device=dev); rs = torch.zeros(1_250, dtype=torch.float64, device=dev)
    so = torch.zeros(4, dtype=torch.float64, device=dev)
    for i in range(20):
        eng.sweep_status(st, rs, so)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        eng.sweep_status(st, rs, so)
    e1.record(); torch.cuda.synchronize()
    status_us = e0.elapsed_time(e1) / 200 * 1e3
    gather_us = None
    try:
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        buf = torch.zeros((1, 1_250 * n + 8), dtype=torch.float64, device=dev)       # one rank's row of the 8-rank message
        for i in range(20):
            dist.all_gather_into_tensor(buf, buf[0])
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            dist.all_gather_into_tensor(buf, buf[0])
        e1.record(); torch.cuda.synchronize()
        gather_us = e0.elapsed_time(e1) / 200 * 1e3
        if own:
            dist.destroy_process_group()
    except Exception as e:          # the proxy must never cost the run its line
        gather_us = None
        print(f"[bench] strong_scaling_proxy: one-rank RCCL all-gather not measured ({e})", file=sys.stderr, flush=True)
    # a 64-byte all-reduce on the one-rank group: what the status-only exchange of a net sharded BY CLUSTER costs per sweep
    # (sharding.solve_sharded: no cluster reads another's variables, so the iterate stays where it is until the end)
    allred_us = None
    try:
        import torch.distributed as dist
        own = not dist.is_initialized()
        if own:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        tiny = torch.zeros(8, dtype=torch.float64, device=dev)
        for i in range(20):
            dist.all_reduce(tiny)
        torch.cuda.synchronize()
        e0.record()
        for i in range(200):
            dist.all_reduce(tiny)
        e1.record(); torch.cuda.synchronize()
        allred_us = e0.elapsed_time(e1) / 200 * 1e3
        if own:
            dist.destroy_process_group()
    except Exception as e:
        print(f"[bench] scaling model: one-rank all-reduce not measured ({e})", file=sys.stderr, flush=True)
    exch_ms = (status_us + (gather_us or 0.0)) * 1e-3
    stat_ms = (status_us + (allred_us or 0.0)) * 1e-3
    t1 = per[10_000]
    return {"what": "a MODEL from single-GPU timings, NOT a measurement on N GPUs (no multi-GPU node was available to this run): "
                    "t(10 000 nodes) / (t(10 000 / G nodes) + exchange), every term measured on this one GPU",
            "route": "resident node records, one fused launch per sweep (the bench step), HIP-event time over 400 back-to-back sweeps",
            "ms_per_sweep": {str(k): v for k, v in per.items()},
            "exchange_us": {"sweep_status_kernel": status_us, "rccl_all_gather_one_rank_lower_bound": gather_us,
                            "rccl_all_reduce_64B_one_rank_lower_bound": allred_us},
            "modelled_speedup_all_gather_per_sweep": {str(g): t1 / (per[10_000 // g] + (exch_ms if g > 1 else 0.0)) for g in (1, 2, 4, 8)},
            "modelled_speedup_status_only_per_sweep": {str(g): t1 / (per[10_000 // g] + (stat_ms if g > 1 else 0.0)) for g in (1, 2, 4, 8)},
            "note": "strong scaling of this net is bound by the per-node dependent chain (one wavefront per node, ~30 us), "
                    "not by the exchange: 1 250 nodes do not fill one GPU's 4 096 resident wavefronts.  status-only: the net sharded "
                    "by cluster (whole leader-follower pairs per rank, sharding.solve_sharded) exchanges the 8-double status per "
                    "sweep and the iterate once at the end; all-gather: node-range sharding, the whole iterate every sweep"}


def committed_counters():
    """HBM traffic and fp64-pipe occupancy of the dominant kernel come from rocprofv3 --pmc passes (own runs: counters cannot
    be collected inside a timed run) -- the committed summary of the latest one, tagged with where it came from.  They
    describe the build they were measured on, not necessarily this run's."""
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        pj = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pj):
            continue
        try:
            d = json.load(open(pj))
            sq = d.get("avi_solve_schur_sq_per_launch", {})
            # (VALU-active + MFMA-busy cycles) / wall cycles per SIMD; SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count in
            # units of 4 cycles, 4 waves share a SIMD
            pipe = (4.0 * sq["SQ_ACTIVE_INST_VALU"] + sq["SQ_VALU_MFMA_BUSY_CYCLES"]) / (4.0 * sq["SQ_WAVE_CYCLES"] / 4.0)
            traffic = d.get("avi_solve_hbm_bytes_per_launch")
            if traffic is None:
                tj = os.path.join(ROOT, "profiles", "traffic.json")
                traffic = json.load(open(tj)).get("avi_solve_hbm_bytes_per_launch") if os.path.exists(tj) else None
            return {"traffic": traffic, "fp64_pipe_busy_frac": pipe,
                    "counters_source": f"profiles/{name} (committed rocprofv3 --pmc summary"
                                       + (f", measured at commit {d['commit']}" if d.get("commit") else "") + "; not measured by this run)"}
        except Exception:
            continue
    return {"traffic": None}


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


def cpu_baseline(np, Q, R, qd, A, B, l, u, w, budget_s=1.0):
    """The CPU oracle (a port -- NOT Julia+PATH, which cannot run here) on the same node records, the host cores of the
    job's CPU quota, OpenMP over nodes.  Bounded: whole passes over the batch until ~1 s of wall time."""
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
    while passes < 3 or (spent < budget_s and passes < 200):
        t0 = time.perf_counter()
        r = ob.solve_avi_batch_colmajor(Mc, N * N, q, lo, hi, z0, np.ascontiguousarray(kind), N, nthreads=cores)
        spent += time.perf_counter() - t0
        solved += cnt - r["nfail"]
        passes += 1
    return {"value": solved / spent, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"the same {cnt}-node batch (n=m={n}; solve only, blocks pre-assembled), {passes} passes = {spent:.2f} s "
                      f"wall on {cores} threads after one untimed pass, CPU restatement oracle/qpn_oracle.c with OpenMP "
                      "-- not Julia+PATH"}


def run_config5(env, args):
    """BASELINE configs[4]: 512 nodes x 256 variables (n = m = 256, N_red = 512), the large per-node KKT path (blocked MFMA
    crash, then Stage B by block principal pivoting with the delayed-update Lemke kernel behind it, csrc/qpn_avi_schur_big*.hip).
    Bound: fp64 matrix/vector pipe; flops per solve = N^3/3 (one factorisation) + 2 N^2 per Lemke pivot,
    the pivots counted by the Lemke kernel on the same records (one untimed sweep)."""
    np, torch = env["np"], env["torch"]
    eng, world, rank, dev = env["eng"], env["world"], env["rank"], env["dev"]
    from qpn_amd import sharding, synthetic
    from qpn_amd.engine import colmajor
    n = m = 256
    p = NPAR
    N = n + m
    total = args.nodes if args.nodes is not None else 512
    lo_id, hi_id = sharding.node_range(total, world, rank)
    cnt = hi_id - lo_id
    Q, R, qd, A, B, l, u = synthetic.synth_nodes(lo_id, cnt, n, m, p)
    w_host = synthetic.shared_params(p)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    drec = (t(colmajor(Q)), t(colmajor(R)), t(qd), t(colmajor(A)), t(colmajor(B)), t(l), t(u))
    ring_host = np.repeat(w_host[None, :], RING, axis=0)
    if not args.fixed_w:
        ring_host = ring_host + 0.25 * np.random.Generator(np.random.Philox(key=[synthetic.SEED, 2 ** 41])).standard_normal((RING, p))
    ring = t(ring_host)
    from qpn_amd import _lib as qlib
    # the work of a solve by SURVEY.md section 8(d)'s count is that of the pivotal method (N^3/3 + 2 N^2 per complementary pivot):
    # the pivots are counted ONCE, untimed, on the Lemke route (QPN_OPT_SYM_ROUTE = 0), so that the figure does not move with the
    # method Stage B runs in the timed loop (block principal pivoting switches ~220 pairs in ~7 rounds where Lemke makes ~100 pivots)
    eng.set_option(qlib.OPT_SYM_ROUTE, 0)
    handle = eng.upload_nodes(*drec)
    ref_piv = float(handle.solve(t(w_host))["pivots"].double().mean().item())
    handle.close()
    eng.set_option(qlib.OPT_SYM_ROUTE, args.sym_route)
    handle = eng.upload_nodes(*drec)
    steps = min(args.steps, 50)
    warmup = min(args.warmup, 5)
    out = None
    x = torch.zeros((cnt, n), dtype=torch.float64, device=dev)
    prewarm_steps = 0 if args.no_prewarm else 200         # ~0.7 s: the GPU's idle power state (see the module docstring)
    piv_sum = torch.zeros((), dtype=torch.float64, device=dev)
    for i in range(prewarm_steps + warmup):
        out = handle.solve(ring[i % RING], out=out, x_out=x)
        # (the warm-up runs the SAME step as the timed loop, pivot-count reduction included: torch loads that reduction kernel's
        #  module on first use -- tens of milliseconds that round 2's line carried inside its timed region)
        piv_sum += out["pivots"].sum()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    piv_sum.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        out = handle.solve(ring[(warmup + i) % RING], out=out, x_out=x)
        piv_sum += out["pivots"].sum()
    ev1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = ev0.elapsed_time(ev1) / steps
    solved = int((out["status"] == 1).sum().item())
    mean_piv = float(piv_sum.item()) / (steps * cnt)
    # the kernel's counter includes the n crash pivots (all free variables enter): those ARE the factorisation (N^3/3);
    # the 2 N^2 term is per complementary (Lemke) pivot after it
    lemke_piv = max(ref_piv - n, 0.0)
    flops = N ** 3 / 3.0 + 2.0 * N * N * lemke_piv
    achieved = flops * cnt / (ms * 1e-3) / 1e12
    # what the Lemke phase EXECUTES: an exchange on the m x (m + 1) Schur dictionary is 2 m (m + 1) flops, not the 2 N^2 of the
    # N x N formula (the crash took the n free variables out first); the factorisation term is left as the formula has it
    flops_exec = N ** 3 / 3.0 + 2.0 * m * (m + 1) * lemke_piv
    traffic = None; traffic_src = None
    try:
        pj = os.path.join(ROOT, "profiles", "r04_c5_pmc_summary.json")
        d5 = json.load(open(pj))
        if total == 512 and world == 1:
            traffic = d5.get("hbm_bytes_per_sweep")
            traffic_src = ("profiles/r04_c5_pmc_summary.json (committed rocprofv3 --pmc summary, sum over the sweep's five kernels"
                           + (f", measured at commit {d5['commit']}" if d5.get("commit") else "") + "; not measured by this run)")
    except Exception:
        pass
    res = {
        "metric": baseline_metric(), "value": solved * steps / dt, "unit": "solves/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "prewarm_steps": prewarm_steps, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic {total}-node x {n}-var QPNet (n=m={n}, N_red={N}, p={p}; BASELINE.json configs[4]); step = "
                               "KKT assembly + cold-start AVI solve + check + active sets + primal write-back over resident node "
                               f"records (qpn_solve_nodes_h -> blocked MFMA crash straight from the records + Stage B by "
                               + ("block principal pivoting, the delayed-update Lemke kernel behind it" if args.sym_route else "the delayed-update Lemke kernel")
                               + f"); ring of {RING} parameter vectors",
                   "nodes": total, "nodes_per_gpu": cnt, "n": n, "m": m, "params": p,
                   "mean_lemke_pivots": lemke_piv, "mean_lemke_pivots_source": "one untimed sweep of the same records on the Lemke route",
                   "mean_pivots_reported_by_timed_route": mean_piv,
                   "stage_b": "block principal pivoting (pivots = n + complementarity pairs switched)" if args.sym_route else "Lemke",
                   "max_resid": float(out["resid"].max().item()), "solved": solved},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                     "traffic": traffic, "counters_source": traffic_src,
                     "algorithmic_bytes_per_step": synthetic.algorithmic_bytes(n, m) * cnt,
                     "kernel": "schur_big2_convert + schur_big2_eliminate + schur_big2_sprod + schur_big_bpp / schur_big_lemke + schur_big2_finish (per step)", "kernel_ms": ms,
                     "flops_per_solve": flops, "solves_per_launch": cnt,
                     "flops_per_solve_with_schur_sized_pivots": flops_exec,
                     "frac_with_schur_sized_pivots": flops_exec * cnt / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
    }
    if world == 1 and not args.no_cpu_baseline and rank == 0:
        k = min(cnt, 16)
        res["cpu_baseline"] = cpu_baseline(np, Q[:k], R[:k], qd[:k], A[:k], B[:k], l[:k], u[:k], w_host, budget_s=5.0)
    handle.close()
    return res


if __name__ == "__main__":
    main()
