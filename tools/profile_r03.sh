#!/usr/bin/env bash
# Round-3 profile recipe (run on the GPU box through gpurun: `bash tools/profile_r03.sh [part]`, part = bench | sizes | c5 | all).
# Kernel traces and counter passes are separate runs (never --pmc together with tracing); the program itself follows `--`.
# Summaries are written under gpurun_out/prof_r03/ and copied into profiles/ by tools/summarize_r03.py.
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_r03"; part="${1:-all}"
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-scaling-proxy"
if [[ $part == bench || $part == all ]]; then
  # the bench: default command and the driver's command
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 $R/bench.py --no-cpu-baseline --no-scaling-proxy > "$O/trace.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-proxy > "$O/trace20.log" 2>&1
  # counters of the bench kernel: HBM bytes (FETCH_SIZE / WRITE_SIZE, own passes) and the SQ sets
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $BENCH > "$O/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $BENCH > "$O/pmc_write.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/pmc_sq" -- $BENCH > "$O/pmc_sq.log" 2>&1 || true
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$O/pmc_sq_b" -- $BENCH > "$O/pmc_sq_b.log" 2>&1 || true
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/pmc_sq_c" -- $BENCH > "$O/pmc_sq_c.log" 2>&1 || true
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/fetch_calib" > "$O/calib_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- "$R/tools/fetch_calib" > "$O/calib_write.log" 2>&1
fi
if [[ $part == sizes || $part == all ]]; then
  # the other size classes (resident-records route, one launch per sweep): 16 (compile-time shape), fused workgroup kernels
  CNT=10000 ROUTES=1 REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/small16" -- python3 $R/tools/mid_rate.py 16 > "$O/small16.log" 2>&1
  for nn in 33 48 64; do
    ROUTES=1 REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/mid$nn" -- python3 $R/tools/mid_rate.py $nn > "$O/mid$nn.log" 2>&1
  done
  for nn in 96 128; do
    ROUTES=1 REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/wg2_$nn" -- python3 $R/tools/mid_rate.py $nn > "$O/wg2_$nn.log" 2>&1
  done
  # the explicit-M node-shaped route (qpn_solve_avi_batch)
  REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/explicit64" -- python3 $R/tools/explicit_rate.py 32x32 > "$O/explicit64.log" 2>&1
  # HBM traffic of the fused mid-size kernel at n = m = 48 (4 000 nodes per launch)
  ROUTES=1 REPS=10 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_fetch_mid48.log" 2>&1
  ROUTES=1 REPS=10 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_write_mid48.log" 2>&1
  ROUTES=1 REPS=10 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d "$O/pmc_sq_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_sq_mid48.log" 2>&1 || true
  if [[ ! -d "$O/calib_fetch" ]]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/fetch_calib" > "$O/calib_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- "$R/tools/fetch_calib" > "$O/calib_write.log" 2>&1
  fi
fi
if [[ $part == c5 || $part == all ]]; then
  # BASELINE config 5: this round's route and round 2's (A/B on the same binary)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace5" -- python3 $R/bench.py --config 5 --no-cpu-baseline > "$O/trace5.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace5_route0" -- python3 $R/bench.py --config 5 --big-route 0 --no-cpu-baseline > "$O/trace5_route0.log" 2>&1
fi
find "$O" -name "*kernel_stats.csv" | head -30
