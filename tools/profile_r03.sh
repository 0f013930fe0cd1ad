#!/usr/bin/env bash
# Round-3 profile recipe (run on the GPU box through gpurun).  Kernel traces and counter passes are separate runs (never
# --pmc together with tracing).  Summaries are written under gpurun_out/prof_r03/ and copied into profiles/ by
# tools/summarize_r03.py.
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_r03"
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
# the bench: default command and the driver's command
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 $R/bench.py --no-cpu-baseline --no-scaling-proxy > "$O/trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-proxy > "$O/trace20.log" 2>&1
# the fused mid-size kernels (resident-records route, one launch per sweep)
for nn in 33 48 64; do
  ROUTES=1 REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/mid$nn" -- python3 $R/tools/mid_rate.py $nn > "$O/mid$nn.log" 2>&1
done
for nn in 96 128; do
  ROUTES=1 REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/wg2_$nn" -- python3 $R/tools/mid_rate.py $nn > "$O/wg2_$nn.log" 2>&1
done
# HBM traffic of the fused mid-size kernel at n = m = 48 (4 000 nodes per launch)
ROUTES=1 REPS=10 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_fetch_mid48.log" 2>&1
ROUTES=1 REPS=10 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_write_mid48.log" 2>&1
ROUTES=1 REPS=10 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d "$O/pmc_sq_mid48" -- python3 $R/tools/mid_rate.py 48 > "$O/pmc_sq_mid48.log" 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/fetch_calib" > "$O/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- "$R/tools/fetch_calib" > "$O/calib_write.log" 2>&1
find "$O" -name "*kernel_stats.csv" | head -20
