#!/usr/bin/env bash
# Round-4 profile recipe (run on the GPU box through gpurun: `bash tools/profile_r04.sh <part>`, part = verify | bench | c5 | mid).
# Kernel traces and counter passes are separate runs (never --pmc together with tracing); the program itself follows `--`.
# Raw output under gpurun_out/prof_r04/; tools/summarize_r04.py turns it into the files under profiles/.
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_r04"; part="${1:-verify}"
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
[[ -x "$R/tools/fetch_calib" ]] || hipcc --offload-arch=gfx950 -O2 -o "$R/tools/fetch_calib" "$R/tools/fetch_calib.hip"
calib() {
  if [[ ! -d "$O/calib_fetch" ]]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/fetch_calib" > "$O/calib_fetch.log" 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- "$R/tools/fetch_calib" > "$O/calib_write.log" 2>&1
  fi
}
pmc_all() {   # pmc_all <tag> <command...>: HBM bytes (own passes) and three SQ sets
  local tag="$1"; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/${tag}_fetch" -- "$@" > "$O/${tag}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/${tag}_write" -- "$@" > "$O/${tag}_write.log" 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/${tag}_sq" -- "$@" > "$O/${tag}_sq.log" 2>&1 || true
  rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$O/${tag}_sq_b" -- "$@" > "$O/${tag}_sq_b.log" 2>&1 || true
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/${tag}_sq_c" -- "$@" > "$O/${tag}_sq_c.log" 2>&1 || true
}
if [[ $part == verify ]]; then
  # A8: the 32-class at the solution / perturbed / shrunk (one trace), counters at the solution; the other classes' rates
  python3 $R/tools/verify_rate.py > "$O/verify_rate.txt" 2>&1
  REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/verify_trace" -- python3 $R/tools/verify_rate.py > "$O/verify_trace.log" 2>&1
  for mode in 0 2; do
    MODE=$mode REPS=40 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/verify_trace_mode$mode" -- python3 $R/tools/verify_rate.py > "$O/verify_trace_mode$mode.log" 2>&1
  done
  calib
  MODE=0 REPS=10 pmc_all verify python3 $R/tools/verify_rate.py
  for shape in "48 48 4000" "64 64 4000" "64 96 2000" "128 200 1024" "256 256 512"; do
    set -- $shape
    N=$1 M=$2 CNT=$3 REPS=10 python3 $R/tools/verify_rate.py >> "$O/verify_rate_other.txt" 2>&1
    N=$1 M=$2 CNT=$3 REPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/verify_trace_$1x$2" -- python3 $R/tools/verify_rate.py > "$O/verify_trace_$1x$2.log" 2>&1
  done
fi
if [[ $part == bench ]]; then
  BENCH="python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-scaling-proxy --outer-loop-pairs 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 $R/bench.py --no-cpu-baseline --no-scaling-proxy --outer-loop-pairs 0 > "$O/trace.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-proxy --outer-loop-pairs 0 > "$O/trace20.log" 2>&1
  calib
  pmc_all bench $BENCH
fi
if [[ $part == c5 ]]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace5" -- python3 $R/bench.py --config 5 --no-cpu-baseline > "$O/trace5.log" 2>&1
  calib
  pmc_all c5 python3 $R/bench.py --config 5 --no-cpu-baseline --steps 12 --warmup 2
fi
if [[ $part == mid ]]; then
  # the mid-size classes (one wavefront per node up to 48, one workgroup per node up to 128) and the wide verify: traces + counters
  calib
  for n in 48 64 96 128; do
    REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/mid${n}_trace" -- python3 $R/tools/mid_rate.py $n > "$O/mid${n}_trace.log" 2>&1
    REPS=6 pmc_all mid$n python3 $R/tools/mid_rate.py $n
  done
  N=256 M=256 CNT=512 MODE=0 REPS=10 pmc_all vwide python3 $R/tools/verify_rate.py
fi
find "$O" -name "*kernel_stats.csv" | head -40
