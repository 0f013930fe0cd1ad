#!/usr/bin/env bash
# Round-2 profile recipe (run on the GPU box through gpurun).  Counters are collected in their own
# passes (never together with tracing), as the guide prescribes.
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_r02"
mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline"
# the trace pass runs the DEFAULT bench command (200 timed steps): its average is the steady-state launch
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 $R/bench.py --no-cpu-baseline > "$O/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $BENCH > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $BENCH > "$O/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/pmc_sq" -- $BENCH > "$O/pmc_sq.log" 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d "$O/pmc_sq_b" -- $BENCH > "$O/pmc_sq_b.log" 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$O/pmc_sq_c" -- $BENCH > "$O/pmc_sq_c.log" 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/calib_fetch" -- "$R/tools/fetch_calib" > "$O/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/calib_write" -- "$R/tools/fetch_calib" > "$O/calib_write.log" 2>&1
find "$O" -name "*.csv" | head -40
# the driver's own command (--steps 20 --warmup 5) and the config-5 line, kernel traces only
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace20" -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/trace20.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace5" -- python3 $R/bench.py --config 5 --no-cpu-baseline > "$O/trace5.log" 2>&1
