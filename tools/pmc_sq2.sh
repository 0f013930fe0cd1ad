#!/usr/bin/env bash
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_sq2"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d "$O/a" -- $BENCH > "$O/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU --output-format csv -d "$O/b" -- $BENCH > "$O/b.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d "$O/c" -- $BENCH > "$O/c.log" 2>&1
