#!/usr/bin/env bash
# kernel trace of `bench.py --config 5` (run on the GPU box through gpurun); prints the top kernels
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/prof_c5"
rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o c5 -- python3 $R/bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > "$O/c5.log" 2>&1
tail -1 "$O/c5.log" | cut -c1-200
find "$O" -name "*kernel_stats.csv" | xargs head -9 | cut -c1-150
