#!/usr/bin/env bash
# Developer aid: kernel breakdown of the mid-size node path (n = m = NN, 4000 nodes) under rocprofv3 --kernel-trace --stats
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/midsize"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
export NN=${NN:-48}
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t$NN" -- python3 $R/tools/mid_one.py > "$O/t$NN.log" 2>&1
f=$(ls $O/t$NN/*/*kernel_stats.csv | head -1)
cut -c1-150 "$f" | head -12
tail -2 "$O/t$NN.log"
