set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/kt"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > "$O/log.txt" 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1); head -4 $f | cut -c1-160
