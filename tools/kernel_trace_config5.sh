set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/kt5"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
CNT=512 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 $R/tools/config5_probe.py > "$O/log.txt" 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-140
