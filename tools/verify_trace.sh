# rocprofv3 kernel trace of tools/verify_rate.py (A8: qpn_verify_nodes at the solution and off it); summary -> gpurun_out/vt/
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/vt"; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/verify_rate.py > "$O/rate.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 $R/tools/verify_rate.py > "$O/log.txt" 2>&1
f=$(ls $O/*/*kernel_stats.csv | head -1); cp "$f" "$O/kernel_stats.csv"; head -12 "$O/kernel_stats.csv" | cut -c1-200; cat "$O/rate.txt"
