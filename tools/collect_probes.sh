#!/usr/bin/env bash
# Runs the developer probes this round's DESIGN.md quotes and collects their output (run on the GPU box).
set -uo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/probes_r01.txt"; : > "$O"
run() { echo "===== $*" >> "$O"; "$@" 2>&1 | grep -v "amdgpu.ids\|warning" | tail -n "${TAILN:-30}" >> "$O"; echo >> "$O"; }
cd "$R"
run ./tools/mfma_valu_overlap
run ./tools/rcp_probe
TAILN=24 FUSED=1 run python3 tools/stamp_profile.py
run bash tools/pivot_cost.sh
run python3 tools/lpt_probe.py
run python3 tools/verify_rate.py
run python3 tools/host_cost.py
CNT=512 run python3 tools/config5_probe.py
for n in 1250 2500 4096 5000 10000 20480 40960; do
  echo "===== bench.py --nodes $n" >> "$O"
  python3 bench.py --nodes $n --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('nodes', d['config']['nodes'], 'value', round(d['value']/1e6,2), 'M solves/s, ms_per_step', round(d['ms_per_step'],4), 'roofline frac', round(d['roofline']['frac'],4))" >> "$O"
done
cat "$O" | tail -5
