#!/usr/bin/env bash
# Round 4: the fuzzers against the CPU oracle on the final library (run on the GPU box); output -> gpurun_out/r04_fuzz.txt
set -uo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04_fuzz.txt"; : > "$O"
run() { echo "===== $*" >> "$O"; timeout -k 10 "${TMO:-300}" "$@" 2>&1 | grep -v "amdgpu.ids\|Warning\|warning" | tail -n "${TAILN:-3}" >> "$O"; echo "rc=$?" >> "$O"; }
cd "$R"
run python3 tools/all_fuzz.py 1200 4041
OPTS=1 run python3 tools/all_fuzz.py 600 4042
run python3 tools/degenerate_fuzz.py
run python3 tools/wg2_fuzz.py
MAXDIM=32 run python3 tools/verify_fuzz.py
MAXDIM=96 run python3 tools/verify_fuzz.py
run python3 tools/bpp_fuzz.py 45
run python3 tools/pieces_fuzz.py
run python3 tools/pools_fuzz.py
run python3 tools/csc_fuzz.py
run python3 tools/refform_fuzz.py
cat "$O"
