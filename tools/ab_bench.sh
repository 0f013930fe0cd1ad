#!/usr/bin/env bash
# Developer aid: alternate bench.py runs over several libraries (QPN_HIP_LIB), print value / ms per step / kernel ms per run.
#   tools/ab_bench.sh <rounds> <lib> [<lib> ...] [-- bench args]
set -uo pipefail
rounds="$1"; shift
libs=(); while (( $# )) && [[ "$1" != "--" ]]; do libs+=("$1"); shift; done
(( $# )) && shift
mkdir -p gpurun_out
for r in $(seq 1 "$rounds"); do
  for lib in "${libs[@]}"; do
    tag="$(basename "$lib" .so)"
    QPN_HIP_LIB="$lib" python bench.py "$@" > "gpurun_out/ab_${tag}_$r.log" 2>&1 || { echo "$tag run $r failed"; tail -3 "gpurun_out/ab_${tag}_$r.log"; exit 1; }
    python - "$tag" "$r" "gpurun_out/ab_${tag}_$r.log" <<'PY'
import json, sys
tag, r, path = sys.argv[1:]
d = json.loads([x for x in open(path) if x.startswith("{")][-1])
print(f"{tag:24s} run {r}: {d['value']/1e6:8.3f} M  {d['ms_per_step']*1e3:8.2f} us/step  kernel {d['roofline'].get('kernel_ms', 0)*1e3:8.2f} us", flush=True)
PY
  done
done
