#!/usr/bin/env bash
set -euo pipefail
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pc${TAG:-}"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for sc in 1.0 3.0; do
  export SCALE=$sc
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d "$O/s$sc" -- python3 $R/tools/pivot_cost.py > "$O/s$sc.log" 2>&1
  grep SCALE "$O/s$sc.log"
  python3 - "$O/s$sc" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'avi_solve_schur' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: sum(v)/len(v)/10000 for k,v in d.items()})
PY
done
