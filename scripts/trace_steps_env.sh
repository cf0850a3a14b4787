#!/bin/bash
# dev helper: per-kernel timeline of the last search step under a few settings of one environment variable
# usage: trace_steps_env.sh VAR v1 v2 ...   (10M x 768 fp16, B = 256, k = 10 through scripts/kbench.py)
cd /tmp; export TMPDIR=/tmp
VAR=$1; shift
for V in "$@"; do
  O=$GRAFT_REPO_ROOT/gpurun_out/trace_env_$V; rm -rf $O
  export $VAR=$V
  echo "== $VAR=$V"
  rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py --steps 8 2>&1 | grep -E "step" | grep -v rocprofv3
  python3 - $O <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None
for r in rows[-15:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3 if prev else 0:7.1f} us")
    prev = e
PY
done
