#!/bin/bash
# dev helper: kernel trace of a few search steps; prints the per-kernel timeline of the last step
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_$1; shift
rm -rf $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/scripts/kbench.py --steps 6 "$@" 2>&1 | grep -E "step|Error|error" 
python3 - $O <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prev = None
for r in rows[-16:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3 if prev else 0:7.1f} us")
    prev = e
PY
