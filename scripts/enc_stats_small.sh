#!/bin/bash
# dev helper: per-kernel averages of a small-batch encoder forward under rocprofv3 (usage: enc_stats_small.sh B L)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/enc_stats_small; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py ${1:-1} ${2:-32} > $O/p.log 2>&1
grep encoder $O/p.log
python3 - $O <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/p/*/*kernel_stats.csv')[0])))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:7.2f}  max {float(r['MaxNs'])/1e3:7.2f}")
PY
