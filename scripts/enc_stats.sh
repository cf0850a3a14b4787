#!/bin/bash
# dev helper: per-kernel averages of the packed / padded encoder forward under rocprofv3 (usage: enc_stats.sh [env assignments...])
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
O=$GRAFT_REPO_ROOT/gpurun_out/enc_stats; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
python3 - $O <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/p/*/*kernel_stats.csv')[0])))
for r in rows[:9]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
rm -rf $O
