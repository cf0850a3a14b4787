#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_enc2; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/slot -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/slot.log 2>&1
export VQA_GEMM_V1=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1 -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/v1.log 2>&1
for v in slot v1; do
f=$(ls $O/$v/*/*kernel_stats.csv | head -1)
echo "== $v"; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
