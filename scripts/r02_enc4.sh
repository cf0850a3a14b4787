#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_enc4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
f=$(ls $O/p/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
cd $GRAFT_REPO_ROOT; for i in 1 2; do python scripts/enc_bench.py 256 32 | grep encoder; done
