#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_enc6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_index_build.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do python scripts/enc_bench.py 256 32 | grep encoder; VQA_GEMM_TILE=2 python scripts/enc_bench.py 256 32 | grep encoder | sed 's/^/all-256x128 /'; done
python scripts/enc_bench.py 256 128 | grep encoder
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
f=$(ls $O/p/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r['Name'][:75].ljust(75), r['Calls'], r['AverageNs'], r['Percentage'])
PY
