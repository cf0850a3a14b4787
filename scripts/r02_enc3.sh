#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_enc3; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in default gns gng gnm gnsm; do
lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval_$v.so
[ "$v" = default ] && lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
VQA_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/$v.log 2>&1
f=$(ls $O/$v/*/*kernel_stats.csv | head -1)
echo "== $v"; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
