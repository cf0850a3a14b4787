#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_enc7; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for t in -1 0 1 2 3 4; do
export VQA_GEMM_TILE=$t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$t -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/t$t.log 2>&1
echo "== forced tile $t: $(grep encoder $O/t$t.log)"
f=$(ls $O/t$t/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv,sys,re
for r in list(csv.DictReader(open(sys.argv[1]))):
    m=re.search(r'gemm_tile_kernelILi(\d)ELi(\d+)ELi(\d+)ELi\dELi\dELi(\d+)E', r['Name'])
    if m: print(f"   EPI{m.group(1)} {m.group(2)}x{m.group(3)} BK{m.group(4)}  calls {r['Calls']}  avg {float(r['AverageNs'])/1e3:.1f} us")
PY
done
