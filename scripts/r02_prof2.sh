#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r02b_fp8 --dtype fp8 > gpurun_out/r02b_fp8.log 2>&1
tail -4 gpurun_out/r02b_fp8.log
O=$GRAFT_REPO_ROOT/gpurun_out/r02b_enc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
head -8 $O/kernel_stats.csv | cut -c1-150
