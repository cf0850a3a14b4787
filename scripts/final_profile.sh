#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r02_fp16 > gpurun_out/r02_fp16.log 2>&1
tail -4 gpurun_out/r02_fp16.log
timeout 1200 python scripts/bench_configs.py --tag r02 > gpurun_out/r02_configs.log 2>&1
tail -5 gpurun_out/r02_configs.log | cut -c1-300
O=$GRAFT_REPO_ROOT/gpurun_out/r02b_enc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export ENC_PACK=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats_packed.csv
