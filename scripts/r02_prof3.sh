#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r02_fp16 > gpurun_out/r02_fp16.log 2>&1
tail -4 gpurun_out/r02_fp16.log
timeout 1200 python scripts/bench_configs.py --tag r02 > gpurun_out/r02_configs.log 2>&1
tail -5 gpurun_out/r02_configs.log | cut -c1-400
