#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r02a_fp16 > gpurun_out/r02a_fp16.log 2>&1
tail -5 gpurun_out/r02a_fp16.log
timeout 1200 python scripts/bench_configs.py --tag r02a > gpurun_out/r02a_configs.log 2>&1
tail -8 gpurun_out/r02a_configs.log
