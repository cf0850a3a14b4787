#!/bin/bash
# The round's profile set, run on the GPU box in one call: fp16 + fp8 bench / rocprof / PMC sets (scripts/profile_round.sh), the
# BASELINE configs legs (scripts/bench_configs.py) and the encoder's kernel stats, padded and packed.  Copy what is to be judged
# from gpurun_out/ into profiles/ afterwards (scripts/collect_profiles.py; gpurun_out/ is scratch).
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh r02_fp16 > gpurun_out/r02_fp16.log 2>&1
tail -4 gpurun_out/r02_fp16.log
bash scripts/profile_round.sh r02_fp8 --dtype fp8 --no-cpu > gpurun_out/r02_fp8.log 2>&1
tail -2 gpurun_out/r02_fp8.log
timeout 1200 python scripts/bench_configs.py --tag r02 > gpurun_out/r02_configs.log 2>&1
tail -5 gpurun_out/r02_configs.log | cut -c1-300
O=$GRAFT_REPO_ROOT/gpurun_out/r02b_enc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/d.log 2>&1
grep encoder $O/d.log
cp $(ls $O/d/*/*kernel_stats.csv | head -1) $O/kernel_stats_padded.csv
export ENC_PACK=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats_packed.csv
rm -rf $O/d $O/p
# the real N = 2 path on ONE device (two ranks share cuda:0 over gloo: HIP search + all-gather + HIP merge as one program)
cd $GRAFT_REPO_ROOT
VQA_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --docs-per-gpu 5000000 --steps 50 --warmup 10 --e2e-steps 10 2> gpurun_out/r02_two_ranks.err | grep '^{' > gpurun_out/r02_two_ranks_one_gpu.json
cut -c1-400 gpurun_out/r02_two_ranks_one_gpu.json
timeout 1500 python scripts/stress_races.py > gpurun_out/r02_race_screen.txt 2>&1; tail -3 gpurun_out/r02_race_screen.txt
