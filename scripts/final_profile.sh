#!/bin/bash
# The round's profile set, run on the GPU box in one call (usage: final_profile.sh [round tag, default r03]): fp16 + fp8 bench /
# rocprof / PMC sets (scripts/profile_round.sh), the encoder's kernel stats (padded and packed), the vendor-GEMM yardstick on the
# encoder's shapes beside gemm_tile_kernel (with the stamped K-step timeline), the K1 loop A/B, the N = 2 and N = 8 programs on
# ONE device (gloo) and the race screen.  Copy what is to be judged from gpurun_out/ into profiles/ afterwards
# (scripts/collect_profiles.py; gpurun_out/ is scratch).
R=${1:-r03}
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh ${R}_fp16 > gpurun_out/${R}_fp16.log 2>&1
tail -4 gpurun_out/${R}_fp16.log
# the same shard through the exact fp16 scan (no int8 sketch)
VQA_SKETCH=0 bash scripts/profile_round.sh ${R}_fp16_exact --no-cpu --no-other > gpurun_out/${R}_fp16_exact.log 2>&1
tail -4 gpurun_out/${R}_fp16_exact.log
bash scripts/profile_round.sh ${R}_fp8 --dtype fp8 --no-cpu --no-other > gpurun_out/${R}_fp8.log 2>&1
tail -2 gpurun_out/${R}_fp8.log
O=$GRAFT_REPO_ROOT/gpurun_out/${R}_enc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/d.log 2>&1
grep encoder $O/d.log
cp $(ls $O/d/*/*kernel_stats.csv | head -1) $O/kernel_stats_padded.csv
export ENC_PACK=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
grep encoder $O/p.log
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats_packed.csv
unset ENC_PACK
rm -rf $O/d $O/p
cd $GRAFT_REPO_ROOT
# the vendor GEMM on the encoder's shapes (yardstick) and gemm_tile_kernel's shapes / variants beside it, one process each
python scripts/probes/blaslt_shapes.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_blaslt_shapes.txt
{ python scripts/gemm_bench.py --no-build --tag dev --shapes=0,1,2,3,5,6,7,9,10,11 --rounds 3
  python scripts/gemm_bench.py --no-build --stamps --problems FFN1 --shapes=1
  python scripts/gemm_bench.py --no-build --stamps --problems FFN2 --shapes=3
  python scripts/gemm_bench.py --no-build --stamps --problems FFN1 --shapes=7 --names start,mma1,reads,dma,mma2,lgkm,vmcnt,bar; } 2>&1 | grep -v amdgpu > gpurun_out/${R}_encoder_gemm_variants.txt
tail -3 gpurun_out/${R}_encoder_gemm_variants.txt
python scripts/ab_loops.py VQA_SKETCH=1 VQA_SKETCH=0,VQA_F16_LOOP=0 VQA_SKETCH=0,VQA_F16_LOOP=1 VQA_SKETCH=0,VQA_F16_LOOP=2 VQA_SKETCH=0,VQA_STAGE_MIN=0 2>&1 | grep -v amdgpu > gpurun_out/${R}_k1_loop_ab.txt
cat gpurun_out/${R}_k1_loop_ab.txt
# the real N = 2 and N = 8 programs on ONE device (ranks share cuda:0 over gloo: HIP search + all-gather + HIP merge as one program)
for N in 2 8; do
  VQA_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N \
    bench.py --gpus $N --docs-per-gpu $((10000000 / N)) --steps 50 --warmup 10 --e2e-steps 10 2> gpurun_out/${R}_${N}_ranks.err | grep '^{' > gpurun_out/${R}_${N}_ranks_one_gpu.json
  cut -c1-300 gpurun_out/${R}_${N}_ranks_one_gpu.json
done
timeout 1500 python scripts/stress_races.py > gpurun_out/${R}_race_screen.txt 2>&1; tail -3 gpurun_out/${R}_race_screen.txt
