#!/bin/bash
# The round's profile set, run on the GPU box (usage: final_profile.sh [round tag, default r04] [part ...]; parts: search enc gemm
# sweeps ranks misc -- default all).  Needs the variant libraries built beforehand on the build host (they travel with the tree):
#   python -c "from vietnamese_qa_system_amd import build as b; b.build_variant('dev', ['VQA_DEV']); b.build_variant('dev_st', ['VQA_DEV', 'VQA_GSTAMPS']); b.build_variant('stamps', ['VQA_STAMPS=1'])"
# Everything lands under gpurun_out/; copy what is to be judged into profiles/ afterwards (scripts/collect_profiles.py).
R=${1:-r05}; shift
PARTS=${@:-search enc gemm sweeps ranks fullranks misc}
cd $GRAFT_REPO_ROOT
has() { [[ " $PARTS " == *" $1 "* ]]; }
LIB=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib

if has search; then
  # fp16 headline (sketch search), the same shard through the exact fp16 scan, fp8: bench line + rocprofv3 kernel stats + PMC passes each
  bash scripts/profile_round.sh ${R}_fp16 > gpurun_out/${R}_fp16.log 2>&1; tail -3 gpurun_out/${R}_fp16.log
  bash scripts/profile_round.sh ${R}_fp16_exact --no-sketch --no-cpu --no-other > gpurun_out/${R}_fp16_exact.log 2>&1; tail -3 gpurun_out/${R}_fp16_exact.log
  bash scripts/profile_round.sh ${R}_fp8 --dtype fp8 --no-cpu --no-other > gpurun_out/${R}_fp8.log 2>&1; tail -2 gpurun_out/${R}_fp8.log
  # kernel timeline of one sketch-search step, and the s_memtime-stamped timeline of the int8 slot loop (main scan, workgroup 5, tile 40)
  bash scripts/trace_steps_env.sh VQA_NOP 0 2>&1 | grep -v amdgpu > gpurun_out/${R}_step_timeline.txt
  VQA_LIB=$LIB/libvqa_retrieval_stamps.so python scripts/stamp_timeline.py --names start,work1,wait1,pre_bar,bar1,work2,issue,bar2 2>&1 | grep -v amdgpu > gpurun_out/${R}_stamps_sketch_loop.txt
  head -14 gpurun_out/${R}_stamps_sketch_loop.txt
fi

if has enc; then
  O=$GRAFT_REPO_ROOT/gpurun_out/${R}_enc; rm -rf $O; mkdir -p $O
  ( cd /tmp; export TMPDIR=/tmp
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/d.log 2>&1
    grep encoder $O/d.log; cp $(ls $O/d/*/*kernel_stats.csv | head -1) $O/kernel_stats_padded.csv
    export ENC_PACK=1
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/scripts/enc_bench.py 256 32 > $O/p.log 2>&1
    grep encoder $O/p.log; cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats_packed.csv
    rm -rf $O/d $O/p )
  ENC_PACK=1 bash scripts/encoder_pmc.sh > gpurun_out/${R}_encoder_pmc.log 2>&1; tail -8 gpurun_out/${R}_encoder_pmc.log
  cp gpurun_out/enc_pmc/summary.json gpurun_out/${R}_encoder_pmc_summary.json
fi

if has gemm; then
  # the vendor GEMM on the encoder's shapes (bare, with its own fused epilogue, and like for like with the element-wise kernels our
  # epilogues fold in) and gemm_tile_kernel's plain forms beside it, padded and packed row counts
  python scripts/probes/blaslt_shapes.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_blaslt_shapes.txt; cat gpurun_out/${R}_blaslt_shapes.txt
  { python scripts/gemm_bench.py --no-build --tag dev --shapes=-1 --rounds 3
    python scripts/gemm_bench.py --no-build --tag dev --shapes=-1 --rounds 3 --m 5114
    python scripts/gemm_bench.py --no-build --tag dev --stamps --problems FFN1 --shapes=5 --m 5114; } 2>&1 | grep -v amdgpu > gpurun_out/${R}_gemm_bench.txt
  tail -12 gpurun_out/${R}_gemm_bench.txt
fi

if has sweeps; then
  bash scripts/sweeps.sh $R > /dev/null 2>&1; tail -30 gpurun_out/${R}_sweeps.txt
  python scripts/probes/k_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_k_and_batch.txt; cat gpurun_out/${R}_k_and_batch.txt
  python scripts/probes/data_shapes_probe.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_data_shapes.txt; cat gpurun_out/${R}_data_shapes.txt
fi

if has ranks; then
  # the real N = 2 and N = 8 programs on ONE device (ranks share cuda:0 over gloo: HIP search + all-gather + HIP merge as one program)
  for N in 2 8; do
    VQA_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N \
      bench.py --gpus $N --docs-per-gpu $((10000000 / N)) --steps 48 --warmup 10 --e2e-steps 10 2> gpurun_out/${R}_${N}_ranks.err | grep '^{' > gpurun_out/${R}_${N}_ranks_one_gpu.json
    cut -c1-300 gpurun_out/${R}_${N}_ranks_one_gpu.json
  done
fi

if has fullranks; then
  # BASELINE configs[3] at FULL size as the real 8-rank program on ONE device: 8 shards of 10M x 768 fp16 (80M rows resident: rows + int8
  # sketch per shard, the row-major copy where the device still has room), ranks time-slice cuda:0 over gloo -- the workload's
  # correctness and plumbing at its real size, not a scaling point; and configs[4]: 8 shards of 12.5M x 768 fp8 (100M rows)
  VQA_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python bench.py --gpus 8 --steps 24 --warmup 6 --no-e2e \
      2> gpurun_out/${R}_configs3_full.err | grep '^{' > gpurun_out/${R}_configs3_80M_eight_ranks_one_gpu.json
  cut -c1-400 gpurun_out/${R}_configs3_80M_eight_ranks_one_gpu.json; tail -2 gpurun_out/${R}_configs3_full.err
  VQA_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python bench.py --gpus 8 --steps 24 --warmup 6 --no-e2e --dtype fp8 --docs-per-gpu 12500000 --verify-queries 4 \
      2> gpurun_out/${R}_configs4_full.err | grep '^{' > gpurun_out/${R}_configs4_100M_fp8_eight_ranks_one_gpu.json
  cut -c1-400 gpurun_out/${R}_configs4_100M_fp8_eight_ranks_one_gpu.json; tail -2 gpurun_out/${R}_configs4_full.err
fi

if has misc; then
  VQA_LIB=$LIB/libvqa_retrieval_dev.so python scripts/ab_loops.py VQA_SKETCH_SX=5 VQA_SKETCH_SX=6 VQA_SKETCH=0 2>&1 | grep -v amdgpu > gpurun_out/${R}_scan_ab.txt; cat gpurun_out/${R}_scan_ab.txt
  timeout 1500 python scripts/stress_races.py > gpurun_out/${R}_race_screen.txt 2>&1; tail -3 gpurun_out/${R}_race_screen.txt
fi
