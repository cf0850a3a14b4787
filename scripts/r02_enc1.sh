#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_enc1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_index_build.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
python scripts/enc_bench.py 256 32 2>&1 | grep encoder
VQA_GEMM_V1=1 python scripts/enc_bench.py 256 32 2>&1 | grep encoder | sed 's/^/v1 /'
done
python scripts/enc_bench.py 256 128 2>&1 | grep encoder
VQA_GEMM_V1=1 python scripts/enc_bench.py 256 128 2>&1 | grep encoder | sed 's/^/v1 /'
