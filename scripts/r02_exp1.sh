#!/bin/bash
# round-2 experiment 1: fine-interleaved K-step variants vs the default fp16 kernel (interleaved rounds) + parity of the variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_exp1; mkdir -p $O
bash scripts/kb_variants.sh 3 default fine fineo1 fineo2 prio fineprio > $O/kb.log 2>&1
VQA_LIB=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval_fine.so timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_fine.log 2>&1
tail -3 $O/pytest_fine.log
cat $O/kb.log
