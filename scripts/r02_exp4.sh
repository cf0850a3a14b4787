#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_exp4; mkdir -p $O
L=$PWD/vietnamese_qa_system_amd/lib
for v in e1 e2 e3 e4; do
VQA_LIB=$L/libvqa_retrieval_$v.so timeout 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -1
done
bash scripts/kb_variants.sh 3 default e1 e2 e3 e4 > $O/kb.log 2>&1
cat $O/kb.log
