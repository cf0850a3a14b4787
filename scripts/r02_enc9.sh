#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$PWD/vietnamese_qa_system_amd/lib
for r in 1 2; do
python scripts/enc_bench.py 256 32 | grep encoder | sed 's/^/plain /'
for v in st1 st2 st3; do VQA_LIB=$L/libvqa_retrieval_$v.so python scripts/enc_bench.py 256 32 | grep encoder | sed "s/^/$v /"; done
done
