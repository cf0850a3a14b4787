#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_exp2; mkdir -p $O
L=$PWD/vietnamese_qa_system_amd/lib
VQA_LIB=$L/libvqa_retrieval_slot.so timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_slot.log 2>&1
tail -3 $O/pytest_slot.log
bash scripts/kb_variants.sh 3 default slot > $O/kb.log 2>&1
cat $O/kb.log
VQA_LIB=$L/libvqa_retrieval_slotst.so python scripts/stamp_timeline.py --show 2 --names start,s1a,s1b,s1end,bar1,s2a,s2b,bar2 > $O/stamps_slot.txt 2>&1
head -14 $O/stamps_slot.txt
