#!/bin/bash
# dev helper: kbench over library variants, R rounds interleaved.  usage: kb_variants.sh R tag tag ... [-- kbench args]
R=$1; shift
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq $R); do
  for v in "${tags[@]}"; do
    lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval_$v.so
    [ "$v" = default ] && lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
    VQA_LIB=$lib python scripts/kbench.py --steps 20 "$@" 2>&1 | grep -v amdgpu.ids | sed "s|$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval_||"
  done
done
