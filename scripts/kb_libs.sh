#!/bin/bash
# dev helper: kbench over library variants ("default" or a tag of lib/libvqa_retrieval_<tag>.so), R rounds interleaved.
# usage: kb_libs.sh R tag tag ... [-- kbench args]
R=$1; shift
tags=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd "$(dirname "$0")/.."
for r in $(seq $R); do
  for v in "${tags[@]}"; do
    lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval_$v.so
    [ "$v" = default ] && lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
    VQA_LIB=$lib python scripts/kbench.py --steps 20 "$@" 2>&1 | grep -v amdgpu.ids
  done
done
