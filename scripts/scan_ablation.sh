#!/bin/bash
# the sketch search's main launch (int8 slot loop) at B = 256 and B = 1 under -DVQA_ABLATE builds
# (bits: score_topk.hip top; since round 6 the slot loop honours every bit).  Results of the ablated builds are wrong: timing only.
# Build first: python scripts/build_variants.py --only=score_topk.hip ab4:VQA_ABLATE=4 ab8:VQA_ABLATE=8 ... (tags = ab<mask>)
# usage: scan_ablation.sh <rounds> mask mask ...   (mask 0 = the product library)
R=$1; shift
cd "$(dirname "$0")/.."
echo "# scan ablation: kbench.py, 10M x 768 fp16 shard, k = 10; 'main kernel' = HIP-event time of the dominant launch (7 050 880 sketch rows)"
echo "# bits: 1 no LDS-DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 32 no Q pieces, 64 no loop barriers"
for r in $(seq $R); do
  for m in "$@"; do
    lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval_ab$m.so
    [ "$m" = 0 ] && lib=$PWD/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
    VQA_LIB=$lib python scripts/kbench.py --steps 20 --b 256,1 2>&1 | grep -v amdgpu.ids | sed "s|libvqa_retrieval||"
  done
done
