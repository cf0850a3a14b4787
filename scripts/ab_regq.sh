#!/bin/bash
# A/B of the two sketch scan kernels on the headline shard: interleaved rounds, one index per variant in one process (ab_loops.py)
cd "$(dirname "$0")/.."
python scripts/ab_loops.py VQA_SKETCH_REGQ=0 VQA_SKETCH_REGQ=1 "$@"
