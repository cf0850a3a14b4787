#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in 1 2 4; do echo -n "rpw=$v "; VQA_LN_RPW=$v python scripts/enc_bench.py 256 32 | grep encoder; done; done
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -x -q 2>&1 | tail -2
