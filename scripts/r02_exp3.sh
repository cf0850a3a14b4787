#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_exp3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1
tail -3 $O/pytest_all.log
bash scripts/kb_variants.sh 3 default old sprio > $O/kb.log 2>&1
cat $O/kb.log
