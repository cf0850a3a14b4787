#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_exp5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/pytest_all.log 2>&1
tail -25 $O/pytest_all.log
