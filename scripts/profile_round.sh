#!/bin/bash
# Profile set for one bench configuration, run on the GPU box:  scripts/profile_round.sh <tag> [bench.py args]
#   1. bench.py (un-profiled)                        -> gpurun_out/<tag>/bench.json
#   2. timeout 300 rocprofv3 --kernel-trace --stats              -> gpurun_out/<tag>/stats/  (per-kernel durations)
#   3. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters + clock | LDS + L2 hits), each under its own timeout:
#      a counter set the hardware cannot collect aborts rocprofv3 and leaves the process hanging  -> gpurun_out/<tag>/pmc_*/
#   4. scripts/summarize_profile.py                   -> gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.json}
# Copy what is to be judged from gpurun_out/<tag>/ into profiles/ afterwards (gpurun_out is scratch).
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 python3 $R/bench.py "$@" > $O/bench.json 2> $O/bench.err
PROF_ARGS="--steps 12 --warmup 3 --no-cpu --no-e2e"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu --no-e2e "$@" > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $PROF_ARGS "$@" > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $PROF_ARGS "$@" > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py $PROF_ARGS "$@" > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_lds -- python3 $R/bench.py $PROF_ARGS "$@" > $O/pmc_lds.log 2>&1
python3 $R/scripts/summarize_profile.py $O
