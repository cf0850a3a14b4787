#!/bin/bash
# dev helper: effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) of the main scoring kernel per library variant
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/clk
mkdir -p $O
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
  VQA_LIB=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/$v -- python scripts/kbench.py --steps 4 $KB_ARGS > $O/$v.log 2>&1
  python3 - $O/$v $v <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'score_topk_kernel<1,' in r['Kernel_Name'] or 'score_topk_kernelILi1E' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
g = agg['GRBM_GUI_ACTIVE'][2:]
clk = sum(x[0] / 8 / x[1] for x in g) / len(g) / 1e3
dur = sum(x[1] for x in g) / len(g) / 1e3
def m(n): 
    a = agg[n][2:]; return sum(x[0] for x in a) / len(a)
wc = m('SQ_WAVE_CYCLES')
print(f"{v:10s} kernel {dur:.3f} ms  clock {clk:.3f} GHz  wait_any {m('SQ_WAIT_ANY')/wc:.2f} wait_inst {m('SQ_WAIT_INST_ANY')/wc:.2f} active {m('SQ_ACTIVE_INST_ANY')/wc:.2f}")
PY
done
