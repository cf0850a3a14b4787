#!/bin/bash
# dev helper: effective shader clock (GRBM_GUI_ACTIVE / 8 / duration) + MFMA busy of the sketch scan's main launch per library variant
# usage: clock_probe2.sh <kernel name substring> tag[:kbench --opt string] ...    (tag "default" = the product library)
export TMPDIR=/tmp
KN=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/clk6
mkdir -p $O
for spec in "$@"; do
  v=${spec%%:*}; opt=""; [ "$spec" != "$v" ] && opt=${spec#*:}
  lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval_$v.so
  [ "$v" = default ] && lib=$GRAFT_REPO_ROOT/vietnamese_qa_system_amd/lib/libvqa_retrieval.so
  tag=$v${opt:+_${opt//[=,]/_}}
  rm -rf $O/$tag
  VQA_LIB=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/$tag -- python3 scripts/kbench.py --steps 6 --opt "$opt" > $O/$tag.log 2>&1
  python3 - $O/$tag $tag "$KN" <<'PY'
import csv, glob, sys, collections
d, v, kn = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if kn in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:50]))
g = agg['GRBM_GUI_ACTIVE']
# the main launch is the longest of the matching dispatches
dmax = max(x[1] for x in g)
sel = lambda n: [x for x in agg[n] if x[1] > 0.6 * dmax][1:]
gg = sel('GRBM_GUI_ACTIVE')
clk = sum(x[0] / 8 / x[1] for x in gg) / len(gg) / 1e3
dur = sum(x[1] for x in gg) / len(gg) / 1e3
m = lambda n: sum(x[0] for x in sel(n)) / max(len(sel(n)), 1)
wc = m('SQ_WAVE_CYCLES')
busy = m('SQ_VALU_MFMA_BUSY_CYCLES') / (m('GRBM_GUI_ACTIVE') / 8 * 256 * 4) if m('GRBM_GUI_ACTIVE') else 0
print(f"{v:28s} main launch {dur:.3f} ms  clock {clk:.3f} GHz  mfma busy {busy:.3f}  busy x clock {busy*clk:.3f} GHz | wait_any {m('SQ_WAIT_ANY')/wc:.2f} wait_inst {m('SQ_WAIT_INST_ANY')/wc:.2f} active {m('SQ_ACTIVE_INST_ANY')/wc:.2f}  ({gg[0][2]})")
PY
done
