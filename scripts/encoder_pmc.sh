#!/bin/bash
# PMC passes over the encoder forward (scripts/enc_bench.py, padded B = 256, L = 32): per-kernel L2 hit rate, LDS bank conflicts,
# MFMA pipe busy and effective clock, HBM fetch bytes -> gpurun_out/enc_pmc/summary.json (copy into profiles/ to keep it).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/enc_pmc; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/l2 -- python3 $R/scripts/enc_bench.py 256 32 > $O/l2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- python3 $R/scripts/enc_bench.py 256 32 > $O/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/enc_bench.py 256 32 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/enc_bench.py 256 32 > $O/write.log 2>&1
python3 - <<'PY'
import collections, csv, glob, json, os, re
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "enc_pmc")
out = collections.defaultdict(dict)
for sub in ("l2", "sq", "fetch", "write"):
    f = glob.glob(os.path.join(O, sub, "*", "*counter_collection.csv"))
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if not re.search(r"gemm_tile|attention_mfma|gemm_skinny|embed_ln|pool_normalize", name):
            continue
        short = re.sub(r"EEvPK.*", "", name).replace("_ZN12_GLOBAL__N_1", "")[:60]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[short]["_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in agg.items():
        for c, v in d.items():
            out[k][(sub + "_kernel_us") if c == "_ns" else c] = sum(v) / len(v) / (1e3 if c == "_ns" else 1)
for k, d in out.items():
    if "TCC_HIT_sum" in d:
        d["l2_hit_frac"] = d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1)
    if "GRBM_GUI_ACTIVE" in d:
        d["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8 / (d["sq_kernel_us"] * 1e3)
        d["mfma_pipe_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (d["GRBM_GUI_ACTIVE"] / 8)
    if "FETCH_SIZE" in d:
        d["hbm_read_mb_corrected"] = d["FETCH_SIZE"] * 1024 * 2 / 1e6
    if "WRITE_SIZE" in d:
        d["hbm_write_mb"] = d["WRITE_SIZE"] * 1024 / 1e6
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for k, d in out.items():
    print(k, {a: round(b, 3) for a, b in d.items() if a in ("l2_hit_frac", "effective_clock_ghz", "mfma_pipe_busy_frac", "hbm_read_mb_corrected", "hbm_write_mb", "sq_kernel_us", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")})
PY
