"""Condense the rocprofv3 output of scripts/profile_round.sh: per-kernel stats CSV + PMC summary of the main scoring kernel.
HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE in separate passes, KiB units,
FETCH_SIZE doubled on gfx950 for wide coalesced streaming reads."""
import re, collections, csv, glob, json, os, shutil, sys

root = sys.argv[1]
out = {}


def rows(sub, suffix):
    f = glob.glob(os.path.join(root, sub, "*", "*" + suffix))
    return list(csv.DictReader(open(f[0]))) if f else []


st = glob.glob(os.path.join(root, "stats", "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(root, "kernel_stats.csv"))
main_name = None
# the dominant launch: the int8 sketch scan score_topk_kernel<2, 3, 0, 0> where a search has one, else the exact main launch
# score_topk_kernel<1, DT, 0, L> (<1, DT, 1, L> is the first stage of a two-stage search, <0, ..> the seed pass)
MAIN = r"score_topk_kernel<1, \d, 0(, \d)?>|score_topk_kernelILi1ELi\dELi0E"
if any(re.search(r"sketch_scan_regq_kernel<", r["Kernel_Name"]) for r in rows("pmc_fetch", "counter_collection.csv")):
    MAIN = r"sketch_scan_regq_kernel<\d+, 0>"  # (<KT, 1> is the same code under the early stages' symbol)
    out["sketch_scan"] = True
elif any(re.search(r"score_topk_kernel<2, 3", r["Kernel_Name"]) for r in rows("pmc_fetch", "counter_collection.csv")):
    MAIN = r"score_topk_kernel<2, 3, 0, \d>"
    out["sketch_scan"] = True
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    agg = collections.defaultdict(list)
    for r in rows(sub, "counter_collection.csv"):
        if re.search(MAIN, r["Kernel_Name"]):
            main_name = re.search(r"(score_topk_kernel|sketch_scan_regq_kernel)(<[^>]*>|IL\w*E)", r["Kernel_Name"]).group(0)
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for name, v in agg.items():
        v = v[3:] if len(v) > 6 else v  # drop the warm-up launches
        out[name] = sum(x[0] for x in v) / len(v)
        out[sub + "_kernel_ms"] = sum(x[1] for x in v) / len(v) / 1e6
out["kernel"] = main_name
# every kernel of the search step, from the same passes: mean HBM read / write bytes per launch (the cross-check of bench.py's
# algorithmic step_bytes_moved: one step = the launches of one vqa_index_search)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for r in rows(sub, "counter_collection.csv"):
        m = re.search(r"(score_topk_kernel<[^>]*>|sketch_scan_regq_kernel<[^>]*>|final_rescore_kernel|rescore_kernel|merge_partials_kernel|sketch_rows_kernel|tile_rows_kernel|sketch_qconst_kernel)", r["Kernel_Name"])
        if m and r["Counter_Name"] == ctr:
            per[m.group(1)][ctr].append(float(r["Counter_Value"]))
out["per_kernel_hbm_bytes_per_launch"] = {
    k: {"launches_seen": len(v.get("FETCH_SIZE", [])), "read_corrected": round(sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [])), 1) * 2048),
        "write": round(sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [])), 1) * 1024)} for k, v in per.items()}
if "FETCH_SIZE" in out:
    out["hbm_read_bytes_corrected"] = out["FETCH_SIZE"] * 1024 * 2
if "WRITE_SIZE" in out:
    out["hbm_write_bytes"] = out["WRITE_SIZE"] * 1024
if "hbm_read_bytes_corrected" in out and "hbm_write_bytes" in out:
    out["hbm_traffic_bytes_per_launch"] = out["hbm_read_bytes_corrected"] + out["hbm_write_bytes"]
if "GRBM_GUI_ACTIVE" in out:
    out["effective_clock_ghz"] = out["GRBM_GUI_ACTIVE"] / 8 / (out["pmc_sq_kernel_ms"] * 1e6)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out:  # counts cycles summed over the SIMDs of the chip (256 CUs x 4)
        out["mfma_pipe_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (out["GRBM_GUI_ACTIVE"] / 8)
if "TCC_HIT_sum" in out:
    out["l2_hit_frac"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
try:
    out["bench"] = json.loads(open(os.path.join(root, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    out["bench_error"] = str(e)
json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench"}, indent=1))
