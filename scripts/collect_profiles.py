"""Copy what is to be judged from gpurun_out/<tag>/ (scratch, written by scripts/profile_round.sh on the GPU box) into profiles/:
<name>_kernel_stats.csv (kernel names cut to 120 characters: torch's templated names run to kilobytes), <name>_pmc_summary.json,
<name>_bench.json; and refresh the workload's entry of profiles/traffic.json.   usage: collect_profiles.py <tag> <name>"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
rows = list(csv.reader(open(os.path.join(src, "kernel_stats.csv"))))
with open(os.path.join(dst, f"{name}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)
summ = json.load(open(os.path.join(src, "pmc_summary.json")))
bench = summ.pop("bench", None)
if summ.get("kernel") in (None, "", "void "):
    summ["kernel"] = next((r[0][:120] for r in rows[1:] if re.search(r"sketch_scan_regq_kernel<\d+, 0>|score_topk_kernel<2, 3, 0, \d>|score_topk_kernel<1, \d, 0(, \d)?>|score_topk_kernelILi1ELi\dELi0E", r[0])), summ.get("kernel"))
json.dump(summ, open(os.path.join(dst, f"{name}_pmc_summary.json"), "w"), indent=1)
if bench:
    json.dump(bench, open(os.path.join(dst, f"{name}_bench.json"), "w"), indent=1)
    c = bench["config"]
    key = f"{c['docs_per_gpu']}x{c['dim']}_{bench['dtype']}_b{c['batch']}_k{c['k']}" + ("_sketch" if summ.get("sketch_scan") else "")
    tpath = os.path.join(dst, "traffic.json")
    t = json.load(open(tpath))
    if "hbm_traffic_bytes_per_launch" in summ:
        t[key] = int(round(summ["hbm_traffic_bytes_per_launch"]))
        t["_source_" + key] = f"profiles/{name}_pmc_summary.json"
        t["_kernel_" + key] = summ.get("kernel")  # bench.py drops the figure when the dominant kernel is another one by now
        json.dump(t, open(tpath, "w"), indent=1)
    print(key, t.get(key))
print("ok", name)
