"""One 80M x 768 fp16 shard (122.9 GB = BASELINE configs[3]'s whole corpus) on ONE MI355X: the index is filled chunk by
chunk (`DeviceIndex.empty` + `set_rows`), so no second full copy ever exists; planted needles check the result.
Prints the per-batch time = the 1-GPU reference point for the 8-GPU strong-scaling statement."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=80_000_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--exact", action="store_true", help="no int8 sketch: the exact fp16 scan")
args = ap.parse_args()
n, d, b, k = args.n, 768, 256, 10
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(7)
q = torch.randn((b, d), generator=gen, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
rng = np.random.default_rng(3)
needles = np.sort(rng.choice(n, size=64, replace=False))  # query i is planted at row needles[i]
ix = DeviceIndex.empty(n, d, id_base=1, dtype="fp16", device=0, sketch=not args.exact)
chunk = 1 << 20
t0 = time.perf_counter()
for c0 in range(0, n, chunk):
    c1 = min(n, c0 + chunk)
    x = torch.randn((c1 - c0, d), generator=gen, device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).half()
    for i, r in enumerate(needles):
        if c0 <= r < c1:
            x[r - c0] = q[i]
    ix.set_rows(c0, x)
torch.cuda.synchronize()
build_s = time.perf_counter() - t0
for _ in range(2):
    s, i, p = ix.search(q, k, return_positions=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    s, i, p = ix.search(q, k, return_positions=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
ok = bool(np.array_equal(p[:64, 0].cpu().numpy(), needles)) and bool((s[:64, 0] > 0.99).all())
print(json.dumps({"rows": n, "index_gb": round(n * d * 2 / 1e9, 1), "build_s": round(build_s, 1), "ms_per_batch": round(ms, 3),
                  "queries_per_s": round(b / ms * 1e3, 1), "needles_found_first": ok,
                  "sketch_scan": int(ix.launch_info(b, k).sketch_scan),
                  # bytes of the dominant launch as that launch reads them (one byte per element through the int8 sketch), never the fp16
                  # shard size over a step that does not read it
                  "main_launch_gb": round(ix.launch_info(b, k).rows_per_launch * d * (1 if ix.launch_info(b, k).sketch_scan else 2) / 1e9, 1),
                  "step_gbs_of_main_launch_bytes": round(ix.launch_info(b, k).rows_per_launch * d * (1 if ix.launch_info(b, k).sketch_scan else 2) / ms / 1e6, 1), "hbm_allocated_gb": round(torch.cuda.memory_allocated() / 1e9, 1),
                  "hbm_in_use_gb": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1)}))
