import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
def unit(gen, n, d):
    x = torch.randn((n, d), generator=gen, device=dev); x = x / x.norm(dim=1, keepdim=True)
    return x.half() if dtype == "fp16" else x
gen = torch.Generator(device=dev); gen.manual_seed(1)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 768
dtype = sys.argv[3] if len(sys.argv) > 3 else "fp16"
x = torch.empty((n, d), dtype=torch.float16 if dtype == 'fp16' else torch.float32, device=dev)
for c0 in range(0, n, 1 << 18): x[c0:c0 + (1 << 18)] = unit(gen, min(n, c0 + (1 << 18)) - c0, d)
q = unit(gen, 256, d)
x[n // 3] = x[5]; x[n - 7] = x[5]; q[0] = x[5]
a = DeviceIndex(x, sketch=False, dtype=dtype); b = DeviceIndex(x, sketch=True, dtype=dtype)
ia, ib = a.launch_info(256, 10), b.launch_info(256, 10)
print("sketch_scan", ia.sketch_scan, ib.sketch_scan, "first_stage_rows", ib.first_stage_rows)
sa, _, pa = a.search(q, 10, return_positions=True); sb, _, pb = b.search(q, 10, return_positions=True)
torch.cuda.synchronize()
print("pos equal:", torch.equal(pa, pb), "max |ds|:", (sa - sb).abs().max().item(), "dups:", pb[0, :3].tolist())
if not torch.equal(pa, pb):
    bad = (pa != pb).nonzero(); print(bad[:5], sa[bad[0,0]], sb[bad[0,0]])
st = b.sketch_stats()
print(f"candidate pairs {st['last_scan_pairs']} ({st['last_scan_pairs'] / 256:.0f} per query), largest region {st['largest_region']}, longest query sub-list {st['longest_sublist']}, overflow {st['overflow']}")
for ix, nm in ((a, "exact"), (b, "sketch")):
    for _ in range(5): ix.search(q, 10)
    ix.set_timing(True); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ix.search(q, 10)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 20 * 1e3
    ms, k = ix.get_timing(); print(f"{nm}: step {el:.3f} ms, main launch {ms / max(k, 1):.3f} ms")
