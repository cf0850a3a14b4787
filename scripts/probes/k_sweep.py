"""Dev probe: search step of a 10M x 768 fp16 shard over k and B (the cliffs of round 3: k = 65, B = 257)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
n, d = int(os.environ.get("N", 10_000_000)), 768
buf = torch.empty((n, d), dtype=torch.float16, device=dev)
for c0 in range(0, n, 1 << 18):
    c1 = min(n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, d), generator=gen, device=dev); x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.half()
ix = DeviceIndex(buf, dtype="fp16")
ex = DeviceIndex(buf, dtype="fp16", sketch=False)
def timed(index, q, k, steps=10):
    for _ in range(3): index.search(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): index.search(q, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for b, k in ((256, 10), (256, 30), (256, 64), (256, 65), (256, 100), (256, 128), (256, 129), (257, 10), (1, 10), (512, 10)):
    q = torch.randn((b, d), generator=gen, device=dev); q = (q / q.norm(dim=1, keepdim=True)).half()
    ts = timed(ix, q, k)
    st = ix.sketch_stats() if ix.launch_info(b, k).sketch_scan else None
    te = timed(ex, q, k, 5)
    s1, _, p1 = ix.search(q, k, return_positions=True); s0, _, p0 = ex.search(q, k, return_positions=True); torch.cuda.synchronize()
    print(f"B {b:4d} k {k:4d}: default {ts:7.3f} ms  exact path {te:7.3f} ms  same rows {bool(torch.equal(p0, p1))}  max|ds| {float((s0 - s1).abs().max()):.1e}  "
          f"state {ix.sketch_state()}  {('pairs ' + str(st['rescored_pairs']) + ' longest sublist ' + str(st['longest_sublist'])) if st else ''}", flush=True)
