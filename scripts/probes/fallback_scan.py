"""Dev probe: where does a sketch search overflow into its exact fallback on unclustered data?  A grid of shard sizes, dimensions,
batch sizes and k; after every search `sketch_state` tells (0 = stayed on the sketch).  Prints the cells that fell back."""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd.index import DeviceIndex
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(7)
bad, cells = [], 0
for dtype, sizes in (("fp16", (1_200_000, 3_000_000, 10_000_000)), ("fp32", (600_000, 1_000_000))):
    for n, d in itertools.product(sizes, (64, 128, 384, 768, 1024)):
        if n * d > 10_000_000 * 768:
            continue
        x = torch.empty((n, d), dtype=torch.float16 if dtype == "fp16" else torch.float32, device=dev)
        for c0 in range(0, n, 1 << 19):
            r = torch.randn((min(n, c0 + (1 << 19)) - c0, d), generator=g, device=dev)
            x[c0:c0 + r.shape[0]] = (r / r.norm(dim=1, keepdim=True)).to(x.dtype)
        ix = DeviceIndex(x, dtype=dtype)
        del x
        for b, k in itertools.product((1, 7, 256, 257), (1, 10, 12, 13, 32, 64, 100, 128)):
            if ix.launch_info(b, k).sketch_scan != 1:
                continue
            q = torch.randn((b, d), generator=g, device=dev)
            q = (q / q.norm(dim=1, keepdim=True)).to(torch.float16 if dtype == "fp16" else torch.float32)
            ix.search(q, k)
            torch.cuda.synchronize()
            cells += 1
            st = ix.sketch_state()
            if st != 0:
                bad.append((dtype, n, d, b, k, st))
                for _ in range(st + 1):  # run the pause out (it counts the searches it applies to: those of at least half this k)
                    ix.search(q[:1], k)
                torch.cuda.synchronize()
        ix.close()
        print(f"{dtype} n={n} d={d}: done ({cells} searches so far, {len(bad)} fell back)", flush=True)
print("FELL BACK:", bad if bad else "none")
