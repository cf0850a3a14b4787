"""Dev probe: the B = 256 search of the 10M-row shard takes 1.87 ms back to back but 1.67 ms right behind the encoder (bench.py's
end-to-end leg).  Is that the clock the chip holds?  Search steps alone, and with a short matrix-multiply burst in front of every step
(sizes 1024^3 ... 8192^3 fp16): time per iteration, the burst's own time, and what is left for the search."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd.index import DeviceIndex
device = torch.device("cuda", 0)
shard = bench.build_shard(torch, 10_000_000, 768, 1234, device, "fp16")
ix = DeviceIndex(shard, id_base=1, dtype="fp16", device=0, sketch=True)
del shard
torch.cuda.empty_cache()
q = torch.randn((256, 768), device=device)
q = (q / q.norm(dim=1, keepdim=True)).half()
steps = 40


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


base = timed(lambda: ix.search(q, 10))
print(f"search alone: {base:.4f} ms per step", flush=True)
for m in (1024, 2048, 4096, 8192):
    a = torch.randn((m, m), device=device, dtype=torch.float16)
    mm = timed(lambda: torch.mm(a, a))
    both = timed(lambda: (torch.mm(a, a), ix.search(q, 10)))
    print(f"burst {m}^3: burst alone {mm:.4f} ms, burst + search {both:.4f} ms -> search part {both - mm:.4f} ms ({(both - mm) / base - 1:+.1%})", flush=True)
