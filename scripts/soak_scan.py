"""Soak of the sketch search on the 10M x 768 fp16 shard: thousands of repeats per batch size, every result compared ON THE DEVICE with
the first one (scores, ids, positions bit for bit).  A rare race in the register-resident scan's ring / barrier structure, the re-scoring
or the selections shows up as a differing repeat.  GPU box:  python scripts/soak_scan.py [--reps 4000]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=4000)
ap.add_argument("--b", default="256,1,129,257,97")
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
d, k = 768, 10
buf = torch.empty((args.n, d), dtype=torch.float16, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, d), generator=g, device=dev)
    buf[c0:c1] = (x / x.norm(dim=1, keepdim=True)).half()
ix = DeviceIndex(buf, dtype="fp16")
bad_total = 0
for b in [int(v) for v in args.b.split(",")]:
    q = torch.randn((b, d), generator=g, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).half()
    s0, i0, p0 = ix.search(q, k, return_positions=True)
    s0, i0, p0 = s0.clone(), i0.clone(), p0.clone()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        s, i, p = ix.search(q, k, return_positions=True)
        bad += ((s != s0).any() | (i != i0).any() | (p != p0).any()).long()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    nb = int(bad.item())
    bad_total += nb
    print(f"B = {b:4d}: {args.reps} repeats, {nb} differ, {el / args.reps * 1e3:.3f} ms each, overflow {ix.sketch_stats()['overflow']}, sketch_state {ix.sketch_state()}", flush=True)
print("TOTAL DIFFERING REPEATS:", bad_total)
