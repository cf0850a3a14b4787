"""Dev helper: A/B of index options (named as rounds 1-4 named their environment switches: VQA_F16_LOOP, VQA_STAGE_MIN, ... -> scripts/_options.py), interleaved rounds in
ONE process on one device (guide rule 24): `python scripts/ab_loops.py VQA_F16_LOOP=0 VQA_F16_LOOP=1 [--n rows] [--rounds R]`.
Every variant is its own index over the same synthetic shard; per variant: median / min step and main-launch time."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex
from _options import options as _options

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", help="comma-separated NAME=VALUE settings per variant, e.g. VQA_F16_LOOP=1,VQA_STAGE_MIN=0")
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--b", type=int, default=256)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
buf = torch.empty((args.n, args.d), dtype=torch.float16 if args.dtype == "fp16" else torch.float32, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, args.d), generator=gen, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.to(buf.dtype)
q = torch.randn((args.b, args.d), generator=gen, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).to(buf.dtype)
idx = []
for v in args.variants:
    kv = dict(s.split("=", 1) for s in v.split(",") if s)
    idx.append(DeviceIndex(buf, dtype=args.dtype, options=_options(kv)))
ref = None
for v, ix in zip(args.variants, idx):
    s, i, _ = ix.search(q, args.k)
    torch.cuda.synchronize()
    if ref is None:
        ref = (s.clone(), i.clone())
    else:
        print(f"{v}: ids equal to the first variant: {torch.equal(i, ref[1])}, scores equal: {torch.equal(s, ref[0])}")
    for _ in range(10):
        ix.search(q, args.k)
step = [[] for _ in idx]
kern = [[] for _ in idx]
for r in range(args.rounds):
    for j, ix in enumerate(idx):
        ix.set_timing(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ix.search(q, args.k)
        ix.get_timing()
        e0.record()
        for _ in range(args.steps):
            ix.search(q, args.k)
        e1.record()
        torch.cuda.synchronize()
        ms, n = ix.get_timing()
        ix.set_timing(False)
        step[j].append(e0.elapsed_time(e1) / args.steps)
        kern[j].append(ms / max(n, 1))
esz = {"fp16": 2, "fp8": 1, "fp32": 4}[args.dtype]
for v, st, kn in zip(args.variants, step, kern):
    print(f"{v:40s} step median {np.median(st):.4f} min {min(st):.4f} ms | main launch median {np.median(kn):.4f} min {min(kn):.4f} ms | "
          f"{args.b / (np.median(st) * 1e-3):.0f} q/s = {args.n * args.d * esz / (np.median(st) * 1e-3) / 1e9 / 8000:.4f} of the HBM-roofline q/s of the index as stored", flush=True)
