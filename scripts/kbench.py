"""Dev helper: time vqa_index_search on a synthetic fp16 shard (no oracle, no checks).  VQA_LIB selects the library."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--b", default="256", help="batch size, or a comma list: one index, one line per batch size")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--opt", default="", help="index options, NAME=VALUE,... (scripts/_options.py vocabulary)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
buf = torch.empty((args.n, args.d), dtype=torch.float16 if args.dtype == "fp16" else torch.float32, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, args.d), generator=gen, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.to(buf.dtype)
from _options import options as _options
ix = DeviceIndex(buf, dtype=args.dtype, options=_options(dict(s.split('=', 1) for s in args.opt.split(',') if s)))
for b in [int(v) for v in str(args.b).split(",")]:
    q = torch.randn((b, args.d), generator=gen, device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).to(buf.dtype)
    ix.set_timing(False)
    for _ in range(3):
        ix.search(q, args.k)
    ix.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ix.search(q, args.k)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps * 1e3
    ms, n = ix.get_timing()
    info = ix.launch_info(b, args.k)
    kms = ms / max(n, 1)
    print(f"{os.path.basename(os.environ.get('VQA_LIB', 'default')):40s} B {b:4d}  step {el:.3f} ms  main kernel {kms:.3f} ms  "
          f"-> {info.bytes_per_launch / (kms * 1e-3) / 1e9:.0f} GB/s of the bytes that launch reads ({'int8 sketch' if info.sketch_scan else args.dtype}), "
          f"{info.flops_per_launch / (kms * 1e-3) / 1e12:.0f} T(FL)OP/s", flush=True)
