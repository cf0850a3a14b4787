"""Dev helper: step time of the 10M x 768 fp16 sketch search over the cascade's level sizes (stage_pct x sketch_mid_pct), one shard in
memory, one index per setting.  usage: stage_sweep.py [--pct 6,8,10,14] [--mid 100,200,300] [--b 256] [--steps 30]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--b", type=int, default=256)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--pct", default="6,8,10,14")
ap.add_argument("--mid", default="100,200,300")
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
buf = torch.empty((args.n, args.d), dtype=torch.float16, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, args.d), generator=gen, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.half()
q = torch.randn((args.b, args.d), generator=gen, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).half()
for rnd in range(2):
    for pct in [int(v) for v in args.pct.split(",")]:
        for mid in [int(v) for v in args.mid.split(",")]:
            ix = DeviceIndex(buf, dtype="fp16", options={"stage_pct": pct, "sketch_mid_pct": mid})
            ix.set_timing(False)
            for _ in range(5):
                ix.search(q, args.k)
            ix.set_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ix.search(q, args.k)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / args.steps * 1e3
            ms, n = ix.get_timing()
            info = ix.launch_info(args.b, args.k)
            print(f"round {rnd} stage_pct {pct:3d} mid_pct {mid:4d}  levels {info.levels}  step {el:.3f} ms  main {ms / max(n, 1):.3f} ms  "
                  f"rest {el - ms / max(n, 1):.3f} ms  overflow {ix.sketch_stats()['overflow']}", flush=True)
            ix.close()
            del ix
