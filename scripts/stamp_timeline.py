"""Dev helper: per-phase cycle timeline of the scoring kernel's K-steps (diagnostic build -DVQA_STAMPS, VQA_LIB selects it).
Prints, for each wave of one workgroup during one tile, the mean cycles between consecutive stamps of a K-step and a
merged timeline of a few K-steps (all stamps are s_memtime values = shader cycles)."""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex
from vietnamese_qa_system_amd import _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=768)
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--names", default="start,p1,reads,dma,mfma,lgkm,vmcnt,barrier")
ap.add_argument("--show", type=int, default=4, help="K-steps printed as a merged timeline")
args = ap.parse_args()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
buf = torch.empty((args.n, args.d), dtype=torch.float16 if args.dtype == "fp16" else torch.float32, device=dev)
for c0 in range(0, args.n, 1 << 18):
    c1 = min(args.n, c0 + (1 << 18))
    x = torch.randn((c1 - c0, args.d), generator=gen, device=dev)
    x /= x.norm(dim=1, keepdim=True)
    buf[c0:c1] = x.to(buf.dtype)
ix = DeviceIndex(buf, dtype=args.dtype)
q = torch.randn((256, args.d), generator=gen, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).to(buf.dtype)
for _ in range(12):
    ix.search(q, 10)
torch.cuda.synchronize()
lib = N.load()
KT = (args.d * 2 + 63) // 64 if args.dtype == "fp16" else (args.d + 127) // 128  # fp8: K-step pairs per tile
if ix.launch_info(256, 10).sketch_scan:  # the stamped launch is the int8 sketch scan: K-steps of 64 one-byte elements
    KT = (args.d + 127) // 128 * 2
out = np.zeros((8, 64, 8), dtype=np.uint64)
lib.vqa_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.vqa_debug_read_stamps(out.ctypes.data, out.size) == 0
t = out[:, :KT, :].astype(np.int64)
names = args.names.split(",")
t0 = t[:, 0, 0].min()
print("K-steps per tile:", KT, " origin:", t0)
print("mean cycles between consecutive stamps of a K-step (rows = waves; last column = whole K-step, start -> next start)")
print("wave  " + "  ".join(f"{names[j]:>7s}->{names[j+1]:<7s}" for j in range(7)) + "   kstep")
for w in range(8):
    d = np.diff(t[w], axis=1).mean(axis=0)
    whole = np.diff(t[w, :, 0]).mean()
    print(f"{w:4d}  " + "  ".join(f"{x:16.0f}" for x in d) + f"  {whole:7.0f}")
print("\nmerged timeline (cycles from the origin): wave:stamp")
ev = []
for w in range(8):
    for k in range(min(args.show, KT)):
        for j in range(8):
            ev.append((int(t[w, k, j] - t0), w, k, names[j]))
for c, w, k, nm in sorted(ev):
    print(f"{c:7d}  " + "    " * w + f"w{w} k{k} {nm}")
