"""Dev helper: one encoder GEMM at a time through `gemm_tile_kernel` (variant library built with -DVQA_DEV, entry point
vqa_dev_gemm), checked against torch.mm and timed beside it (the vendor GEMM: a measuring stick, never part of the product).

    python scripts/gemm_bench.py [--stamps] [--shapes 1,5] [--problems FFN1,QKV] [--defs VQA_X=1 ...]

--stamps builds with -DVQA_GSTAMPS and prints the per-phase cycle timeline of one workgroup's first tile
(slot 1: reads | DMA issue | lgkmcnt | vmcnt | barrier;  slot 2: MFMAs | barrier)."""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--shapes", default="-1")
ap.add_argument("--problems", default="QKV,out,FFN1,FFN2")
ap.add_argument("--defs", nargs="*", default=[])
ap.add_argument("--tag", default="dev")
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--m", type=int, default=8192)
ap.add_argument("--no-build", action="store_true")
ap.add_argument("--names", default="start,reads,dma,lgkm,vmcnt,bar1,s2work,s2end",
                help="stamp names (one-barrier loop: start,mma1,reads,dma,mma2,lgkm,vmcnt,bar)")
args = ap.parse_args()

from vietnamese_qa_system_amd import build
tag = args.tag + ("_st" if args.stamps else "")
lib_path = os.path.join(build.LIB_DIR, f"libvqa_retrieval_{tag}.so")
if not args.no_build:
    lib_path = build.build_variant(tag, ["VQA_DEV"] + (["VQA_GSTAMPS"] if args.stamps else []) + list(args.defs))
import torch
lib = ctypes.CDLL(lib_path)
lib.vqa_dev_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
lib.vqa_last_error.restype = ctypes.c_char_p
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
M = args.m
PROBLEMS = {"QKV": (M, 2304, 768, 0), "out": (M, 768, 768, 2), "FFN1": (M, 3072, 768, 1), "FFN2": (M, 768, 3072, 2)}
SHAPES = {-1: "auto", 0: "256x288/32", 1: "256x192/32", 2: "256x128/64", 3: "128x192/64", 4: "256x128/32", 5: "256x256/32", 6: "128x128/64",
          7: "256x192/32 1bar", 8: "256x256/32 1bar", 9: "256x128/64 1bar", 10: "128x192/64 1bar", 11: "128x128/64 1bar",
          12: "256x288/32 tiled", 13: "256x192/32 tiled", 14: "256x128/64 tiled", 15: "128x192/64 tiled", 16: "256x256/32 tiled",
          17: "128x128/64 1bar tiled"}


def tiled(t):
    """[rows, k] fp16 -> the tiled operand layout: 256-row tiles x 64-byte K-blocks (rows padded to 256)"""
    rows, k = t.shape
    rp = (rows + 255) // 256 * 256
    p = torch.zeros((rp, k), device=t.device, dtype=t.dtype)
    p[:rows] = t
    return p.view(rp // 256, 256, k // 32, 32).permute(0, 2, 1, 3).contiguous()
stream = torch.cuda.current_stream().cuda_stream
for name in args.problems.split(","):
    m, n, k, epi = PROBLEMS[name]
    mp = (m + 255) // 256 * 256
    a = torch.zeros((mp, k), device=dev, dtype=torch.float16)
    a[:m] = torch.randn((m, k), generator=g, device=dev).half()
    w = (torch.randn((n, k), generator=g, device=dev) * 0.03).half()
    bias = torch.randn((n,), generator=g, device=dev)
    r = torch.zeros((mp, n), device=dev, dtype=torch.float16)
    r[:m] = torch.randn((m, n), generator=g, device=dev).half()
    c = torch.zeros((mp, n), device=dev, dtype=torch.float16)
    ref = a[:m].float() @ w.float().t() + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + r[:m].float()
    at, wt = tiled(a), tiled(w)
    ops = lambda sh: (at.data_ptr(), wt.data_ptr()) if sh >= 12 else (a.data_ptr(), w.data_ptr())
    for _ in range(5):
        torch.mm(a[:m], w.t())
    times = {}
    for sh in [int(x) for x in args.shapes.split(",")]:
        c.zero_()
        rc = lib.vqa_dev_gemm(*ops(sh), bias.data_ptr(), r.data_ptr(), c.data_ptr(), m, n, k, epi, sh, stream)
        if rc != 0:
            print(f"{name} shape {SHAPES.get(sh, sh)}: rc {rc} {lib.vqa_last_error().decode()}")
            continue
        torch.cuda.synchronize()
        err = (c[:m].float() - ref).abs().max().item()
        times[sh] = [err, []]
    times["mm"] = [0.0, []]
    for _ in range(args.rounds):
        for sh in times:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                if sh == "mm":
                    torch.mm(a[:m], w.t())
                else:
                    lib.vqa_dev_gemm(*ops(sh), bias.data_ptr(), r.data_ptr(), c.data_ptr(), m, n, k, epi, sh, stream)
            e1.record()
            torch.cuda.synchronize()
            times[sh][1].append(e0.elapsed_time(e1) / args.reps * 1e3)
    for sh, (err, ts) in times.items():
        us = float(np.median(ts))
        label = "torch.mm (no epilogue)" if sh == "mm" else f"tile {SHAPES.get(sh, sh)} epi {epi}"
        print(f"{name:5s} {m}x{n}x{k}  {label:28s} {us:7.1f} us (min {min(ts):6.1f})  {2 * m * n * k / us / 1e6:6.0f} TF/s  max|err| {err:.3g}", flush=True)
    if args.stamps:
        lib.vqa_dev_read_gstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
        out = np.zeros((8, 64, 8), dtype=np.uint64)
        assert lib.vqa_dev_read_gstamps(out.ctypes.data, out.size) == 0
        t = out.astype(np.int64)
        kt = int((t[0, :, 0] > 0).sum())
        names = args.names.split(",")
        print(f"  stamps of the LAST launch ({name}), workgroup 37, first tile, {kt} K-steps; mean cycles between stamps (rows = waves)")
        print("  wave  " + "  ".join(f"{names[j]:>6s}->{names[j + 1]:<6s}" for j in range(7)) + "   kstep")
        for wv in range(8):
            d = np.diff(t[wv, 2:kt], axis=1).mean(axis=0)
            whole = np.diff(t[wv, 2:kt, 0]).mean()
            print(f"  {wv:4d}  " + "  ".join(f"{x:14.0f}" for x in d) + f"  {whole:7.0f}")
