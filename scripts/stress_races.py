"""Race screen for the barrier / counted-vmcnt structures (scoring slot loop, encoder tile GEMM): the same launch repeated many
times at several sizes must give bit-identical results every time (an LDS-DMA read placed by luck instead of by the
vmcnt / barrier count shows up as a rare wrong tile that comes and goes with load).  GPU box:  python scripts/stress_races.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import encoder as E
from vietnamese_qa_system_amd.encoder import QuestionEncoder
from vietnamese_qa_system_amd.index import DeviceIndex

dev = torch.device("cuda", 0)
bad = 0


def rows(n, d, seed, dtype):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    buf = torch.empty((n, d), dtype=dtype, device=dev)
    for c0 in range(0, n, 1 << 18):
        x = torch.randn((min(n, c0 + (1 << 18)) - c0, d), generator=g, device=dev)
        buf[c0:c0 + x.shape[0]] = (x / x.norm(dim=1, keepdim=True)).to(dtype)
    return buf


for dtype, n, d, b, k, reps in (("fp16", 10_000_000, 768, 256, 10, 150), ("fp16", 3_333_333, 768, 200, 12, 150), ("fp16", 70_001, 64, 300, 10, 300),
                               ("fp16", 2_000_000, 100, 256, 5, 150), ("fp8", 5_000_000, 768, 256, 10, 150), ("fp32", 1_000_000, 768, 256, 10, 100),
                               ("fp16", 1_000_000, 768, 256, 100, 60)):
    x = rows(n, d, 7, torch.float16 if dtype == "fp16" else torch.float32)
    q = rows(b, d, 8, x.dtype)
    ix = DeviceIndex(x, dtype=dtype)
    s0, i0, _ = ix.search(q, k)
    torch.cuda.synchronize()
    s0, i0 = s0.clone(), i0.clone()
    t0 = time.perf_counter(); diff = 0
    for r in range(reps):
        s, i, _ = ix.search(q, k)
        if not (torch.equal(s, s0) and torch.equal(i, i0)):
            diff += 1
    torch.cuda.synchronize()
    print(f"search {dtype} {n}x{d} B={b} k={k}: {reps} repeats, {diff} differ, {(time.perf_counter() - t0) / reps * 1e3:.2f} ms each", flush=True)
    bad += diff
    ix.close(); del x, ix
    torch.cuda.empty_cache()

cfg = dict(E.PHOBERT_BASE, vocab_size=8000)
w = E.synthetic_weights(cfg, seed=3)
enc = QuestionEncoder(w, cfg, max_tokens=256 * 128)
for b, l, reps in ((256, 32, 150), (256, 128, 40), (96, 48, 100), (40, 32, 100), (1000, 16, 80), (1, 32, 300), (2, 32, 300), (3, 16, 300)):  # the last three: the latency form
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=b + l)
    ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for real in (0, int(mask.sum())):
        ref = enc.forward(ids_t, mask_t, pooling="mean", real_tokens=real).clone()
        diff = 0
        for r in range(reps):
            out = enc.forward(ids_t, mask_t, pooling="mean", real_tokens=real)
            if not torch.equal(out, ref):
                diff += 1
        torch.cuda.synchronize()
        print(f"encoder B={b} L={l} real_tokens={real}: {reps} repeats, {diff} differ", flush=True)
        bad += diff
enc.close()

# K4 (csrc/tiny_search.hip): per-workgroup lists -> ticket -> merge by the last arriver, ordered WITHOUT device-scope fences.  Two question
# sets alternate call by call, so a list read before its writer's store landed would be the OTHER set's (a stale result, not a repeat
# of the right one); every call must equal the general launches' answer for its set.
rng = np.random.default_rng(11)
for dtype, n, d, b, k, reps in (("fp16", 131072, 768, 1, 1, 3000), ("fp16", 131072, 768, 4, 16, 1500), ("fp16", 16384, 768, 4, 16, 2000),
                                ("fp16", 5000, 768, 1, 10, 3000), ("fp32", 16385, 768, 16, 4, 1500), ("fp32", 100000, 256, 1, 10, 2000)):
    x = rng.standard_normal((n, d)).astype(np.float32)
    qs = [rng.standard_normal((b, d)).astype(np.float32) for _ in range(2)]
    gen = DeviceIndex(x, dtype=dtype, options={"one_launch": 0})
    want = [gen.search_host(q, k, normalize=True, return_positions=True) for q in qs]
    gen.close()
    one = DeviceIndex(x, dtype=dtype)
    diff = 0
    t0 = time.perf_counter()
    for r in range(reps):
        got = one.search_host(qs[r & 1], k, normalize=True, return_positions=True)
        if not all(np.array_equal(u, v) for u, v in zip(got, want[r & 1])):
            diff += 1
    print(f"one-launch search {dtype} {n}x{d} B={b} k={k}: {reps} alternating calls, {diff} differ, {(time.perf_counter() - t0) / reps * 1e6:.1f} us each", flush=True)
    bad += diff
    one.close()
print("TOTAL DIFFERING RUNS:", bad)
sys.exit(1 if bad else 0)
