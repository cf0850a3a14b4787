"""Dev probe: practical ceilings of this MI355X on random data -- library GEMM rate (hipBLASLt via torch.matmul) and a
streaming read -- to put the fused kernel's numbers in context.  Not part of the product path."""
import time, torch
dev = torch.device("cuda", 0)
def timeit(f, n=10, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
g = torch.Generator(device=dev); g.manual_seed(0)
import os
ONLY_FP8 = os.environ.get("ONLY_FP8") == "1"
if not ONLY_FP8:
    a = torch.randn((8192, 8192), device=dev, generator=g).half(); b = torch.randn((8192, 8192), device=dev, generator=g).half()
    t = timeit(lambda: a @ b.T, 20)
    print(f"hipBLASLt fp16 8192^3 NT random normal: {t*1e3:.3f} ms  {2*8192**3/t/1e12:.0f} TFLOP/s")
    az = torch.zeros_like(a); bz = torch.zeros_like(b)
    t = timeit(lambda: az @ bz.T, 20)
    print(f"hipBLASLt fp16 8192^3 NT zeros:         {t*1e3:.3f} ms  {2*8192**3/t/1e12:.0f} TFLOP/s")
    del a, b, az, bz
    n = 10_000_000
    x = torch.empty((n, 768), device=dev, dtype=torch.float16)
    for c in range(0, n, 1 << 20):
        y = torch.randn((min(n, c + (1 << 20)) - c, 768), device=dev, generator=g); y /= y.norm(dim=1, keepdim=True); x[c:c + y.shape[0]] = y.half()
    q = torch.randn((256, 768), device=dev, generator=g); q = (q / q.norm(dim=1, keepdim=True)).half()
    out = torch.empty((256, n), device=dev, dtype=torch.float16)
    t = timeit(lambda: torch.matmul(q, x.T, out=out), 10)
    print(f"hipBLASLt [256,768]x[768,10M] -> fp16 scores (no top-k): {t*1e3:.3f} ms  {2*256*768*n/t/1e12:.0f} TFLOP/s  {n*768*2/t/1e9:.0f} GB/s of index")
    t2 = timeit(lambda: torch.topk(out, 10, dim=1), 3, 1)
    print(f"torch.topk(k=10) over the [256,10M] fp16 score matrix: {t2*1e3:.3f} ms")
    del out
    t = timeit(lambda: x.view(torch.int32).sum(), 10)
    print(f"torch sum over the 15.36 GB index (streaming read): {t*1e3:.3f} ms  {n*768*2/t/1e9:.0f} GB/s")
    y = torch.empty_like(x[: n // 2])
    t = timeit(lambda: y.copy_(x[: n // 2]), 10)
    print(f"device copy 7.68 GB: {t*1e3:.3f} ms  read+write {2*(n//2)*768*2/t/1e9:.0f} GB/s")

# fp8 (OCP e4m3) library GEMM on random data, for the fp8 index's context
try:
    a8 = (torch.randn((8192, 8192), device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
    b8 = (torch.randn((8192, 8192), device=dev, generator=g) * 0.5).to(torch.float8_e4m3fn)
    one = torch.ones((), device=dev)
    f = lambda: torch._scaled_mm(a8, b8.T, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
    t = timeit(f, 20)
    print(f"hipBLASLt fp8 e4m3 8192^3 NT random normal: {t*1e3:.3f} ms  {2*8192**3/t/1e12:.0f} TFLOP/s")
except Exception as e:  # noqa: BLE001
    print("fp8 library GEMM not available:", repr(e)[:200])
