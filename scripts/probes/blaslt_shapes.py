"""Dev probe: torch.mm (hipBLASLt / rocBLAS) on the encoder's four GEMM shapes, fp16, random normal data -- a measuring stick for
gemm_tile_kernel (which also applies bias / GELU / residual / LayerNorm terms in its epilogue).  Not used by the product."""
import torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
for name, (m, n, k) in {"QKV": (8192, 2304, 768), "out-proj": (8192, 768, 768), "FFN1": (8192, 3072, 768), "FFN2": (8192, 768, 3072),
                        "QKV packed": (5114, 2304, 768), "FFN1 packed": (5114, 3072, 768), "FFN2 packed": (5114, 768, 3072)}.items():
    a = torch.randn((m, k), generator=g, device=dev).half()
    w = torch.randn((n, k), generator=g, device=dev).half() * 0.03
    for _ in range(5): torch.mm(a, w.t())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): torch.mm(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{name:12s} {m}x{n}x{k}: {us:7.1f} us  {2 * m * n * k / us / 1e6:7.0f} TFLOP/s")
