"""Dev probe: the vendor GEMM (torch -> hipBLASLt / rocBLAS) on the encoder's GEMM shapes, fp16, random normal data -- a measuring
stick for gemm_tile_kernel, never part of the product.  Three forms per shape, because gemm_tile_kernel's launches are not bare
GEMMs: (a) torch.mm, no epilogue; (b) the vendor's own fused epilogue where it has one (torch.addmm = + bias; torch._addmm_activation
= + bias + GELU, the FFN1 form); (c) what a vendor-GEMM forward would really run for that launch -- the bare GEMM plus the element-wise
kernels that gemm_tile_kernel folds into its epilogue (bias + erf GELU for FFN1; bias + residual add + LayerNorm for out-proj / FFN2)."""
import torch
import torch.nn.functional as F
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, (m, n, k, kind) in {"QKV": (8192, 2304, 768, "bias"), "out-proj": (8192, 768, 768, "res_ln"), "FFN1": (8192, 3072, 768, "gelu"),
                              "FFN2": (8192, 768, 3072, "res_ln"), "QKV packed": (5114, 2304, 768, "bias"),
                              "out-proj packed": (5114, 768, 768, "res_ln"), "FFN1 packed": (5114, 3072, 768, "gelu"),
                              "FFN2 packed": (5114, 768, 3072, "res_ln")}.items():
    a = torch.randn((m, k), generator=g, device=dev).half()
    w = torch.randn((n, k), generator=g, device=dev).half() * 0.03
    bias = torch.randn((n,), generator=g, device=dev).half()
    res = torch.randn((m, n), generator=g, device=dev).half()
    gamma, beta = torch.ones(n, device=dev).half(), torch.zeros(n, device=dev).half()
    wt = w.t()
    bare = timed(lambda: torch.mm(a, wt))
    if kind == "gelu":
        fused = timed(lambda: torch._addmm_activation(bias, a, wt, use_gelu=True))
        full = timed(lambda: F.gelu(torch.addmm(bias, a, wt)))
        what = "+bias+GELU(tanh) epilogue", "addmm + erf GELU kernel"
    elif kind == "bias":
        fused = timed(lambda: torch.addmm(bias, a, wt))
        full = fused
        what = "+bias epilogue", "the same"
    else:
        fused = timed(lambda: torch.addmm(bias, a, wt))
        full = timed(lambda: F.layer_norm(torch.addmm(bias, a, wt) + res, (n,), gamma, beta, 1e-5))
        what = "+bias epilogue", "addmm + residual add + LayerNorm kernels"
    print(f"{name:16s} {m}x{n}x{k}: bare {bare:6.1f} us ({2 * m * n * k / bare / 1e6:5.0f} TFLOP/s) | vendor fused ({what[0]}) {fused:6.1f} us | "
          f"like for like ({what[1]}) {full:6.1f} us", flush=True)
