"""Dev probe: vqa_index_search_host on small fp16 shards, one question, k = 1 -- the ONE-kernel form (csrc/tiny_search.hip) against the
general launches (options one_launch = 0): wall time of the raw C call (arguments prepared once), of DeviceIndex.search_host and of
Embeddings.search.  Under `rocprofv3 --kernel-trace --stats` the kernel table gives the device side."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vietnamese_qa_system_amd import Embeddings, _native as N
from vietnamese_qa_system_amd.index import DeviceIndex

rng = np.random.default_rng(0)
d = 768
reps = int(os.environ.get("REPS", "200"))


def wall(fn):
    for _ in range(10):
        fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    return f"{np.median(t):6.1f} us (p10 {np.percentile(t, 10):5.1f}, p90 {np.percentile(t, 90):5.1f})"


for n in (1000, 5000, 16384, 20000, 50000, 131072):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = rng.standard_normal((1, d)).astype(np.float32)
    q16 = rng.standard_normal((16, d)).astype(np.float32)
    for dtype, one in ((t, o) for t in os.environ.get("DTYPES", "fp16,fp32").split(",") for o in (1, 0)):
        ix = DeviceIndex(x, id_base=1, dtype=dtype, device=0, options={"one_launch": one})
        s, i = np.empty((1, 1), np.float32), np.empty((1, 1), np.int64)
        stream = torch.cuda.current_stream(0).cuda_stream
        lib = ix._lib
        raw = lambda: lib.vqa_index_search_host(ix._handle, q.ctypes.data, N.VQA_F32, 1, 1, 1, s.ctypes.data, i.ctypes.data, None, stream)
        print(f"n={n:6d} {dtype} one_launch={one}: C call {wall(raw)} | DeviceIndex.search_host {wall(lambda: ix.search_host(q, 1, normalize=True))}", flush=True)
        print("      " + " | ".join(f"B={b} k={k}: {wall(lambda: ix.search_host(q16[:b], k, normalize=True))}" for b, k in ((1, 10), (4, 16), (16, 4))), flush=True)
        ix.close()
emb = Embeddings(dtype="fp16", device=0)
x = rng.standard_normal((5000, d)).astype(np.float32)
emb.index_vectors(np.arange(1, 5001), x)
q = rng.standard_normal(d).astype(np.float32)
print(f"Embeddings.search(vector, 1), 5000 docs: {wall(lambda: emb.search(q, 1))}")
