"""Differential fuzz of the one-launch search (csrc/tiny_search.hip) against the general launches: random shard sizes (1 ... 262 144 rows,
clustered around the workgroup and tile edges), row lengths (any d, not only multiples of 8), storage types, id vectors, question counts
and k inside its limits, raw / normalised fp32 and fp16 questions, host and device resident, duplicated rows.  Every case must return
the general path's scores, ids and positions bit for bit.  GPU box:  python scripts/fuzz_one_launch.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vietnamese_qa_system_amd.index import DeviceIndex

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
edges = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 5000, 16383, 16384, 16385, 16447, 16448, 16449,
         50000, 65535, 65536, 131071, 131072, 131073, 200000, 262143, 262144]
bad = 0
t0 = time.time()
for c in range(cases):
    n = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 262145))
    d = int(rng.choice([8, 24, 64, 100, 128, 200, 384, 768, 1024])) if rng.random() < 0.7 else int(rng.integers(1, 400))
    if n * d > 60_000_000:
        d = max(1, 60_000_000 // n)
    dtype = "fp16" if rng.random() < 0.6 else "fp32"
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    if n > 4:  # duplicated rows: equal scores, positions ascending
        for _ in range(3):
            i, j = rng.integers(0, n, 2)
            x[i] = x[j]
    ids = (rng.permutation(n).astype(np.int64) * 3 + 5) if rng.random() < 0.5 else None
    one = DeviceIndex(x, ids=ids, id_base=1, dtype=dtype, device=0)
    gen = DeviceIndex(x, ids=ids, id_base=1, dtype=dtype, device=0, options={"one_launch": 0})
    for _ in range(4):
        k = int(rng.integers(1, 33))
        b = int(rng.integers(1, min(16, 64 // k) + 1))
        q = rng.standard_normal((b, d)).astype(np.float32)
        if rng.random() < 0.3:
            q[0] = x[rng.integers(0, n)] * 3
        form = rng.integers(0, 4)
        if form == 0:
            qa, norm = q, True
        elif form == 1:
            qa, norm = q, False
        elif form == 2:
            qa, norm = q.astype(np.float16), False
        else:
            qa, norm = torch.from_numpy(q).cuda(), bool(rng.integers(0, 2))
        a = one.search_host(qa, k, normalize=norm, return_positions=True)
        g = gen.search_host(qa.cpu().numpy() if isinstance(qa, torch.Tensor) else qa, k, normalize=norm, return_positions=True)
        if not all(np.array_equal(u, v, equal_nan=True) for u, v in zip(a, g)):
            bad += 1
            print(f"MISMATCH case {c}: n={n} d={d} {dtype} ids={ids is not None} b={b} k={k} form={form}", flush=True)
    one.close()
    gen.close()
print(f"{cases} shards x 4 calls, {bad} mismatching calls, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
