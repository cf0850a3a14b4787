#!/usr/bin/env python3
"""Runs the BASELINE.json configs that fit one MI355X and writes profiles/<tag>_configs.json.

    configs[0]  1k x 768 random index, cosine top-10 through the Embeddings API (plumbing) + the CPU oracle beside it
    configs[1]  PhoBERT-base-shaped question encoder (random weights) + 1M x 768 fp32 index, top-10
    configs[2]  10M x 768 fp16 index, batch 256 (the headline; bench.py reports it with roofline / cpu_baseline)
    configs[4]  per-GPU share of 100M x 768 fp8 on 8 GPUs = 12.5M rows, fp8 MFMA scoring, recall@10 vs fp32 on a prefix

configs[3] (80M rows over 8 GPUs) needs the 8-GPU node: `bench.py --gpus 8` is its per-rank 10M-row shard.
The oracle is used as the checker / CPU baseline only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import encoder as E  # noqa: E402
from oracle import retrieval as R  # noqa: E402
from vietnamese_qa_system_amd import Embeddings  # noqa: E402
from vietnamese_qa_system_amd.encoder import QuestionEncoder  # noqa: E402
from vietnamese_qa_system_amd.index import DeviceIndex  # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def unit_rows(n, d, seed, dtype=torch.float32, chunk=1 << 18):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    buf = torch.empty((n, d), dtype=dtype, device="cuda")
    for c0 in range(0, n, chunk):
        x = torch.randn((min(n, c0 + chunk) - c0, d), generator=gen, device="cuda")
        buf[c0:c0 + x.shape[0]] = (x / x.norm(dim=1, keepdim=True)).to(dtype)
    return buf


def config0():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 768)).astype(np.float32)
    q = rng.standard_normal((256, 768)).astype(np.float32)
    emb = Embeddings(min_score=None)
    emb.index_vectors(list(range(1, 1001)), x)
    res = emb.batchsearch(q, 10)
    ms = timed(lambda: emb.batchsearch(q, 10), 20)
    t0 = time.perf_counter()
    _, ref_i, _ = R.search(R.l2_normalize(q), R.l2_normalize(x), 10, id_base=1)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    got = np.array([[h[0] for h in r] for r in res])
    return {"config": "1k x 768 random index, cosine top-10 via the Embeddings API", "batch": 256, "ms_per_batch_gpu_api": round(ms, 3),
            "queries_per_s_gpu_api": round(256 / ms * 1e3, 1), "cpu_oracle_ms": round(cpu_ms, 2),
            "recall_at_10_vs_cpu_fp32": R.recall_at_k(got, ref_i)}


def config1(n=1_000_000, b=256, l=32):
    cfg = dict(E.PHOBERT_BASE)
    w = E.synthetic_weights(cfg, seed=0)
    ids, mask = E.synthetic_tokens(cfg, b, l, seed=1)
    enc = QuestionEncoder(w, cfg, max_tokens=b * l)
    rows = unit_rows(n, 768, 11)
    ix = DeviceIndex(rows, id_base=1, dtype="fp32")
    ids_t, mask_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()

    def encode():
        return enc.forward(ids_t, mask_t, pooling="cls")

    def e2e():
        return ix.search(encode(), 10)

    enc_ms = timed(encode, 10)
    e2e_ms = timed(e2e, 10)
    qv = encode()
    ix.set_timing(True)
    s, i, p = ix.search(qv, 10, return_positions=True)
    torch.cuda.synchronize()
    k_ms, _ = ix.get_timing()
    kn = int(ix.launch_info(b, 10).rows_per_launch)
    # parity: encoder vs the fp64 oracle on 8 sequences; scoring vs the oracle on the SAME (GPU-encoded) queries
    ref_q = E.encode(w, cfg, ids[:8], mask[:8], pooling="cls")
    cos = float(((qv[:8].cpu().numpy() * ref_q).sum(1) / np.linalg.norm(ref_q, axis=1)).min())
    xs = rows[:200_000].cpu().numpy()
    pre = DeviceIndex(rows[:200_000], id_base=1, dtype="fp32")
    _, _, pp = pre.search(qv[:32], 10, return_positions=True)
    torch.cuda.synchronize()
    _, _, ref_p = R.search(qv[:32].cpu().numpy(), xs, 10)
    tokens = int(mask.sum())
    # CLS pooling: the last layer's out-projection + FFN run on the b first rows only
    flops = b * l * (12 * (2 * 768 * 2304 + 4 * l * 768) + 11 * 2 * (768 * 768 + 2 * 768 * 3072)) + b * 2 * (768 * 768 + 2 * 768 * 3072)
    return {"config": f"PhoBERT-base-shaped encoder (random weights, B={b}, L={l}, {tokens} real tokens) + {n} x 768 fp32 index, top-10",
            "encoder_ms": round(enc_ms, 3), "encoder_tflops": round(flops / enc_ms / 1e9, 1),
            "encoder_roofline": {"bound": "mfma", "achieved": round(flops / enc_ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": round(flops / enc_ms / 1e9 / 2500.0, 4),
                                 "note": "whole forward over its wall time; flops = B L (12 (2 H 3H + 4 L H) + 11 x 2 (H H + 2 H F)) + B x 2 (H H + 2 H F): the last layer past attention runs on B rows"},
            "scoring_kernel_ms": round(k_ms, 3),
            "scoring_kernel_rows": kn, "scoring_fp32_mfma_tflops": round(2 * 256 * kn * 768 / k_ms / 1e9, 1), "end_to_end_ms": round(e2e_ms, 3),
            "queries_per_s_end_to_end": round(b / e2e_ms * 1e3, 1), "encoder_min_cosine_vs_fp64_oracle": cos,
            "recall_at_10_vs_cpu_oracle_200k_prefix": R.recall_at_k(pp.cpu().numpy(), ref_p)}


def config4(n=12_500_000, b=256):
    rows = unit_rows(n, 768, 1234)
    q = unit_rows(b, 768, 99)
    ix = DeviceIndex(rows, id_base=1, dtype="fp8")
    ms = timed(lambda: ix.search(q, 10), 10)
    ix.set_timing(True)
    ix.search(q, 10)
    k_ms, _ = ix.get_timing()
    kn = int(ix.launch_info(b, 10).rows_per_launch)  # rows the timed (main) launch covers: two-stage search scores the first 10 % apart
    # recall of the fp8 index against the fp32 reference on a 1M-row prefix (both on the GPU path; the fp32 path is
    # itself checked against the oracle in tests/test_gpu_dtypes.py), and against the CPU oracle on a 200k prefix
    pre8 = DeviceIndex(rows[:1_000_000], dtype="fp8")
    pre32 = DeviceIndex(rows[:1_000_000], dtype="fp32")
    _, i8, _ = pre8.search(q, 10)
    _, i32, _ = pre32.search(q, 10)
    torch.cuda.synchronize()
    xs = rows[:200_000].cpu().numpy()
    small = DeviceIndex(rows[:200_000], dtype="fp8")
    _, _, p8 = small.search(q[:32], 10, return_positions=True)
    torch.cuda.synchronize()
    qn = q[:32].cpu().numpy()
    _, _, ref8 = R.search(R.e4m3_decode(R.e4m3_encode(qn * 16)), R.e4m3_encode(xs * 16), 10, dtype=R.DTYPE_FP8_E4M3)
    return {"config": f"per-GPU share of 100M x 768 fp8 (e4m3) on 8 GPUs = {n} rows, batch {b}, top-10", "ms_per_batch": round(ms, 3),
            "queries_per_s": round(b / ms * 1e3, 1), "scoring_kernel_ms": round(k_ms, 3),
            "scoring_kernel_rows": kn, "hbm_gbs": round(kn * 768 / k_ms / 1e6, 1), "mfma_tflops": round(2 * 256 * kn * 768 / k_ms / 1e9, 1),
            "recall_at_10_fp8_vs_fp32_index_1M_prefix": R.recall_at_k(i8.cpu().numpy(), i32.cpu().numpy()),
            "recall_at_10_vs_cpu_oracle_same_codes_200k_prefix": R.recall_at_k(p8.cpu().numpy(), ref8)}


def index_build(n_docs=65536, l_max=128):
    """f3: Embeddings.index(list[dict]) (heavy_ranker.py:86) through a host tokenizer stand-in + the HIP encoder: docs/s."""
    import zlib
    from vietnamese_qa_system_amd.encoder import TextEncoder
    cfg = dict(E.PHOBERT_BASE, vocab_size=8000)
    w = E.synthetic_weights(cfg, seed=3)
    enc = QuestionEncoder(w, cfg, max_tokens=256 * l_max)
    rng = np.random.default_rng(77)
    words = [f"từ{j}" for j in range(3000)]
    lens = rng.integers(20, 120, size=n_docs)
    docs = [{"id": i + 1, "text": " ".join(words[j] for j in rng.integers(0, 3000, size=lens[i]))} for i in range(n_docs)]

    def tok(texts):
        rows = [[0] + [3 + (zlib.crc32(t.encode()) % 7997) for t in x.split()][:l_max - 2] + [2] for x in texts]
        width = max(len(r) for r in rows)
        ids = np.full((len(rows), width), cfg["pad_id"], dtype=np.int32)
        mask = np.zeros_like(ids)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
            mask[i, :len(r)] = 1
        return ids, mask

    t0 = time.perf_counter()
    toks = [tok([d["text"] for d in docs[c:c + 256]]) for c in range(0, n_docs, 256)]
    tok_s = time.perf_counter() - t0
    it = iter(toks)
    emb = Embeddings(encoder=TextEncoder(lambda texts: next(it), enc, pooling="mean", batch_size=256), dtype="fp16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb.index(docs, batch_size=256)
    torch.cuda.synchronize()
    enc_s = time.perf_counter() - t0
    tokens = int(sum(m.sum() for _, m in toks))
    padded = int(sum(m.size for _, m in toks))
    return {"config": f"Embeddings.index of {n_docs} synthetic documents (20-120 words, L <= {l_max}) through the HIP encoder, mean pooling",
            "docs": n_docs, "real_tokens": tokens, "padded_tokens": padded, "host_tokenizer_seconds": round(tok_s, 2),
            "encode_and_index_seconds": round(enc_s, 3), "docs_per_s_encode_and_index": round(n_docs / enc_s, 1),
            "docs_per_s_with_host_tokenizer": round(n_docs / (enc_s + tok_s), 1), "padded_tokens_per_s": round(padded / enc_s, 1)}


def load_rate(n=10_000_000, d=768):
    """f1: save a fp16 shard, load it back (file -> pinned host buffers -> HBM, double buffered): GB/s.  The files were just
    written, so they come from the page cache: this prices the host -> device path, not the disk."""
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 2.5 * n * d * 2 else tempfile.gettempdir()
    free = shutil.disk_usage(base).free
    while n * d * 2 * 1.2 > free and n > 1_000_000:
        n //= 2
    path = tempfile.mkdtemp(prefix="vqa_load_", dir=base)
    try:
        rows = unit_rows(n, d, 5, dtype=torch.float16)
        emb = Embeddings(dtype="fp16", min_score=None)
        emb.index_vectors(None, rows)
        q = unit_rows(16, d, 6)
        before = emb.batchsearch(q, 5)
        t0 = time.perf_counter()
        emb.save(path)
        save_s = time.perf_counter() - t0
        del emb, rows
        torch.cuda.empty_cache()
        back = Embeddings(min_score=None).load(path)
        same = back.batchsearch(q, 5) == before
        st = back.load_stats
        return {"config": f"save + load of a {n} x {d} fp16 shard ({n * d * 2 / 1e9:.2f} GB) under {base}", "save_seconds": round(save_s, 2),
                "save_gb_per_s": round(n * d * 2 / save_s / 1e9, 2), "load_seconds": round(st["seconds"], 2),
                "load_gb_per_s": round(st["gb_per_s"], 2), "results_identical_after_load": bool(same)}
    finally:
        shutil.rmtree(path, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--only", default="0,1,4,build,load")
    args = ap.parse_args()
    out = {}
    for c in args.only.split(","):
        out[f"configs[{c}]"] = {"0": config0, "1": config1, "4": config4, "build": index_build, "load": load_rate}[c]()
        torch.cuda.empty_cache()
        print(json.dumps({f"configs[{c}]": out[f"configs[{c}]"]}), flush=True)
    path = os.path.join(ROOT, "gpurun_out", f"{args.tag}_configs.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
