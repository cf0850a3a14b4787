"""Dev probe: the reference's loop body (heavy_ranker.py:98-101) -- ONE question through its two retrievers (MiniLM-L12 shape, d = 384,
and XLM-R-base shape, d = 768; random weights, a stand-in tokenizer, 5000 documents each): the two `search(question, 1)` calls one after
the other against `heavy_ranker.rank_query` (the two forwards on a stream each)."""
import os, sys, time, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd import Embeddings, heavy_ranker
from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE, TextEncoder

device = torch.device("cuda", 0)
rng = np.random.default_rng(0)
embs = []
for cfg in (MINILM_L12, XLMR_BASE):
    enc, *_ = bench.make_encoder(torch, device, 0, 1, 32, max_tokens=64, cfg=cfg)

    def tokenize(texts, cfg=cfg):
        ids = np.full((len(texts), 32), cfg["pad_id"], np.int32)
        mask = np.zeros((len(texts), 32), np.int32)
        for i, t in enumerate(texts):
            toks = ([0] + [5 + zlib.crc32(x.encode()) % (cfg["vocab_size"] - 5) for x in t.split()] * 4)[:31] + [2]
            ids[i, :len(toks)] = toks
            mask[i, :len(toks)] = 1
        return ids, mask

    emb = Embeddings(content=False, encoder=TextEncoder(tokenize, enc, pooling="mean"))
    emb.index_vectors(np.arange(1, 5001), rng.standard_normal((5000, cfg["hidden"])).astype(np.float32))
    embs.append(emb)
text = "xin chao day la mot cau hoi ve luat giao thong duong bo"
a, b = embs
seq = lambda: (a.search(text, 1), b.search(text, 1))
par = lambda: heavy_ranker.rank_query(a, b, text, 1)
assert seq() == par(), (seq(), par())
for name, fn in (("two search() calls, one after the other", seq), ("heavy_ranker.rank_query (two streams)", par)):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(100):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{name:44s}: median {np.median(ts):.3f} ms  p10 {np.percentile(ts, 10):.3f}  p90 {np.percentile(ts, 90):.3f}", flush=True)
