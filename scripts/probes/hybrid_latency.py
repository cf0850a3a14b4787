"""Dev probe: `Embeddings(hybrid=True, content=True)` -- the constructor of heavy_ranker.py:78 -- asked one text question with limit 1:
wall clock, and its parts (dense half, BM25 half, document join).  PhoBERT-base-shaped encoder, stand-in tokenizer, 5000 documents of 40
words from a 3000-word vocabulary."""
import os, sys, time, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vietnamese_qa_system_amd import Embeddings
from vietnamese_qa_system_amd.encoder import PHOBERT_BASE, TextEncoder
device = torch.device("cuda", 0)
cfg = PHOBERT_BASE
enc, *_ = bench.make_encoder(torch, device, 0, 64, 64, cfg=cfg)


def tokenize(texts):
    ids = np.full((len(texts), 48), cfg["pad_id"], np.int32)
    mask = np.zeros((len(texts), 48), np.int32)
    for i, t in enumerate(texts):
        toks = ([0] + [5 + zlib.crc32(x.encode()) % (cfg["vocab_size"] - 5) for x in t.split()])[:47] + [2]
        ids[i, :len(toks)] = toks
        mask[i, :len(toks)] = 1
    return ids, mask


rng = np.random.default_rng(0)
words = [f"tu{i}" for i in range(3000)]
docs = [{"id": i + 1, "text": " ".join(rng.choice(words, 40)), "source": "s"} for i in range(5000)]
for hybrid in (True, False):
    emb = Embeddings(hybrid=hybrid, content=True, encoder=TextEncoder(tokenize, enc, pooling="mean"))
    emb.index(docs)
    question = " ".join(rng.choice(words, 12))
    for _ in range(5):
        r = emb.search(question, 1)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        r = emb.search(question, 1)
        ts.append(time.perf_counter() - t0)
    print(f"hybrid={hybrid}: search(question, 1) median {np.median(ts) * 1e3:.3f} ms  -> {r[0]['id']} {r[0]['score']:.4f}", flush=True)
    if hybrid:
        t0 = time.perf_counter()
        for _ in range(50):
            emb._sparse.search(question, 10)
        print(f"   BM25 half alone: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
