"""Dev probe: wall-clock latency of the Embeddings API at the reference's own scale -- one query at a time
(heavy_ranker.py:97-101) against a few thousand documents: vector queries and text queries through a PhoBERT-base-shaped
random-weight encoder with a stand-in tokenizer."""
import os, sys, time, zlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import encoder as E
from vietnamese_qa_system_amd import Embeddings
from vietnamese_qa_system_amd.encoder import QuestionEncoder, TextEncoder

cfg = dict(E.PHOBERT_BASE)
w = E.synthetic_weights(cfg, seed=0)
enc = QuestionEncoder(w, cfg, max_tokens=256 * 32)


def tokenize(texts):
    ids = np.full((len(texts), 32), cfg["pad_id"], np.int32)
    mask = np.zeros((len(texts), 32), np.int32)
    for i, t in enumerate(texts):
        toks = [0] + [5 + zlib.crc32(x.encode()) % (cfg["vocab_size"] - 5) for x in t.split()][:30] + [2]
        ids[i, :len(toks)] = toks
        mask[i, :len(toks)] = 1
    return ids, mask


n = 5000
rng = np.random.default_rng(0)
vecs = rng.standard_normal((n, 768)).astype(np.float32)
emb = Embeddings(content=False, encoder=TextEncoder(tokenize, enc, pooling="cls"))  # text queries of <= 64 questions: forward_host + search_host
emb.index_vectors(np.arange(1, n + 1), vecs)
qv = rng.standard_normal((1, 768)).astype(np.float32)
text = "xin chao day la mot cau hoi ve luat giao thong duong bo"
for name, fn in (("vector query, limit 1", lambda: emb.search(qv[0], 1)), ("text query, limit 1", lambda: emb.search(text, 1)),
                 ("text query, limit 3", lambda: emb.search(text, 3)),
                 ("256 vector queries (batchsearch)", lambda: emb.batchsearch(np.repeat(qv, 256, 0), 10))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(50):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print(f"{name:36s}: median {np.median(t):.3f} ms  p10 {np.percentile(t, 10):.3f}  p90 {np.percentile(t, 90):.3f}", flush=True)
