"""Dev probe: where the HOST time of the reference's call goes -- `Embeddings(hybrid=True, content=True)` built, saved, loaded
(heavy_ranker.py:78-94), then `search(question, 1)` x 300 under cProfile (PhoBERT-base-shaped encoder, stand-in tokenizer, 5000 documents)."""
import cProfile, os, pstats, sys, tempfile, time, zlib
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
te = TextEncoder(tokenize, enc, pooling="mean")
built = Embeddings(hybrid=True, content=True, encoder=te)
built.index(docs)
d = tempfile.mkdtemp()
built.save(d)
emb = Embeddings(encoder=te).load(d)
question = " ".join(rng.choice(words, 12))
for _ in range(10):
    emb.search(question, 1)
ts = []
for _ in range(100):
    t0 = time.perf_counter()
    r = emb.search(question, 1)
    ts.append(time.perf_counter() - t0)
print(f"loaded hybrid + content index: search(question, 1) median {np.median(ts) * 1e3:.3f} ms -> {r[0]['id']} {r[0]['score']:.4f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    emb.search(question, 1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
