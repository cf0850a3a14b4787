"""f3 for real: ``Embeddings(...).index(list[dict])`` (heavy_ranker.py:86) through the HIP
encoder at index-build batch sizes -- host tokenizer hand-off, TextEncoder chunking to the encoder workspace, vectors
into the shard, save, load, search -- with the stored vectors checked against the fp64 oracle encoder.  Needs an MI355X."""
import time
import zlib

import numpy as np
import pytest
import torch

from oracle import encoder as E

pytestmark = pytest.mark.gpu

N_DOCS, L_MAX = 20480, 128
CFG = dict(E.PHOBERT_BASE, vocab_size=8000)  # PhoBERT-base layers (768 / 12 heads / 3072, 12 layers); a short vocabulary keeps
                                             # the host-side weight generation and the oracle's embedding table small


def _docs(n):
    rng = np.random.default_rng(77)
    words = [f"từ{j}" for j in range(3000)]
    lens = rng.integers(20, 120, size=n)
    return [{"id": i + 1, "text": " ".join(words[j] for j in rng.integers(0, 3000, size=lens[i])), "source": "synthetic"}
            for i in range(n)]


def _tokenizer(texts):
    """Whitespace words -> ids in [3, vocab) (a stand-in for PhoBERT's segmentation + BPE, which needs files that are not
    available offline): <s> = 0, </s> = 2, pad = 1, truncated to L_MAX, padded to the batch's longest sequence."""
    rows = [[0] + [3 + (zlib.crc32(w.encode()) % (CFG["vocab_size"] - 3)) for w in t.split()][:L_MAX - 2] + [2] for t in texts]
    width = max(len(r) for r in rows)
    ids = np.full((len(rows), width), CFG["pad_id"], dtype=np.int32)
    mask = np.zeros((len(rows), width), dtype=np.int32)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
        mask[i, :len(r)] = 1
    return ids, mask


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def test_index_documents_through_the_hip_encoder(native_lib, tmp_path):
    from vietnamese_qa_system_amd import Embeddings
    from vietnamese_qa_system_amd.encoder import QuestionEncoder, TextEncoder
    w = E.synthetic_weights(CFG, seed=3)
    enc = QuestionEncoder(w, CFG, max_tokens=16384)  # the default workspace: 256 x 128 tokens do NOT fit in one call
    text_encoder = TextEncoder(_tokenizer, enc, pooling="mean", normalize=True, batch_size=256)
    docs = _docs(N_DOCS)
    emb = Embeddings(content=True, encoder=text_encoder, dtype="fp16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb.index(docs)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    assert emb.count() == N_DOCS
    print(f"index build: {N_DOCS} docs in {build_s:.2f} s = {N_DOCS / build_s:.0f} docs/s (host tokenizer included)")
    # stored vectors against the fp64 oracle encoder on a sample of documents (cosine >= 0.999)
    sample = [0, 1, 777, N_DOCS - 1]
    rows, _ = emb._index.get_rows()
    ids, mask = _tokenizer([docs[i]["text"] for i in sample])
    ref = E.encode(w, CFG, ids, mask, pooling="mean")
    got = rows[sample].astype(np.float32)
    assert _cos(got, ref).min() > 0.999, _cos(got, ref)
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 2e-3  # unit rows (fp16 rounding)
    # searching with a document's own text: the scores are the brute-force scores of the stored rows (random-init encoders
    # give near-parallel vectors, so ranks inside 1e-5 may swap: scores are compared, and the document itself must score
    # within fp16 noise of the best), texts come back through the doc-text join; save / load keep all of it
    picks = (5, 4242, 20000)
    queries = [docs[i]["text"] for i in picks]
    res = emb.batchsearch(queries, 3)
    qv = text_encoder(queries).cpu().numpy().astype(np.float16).astype(np.float32)
    brute = np.sort(qv @ rows.astype(np.float32).T, axis=1)[:, ::-1][:, :3]
    assert np.abs(np.array([[h["score"] for h in r] for r in res]) - brute).max() < 1e-4
    for r, i in zip(res, picks):
        self_score = float(qv[picks.index(i)] @ rows[i].astype(np.float32))
        assert self_score > 0.99 and r[0]["score"] - self_score < 5e-3
        assert all(h["text"] == docs[h["id"] - 1]["text"] for h in r)
    emb.save(str(tmp_path / "ix"))
    back = Embeddings(encoder=text_encoder).load(str(tmp_path / "ix"))
    assert back.load_stats["bytes"] == N_DOCS * 768 * 2
    res2 = back.batchsearch(queries, 3)
    assert [[h["id"] for h in r] for r in res2] == [[h["id"] for h in r] for r in res]
    assert res2[2][0]["text"] == docs[res2[2][0]["id"] - 1]["text"]
    # a re-save of the LOADED index keeps the documents
    back.save(str(tmp_path / "ix2"))
    again = Embeddings(encoder=text_encoder).load(str(tmp_path / "ix2"))
    top = again.search(queries[0], 1)[0]
    assert top["text"] == docs[top["id"] - 1]["text"]
    enc.close()


def test_out_of_range_token_ids_are_refused(native_lib):
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    tiny = dict(vocab_size=100, hidden=64, layers=1, heads=4, ffn=128, max_pos=40, type_vocab=1, pad_id=1, ln_eps=1e-5)
    w = E.synthetic_weights(tiny, seed=1)
    enc = QuestionEncoder(w, tiny, max_tokens=64)
    ids = np.array([[0, 5, 100, 2]], dtype=np.int32)  # 100 = one past the table
    mask = np.ones_like(ids)
    with pytest.raises(ValueError, match="token ids"):
        enc.forward(ids, mask)                         # host ids: refused before anything is launched
    with pytest.raises(ValueError, match="token ids"):
        enc.forward(np.array([[0, -3, 2, 1]], dtype=np.int32), mask)
    # device-resident ids: the kernel embeds them as pad (no out-of-bounds read) and the NEXT call reports it
    bad = torch.from_numpy(ids).cuda()
    out = enc.forward(bad, torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    good = np.array([[0, 5, 7, 2]], dtype=np.int32)
    with pytest.raises(ValueError, match="token ids outside"):
        enc.forward(good, mask)
    ref = E.encode(w, tiny, good, mask, pooling="cls")
    got = enc.forward(good, mask).cpu().numpy()        # the flag was consumed: the handle works again
    assert _cos(got, ref).min() > 0.999
    enc.close()
