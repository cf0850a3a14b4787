"""The txtai-shaped Embeddings boundary end to end on the GPU (vectors in, results out), the on-disk format, the
content join, the driver counterpart, and the element-wise helpers -- against the oracle."""
import numpy as np
import pytest
import torch

from oracle import retrieval as R

pytestmark = pytest.mark.gpu


def _corpus(n=3000, d=768, b=20, seed=21):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d)).astype(np.float32), rng.standard_normal((b, d)).astype(np.float32)


def test_normalize_convert_matches_oracle(native_lib):
    from vietnamese_qa_system_amd.index import DeviceIndex
    x, _ = _corpus(2000, 100)
    x[5] = 0  # zero row stays zero
    ix = DeviceIndex(x, dtype="fp16", normalize=True)
    rows, ids = ix.get_rows()
    ref = R.l2_normalize(x)
    assert ids is None and rows.shape == (2000, 100)
    # the norm's summation order differs (wave shuffle vs numpy pairwise): allow one fp16 ulp
    assert np.abs(rows.astype(np.float32) - ref.astype(np.float16).astype(np.float32)).max() <= 2 ** -11
    assert (rows == ref.astype(np.float16)).mean() > 0.999
    assert np.all(rows[5] == 0)
    ix.close()


def test_set_rows_get_rows_roundtrip_chunked(native_lib):
    from vietnamese_qa_system_amd.index import DeviceIndex
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1000, 72)).astype(np.float16)  # d not a multiple of 64; ragged last tile
    ids = rng.permutation(10_000)[:1000].astype(np.int64)
    ix = DeviceIndex.empty(1000, 72, with_ids=True)
    ix.set_rows(0, x[:300], ids[:300])
    ix.set_rows(300, torch.from_numpy(x[300:]).cuda(), ids[300:])  # device source
    rows, got_ids = ix.get_rows()
    assert np.array_equal(rows, x) and np.array_equal(got_ids, ids)
    q = rng.standard_normal((3, 72)).astype(np.float16)
    s, i, p = ix.search(torch.from_numpy(q).cuda(), 5, return_positions=True)
    torch.cuda.synchronize()
    R.check_topk(s.cpu().numpy(), p.cpu().numpy(), R.full_scores(q.astype(np.float32), x, R.DTYPE_F16), 5, score_tol=2e-4,
                 tie_tol=2e-5)  # un-normalised rows: scores ~ 70 x larger than unit vectors
    assert np.array_equal(i.cpu().numpy(), ids[p.cpu().numpy()])
    with pytest.raises(ValueError):
        ix.set_rows(900, x[:200], ids[:200])
    ix.close()


def test_embeddings_vectors_in_results_out(native_lib):
    from vietnamese_qa_system_amd import Embeddings
    x, q = _corpus()
    ids = list(range(1, x.shape[0] + 1))  # sqlite rowids (setup_db.py:14)
    emb = Embeddings(min_score=None)
    emb.index_vectors(ids, x)
    assert emb.count() == 3000
    res = emb.batchsearch(q, 10)
    x16 = R.l2_normalize(x).astype(np.float16)
    q16 = R.l2_normalize(q).astype(np.float16)
    ref_s, ref_i, _ = R.search(q16.astype(np.float32), x16, 10, dtype=R.DTYPE_F16, id_base=1)
    got_i = np.array([[h[0] for h in r] for r in res])
    got_s = np.array([[h[1] for h in r] for r in res], dtype=np.float32)
    assert R.recall_at_k(got_i, ref_i) > 0.995  # device-side normalisation may flip a last-ulp near-tie
    assert np.abs(got_s - ref_s).max() < 1e-4
    assert all(isinstance(h[0], int) and isinstance(h[1], float) for h in res[0])
    one = emb.search(q[0], 1)  # heavy_ranker.py:98 calling pattern: one query, limit 1
    assert one[0][0] == res[0][0][0]
    assert len(emb.search(q[0])) == 3  # txtai default limit
    with pytest.raises(ValueError):
        emb.search(q[:2], 1)
    with pytest.raises(ValueError):
        emb.batchsearch(q[:, :10], 1)


def test_embeddings_min_score_filter_is_host_side(native_lib):
    from vietnamese_qa_system_amd import Embeddings
    x, q = _corpus(50, 64, 4)
    emb = Embeddings()  # txtai default: drop score <= 0
    emb.index_vectors(None, x)
    for r in emb.batchsearch(q, 12):
        assert all(sc > 0 for _, sc in r) and len(r) <= 12
    emb.min_score = None
    assert all(len(r) == 12 for r in emb.batchsearch(q, 12))


def test_embeddings_save_load_content_and_driver(native_lib, tmp_path):
    from vietnamese_qa_system_amd import Embeddings, heavy_ranker, docstore
    x, q = _corpus(400, 128, 6)
    docs = [{"id": i + 1, "text": f"tài liệu số {i + 1}", "source": "wiki"} for i in range(400)]
    emb = Embeddings(content=True, path="sentence-transformers/paraphrase-multilingual-mpnet-base-v2", min_score=None)
    emb.index(docs, vectors=x)
    before = emb.batchsearch(q, 5)
    assert set(before[0][0]) == {"id", "text", "score"} and before[0][0]["text"] == docs[before[0][0]["id"] - 1]["text"]
    emb.save(str(tmp_path / "mpnet"))
    emb2 = Embeddings().load(str(tmp_path / "mpnet"))  # heavy_ranker.py:91-94
    emb2.min_score = None
    after = emb2.batchsearch(q, 5)
    assert after == before
    assert emb2.content and emb2.count() == 400
    # driver counterpart of heavy_ranker.py:97-115 with two "models" (same index twice -> always agreeing ids)
    db = str(tmp_path / "documents.db")
    docstore.write_documents(db, docs)
    out = heavy_ranker.rank_queries(emb2, emb2, (q, q), db)
    for row in out:
        assert row["id_a"] == row["id_b"] and row["doc_a"] == docs[row["id_a"] - 1]["text"]
        assert row["match"] == (row["score_a"] + row["score_b"] > 0.4)


def test_embeddings_string_ids_and_text_encoder_hook(native_lib):
    from vietnamese_qa_system_amd import Embeddings
    x, _ = _corpus(64, 64, 1)
    table = {f"text {i}": torch.from_numpy(x[i]) for i in range(64)}

    def fake_encoder(texts):  # stands in for a tokenizer + question encoder: list[str] -> [B, d]
        return torch.stack([table[t] for t in texts])

    emb = Embeddings(encoder=fake_encoder, min_score=None)
    emb.index([(f"doc-{i}", f"text {i}", None) for i in range(64)])
    hit = emb.search("text 17", 1)[0]
    assert hit[0] == "doc-17" and abs(hit[1] - 1.0) < 2e-3
    with pytest.raises(RuntimeError, match="no text encoder"):
        Embeddings().index([{"id": 1, "text": "x"}])


def test_hybrid_search_combines_dense_and_bm25(native_lib, tmp_path):
    """``Embeddings(hybrid=True, content=True)`` (heavy_ranker.py:78): the result is the convex combination of the dense
    score and the normalised BM25 score over 10 x limit candidates of each half; survives save/load."""
    from vietnamese_qa_system_amd import Embeddings
    from vietnamese_qa_system_amd.sparse import BM25Index, combine
    x, _ = _corpus(200, 64, 1)
    words = ["hà nội", "sài gòn", "huế", "đà nẵng", "cần thơ", "hải phòng", "sông hồng", "phở bò", "bánh mì", "cà phê"]
    texts = [f"{words[i % 10]} {words[(i * 7 + 3) % 10]} tài liệu {i}" for i in range(200)]
    queries = ["phở bò hà nội", "cà phê sài gòn", "tài liệu 42"]
    qv = {t: torch.from_numpy(x[17 * (j + 1)]) for j, t in enumerate(queries)}  # query j embeds exactly like document 17 (j + 1)
    table = {t: torch.from_numpy(x[i]) for i, t in enumerate(texts)}

    def fake_encoder(ts):
        return torch.stack([qv[t] if t in qv else table[t] for t in ts])

    docs = [{"id": i + 1, "text": t, "source": "wiki"} for i, t in enumerate(texts)]
    hyb = Embeddings(hybrid=True, content=True, encoder=fake_encoder)
    hyb.index(docs)
    dense_only = Embeddings(encoder=fake_encoder, min_score=None)
    dense_only.index(docs)
    bm25 = BM25Index().index(texts)
    limit = 3
    got = hyb.batchsearch(queries, limit)
    dense = dense_only.batchsearch(queries, 10 * limit)
    for j, query in enumerate(queries):
        d = [(i, s) for i, s in dense[j] if s > 0]
        sp = [(r + 1, s) for r, s in bm25.search(query, 10 * limit)]
        want = combine(d, sp, limit, 0.5, True)
        assert [h["id"] for h in got[j]] == [u for u, _ in want]
        assert np.allclose([h["score"] for h in got[j]], [s for _, s in want], atol=1e-6)
        assert got[j][0]["text"] == texts[got[j][0]["id"] - 1]
    # the keyword half changes the ranking: dense alone puts document 17 (j + 1) + 1 first with score ~1
    assert dense[0][0][0] == 18 and got[0][0]["id"] != 18 or got[0][0]["score"] < 0.99
    hyb.save(str(tmp_path / "hyb"))
    back = Embeddings(encoder=fake_encoder).load(str(tmp_path / "hyb"))
    assert back.hybrid and back.batchsearch(queries, limit) == got
    # vector queries carry no text: dense scores only
    assert len(hyb.batchsearch(x[:2], 2)[0]) == 2


def test_hybrid_search_against_hand_computed_expectation(native_lib):
    """Expectation worked out by hand in tests/test_sparse.py (HAND_*), independent of sparse.combine / BM25Index: documents
    embed as unit basis vectors, the query as 0.8 e0 + 0.6 e2, so the dense cosines are exactly 0.8 (doc 0) and 0.6 (doc 2)."""
    from test_sparse import HAND_DOCS, HAND_HYBRID  # tests/ is on sys.path (pytest rootdir-relative imports)
    from vietnamese_qa_system_amd import Embeddings
    basis = np.eye(8, dtype=np.float32)
    qvec = 0.8 * basis[0] + 0.6 * basis[2]

    def enc(texts):
        return torch.stack([torch.from_numpy(qvec if t == "đen mèo" else basis[HAND_DOCS.index(t)]) for t in texts])

    emb = Embeddings(hybrid=True, content=True, encoder=enc, dtype="fp32")  # fp32 rows: 0.8 and 0.6 are not fp16 numbers
    emb.index([{"id": i, "text": t} for i, t in enumerate(HAND_DOCS)])
    got = emb.search("đen mèo", 3)
    assert [h["id"] for h in got] == [u for u, _ in HAND_HYBRID]
    assert np.allclose([h["score"] for h in got], [s for _, s in HAND_HYBRID], atol=2e-5)
    assert [h["text"] for h in got] == [HAND_DOCS[u] for u, _ in HAND_HYBRID]


def test_index_streams_every_document_through_the_encoder_exactly_once(native_lib):
    """``Embeddings.index(documents)`` feeds the shard chunk by chunk from the encoder (row-producer path).  The encoder hook
    may be stateful (a tokenizer reading from a stream) and an encoder call costs a forward: every document is asked for
    once, in order, in chunks of ``max(8 * batch_size, 65 536)`` rows (the first chunk written fixes the centre of the shard's sketch:
    it must be a fair sample) -- no one-row probe to learn the dimension."""
    from vietnamese_qa_system_amd import Embeddings
    d, n = 48, 140_000
    rng = np.random.default_rng(3)
    table = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    calls = []

    def encoder(texts):
        calls.append(len(texts))
        rows = [int(t.split()[1]) for t in texts]
        return torch.from_numpy(table[rows]).cuda()

    docs = [{"id": i + 1, "text": f"doc {i}"} for i in range(n)]
    emb = Embeddings(encoder=encoder, content=True)
    emb.index(docs, batch_size=16)
    assert calls == [65536, 65536, 8928]
    calls.clear()
    hit = emb.search("doc 517", 1)[0]
    assert calls == [1] and hit["id"] == 518 and hit["text"] == "doc 517" and abs(hit["score"] - 1.0) < 2e-3


# ---- the latency path (vqa_index_search_host): the reference asks ONE question per call with limit = 1 (heavy_ranker.py:97-101)
@pytest.mark.parametrize("dtype,n,options", [("fp16", 1000, None), ("fp32", 5000, None), ("fp8", 3000, None),
                                             ("fp16", 200_000, {"stage_min_tiles": 2})])  # the last one: a sketch shard
def test_host_latency_path_returns_the_batch_paths_bits(native_lib, dtype, n, options):
    """Host-resident queries of 1 / 17 / 64 rows through vqa_index_search_host (pinned device-mapped staging, normalisation on the
    device, results written into pinned memory, polled completion) == the torch path on the same index, bit for bit -- scores,
    ids, positions -- and == the oracle."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    rng = np.random.default_rng(5)
    d = 768
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    q = rng.standard_normal((64, d)).astype(np.float32)  # NOT normalised: the call does it
    ix = DeviceIndex(x, id_base=1, dtype=dtype, device=0, options=options)
    if options:
        assert ix.launch_info(64, 10).sketch_scan == 1
    for b, k in ((1, 1), (1, 10), (17, 10), (64, 30)):
        s_h, i_h, p_h = ix.search_host(q[:b], k, normalize=True, return_positions=True)
        qn = torch.empty((b, d), dtype=torch.float32, device="cuda")
        N = ix._lib
        N.vqa_normalize_convert(torch.from_numpy(q[:b]).cuda().data_ptr(), b, d, 1, 0, qn.data_ptr(), torch.cuda.current_stream().cuda_stream)
        s_t, i_t, p_t = ix.search(qn, k, return_positions=True)
        torch.cuda.synchronize()
        assert np.array_equal(s_h, s_t.cpu().numpy()) and np.array_equal(i_h, i_t.cpu().numpy()) and np.array_equal(p_h, p_t.cpu().numpy())
        assert np.array_equal(i_h, p_h + 1)
    # fp16 host queries, no normalisation; growing the pinned buffer (B = 300 > the first allocation's share) keeps working
    q16 = R.l2_normalize(q).astype(np.float16)
    big = np.tile(q16, (5, 1))[:300]
    s_h, i_h = ix.search_host(big, 10)
    s_t, i_t, _ = ix.search(torch.from_numpy(big).cuda(), 10)
    torch.cuda.synchronize()
    assert np.array_equal(s_h, s_t.cpu().numpy()) and np.array_equal(i_h, i_t.cpu().numpy())
    if dtype != "fp8":
        stored = x.astype(np.float16) if dtype == "fp16" else x
        R.check_topk(s_h[:64], i_h[:64] - 1, R.full_scores(q16[:64].astype(np.float32), stored, R.DTYPE_F16 if dtype == "fp16" else R.DTYPE_F32), 10,
                     score_tol=1e-5, tie_tol=2e-6)
    with pytest.raises(ValueError):
        ix.search_host(q[:1].astype(np.float64), 1)
    with pytest.raises(ValueError):
        ix.search_host(q16[:1], 1, normalize=True)  # only fp32 queries are normalised by the call
    ix.close()


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("n,d,with_ids", [(1, 768, False), (63, 64, True), (257, 100, False), (5000, 768, True), (16384, 768, False),
                                          (16385, 256, True), (40000, 768, False), (131072, 128, False), (131073, 64, True), (262144, 96, False)])
def test_one_launch_search_returns_the_general_paths_bits(native_lib, n, d, with_ids, dtype):
    """K4 (csrc/tiny_search.hip): on an fp16 or fp32 shard of <= 262 144 rows, vqa_index_search_host with <= 16 questions, k <= 32 and questions x k <= 64 is ONE
    kernel -- normalise, score, select, merge.  Same scores / ids / positions, bit for bit, as the general launches on a handle with
    options.one_launch = 0, for raw fp32 questions (normalised by the call or not) and fp16 questions; rows stored twice tie and come
    back in position order; fewer rows than k: padding; 50 calls in a row (the ticket returns to zero) agree; and == the oracle."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    rng = np.random.default_rng(n + d)
    x = R.l2_normalize(rng.standard_normal((n, d)).astype(np.float32))
    if n >= 257:
        x[n // 2] = x[3]   # ties across workgroups ...
        x[5] = x[3]        # ... and inside one
    ids = (np.arange(n, dtype=np.int64) * 7 + 11) if with_ids else None
    q = rng.standard_normal((16, d)).astype(np.float32)
    q[0] = x[min(3, n - 1)] * 2.5
    kw = dict(dtype=dtype, device=0)
    one = DeviceIndex(x, ids=ids, id_base=1, **kw)
    gen = DeviceIndex(x, ids=ids, id_base=1, options={"one_launch": 0}, **kw)
    for b, k in ((1, 1), (1, 3), (1, 16), (4, 16), (16, 4), (5, 12), (8, 8), (2, 5), (1, 17), (1, 32), (2, 32), (3, 20), (16, 16)):  # (the last one: past questions x k <= 64, the general launches)
        for norm in (True, False):
            a = one.search_host(q[:b], k, normalize=norm, return_positions=True)
            g = gen.search_host(q[:b], k, normalize=norm, return_positions=True)
            for u, v in zip(a, g):
                assert np.array_equal(u, v), (b, k, norm)
            if n < k:
                assert (a[1][:, n:] == -1).all() and np.isneginf(a[0][:, n:]).all() and (a[2][:, n:] == -1).all()
            if n >= 257 and k >= 3 and norm:
                assert list(a[2][0, :3]) == [3, 5, n // 2]
            if with_ids:
                live = a[2] >= 0
                assert np.array_equal(a[1][live], a[2][live] * 7 + 11)
    q16 = R.l2_normalize(q).astype(np.float16)
    a, g = one.search_host(q16[:9], 10, return_positions=True), gen.search_host(q16[:9], 10, return_positions=True)
    for u, v in zip(a, g):
        assert np.array_equal(u, v)
    if n >= 10:
        stored, code = (x.astype(np.float16), R.DTYPE_F16) if dtype == "fp16" else (x, R.DTYPE_F32)
        R.check_topk(a[0], a[2], R.full_scores(q16[:9].astype(np.float32), stored, code), 10, score_tol=1e-5, tie_tol=2e-6)
    # device-resident questions (the encoder's output): fp32 normalised by the call, and an fp16 slice whose address is no multiple of 16
    qd, qd16 = torch.from_numpy(q).cuda(), torch.from_numpy(q16).cuda()
    for dev_q, host_q, norm in ((qd[:5], q[:5], True), (qd16[1:4], q16[1:4], False), (qd[2:3], q[2:3], False)):
        a, g = one.search_host(dev_q, 7, normalize=norm, return_positions=True), gen.search_host(host_q, 7, normalize=norm, return_positions=True)
        for u, v in zip(a, g):
            assert np.array_equal(u, v)
    first = one.search_host(q[:3], 4, normalize=True, return_positions=True)
    for _ in range(50):
        again = one.search_host(q[:3], 4, normalize=True, return_positions=True)
        for u, v in zip(first, again):
            assert np.array_equal(u, v)
    # past its limits the same call takes the general launches
    s17 = one.search_host(np.tile(q, (2, 1))[:17], 3, normalize=True)
    g17 = gen.search_host(np.tile(q, (2, 1))[:17], 3, normalize=True)
    assert np.array_equal(s17[0], g17[0]) and np.array_equal(s17[1], g17[1])
    one.close()
    gen.close()


def test_embeddings_search_of_one_vector_takes_the_latency_path(native_lib):
    """Embeddings.search(vector, 1) -- the reference's call shape -- goes through vqa_index_search_host and returns what batchsearch
    of a device tensor returns; lists of floats and float64 arrays too."""
    from vietnamese_qa_system_amd import Embeddings
    rng = np.random.default_rng(9)
    x = rng.standard_normal((5000, 768)).astype(np.float32)
    q = rng.standard_normal((8, 768)).astype(np.float32)
    emb = Embeddings(dtype="fp16", device=0)
    emb.index_vectors(list(range(1, 5001)), x)
    via_torch = emb.batchsearch(torch.from_numpy(q), 3)
    assert [emb.search(q[i], 3) for i in range(8)] == via_torch
    assert emb.search(q[2].tolist(), 3) == via_torch[2] and emb.search(q[2].astype(np.float64), 3) == via_torch[2]
    assert emb.batchsearch(q, 3) == via_torch
    one = emb.search(q[0], 1)
    assert len(one) == 1 and one[0] == via_torch[0][0]


def test_text_questions_take_two_library_calls_and_return_the_general_paths_results(native_lib):
    """`Embeddings.search(question, 1)` (heavy_ranker.py:98) with the encoder built here: tokenizer -> vqa_encoder_forward_host (host token ids
    in, device vectors out) -> vqa_index_search_host (device vectors in, host results out).  Same results as encoding through the torch
    path and searching the vectors; a question longer than the fast path's workspace, and more questions than it takes, fall back."""
    from oracle import encoder as E
    from vietnamese_qa_system_amd import Embeddings
    from vietnamese_qa_system_amd.encoder import QuestionEncoder, TextEncoder
    cfg = dict(E.PHOBERT_BASE, layers=2, vocab_size=5000)
    w = E.synthetic_weights(cfg, seed=8, layers=2)
    ids, mask = E.synthetic_tokens(cfg, 80, 24, seed=3)
    table = {f"q{i}": (ids[i], mask[i]) for i in range(80)}
    calls = []

    def tokenizer(texts):
        calls.append(len(texts))
        return np.stack([table[t][0] for t in texts]), np.stack([table[t][1] for t in texts])

    enc = QuestionEncoder(w, cfg, max_tokens=80 * 24)
    te = TextEncoder(tokenizer, enc, pooling="mean")
    rng = np.random.default_rng(1)
    docs = rng.standard_normal((4000, 768)).astype(np.float32)
    emb = Embeddings(encoder=te, min_score=None)
    emb.index_vectors(list(range(1, 4001)), docs)
    texts = [f"q{i}" for i in range(80)]

    def general(sub):  # the torch path on the SAME call shape (the encoder picks its kernels by the number of positions): vectors, then search
        return emb.batchsearch(te(sub), 3)

    for i in (0, 1, 17):
        assert emb.search(texts[i], 3) == general([texts[i]])[0]   # one question: the fast path, the latency form of the encoder
    assert emb.batchsearch(texts[:9], 3) == general(texts[:9])     # nine questions: small-batch kernels
    want = general(texts)
    assert emb.batchsearch(texts, 3) == want                       # 80 > 64 questions: the general path itself
    # the kernels differ between call shapes by fp16 rounding only: same documents, scores within 1e-3
    nine = emb.batchsearch(texts[:9], 3)
    assert [[d for d, _ in r] for r in nine] == [[d for d, _ in r] for r in want[:9]]
    assert max(abs(a[1] - b[1]) for ra, rb in zip(nine, want[:9]) for a, b in zip(ra, rb)) < 1e-3
    one = emb.search(texts[5], 1)
    assert len(one) == 1 and one[0] == general([texts[5]])[0][0]
    # device-resident vectors through the host-result entry: what the fast path's second call is
    vecs = te(texts[:7])
    s_h, i_h = emb._index.search_host(vecs.contiguous(), 3, normalize=True)
    assert [[(int(i), float(s)) for i, s in zip(ir, sr)] for ir, sr in zip(i_h, s_h)] == [list(r) for r in general(texts[:7])]
    with pytest.raises(ValueError, match="token id"):
        bad = ids[:1].copy()
        bad[0, 1] = 5000
        enc.forward_host(bad, mask[:1], torch.empty((1, 768), dtype=torch.float32, device="cuda"))
    enc.close()


def test_rank_query_runs_the_two_retrievers_side_by_side_and_returns_what_two_searches_return(native_lib):
    """heavy_ranker.py:98-101 asks its two retrievers the same question one after the other.  `heavy_ranker.rank_query` enqueues the
    two encoder forwards on a stream each (`Embeddings.search_begin`), then completes the two searches (`search_end`): the results are
    those of two `search(question, limit)` calls -- two different models (hidden 768 / 384), content on and off; a retriever whose
    encoder is a plain callable takes the ordinary path inside `search_end`."""
    from oracle import encoder as E
    from vietnamese_qa_system_amd import Embeddings, heavy_ranker
    from vietnamese_qa_system_amd.encoder import QuestionEncoder, TextEncoder
    rng = np.random.default_rng(2)
    embs, encs = [], []
    for j, base in enumerate((E.PHOBERT_BASE, E.MINILM_L12)):
        cfg = dict(base, layers=2, vocab_size=5000)
        w = E.synthetic_weights(cfg, seed=30 + j, layers=2)
        ids, mask = E.synthetic_tokens(cfg, 12, 20, seed=5)
        table = {f"q{i}": (ids[i], mask[i]) for i in range(12)}
        enc = QuestionEncoder(w, cfg, max_tokens=64)
        te = TextEncoder(lambda texts, table=table: (np.stack([table[t][0] for t in texts]), np.stack([table[t][1] for t in texts])), enc, pooling="mean")
        emb = Embeddings(encoder=te, min_score=None, content=bool(j), hybrid=bool(j))  # (the second one as heavy_ranker.py:78 builds it)
        docs = rng.standard_normal((3000, cfg["hidden"])).astype(np.float32)
        if j:
            emb.index([{"id": i + 1, "text": f"q{i % 12}"} for i in range(24)])  # (content=True: through the text route; a small corpus the stand-in tokenizer knows)
        else:
            emb.index_vectors(list(range(1, 3001)), docs)
        embs.append(emb)
        encs.append(enc)
    a, b = embs
    for i in range(12):
        for limit in (1, 3):
            want = (a.search(f"q{i}", limit), b.search(f"q{i}", limit))
            assert heavy_ranker.rank_query(a, b, f"q{i}", limit) == want
    # hybrid=True with the encoder built here: the dense half takes the two library calls as well; same results as the torch route
    for i in (0, 4, 9):
        for limit in (1, 3):
            assert b.search(f"q{i}", limit) == b._hybrid(b._query_vectors([f"q{i}"]), [f"q{i}"], limit)[0]
    tok = a.search_begin("q3")
    assert tok[1] is not None and a.search_end(tok, 2) == a.search("q3", 2)
    plain = Embeddings(encoder=lambda texts: torch.ones((len(texts), 768), device="cuda"), min_score=None)
    plain.index_vectors(list(range(1, 101)), rng.standard_normal((100, 768)).astype(np.float32))
    tok = plain.search_begin("anything")
    assert tok[1] is None and plain.search_end(tok, 1) == plain.search("anything", 1)
    assert heavy_ranker.rank_query(a, plain, "q1", 1) == (a.search("q1", 1), plain.search("q1", 1))
    with pytest.raises(ValueError):
        a.search_begin(np.zeros(768, np.float32))
    for enc in encs:
        enc.close()
