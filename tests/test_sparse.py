"""BM25 half of ``Embeddings(hybrid=True)`` (host side, no GPU): scoring against a plain restatement of the formula,
normalisation, the hybrid merge rule, persistence.  The reference delegates this to txtai (unpinned, not available here):
the formulas are recalled, see ``vietnamese_qa_system_amd/sparse.py``."""
import math

import numpy as np

from vietnamese_qa_system_amd.sparse import BM25Index, combine, tokenize

DOCS = [
    "Hà Nội là thủ đô của Việt Nam",
    "Thành phố Hồ Chí Minh là thành phố lớn nhất Việt Nam",
    "Paris is the capital of France",
    "Sông Hồng chảy qua Hà Nội",
    "the the the capital capital",
    "",
]


def _plain_bm25(docs, query, k1=1.2, b=0.75):
    toks = [tokenize(d) for d in docs]
    n = len(docs)
    avgdl = sum(len(t) for t in toks) / n
    out = np.zeros(n)
    for term in tokenize(query):
        df = sum(term in t for t in toks)
        if df == 0:
            continue
        idf = math.log(1 + (n - df + 0.5) / (df + 0.5))
        for i, t in enumerate(toks):
            tf = t.count(term)
            if tf:
                out[i] += idf * tf * (k1 + 1) / (tf + k1 * (1 - b + b * len(t) / avgdl))
    return out


def test_tokenize_keeps_vietnamese_words_and_drops_single_letters():
    assert tokenize("Hà Nội, là thủ-đô a b!") == ["hà", "nội", "là", "thủ", "đô"]


def test_raw_scores_match_the_formula():
    ix = BM25Index(normalize=False).index(DOCS)
    for q in ("thủ đô Hà Nội", "capital", "thành phố thành phố", "không có từ nào"):
        ref = _plain_bm25(DOCS, q)
        got = ix.search(q, 10)
        want = [i for i in np.lexsort((np.arange(len(DOCS)), -ref)) if ref[i] > 0]
        assert [r for r, _ in got] == want
        assert np.allclose([s for _, s in got], ref[want], rtol=1e-5)


def test_normalised_scores_are_in_unit_range_and_keep_the_order():
    raw = BM25Index(normalize=False).index(DOCS)
    nrm = BM25Index(normalize=True).index(DOCS)
    a, b = raw.search("thủ đô Hà Nội capital", 5), nrm.search("thủ đô Hà Nội capital", 5)
    assert [r for r, _ in a] == [r for r, _ in b]
    assert all(0 < s <= 1 for _, s in b)
    maxscore = min(a[0][1] + raw.avgscore, 6 * raw.avgscore)
    assert np.allclose([s for _, s in b], [min(s / maxscore, 1.0) for _, s in a], rtol=1e-5)


def test_limit_and_empty_cases():
    ix = BM25Index().index(DOCS)
    assert len(ix.search("việt nam", 1)) == 1
    assert ix.search("", 3) == [] and ix.search("zzz", 3) == [] and ix.search("việt", 0) == []
    assert BM25Index().index([]).search("việt", 3) == []


def test_combine_is_a_convex_combination_or_rrf():
    dense = [("a", 0.9), ("b", 0.5), ("c", 0.4)]
    sparse = [("c", 1.0), ("a", 0.2), ("d", 0.1)]
    got = combine(dense, sparse, 3, 0.5, normalized=True)
    assert [u for u, _ in got] == ["c", "a", "b"]
    assert np.allclose([s for _, s in got], [0.7, 0.55, 0.25])
    rrf = combine(dense, sparse, 4, 0.5, normalized=False)
    assert dict(rrf)["a"] == 0.5 / 1 + 0.5 / 2 and dict(rrf)["c"] == 0.5 / 3 + 0.5 / 1
    assert [u for u, _ in combine(dense, sparse, 2, 1.0)] == ["a", "b"]  # weight 1: the dense half alone


def test_save_load_roundtrip(tmp_path):
    ix = BM25Index().index(DOCS)
    ix.save(str(tmp_path))
    back = BM25Index.load(str(tmp_path))
    for q in ("thủ đô", "capital of France", "thành phố"):
        assert back.search(q, 4) == ix.search(q, 4)
    assert BM25Index.load(str(tmp_path / "missing")) is None


# ---- expectations worked out by hand (not through sparse.py): 5 documents, 13 tokens, 6 terms, avgdl = 2.6 -----------------
HAND_DOCS = ["mèo đen", "mèo trắng mèo", "chó đen", "chó trắng chó chó", "cá vàng"]
# idf(df = 2) = ln(1 + 3.5 / 2.5) = ln 2.4 = 0.8754687;  idf(df = 1) = ln(1 + 4.5 / 1.5) = ln 4 = 1.3862944
# query "mèo đen":
#   doc 0 (len 2): each term tf = 1: 0.8754687 * 2.2 / (1 + 1.2 * (0.25 + 0.75 * 2 / 2.6)) = 0.9667338, two terms: 1.9334676
#   doc 1 (len 3): mèo tf = 2:      0.8754687 * 2 * 2.2 / (2 + 1.2 * (0.25 + 0.75 * 3 / 2.6)) = 1.1538436
#   doc 2 (len 2): đen tf = 1:      0.9667338
HAND_RAW = [(0, 1.9334676), (1, 1.1538436), (2, 0.9667338)]
# normalisation: avgfreq = 13 tokens / 6 terms = 2.1666667, avgidf = (4 * 0.8754687 + 2 * 1.3862944) / 6 = 1.0457439,
#   avgscore = 1.0457439 * 2.1666667 * 2.2 / (2.1666667 + 1.2) = 1.4806078, maxscore = min(1.9334676 + 1.4806078, 6 * 1.4806078) = 3.4140754
HAND_NORM = [(0, 0.5663225), (1, 0.3379666), (2, 0.2831612)]
# hybrid, weights 0.5 / 0.5, dense cosine 0.8 for doc 0 and 0.6 for doc 2 (nothing else positive):
#   doc 0: 0.4 + 0.2831612 = 0.6831612;  doc 2: 0.3 + 0.1415806 = 0.4415806;  doc 1: 0.1689833
HAND_HYBRID = [(0, 0.6831612), (2, 0.4415806), (1, 0.1689833)]


def test_bm25_against_hand_computed_scores():
    raw = BM25Index(normalize=False).index(HAND_DOCS).search("mèo đen", 5)
    assert [r for r, _ in raw] == [r for r, _ in HAND_RAW]
    assert np.allclose([s for _, s in raw], [s for _, s in HAND_RAW], atol=2e-6)
    nrm = BM25Index(normalize=True).index(HAND_DOCS)
    assert abs(nrm.avgscore - 1.4806078) < 2e-6 and abs(nrm.avgdl - 2.6) < 1e-6
    got = nrm.search("mèo đen", 5)
    assert [r for r, _ in got] == [r for r, _ in HAND_NORM]
    assert np.allclose([s for _, s in got], [s for _, s in HAND_NORM], atol=2e-6)


def test_hybrid_merge_against_hand_computed_scores():
    sparse = BM25Index(normalize=True).index(HAND_DOCS).search("mèo đen", 30)
    got = combine([(0, 0.8), (2, 0.6)], sparse, 3, 0.5, True)
    assert [u for u, _ in got] == [u for u, _ in HAND_HYBRID]
    assert np.allclose([s for _, s in got], [s for _, s in HAND_HYBRID], atol=2e-6)
