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
