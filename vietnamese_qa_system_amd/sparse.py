"""BM25 keyword index: the sparse half of ``Embeddings(hybrid=True)`` (reference: the commented index build at
``inference_pipeline/db_utils/heavy_ranker.py:78``, ``Embeddings(hybrid=True, content=True, path=...)``).

In the reference this lives inside txtai (unpinned, ``requirements.txt:74``), whose source is not available here; what
follows restates txtai's BM25 scoring FROM MEMORY and is therefore *parity unpinned* (SURVEY.md section 8f-4):

* tokens: lower-cased ``\\w+`` runs longer than one character (txtai's default tokenizer also drops an English stop list,
  which does nothing for Vietnamese text and is not reproduced);
* idf(t) = ln(1 + (N - df + 0.5) / (df + 0.5));  score(tf) = idf * tf * (k1 + 1) / (tf + k1 * (1 - b + b * dl / avgdl)),
  k1 = 1.2, b = 0.75;
* ``normalize=True`` (what ``hybrid=True`` configures): scores are divided by
  ``min(best + avgscore, 6 * avgscore)`` and clipped to 1, where ``avgscore`` is the score of an average term (frequency
  = total tokens / vocabulary size, idf = mean idf) in an average document -- so that they can be mixed with cosine scores by
  a convex combination (weights 0.5 / 0.5).

Host-side numpy: the sparse half is a few postings lists per query and is not on the MI355X hot path (the dense scan is).
"""
from __future__ import annotations

import json
import math
import os
import re
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_TOKEN = re.compile(r"\w+", re.UNICODE)

SPARSE_FILE = "sparse.npz"
VOCAB_FILE = "sparse_vocab.json"


def tokenize(text: str) -> List[str]:
    return [t for t in _TOKEN.findall(str(text).lower()) if len(t) > 1]


class BM25Index:
    """Postings in CSR form: term t owns ``docs[indptr[t]:indptr[t + 1]]`` (row positions, ascending) and their ``tf``."""

    def __init__(self, k1: float = 1.2, b: float = 0.75, normalize: bool = True,
                 tokenizer: Callable[[str], List[str]] = tokenize):
        self.k1, self.b, self.normalize, self.tokenizer = float(k1), float(b), bool(normalize), tokenizer
        self.vocab: Dict[str, int] = {}
        self.indptr = np.zeros(1, np.int64)
        self.docs = np.zeros(0, np.int64)
        self.tf = np.zeros(0, np.float32)
        self.doclen = np.zeros(0, np.float32)
        self.idf = np.zeros(0, np.float32)
        self.avgdl = 0.0
        self.avgscore = 0.0

    # ---- build ------------------------------------------------------------------------------------------------------
    def index(self, texts: Iterable[str]) -> "BM25Index":
        vocab: Dict[str, int] = {}
        rows: List[int] = []
        terms: List[int] = []
        counts: List[int] = []
        doclen: List[int] = []
        for pos, text in enumerate(texts):
            toks = self.tokenizer(text)
            doclen.append(len(toks))
            local: Dict[int, int] = {}
            for t in toks:
                tid = vocab.setdefault(t, len(vocab))
                local[tid] = local.get(tid, 0) + 1
            for tid, c in local.items():
                rows.append(pos)
                terms.append(tid)
                counts.append(c)
        n, v = len(doclen), len(vocab)
        self.vocab = vocab
        self.doclen = np.asarray(doclen, np.float32)
        terms_a = np.asarray(terms, np.int64)
        order = np.lexsort((np.asarray(rows, np.int64), terms_a))  # by term, then row position
        self.docs = np.asarray(rows, np.int64)[order]
        self.tf = np.asarray(counts, np.float32)[order]
        df = np.bincount(terms_a, minlength=v).astype(np.int64)
        self.indptr = np.concatenate([[0], np.cumsum(df)]).astype(np.int64)
        self.idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5)).astype(np.float32)
        self.avgdl = float(self.doclen.mean()) if n else 0.0
        # the score of an average term in an average document; txtai: avgfreq = total tokens / vocabulary size (the mean
        # CORPUS frequency of a term), avgidf = mean idf over the vocabulary [recalled]
        avgfreq = float(self.doclen.sum()) / v if v else 0.0
        avgidf = float(self.idf.mean()) if v else 0.0
        self.avgscore = self._score(avgfreq, avgidf, self.avgdl) if n else 0.0
        return self

    def _score(self, tf, idf, dl):
        norm = self.k1 * (1.0 - self.b + self.b * (dl / self.avgdl if self.avgdl > 0 else 1.0))
        return idf * tf * (self.k1 + 1.0) / (tf + norm)

    def __len__(self) -> int:
        return int(self.doclen.shape[0])

    # ---- search -----------------------------------------------------------------------------------------------------
    def search(self, query: str, limit: int) -> List[Tuple[int, float]]:
        """``[(row position, score)]``, best first (ties by row position), at most ``limit`` rows with a score > 0."""
        n = len(self)
        if n == 0 or limit <= 0:
            return []
        acc = np.zeros(n, np.float32)
        for t in self.tokenizer(query):  # a repeated query term counts each time it occurs, as in txtai
            tid = self.vocab.get(t)
            if tid is None:
                continue
            lo, hi = self.indptr[tid], self.indptr[tid + 1]
            rows = self.docs[lo:hi]
            acc[rows] += self._score(self.tf[lo:hi], self.idf[tid], self.doclen[rows]).astype(np.float32)
        hit = np.flatnonzero(acc > 0)
        if hit.size == 0:
            return []
        order = hit[np.lexsort((hit, -acc[hit].astype(np.float64)))][:limit]
        scores = acc[order].astype(np.float64)
        if self.normalize and self.avgscore > 0:
            maxscore = min(float(scores[0]) + self.avgscore, 6.0 * self.avgscore)
            scores = np.minimum(scores / maxscore, 1.0)
        return [(int(r), float(s)) for r, s in zip(order, scores)]

    def batchsearch(self, queries: Sequence[str], limit: int) -> List[List[Tuple[int, float]]]:
        return [self.search(q, limit) for q in queries]

    # ---- persistence ------------------------------------------------------------------------------------------------
    def save(self, path: str) -> None:
        np.savez(os.path.join(path, SPARSE_FILE), indptr=self.indptr, docs=self.docs, tf=self.tf, doclen=self.doclen, idf=self.idf,
                 params=np.asarray([self.k1, self.b, float(self.normalize), self.avgdl, self.avgscore], np.float64))
        with open(os.path.join(path, VOCAB_FILE), "w", encoding="utf-8") as f:
            json.dump(sorted(self.vocab, key=self.vocab.get), f, ensure_ascii=False)

    @classmethod
    def load(cls, path: str) -> Optional["BM25Index"]:
        fn = os.path.join(path, SPARSE_FILE)
        if not os.path.isfile(fn):
            return None
        z = np.load(fn)
        k1, b, norm, avgdl, avgscore = z["params"].tolist()
        ix = cls(k1, b, bool(norm))
        ix.indptr, ix.docs, ix.tf, ix.doclen, ix.idf = z["indptr"], z["docs"], z["tf"], z["doclen"], z["idf"]
        ix.avgdl, ix.avgscore = float(avgdl), float(avgscore)
        with open(os.path.join(path, VOCAB_FILE), encoding="utf-8") as f:
            ix.vocab = {t: i for i, t in enumerate(json.load(f))}
        return ix


def combine(dense: List[Tuple[object, float]], sparse: List[Tuple[object, float]], limit: int, weight: float = 0.5,
            normalized: bool = True) -> List[Tuple[object, float]]:
    """txtai's hybrid merge [recalled]: per id, ``weight * dense + (1 - weight) * sparse`` when the sparse scores are
    normalised, reciprocal-rank fusion ``sum(w / (rank + 1))`` otherwise; best first, ties by first appearance."""
    merged: Dict[object, float] = {}
    for w, results in ((weight, dense), (1.0 - weight, sparse)):
        if w <= 0:
            continue
        for rank, (uid, score) in enumerate(results):
            merged[uid] = merged.get(uid, 0.0) + (score * w if normalized else w / (rank + 1))
    return sorted(merged.items(), key=lambda kv: kv[1], reverse=True)[:limit]
