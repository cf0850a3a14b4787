"""Counterpart of the reference's retrieval driver ``inference_pipeline/db_utils/heavy_ranker.py``.

The reference is a module-level script: load documents from sqlite (:70-76), load two saved indexes (:91-94), then
for each sample query search both with ``limit=1`` (:98-101), join the document text by id (:102-109) and report a
"match" when both models return the same id and ``score_a + score_b > 0.4`` (:110-115).  Here the same steps are
functions; the per-query Python loop becomes one ``batchsearch`` per model (the GPU path scores the whole batch in
one pass over the index) and the per-hit ``SELECT`` becomes one batched join.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from . import docstore

AGREEMENT_THRESHOLD = 0.4  # heavy_ranker.py:110


def agreement(uid_a, score_a: float, uid_b, score_b: float, threshold: float = AGREEMENT_THRESHOLD) -> bool:
    """``heavy_ranker.py:110``: ``uid_paraphrase == uid_qa and score_paraphrase + score_qa > 0.4``."""
    return uid_a == uid_b and (score_a + score_b) > threshold


def _top1(result: list):
    if not result:
        return None, 0.0
    hit = result[0]
    if isinstance(hit, dict):
        return hit["id"], hit["score"]
    return hit[0], hit[1]


_streams = {}


def rank_query(embeddings_a, embeddings_b, question: str, limit: int = 1):
    """One pass of the reference's loop body (:98-101) -- the same question through both retrievers -- with the two encoder forwards
    side by side: each model's launches go to a stream of its own (one question occupies a fraction of the device), the two searches
    follow on those streams.  Returns ``(result_a, result_b)`` exactly as ``embeddings_x.search(question, limit)`` would."""
    import torch
    streams = []
    for slot, emb in enumerate((embeddings_a, embeddings_b)):
        key = (emb.device, slot)
        if key not in _streams:
            # a high-priority and a normal stream: two streams of ONE priority share a hardware queue under the runtime's defaults and
            # their launches do not overlap (scripts/probes/two_streams_probe.py: 0.82 ms for the two forwards, 0.59 this way)
            _streams[key] = torch.cuda.Stream(device=emb.device, priority=-1 if slot == 0 else 0)
        streams.append(_streams[key])
    tokens = []
    for emb, st in zip((embeddings_a, embeddings_b), streams):
        with torch.cuda.stream(st):
            tokens.append(emb.search_begin(question))
    out = []
    for emb, st, tok in zip((embeddings_a, embeddings_b), streams, tokens):
        with torch.cuda.stream(st):
            out.append(emb.search_end(tok, limit))
    return tuple(out)


def rank_queries(embeddings_a, embeddings_b, queries: Sequence, database_path: Optional[str] = None) -> List[dict]:
    """Both retrievers' best hit per query, the joined document text and the agreement flag (:97-115).

    ``queries`` is a list of strings (needs encoders on the Embeddings objects) or a pair of [B, d] arrays
    ``(queries_for_a, queries_for_b)`` of precomputed query embeddings."""
    if isinstance(queries, tuple):
        qa, qb = queries
    else:
        qa = qb = list(queries)
    ra = embeddings_a.batchsearch(qa, 1)
    rb = embeddings_b.batchsearch(qb, 1)
    hits = [(_top1(a), _top1(b)) for a, b in zip(ra, rb)]
    docs = {}
    if database_path is not None:
        ids = [u for (ua, _), (ub, _) in hits for u in (ua, ub) if isinstance(u, int)]
        docs = docstore.fetch_docs(database_path, ids)
    out = []
    for i, ((ua, sa), (ub, sb)) in enumerate(hits):
        out.append({"query": qa[i] if isinstance(qa, list) else i, "id_a": ua, "score_a": sa, "doc_a": docs.get(ua),
                    "id_b": ub, "score_b": sb, "doc_b": docs.get(ub), "match": agreement(ua, sa, ub, sb),
                    "match_score": sa + sb})
    return out
