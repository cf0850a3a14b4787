"""txtai-shaped ``Embeddings`` object: the drop-in boundary of the retrieval hot path.

The reference drives its retriever only through this duck type (``inference_pipeline/db_utils/heavy_ranker.py``):

    Embeddings(hybrid=True, content=True, path=<hf model>)      # :78-83 (commented build step)
    .index(list[{"id", "text", "source"}])                      # :76, :86
    .save(dir) / Embeddings().load(dir)                         # :87 / :91-94
    .search(query_str, limit) -> [{"id", "score", "text"}]      # :98-101 (reads ['id'], ['score'])

Same names, argument meaning and result shapes here; underneath, scoring + top-k run in the HIP library on the
MI355X (no CPU fallback), the corpus lives in HBM, and with ``torch.distributed`` initialised the rows are sharded
across the ranks with an RCCL all-gather of per-shard candidates (:mod:`.sharded`).

Conventions kept from txtai [recalled, SURVEY.md section 8b]: results sorted by score descending; default ``limit=3``;
``content=False`` -> ``[(id, score)]``, ``content=True`` -> ``[{"id", "text", "score"}]``; results with
``score <= 0`` are dropped -- a HOST-side filter (``min_score=0.0``, exclusive; ``None`` disables it), never part
of the kernel.  ``hybrid=True`` (``heavy_ranker.py:78``) adds a host-side BM25 keyword half (:mod:`.sparse`, txtai's
scoring restated from memory: parity unpinned) and returns the 0.5 / 0.5 convex combination of the two scores for text
queries; vector queries are dense only.
"""
from __future__ import annotations

import json
import os
import shutil
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from . import docstore
from .index import _STORE_NP, DeviceIndex, resolve_dtype
from .sharded import ShardedSearcher, shard_bounds
from .sparse import BM25Index, combine

META_FILE, IDS_FILE, DOCS_FILE = "meta.json", "ids.i64", "documents.db"
VECTOR_FILES = {"float16": "vectors.f16", "float32": "vectors.f32"}


def _e4m3_decode_table() -> np.ndarray:
    """OCP e4m3fn code -> value (the 256 code points); used to export an fp8 index as exactly representable fp16."""
    c = np.arange(256, dtype=np.int64)
    e, m = (c >> 3) & 0xF, c & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    v = np.where(c & 0x80, -v, v)
    v[(c & 0x7F) == 0x7F] = np.nan
    return v


FORMAT_VERSION = 1
HOST_PATH_MAX_QUERIES = 64  # host-resident query batches up to this size take vqa_index_search_host (larger ones: the torch path, asynchronous staging)

Query = Union[str, np.ndarray, torch.Tensor, Sequence[float]]


class Embeddings:
    def __init__(self, config: Optional[dict] = None, *, path: Optional[str] = None, content: bool = False,
                 hybrid: bool = False, dtype: str = "fp16", device: Optional[int] = None, pooling: str = "mean",
                 normalize: bool = True, encoder: Optional[Callable[[List[str]], torch.Tensor]] = None,
                 min_score: Optional[float] = 0.0, group=None, tokenizer=None, max_tokens: int = 1024 * 32):
        """``path`` (``heavy_ranker.py:78-83``): a LOCAL model directory (Hugging Face / sentence-transformers layout) is loaded
        into the HIP question encoder (:meth:`QuestionEncoder.from_pretrained`) when the first text is encoded -- pooling and
        normalisation as its ``modules.json`` says, else the ``pooling`` / ``normalize`` arguments.  ``tokenizer``:
        ``texts -> (input_ids, attention_mask)``; default: ``transformers.AutoTokenizer`` of the same directory (host side).  A
        path that is not a directory (a hub name) is only recorded: there is no network to resolve it."""
        cfg = dict(config or {})
        self.path = cfg.get("path", path)
        self.tokenizer = tokenizer
        self.max_tokens = int(max_tokens)
        self.content = bool(cfg.get("content", content))
        self.hybrid = bool(cfg.get("hybrid", hybrid))
        self.dtype = cfg.get("dtype", dtype)
        resolve_dtype(self.dtype)
        self.pooling = cfg.get("pooling", pooling)
        if self.pooling not in ("cls", "mean"):
            raise ValueError(f"pooling must be 'cls' or 'mean', got {self.pooling!r}")
        self.normalize = bool(cfg.get("normalize", normalize))
        self.min_score = cfg.get("min_score", min_score)
        self.encoder = encoder
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = int(device) if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
        # hybrid=True (heavy_ranker.py:78): dense + BM25 keyword scores, convex combination with `weights` (txtai default 0.5)
        self.weights = float(cfg.get("weights", 0.5))
        self._sparse: Optional[BM25Index] = None
        self._row_ids: Optional[np.ndarray] = None  # device-level id of every global row position (sparse hits -> ids)
        self._index: Optional[DeviceIndex] = None
        self._qbuf: Optional[torch.Tensor] = None  # query vectors of the text fast path (encoder output -> search input, on the device)
        self._searcher: Optional[ShardedSearcher] = None
        self._host_ids: Optional[list] = None  # non-integer external ids, by row position
        self._docs_db: Optional[str] = None
        self._docs_mem: Optional[dict] = None   # device-level id -> document (lookup at search time)
        self._docs_rows: Optional[list] = None  # [(device-level id, document)] of the rows this rank was given
        self._lo = 0                            # first global row of this rank's shard
        self.load_stats: Optional[dict] = None  # filled by load(): bytes, seconds, GB/s of the file -> HBM stream
        self.n = 0
        self.d = 0

    # ---- building -------------------------------------------------------------------------------------------------
    @staticmethod
    def _normalise_documents(documents: Iterable) -> List[dict]:
        """txtai accepts dicts, (id, data, tags) tuples and bare strings; the reference passes dicts
        ``{"id", "text", "source"}`` (``heavy_ranker.py:76``)."""
        out = []
        for pos, doc in enumerate(documents):
            if isinstance(doc, dict):
                d = dict(doc)
                d.setdefault("id", pos)
            elif isinstance(doc, (tuple, list)):
                d = {"id": doc[0], "text": doc[1]}
                if len(doc) > 2 and doc[2] is not None:
                    d["source"] = doc[2]
            else:
                d = {"id": pos, "text": doc}
            out.append(d)
        return out

    def _model_dir(self) -> Optional[str]:
        """``path=`` as a local directory: the directory itself, or the hub name's snapshot in the local Hugging Face /
        sentence-transformers cache (``pretrained.resolve_model_path``: the reference passes hub names, ``heavy_ranker.py:80,83``)."""
        from .pretrained import resolve_model_path
        return resolve_model_path(self.path) if self.path else None

    def _encoder_from_path(self):
        """``path=<local model directory>`` -> ``TextEncoder`` over the HIP encoder (built once, on first use)."""
        from .encoder import QuestionEncoder, TextEncoder
        model_dir = self._model_dir()
        enc = QuestionEncoder.from_pretrained(model_dir, device=self.device, max_tokens=self.max_tokens)
        self._auto_encoder_path = self.path  # (load() of an index built with another model drops this encoder)
        tok = self.tokenizer
        if tok is None:
            try:
                from transformers import AutoTokenizer  # host-side tokenisation only; the forward is ours
            except ImportError as e:  # pragma: no cover
                raise RuntimeError("no tokenizer= was given and transformers is not importable for AutoTokenizer") from e
            hf_tok = AutoTokenizer.from_pretrained(model_dir)
            cap = max(8, min(128, int(enc.config["max_pos"]) - 2 - int(enc.config["pad_id"])))

            def tok(texts):
                e = hf_tok(list(texts), padding=True, truncation=True, max_length=cap, return_tensors="np")
                return e["input_ids"], e["attention_mask"]
        if enc.pooling is not None:
            self.pooling = enc.pooling
        # (txtai L2-normalises every embedding whatever the model's own modules say: cosine = inner product of unit rows)
        return TextEncoder(tok, enc, pooling=self.pooling, normalize=self.normalize or bool(enc.normalize))

    def _encode(self, texts: List[str]) -> torch.Tensor:
        if self.encoder is None and self._model_dir():
            self.encoder = self._encoder_from_path()
        if self.encoder is None:
            raise RuntimeError("this Embeddings object has no text encoder: pass encoder= (see encoder.TextEncoder), a path= that is "
                               f"a local model directory or a hub name already in the local Hugging Face cache (path={self.path!r} is "
                               "neither: nothing can be downloaded here), or index/search with vectors")
        v = self.encoder(texts)
        if not isinstance(v, torch.Tensor) or v.dim() != 2 or v.shape[0] != len(texts):
            raise ValueError("encoder must return a [len(texts), d] tensor")
        return v

    def index(self, documents: Iterable, vectors=None, batch_size: int = 512) -> None:
        """Build the index from documents (``heavy_ranker.py:86``); ``vectors`` [n, d] skips the encoder."""
        docs = self._normalise_documents(documents)
        total = None
        if vectors is None:
            # the encoder's output goes chunk by chunk straight into the shard (row-producer path): no [n, d] fp32 copy of
            # the corpus ever exists (30 GB at 10M documents), and with torch.distributed initialised every rank encodes
            # only the documents of its own row shard
            texts = [str(d.get("text", "")) for d in docs]
            total = len(docs)

            def vectors(lo: int, hi: int) -> torch.Tensor:
                return self._encode(texts[lo:hi]).float()
            if total == 0:
                vectors, total = torch.zeros((0, 1)), None
        self.index_vectors([d["id"] for d in docs], vectors, documents=docs if self.content else None, total=total,
                           # (the shard's sketch takes its centre from the FIRST chunk written: at least 65 536 rows of it)
                           chunk_rows=max(max(int(batch_size), 1) * 8, 1 << 16))
        if self.hybrid:  # every rank indexes the whole corpus' text (host side, small next to the vectors)
            self._sparse = BM25Index().index(str(d.get("text", "")) for d in docs)

    def index_vectors(self, ids: Optional[Sequence], vectors, documents: Optional[List[dict]] = None, *, local: bool = False,
                      total: Optional[int] = None, chunk_rows: int = 1 << 18) -> None:
        """Index precomputed embeddings (float32, or float16 values already normalised).

        * default: ``vectors`` [n, d] is the WHOLE corpus; with ``torch.distributed`` initialised every rank passes the same
          arrays and keeps its contiguous row shard (``shard_bounds``).
        * ``local=True``: ``vectors`` (and ``ids``, ``documents``) are only THIS rank's rows [lo, hi) of a corpus of ``total``
          rows -- no rank ever holds the whole corpus (80M x 768 fp16 = 123 GB).  ``total`` defaults to the sum over ranks.
        * ``vectors`` may also be a callable ``rows(lo, hi) -> [hi - lo, d] array`` (needs ``total``); it is asked for this
          rank's rows in chunks of ``chunk_rows`` and the shard is filled chunk by chunk without a second copy.  ``ids`` /
          ``documents`` then cover either this rank's rows or the whole corpus (told apart by their length).
        """
        self._host_ids = None  # ids of an earlier index() on this object must not leak into the new one
        producer = vectors if callable(vectors) else None
        if producer is not None:
            if total is None:
                raise ValueError("a row producer needs total= (rows of the whole corpus)")
            n = int(total)
            lo, hi = shard_bounds(n, self.world, self.rank)
            # the shard's first chunk tells the dimension and is stored as it is (no one-row probe: a producer may be stateful --
            # a tokenizer feeding from a stream -- and an encoder call for one row would be paid twice); an empty shard asks row 0
            first = producer(lo, min(hi, lo + chunk_rows)) if hi > lo else producer(0, min(1, n))
            first = torch.from_numpy(np.ascontiguousarray(first)) if isinstance(first, np.ndarray) else first
            d, v = int(first.shape[1]), None
            vdtype = first.dtype
        else:
            v = torch.from_numpy(np.ascontiguousarray(vectors)) if isinstance(vectors, np.ndarray) else vectors
            if v.dim() != 2:
                raise ValueError("vectors must be [n, d]")
            d, vdtype = int(v.shape[1]), v.dtype
            if local:
                mine = int(v.shape[0])
                if total is None:
                    if self.world > 1 or (dist.is_initialized() and os.environ.get("VQA_ALWAYS_GATHER") == "1"):
                        t = torch.tensor([mine], dtype=torch.int64)
                        if dist.get_backend(self.group) == "nccl":
                            t = t.cuda(self.device)
                        dist.all_reduce(t, group=self.group)
                        total = int(t.item())
                    else:
                        total = mine
                n = int(total)
                lo, hi = shard_bounds(n, self.world, self.rank)
                if hi - lo != mine:
                    raise ValueError(f"rank {self.rank} holds rows [{lo}, {hi}) of {n} but was given {mine} vectors")
            else:
                n = int(v.shape[0])
                lo, hi = shard_bounds(n, self.world, self.rank)
        part = local or producer is not None  # ids / documents cover only this rank's rows
        if producer is not None and not local and (len(ids) if ids is not None else len(documents) if documents is not None
                                                   else hi - lo) == n:
            part = False  # a row producer with the whole corpus' ids / documents (what index(documents) passes)
        if documents is not None and len(documents) != (hi - lo if part else n):
            raise ValueError(f"{len(documents)} documents for {(hi - lo) if part else n} vectors")
        if ids is not None and len(ids) != (hi - lo if part else n):
            raise ValueError(f"{len(ids)} ids for {(hi - lo) if part else n} vectors")
        int_ids = ids is None or all(isinstance(i, (int, np.integer)) for i in ids)
        dev_ids = None
        id_base = 0
        if ids is None:
            id_base = lo
        elif int_ids:
            arr = np.asarray(ids, dtype=np.int64)
            if not part and n and np.array_equal(arr, np.arange(arr[0], arr[0] + n)):
                id_base = int(arr[0]) + lo  # contiguous ids (sqlite rowids): no id vector needed
            else:
                dev_ids = arr if part else arr[lo:hi]
        else:
            if part:
                raise ValueError("non-integer ids need the whole id list on every rank: pass the full arrays (local=False)")
            self._host_ids = list(ids)
            id_base = lo  # the device returns global row positions, mapped through _host_ids on the host
        # what the dense path reports for global row position p (the sparse half speaks row positions)
        if part:
            self._row_ids = None  # hybrid search needs the whole corpus' ids on every rank
        elif not int_ids:
            self._row_ids = np.arange(n, dtype=np.int64)
        elif dev_ids is None:
            self._row_ids = np.arange(n, dtype=np.int64) + (id_base - lo)
        else:
            self._row_ids = np.asarray(ids, dtype=np.int64)
        self._sparse = None
        if self._index is not None:
            self._index.close()
        normalize = self.normalize and vdtype == torch.float32
        if producer is None:
            rows = v if part else v[lo:hi]
            self._index = DeviceIndex(rows, ids=dev_ids, id_base=id_base, dtype=self.dtype, device=self.device, normalize=normalize)
        else:
            self._index = DeviceIndex.empty(hi - lo, d, id_base=id_base, dtype=self.dtype, device=self.device,
                                            with_ids=dev_ids is not None)
            for c0 in range(lo, hi, chunk_rows):
                c1 = min(hi, c0 + chunk_rows)
                rows = first if c0 == lo else producer(c0, c1)
                self._index.set_rows(c0 - lo, rows, dev_ids[c0 - lo:c1 - lo] if dev_ids is not None else None, normalize=normalize)
        self._searcher = ShardedSearcher(self._local_search, None, self.group)
        self.n, self.d, self._lo = n, d, lo
        self._docs_db = None
        self._docs_rows = self._docs_mem = None
        if documents is not None:
            # key = what the device reports for the row: its global position (ids=None or non-integer ids, which map through
            # _host_ids afterwards), else the integer id given for it -- never the document's own "id" field, which need
            # not be an integer nor agree with `ids`
            base = lo if part else 0
            by_pos = ids is None or not int_ids
            self._docs_rows = [(base + i if by_pos else int(ids[i]), d_) for i, d_ in enumerate(documents)]
            self._docs_mem = dict(self._docs_rows)

    def _local_search(self, q: torch.Tensor, k: int, out_s: torch.Tensor, out_i: torch.Tensor) -> None:
        self._index.search(q, k, out=(out_s, out_i))

    def count(self) -> int:
        return self.n

    # ---- searching ------------------------------------------------------------------------------------------------
    def _query_vectors(self, queries) -> torch.Tensor:
        if isinstance(queries, torch.Tensor):
            v = queries
        elif isinstance(queries, np.ndarray):
            v = torch.from_numpy(np.ascontiguousarray(queries))
        elif len(queries) and isinstance(queries[0], str):
            v = self._encode(list(queries))
        else:
            v = torch.as_tensor(np.asarray(queries, dtype=np.float32))
        if v.dim() == 1:
            v = v[None, :]
        if v.dtype not in (torch.float32, torch.float16):
            v = v.float()
        dev = torch.device("cuda", self.device)
        v = v.to(dev).contiguous()
        if self.normalize and v.dtype == torch.float32 and v.shape[0]:
            # x / ||x|| by the library's HIP kernel (zero rows stay zero): no torch arithmetic on the query path
            out = torch.empty_like(v)
            with torch.cuda.device(dev):
                N.check(N.load().vqa_normalize_convert(v.data_ptr(), int(v.shape[0]), int(v.shape[1]), 1, N.VQA_F32, out.data_ptr(),
                                                       torch.cuda.current_stream(dev).cuda_stream), "vqa_normalize_convert")
            v = out
        return v

    def batchsearch(self, queries, limit: int = 3) -> List[list]:
        """txtai ``batchsearch``: one result list per query; ``queries`` is a list of strings or a [B, d] array."""
        if self._index is None:
            raise RuntimeError("the index is empty: call index()/load() first")
        if isinstance(queries, str):
            queries = [queries]
        limit = int(limit)
        # host-resident vector queries against one shard (the reference asks one question per call, heavy_ranker.py:97-101): the
        # library's latency entry -- one call, no torch tensor on the way in or out
        if isinstance(queries, np.ndarray) and queries.dtype == np.float64:
            queries = queries.astype(np.float32)
        if (isinstance(queries, torch.Tensor) and queries.dim() == 2 and queries.dtype in (torch.float32, torch.float16) and
                0 < queries.shape[0] <= HOST_PATH_MAX_QUERIES and not self._searcher.collective):
            # a few vectors as a tensor: on this device as they are (the entry takes device pointers too), host tensors as arrays
            if queries.is_cuda and queries.device.index == self.device:
                queries = queries.contiguous()
            elif not queries.is_cuda:
                queries = queries.numpy()
        if (isinstance(queries, (np.ndarray, torch.Tensor)) and queries.ndim == 2 and queries.dtype in (np.float32, np.float16, torch.float32, torch.float16) and
                0 < queries.shape[0] <= HOST_PATH_MAX_QUERIES and not self._searcher.collective and
                (isinstance(queries, np.ndarray) or (queries.is_cuda and queries.device.index == self.device))):
            if queries.shape[1] != self.d:
                raise ValueError(f"query dimension {queries.shape[1]} != index dimension {self.d}")
            is_f32 = queries.dtype in (np.float32, torch.float32)
            scores, ids = self._index.search_host(queries, limit, normalize=self.normalize and is_f32)
            return self._format(scores, ids)
        # a few text questions against one shard, encoder built here (TextEncoder over QuestionEncoder): tokenizer -> ONE library call for
        # the forward (host ids in, device vectors out) -> ONE for the search (device vectors in, host results out)
        if self._text_fast_path(queries):
            return self._search_texts_fast(list(queries), limit)
        if self._text_fast_path(queries, hybrid_ok=True):  # hybrid=True (heavy_ranker.py:78): the same two library calls for the dense half
            qd = self._encode_texts_fast(list(queries))
            if qd is not None:
                return self._hybrid(qd, list(queries), limit, host_results=True)
        q = self._query_vectors(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        texts = list(queries) if not isinstance(queries, (np.ndarray, torch.Tensor)) and len(queries) and isinstance(queries[0], str) else None
        if self.hybrid and self._sparse is not None and texts is not None and 0.0 < self.weights < 1.0:
            return self._hybrid(q, texts, limit)
        if 0 < q.shape[0] <= HOST_PATH_MAX_QUERIES and not self._searcher.collective:
            # a few questions through a caller-supplied encoder (any callable): its device vectors, already normalised above, through the
            # host-result entry -- no torch tensor for the results, no copy operation
            scores, ids = self._index.search_host(q, limit, normalize=False)
            return self._format(scores, ids)
        scores, ids = self._searcher.search(q, limit)
        torch.cuda.current_stream(q.device).synchronize()
        return self._format(scores.cpu().numpy(), ids.cpu().numpy())

    def _text_fast_path(self, queries, hybrid_ok: bool = False) -> bool:
        from .encoder import QuestionEncoder, TextEncoder
        if isinstance(queries, (np.ndarray, torch.Tensor)) or not len(queries) or not isinstance(queries[0], str):
            return False
        if len(queries) > HOST_PATH_MAX_QUERIES or self._searcher.collective:
            return False
        if not hybrid_ok and self.hybrid and self._sparse is not None and 0.0 < self.weights < 1.0:
            return False
        if self.encoder is None and self._model_dir():
            self.encoder = self._encoder_from_path()
        enc = self.encoder
        return isinstance(enc, TextEncoder) and isinstance(enc.encoder, QuestionEncoder) and enc.encoder.device == self.device

    def _encode_texts_fast(self, texts: List[str]):
        """The first of the fast path's two library calls: tokenizer, then the encoder's host entry ENQUEUED on the current stream (no
        wait).  Returns the device vectors, or None for a question longer than the encoder's workspace (the general path slices it)."""
        te = self.encoder
        ids, mask = te.tokenizer(texts)
        ids = ids.cpu().numpy() if isinstance(ids, torch.Tensor) else np.asarray(ids)
        mask = mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
        if ids.ndim != 2:
            raise ValueError("the tokenizer must return [B, L] input_ids")
        if ids.size > te.encoder.max_tokens:
            return None
        h = int(te.encoder.config["hidden"])
        if h != self.d:
            raise ValueError(f"query dimension {h} != index dimension {self.d}")
        if self._qbuf is None or self._qbuf.shape[1] != h:
            self._qbuf = torch.empty((HOST_PATH_MAX_QUERIES, h), dtype=torch.float32, device=torch.device("cuda", self.device))
        return te.encoder.forward_host(ids, mask, self._qbuf, pooling=te.pooling, normalize=te.normalize)

    def _search_texts_fast(self, texts: List[str], limit: int) -> List[list]:
        q = self._encode_texts_fast(texts)
        if q is None:
            return self.batchsearch(self._encode(texts), limit)
        scores, out_ids = self._index.search_host(q, limit, normalize=self.normalize)
        return self._format(scores, out_ids)

    def search_begin(self, query: str):
        """First half of ``search(query, limit)`` for a caller that asks SEVERAL retrievers the same question (``heavy_ranker.py:98-100``
        asks two, one after the other): the question's encoder forward is enqueued on the current stream and the call returns at once,
        so the forwards of two models -- each far too small to fill the device -- run side by side when the caller gives every
        retriever a stream of its own (``heavy_ranker.rank_query``).  ``search_end`` completes the call.  Not part of txtai.

        ONE question may be outstanding per object: the returned token's vector lives in this object's single staging buffer, so a second
        ``search_begin`` before the first token's ``search_end`` is refused (it would overwrite the first question's vector: ADVICE r5)."""
        if self._index is None:
            raise RuntimeError("the index is empty: call index()/load() first")
        if not isinstance(query, str):
            raise ValueError("search_begin() takes one text question")
        if getattr(self, "_begun", None) is not None:
            raise RuntimeError("search_begin(): the previous question of this Embeddings object has not been completed by search_end()")
        q = self._encode_texts_fast([query]) if self._text_fast_path([query], hybrid_ok=True) else None
        token = (query, q)
        self._begun = token if q is not None else None  # (a question off the fast path holds no buffer)
        return token

    def search_end(self, token, limit: int = 3) -> list:
        """Second half: the search of the vector ``search_begin`` left on the device (waits for it), or the whole ``search`` when the
        question did not take the fast path."""
        query, q = token
        if q is None:
            return self.search(query, limit)
        if getattr(self, "_begun", None) is not token:
            raise RuntimeError("search_end(): this token is not the outstanding question of this Embeddings object")
        self._begun = None
        if not self._text_fast_path([query]):  # hybrid=True: the dense candidates from q, the BM25 half on the host
            return self._hybrid(q, [query], int(limit), host_results=True)[0]
        scores, out_ids = self._index.search_host(q, int(limit), normalize=self.normalize)
        return self._format(scores, out_ids)[0]

    def _hybrid(self, q: torch.Tensor, texts: List[str], limit: int, host_results: bool = False) -> List[list]:
        """txtai's hybrid search [recalled, see sparse.py]: 10 x limit candidates from each half, per-id convex combination
        ``weights * dense + (1 - weights) * bm25`` (BM25 normalised to 0..1), best ``limit``.  ``host_results``: ``q`` is the device output
        of the encoder's host entry and the dense candidates come through the index's host-result entry (one question: one kernel)."""
        cand = min(10 * limit, 1024, max(self.n, 1))
        sparse_all = None
        if host_results:
            # the encoder forward that produces q is still running on the device: the BM25 half is computed under it, the wait comes after
            sparse_all = [[(int(self._row_ids[r]), s) for r, s in self._sparse.search(text, cand)] for text in texts]
            ds, di = self._index.search_host(q, cand, normalize=self.normalize)
        else:
            scores, ids = self._searcher.search(q, cand)
            torch.cuda.current_stream(q.device).synchronize()
            ds, di = scores.cpu().numpy(), ids.cpu().numpy()
        out_s = np.full((len(texts), limit), -np.inf, dtype=np.float32)
        out_i = np.full((len(texts), limit), -1, dtype=np.int64)
        for b, text in enumerate(texts):
            dense = [(int(i), float(s)) for i, s in zip(di[b], ds[b]) if i >= 0 and s > 0]  # txtai drops dense scores <= 0
            sparse = sparse_all[b] if sparse_all is not None else [(int(self._row_ids[r]), s) for r, s in self._sparse.search(text, cand)]
            for j, (uid, sc) in enumerate(combine(dense, sparse, limit, self.weights, self._sparse.normalize)):
                out_i[b, j], out_s[b, j] = uid, sc
        return self._format(out_s, out_i)

    def search(self, query: Query, limit: int = 3) -> list:
        """``heavy_ranker.py:98``: the ``limit`` best documents for one query, best first."""
        if not isinstance(query, str):
            arr = query if isinstance(query, (np.ndarray, torch.Tensor)) else np.asarray(query, dtype=np.float32)
            if arr.ndim != 1:
                raise ValueError("search() takes ONE query (a string or a 1-D vector); use batchsearch() for several")
            query = arr
        return self.batchsearch([query] if isinstance(query, str) else query[None, :], limit)[0]

    def _format(self, scores: np.ndarray, ids: np.ndarray) -> List[list]:
        keep = ids >= 0
        if self.min_score is not None:
            keep &= scores > self.min_score
        texts = {}
        if self.content:
            wanted = ids[keep].tolist()
            if self._docs_db is not None:
                texts = docstore.fetch_docs(self._docs_db, wanted)
            elif self._docs_mem is not None:
                texts = {i: self._docs_mem[i].get("text") for i in set(wanted) if i in self._docs_mem}
        # plain Python lists first: per-element numpy indexing costs more than the GPU search of a 256-query batch
        il, sl, kl = ids.tolist(), scores.tolist(), keep.tolist()
        ext = (lambda i: self._host_ids[i]) if self._host_ids is not None else (lambda i: i)
        if self.content:
            return [[{"id": ext(i), "text": texts.get(i), "score": s} for i, s, k in zip(ir, sr, kr) if k]
                    for ir, sr, kr in zip(il, sl, kl)]
        return [[(ext(i), s) for i, s, k in zip(ir, sr, kr) if k] for ir, sr, kr in zip(il, sl, kl)]

    def _ext_id(self, device_id: int):
        return self._host_ids[device_id] if self._host_ids is not None else device_id

    # ---- persistence -----------------------------------------------------------------------
    def save(self, path: str) -> None:
        """``heavy_ranker.py:87``: ``<path>/meta.json`` + ``vectors.f16`` / ``vectors.f32`` (row-major, mmap-able, shardable by byte range)
        + ``ids.i64`` (+ ``documents.db`` with the reference's table schema when ``content=True``).  With
        ``torch.distributed`` initialised every rank writes the byte range of its own row shard into the same files
        (rank 0 creates them at full size first); rank 0 writes the metadata and the documents."""
        if self._index is None:
            raise RuntimeError("nothing to save: the index is empty")
        os.makedirs(path, exist_ok=True)
        has_ids = self._index.has_ids
        vec_dtype = "float32" if self._index.dtype == N.VQA_F32 else "float16"
        itemsize = np.dtype(vec_dtype).itemsize
        vec_path, ids_path = os.path.join(path, VECTOR_FILES[vec_dtype]), os.path.join(path, IDS_FILE)
        lo, hi = shard_bounds(self.n, self.world, self.rank)
        if self.rank == 0:  # files at their final size, so that every rank can write its own range
            with open(vec_path, "wb") as f:
                f.truncate(self.n * self.d * itemsize)
            if has_ids:
                with open(ids_path, "wb") as f:
                    f.truncate(self.n * 8)
            elif os.path.exists(ids_path):
                os.remove(ids_path)  # left by an earlier save of an index WITH an id vector into the same directory
        if self.world > 1:
            dist.barrier(group=self.group)
        # chunks leave the device into one of two pinned buffers (a DMA) and are written by a few threads, each its own byte
        # range (os.pwritev releases the GIL), while the next chunk is fetched into the other buffer
        step = min(1 << 18, max(hi - lo, 1))
        nthreads = max(1, min(int(os.environ.get("VQA_LOAD_THREADS", "4")), 16))
        store_t = {np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32,
                   np.dtype(np.uint8): torch.uint8}[np.dtype(_STORE_NP[self._index.dtype])]
        with open(vec_path, "r+b", buffering=0) as fv, ThreadPoolExecutor(max_workers=nthreads) as pool:
            fi = open(ids_path, "r+b") if has_ids else None
            fd = fv.fileno()

            def write_range(view: memoryview, offset: int) -> None:
                done = 0
                while done < len(view):
                    done += os.pwritev(fd, [view[done:]], offset + done)

            try:
                pinned, jobs = [None, None], [[], []]
                for i, c0 in enumerate(range(0, hi - lo, step)):
                    c = min(step, hi - lo - c0)
                    b = i & 1
                    for job in jobs[b]:  # the writes out of this buffer, two chunks ago
                        job.result()
                    jobs[b] = []
                    if pinned[b] is None:
                        pinned[b] = torch.empty((step, self.d), dtype=store_t, pin_memory=True).numpy()
                    rows, ids = self._index.get_rows(c0, c, out=pinned[b][:c])
                    if rows.dtype == np.uint8:
                        # fp8 index: codes of 16 * x -> x as fp16 (3 mantissa bits, exponents down to 2^-13: exact), so that
                        # load() re-encodes to the very same codes
                        from .index import FP8_SCALE
                        rows = (_e4m3_decode_table()[rows] / FP8_SCALE).astype(np.float16)
                    view = memoryview(np.ascontiguousarray(rows)).cast("B")
                    base = (lo + c0) * self.d * itemsize
                    cut = -(-(-(-len(view) // nthreads)) // 4096) * 4096
                    jobs[b] = [pool.submit(write_range, view[o:o + cut], base + o) for o in range(0, len(view), cut)]
                    if fi is not None:
                        fi.seek((lo + c0) * 8)
                        fi.write(ids.tobytes())
                for b in (0, 1):
                    for job in jobs[b]:
                        job.result()
            finally:
                if fi is not None:
                    fi.close()
        docs_rows = self._docs_rows
        if self.content and docs_rows is not None and self.world > 1 and len(docs_rows) != self.n:
            # every rank holds the documents of its own rows only (index_vectors(local=True)): rank 0 collects them
            gathered = [None] * self.world if self.rank == 0 else None
            dist.gather_object(docs_rows, gathered, dst=0, group=self.group)
            if self.rank == 0:
                docs_rows = [pd for part in gathered for pd in part]
        if self.rank == 0:
            meta = {"format": FORMAT_VERSION, "n": self.n, "d": self.d, "dtype": self.dtype, "normalize": self.normalize,
                    "pooling": self.pooling, "path": self.path, "content": self.content, "hybrid": self.hybrid,
                    "id_base": self._index.id_base - self._lo, "has_ids": has_ids, "host_ids": self._host_ids,
                    "vector_dtype": vec_dtype, "weights": self.weights}
            with open(os.path.join(path, META_FILE), "w") as f:
                json.dump(meta, f)
            if self._sparse is not None:
                self._sparse.save(path)
            if self.content:
                db = os.path.join(path, DOCS_FILE)
                if docs_rows is not None:
                    if os.path.exists(db):
                        os.remove(db)
                    # one row per corpus row, keyed by what the device reports for it (row position for string ids)
                    docstore.write_documents(db, [{"id": key, "text": doc.get("text"), "source": doc.get("source")}
                                                  for key, doc in docs_rows])
                elif self._docs_db is not None and os.path.abspath(self._docs_db) != os.path.abspath(db):
                    shutil.copyfile(self._docs_db, db)  # a loaded index keeps its documents on re-save
        if self.world > 1:
            dist.barrier(group=self.group)

    def load(self, path: str) -> "Embeddings":
        """``heavy_ranker.py:92,94``: ``Embeddings().load(dir)``; returns ``self``.  Every rank reads only the byte range of
        its row shard; the rows stream file -> pinned host buffers -> HBM with the next chunk's file read overlapping the
        previous chunk's copy (``load_stats`` reports the rate)."""
        meta_path = os.path.join(path, META_FILE)
        if not os.path.isfile(meta_path):
            raise FileNotFoundError(f"{meta_path} not found: {path!r} is not a saved index")
        with open(meta_path) as f:
            meta = json.load(f)
        if meta.get("format") != FORMAT_VERSION:
            raise ValueError(f"unsupported index format {meta.get('format')}")
        n, d = int(meta["n"]), int(meta["d"])
        self.dtype, self.normalize, self.pooling = meta["dtype"], meta["normalize"], meta["pooling"]
        if getattr(self, "_auto_encoder_path", None) not in (None, meta["path"]):
            self.encoder, self._auto_encoder_path = None, None  # built from the path of ANOTHER model: the loaded index names its own
        self.path, self.content, self.hybrid = meta["path"], meta["content"], meta["hybrid"]
        self._host_ids = meta.get("host_ids")
        lo, hi = shard_bounds(n, self.world, self.rank)
        vec_dtype = meta.get("vector_dtype", "float16")
        np_dtype = np.dtype(vec_dtype)
        vec_path = os.path.join(path, VECTOR_FILES[vec_dtype])
        expected = n * d * np_dtype.itemsize
        if os.path.getsize(vec_path) != expected:
            raise ValueError(f"{vec_path}: {os.path.getsize(vec_path)} bytes, expected {expected}")
        ids = None
        if meta["has_ids"]:
            ids = np.fromfile(os.path.join(path, IDS_FILE), dtype=np.int64, count=hi - lo, offset=lo * 8)
        if self._index is not None:
            self._index.close()
        self._index = DeviceIndex.empty(hi - lo, d, id_base=int(meta["id_base"]) + lo, dtype=self.dtype, device=self.device,
                                        with_ids=ids is not None)
        self.load_stats = self._stream_rows(vec_path, np_dtype, lo, hi, d, ids)
        self._searcher = ShardedSearcher(self._local_search, None, self.group)
        self.n, self.d, self._lo = n, d, lo
        db = os.path.join(path, DOCS_FILE)
        self._docs_db = db if os.path.isfile(db) else None
        self._docs_mem = self._docs_rows = None
        self.weights = float(meta.get("weights", 0.5))
        self._sparse = BM25Index.load(path) if self.hybrid else None
        if self._host_ids is not None:
            self._row_ids = np.arange(n, dtype=np.int64)
        elif meta["has_ids"]:
            self._row_ids = np.fromfile(os.path.join(path, IDS_FILE), dtype=np.int64) if self.hybrid else None
        else:
            self._row_ids = np.arange(n, dtype=np.int64) + int(meta["id_base"])
        return self

    def _stream_rows(self, vec_path: str, np_dtype, lo: int, hi: int, d: int, ids, chunk_rows: int = 1 << 18) -> dict:
        """File rows [lo, hi) -> the shard, double buffered: while chunk i travels pinned host -> device (async copy on a side
        stream) and is transposed into the tiled layout, chunk i + 1 is read from the file into the other pinned buffer (by
        VQA_LOAD_THREADS = 4 reader threads, each its own byte range)."""
        dev = torch.device("cuda", self.device)
        tdt = torch.float32 if np_dtype == np.float32 else torch.float16
        rows_total = hi - lo
        t0 = time.perf_counter()
        if rows_total == 0:
            return {"bytes": 0, "seconds": 0.0, "gb_per_s": 0.0}
        chunk_rows = min(chunk_rows, rows_total)
        with torch.cuda.device(dev):
            pinned = [torch.empty((chunk_rows, d), dtype=tdt, pin_memory=True) for _ in range(2)]
            staged = [torch.empty((chunk_rows, d), dtype=tdt, device=dev) for _ in range(2)]
            copy_stream = torch.cuda.Stream(device=dev)
            done = [None, None]  # event: the copy out of pinned[b] has finished
            pending = None       # (buffer, first row, count) copied but not yet written into the index
            row_bytes = d * np_dtype.itemsize
            # a chunk is read by a few threads at once, each its own byte range straight into the pinned buffer (os.preadv
            # releases the GIL; one thread copies a page-cache resident file at 8-10 GB/s, the host -> device link takes 5x that)
            nthreads = max(1, min(int(os.environ.get("VQA_LOAD_THREADS", "4")), 16))

            def read_range(fd: int, view: memoryview, offset: int) -> None:
                got = 0
                while got < len(view):
                    r = os.preadv(fd, [view[got:]], offset + got)
                    if not r:
                        raise ValueError(f"{vec_path}: unexpected end of file")
                    got += r

            with open(vec_path, "rb", buffering=0) as f, ThreadPoolExecutor(max_workers=nthreads) as pool:
                fd = f.fileno()
                for i, c0 in enumerate(range(0, rows_total, chunk_rows)):
                    b = i & 1
                    c = min(chunk_rows, rows_total - c0)
                    if done[b] is not None:
                        done[b].synchronize()
                    view = memoryview(pinned[b].numpy()).cast("B")[:c * row_bytes]
                    base = (lo + c0) * row_bytes
                    step = -(-len(view) // nthreads)
                    step = -(-step // 4096) * 4096  # page-sized cuts
                    jobs = [pool.submit(read_range, fd, view[o:o + step], base + o) for o in range(0, len(view), step)]
                    for job in jobs:
                        job.result()
                    if pending is not None:  # the previous chunk: copy done? then transpose it into the shard
                        pb, p0, pc = pending
                        done[pb].synchronize()
                        self._index.set_rows(p0, staged[pb][:pc], ids[p0:p0 + pc] if ids is not None else None)
                    with torch.cuda.stream(copy_stream):
                        staged[b][:c].copy_(pinned[b][:c], non_blocking=True)
                        done[b] = torch.cuda.Event()
                        done[b].record(copy_stream)
                    pending = (b, c0, c)
            pb, p0, pc = pending
            done[pb].synchronize()
            self._index.set_rows(p0, staged[pb][:pc], ids[p0:p0 + pc] if ids is not None else None)
            torch.cuda.synchronize(dev)
        sec = time.perf_counter() - t0
        nbytes = rows_total * d * np_dtype.itemsize
        return {"bytes": nbytes, "seconds": sec, "gb_per_s": nbytes / sec / 1e9}
