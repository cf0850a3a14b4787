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
of the kernel.  ``hybrid=True`` is accepted for call compatibility but only the dense score is computed (the BM25
half of txtai's hybrid score is out of scope); a warning says so.
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Callable, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.distributed as dist

from . import docstore
from .index import DeviceIndex, resolve_dtype
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

Query = Union[str, np.ndarray, torch.Tensor, Sequence[float]]


class Embeddings:
    def __init__(self, config: Optional[dict] = None, *, path: Optional[str] = None, content: bool = False,
                 hybrid: bool = False, dtype: str = "fp16", device: Optional[int] = None, pooling: str = "mean",
                 normalize: bool = True, encoder: Optional[Callable[[List[str]], torch.Tensor]] = None,
                 min_score: Optional[float] = 0.0, group=None):
        cfg = dict(config or {})
        self.path = cfg.get("path", path)
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
        self._searcher: Optional[ShardedSearcher] = None
        self._host_ids: Optional[list] = None  # non-integer external ids, by row position
        self._docs_db: Optional[str] = None
        self._docs_mem: Optional[dict] = None
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

    def _encode(self, texts: List[str]) -> torch.Tensor:
        if self.encoder is None:
            raise RuntimeError("this Embeddings object has no text encoder: pass encoder= (see encoder.TextEncoder) "
                               "or index/search with vectors")
        v = self.encoder(texts)
        if not isinstance(v, torch.Tensor) or v.dim() != 2 or v.shape[0] != len(texts):
            raise ValueError("encoder must return a [len(texts), d] tensor")
        return v

    def index(self, documents: Iterable, vectors=None, batch_size: int = 512) -> None:
        """Build the index from documents (``heavy_ranker.py:86``); ``vectors`` [n, d] skips the encoder."""
        docs = self._normalise_documents(documents)
        if vectors is None:
            chunks = []
            for c0 in range(0, len(docs), batch_size):
                chunks.append(self._encode([str(d.get("text", "")) for d in docs[c0:c0 + batch_size]]).float())
            vectors = torch.cat(chunks) if chunks else torch.zeros((0, 1))
        self.index_vectors([d["id"] for d in docs], vectors, documents=docs if self.content else None)
        if self.hybrid:  # every rank indexes the whole corpus' text (host side, small next to the vectors)
            self._sparse = BM25Index().index(str(d.get("text", "")) for d in docs)

    def index_vectors(self, ids: Optional[Sequence], vectors, documents: Optional[List[dict]] = None) -> None:
        """Index precomputed embeddings [n, d] (float32, or float16 values already normalised).  With
        ``torch.distributed`` initialised every rank passes the SAME full arrays and keeps its contiguous row shard."""
        v = torch.from_numpy(np.ascontiguousarray(vectors)) if isinstance(vectors, np.ndarray) else vectors
        if v.dim() != 2:
            raise ValueError("vectors must be [n, d]")
        n, d = int(v.shape[0]), int(v.shape[1])
        if ids is not None and len(ids) != n:
            raise ValueError(f"{len(ids)} ids for {n} vectors")
        lo, hi = shard_bounds(n, self.world, self.rank)
        int_ids = ids is None or all(isinstance(i, (int, np.integer)) for i in ids)
        dev_ids = None
        id_base = 0
        if ids is None:
            id_base = lo
        elif int_ids:
            arr = np.asarray(ids, dtype=np.int64)
            if n and np.array_equal(arr, np.arange(arr[0], arr[0] + n)):
                id_base = int(arr[0]) + lo  # contiguous ids (sqlite rowids): no id vector needed
            else:
                dev_ids = arr[lo:hi]
        else:
            self._host_ids = list(ids)
            id_base = lo  # the device returns global row positions, mapped through _host_ids on the host
        # what the dense path reports for global row position p (the sparse half speaks row positions)
        if not int_ids:
            self._row_ids = np.arange(n, dtype=np.int64)
        elif dev_ids is None:
            self._row_ids = np.arange(n, dtype=np.int64) + (id_base - lo)
        else:
            self._row_ids = np.asarray(ids, dtype=np.int64)
        self._sparse = None
        if self._index is not None:
            self._index.close()
        normalize = self.normalize and v.dtype == torch.float32
        self._index = DeviceIndex(v[lo:hi], ids=dev_ids, id_base=id_base, dtype=self.dtype, device=self.device,
                                  normalize=normalize)
        self._searcher = ShardedSearcher(self._local_search, None, self.group)
        self.n, self.d = n, d
        self._docs_db = None
        self._docs_mem = {d_["id"]: d_ for d_ in documents} if documents is not None else None

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
        v = v.to(torch.device("cuda", self.device))
        if self.normalize and v.dtype == torch.float32:
            v = v / v.norm(dim=1, keepdim=True).clamp_min(torch.finfo(torch.float32).tiny)
        return v

    def batchsearch(self, queries, limit: int = 3) -> List[list]:
        """txtai ``batchsearch``: one result list per query; ``queries`` is a list of strings or a [B, d] array."""
        if self._index is None:
            raise RuntimeError("the index is empty: call index()/load() first")
        if isinstance(queries, str):
            queries = [queries]
        q = self._query_vectors(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"query dimension {q.shape[1]} != index dimension {self.d}")
        limit = int(limit)
        texts = list(queries) if not isinstance(queries, (np.ndarray, torch.Tensor)) and len(queries) and isinstance(queries[0], str) else None
        if self.hybrid and self._sparse is not None and texts is not None and 0.0 < self.weights < 1.0:
            return self._hybrid(q, texts, limit)
        scores, ids = self._searcher.search(q, limit)
        torch.cuda.current_stream(q.device).synchronize()
        return self._format(scores.cpu().numpy(), ids.cpu().numpy())

    def _hybrid(self, q: torch.Tensor, texts: List[str], limit: int) -> List[list]:
        """txtai's hybrid search [recalled, see sparse.py]: 10 x limit candidates from each half, per-id convex combination
        ``weights * dense + (1 - weights) * bm25`` (BM25 normalised to 0..1), best ``limit``."""
        cand = min(10 * limit, 1024, max(self.n, 1))
        scores, ids = self._searcher.search(q, cand)
        torch.cuda.current_stream(q.device).synchronize()
        ds, di = scores.cpu().numpy(), ids.cpu().numpy()
        out_s = np.full((len(texts), limit), -np.inf, dtype=np.float32)
        out_i = np.full((len(texts), limit), -1, dtype=np.int64)
        for b, text in enumerate(texts):
            dense = [(int(i), float(s)) for i, s in zip(di[b], ds[b]) if i >= 0 and s > 0]  # txtai drops dense scores <= 0
            sparse = [(int(self._row_ids[r]), s) for r, s in self._sparse.search(text, cand)]
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
                texts = {i: self._docs_mem[self._ext_id(i)].get("text") for i in set(wanted) if self._ext_id(i) in self._docs_mem}
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
        + ``ids.i64`` (+ ``documents.db`` with the reference's table schema when ``content=True``)."""
        if self._index is None:
            raise RuntimeError("nothing to save: the index is empty")
        if self.world != 1:
            raise NotImplementedError("save() from a sharded Embeddings: gather on one rank first")
        os.makedirs(path, exist_ok=True)
        rows, ids = self._index.get_rows()
        if rows.dtype == np.uint8:
            # fp8 index: codes of 16 * x -> x as fp16 (3 mantissa bits, exponents down to 2^-13: exact), so that
            # load() re-encodes to the very same codes
            from .index import FP8_SCALE
            rows = (_e4m3_decode_table()[rows] / FP8_SCALE).astype(np.float16)
        vec_dtype = "float32" if rows.dtype == np.float32 else "float16"
        rows.tofile(os.path.join(path, VECTOR_FILES[vec_dtype]))
        if ids is not None:
            ids.tofile(os.path.join(path, IDS_FILE))
        meta = {"format": FORMAT_VERSION, "n": self.n, "d": self.d, "dtype": self.dtype, "normalize": self.normalize,
                "pooling": self.pooling, "path": self.path, "content": self.content, "hybrid": self.hybrid,
                "id_base": self._index.id_base, "has_ids": ids is not None, "host_ids": self._host_ids,
                "vector_dtype": vec_dtype}
        meta["weights"] = self.weights
        with open(os.path.join(path, META_FILE), "w") as f:
            json.dump(meta, f)
        if self._sparse is not None:
            self._sparse.save(path)
        if self.content and self._docs_mem is not None:
            db = os.path.join(path, DOCS_FILE)
            if os.path.exists(db):
                os.remove(db)
            docs = []
            for pos, (ext, doc) in enumerate(self._docs_mem.items()):
                key = pos if self._host_ids is not None else int(ext)
                docs.append({"id": key, "text": doc.get("text"), "source": doc.get("source")})
            docstore.write_documents(db, docs)

    def load(self, path: str) -> "Embeddings":
        """``heavy_ranker.py:92,94``: ``Embeddings().load(dir)``; returns ``self``."""
        meta_path = os.path.join(path, META_FILE)
        if not os.path.isfile(meta_path):
            raise FileNotFoundError(f"{meta_path} not found: {path!r} is not a saved index")
        with open(meta_path) as f:
            meta = json.load(f)
        if meta.get("format") != FORMAT_VERSION:
            raise ValueError(f"unsupported index format {meta.get('format')}")
        n, d = int(meta["n"]), int(meta["d"])
        self.dtype, self.normalize, self.pooling = meta["dtype"], meta["normalize"], meta["pooling"]
        self.path, self.content, self.hybrid = meta["path"], meta["content"], meta["hybrid"]
        self._host_ids = meta.get("host_ids")
        lo, hi = shard_bounds(n, self.world, self.rank)
        vec_dtype = meta.get("vector_dtype", "float16")
        np_dtype = np.dtype(vec_dtype)
        vec_path = os.path.join(path, VECTOR_FILES[vec_dtype])
        expected = n * d * np_dtype.itemsize
        if os.path.getsize(vec_path) != expected:
            raise ValueError(f"{vec_path}: {os.path.getsize(vec_path)} bytes, expected {expected}")
        rows = np.memmap(vec_path, dtype=np_dtype, mode="r", shape=(n, d)) if n else np.zeros((0, d), np_dtype)
        ids = None
        if meta["has_ids"]:
            ids = np.fromfile(os.path.join(path, IDS_FILE), dtype=np.int64)[lo:hi]
        if self._index is not None:
            self._index.close()
        self._index = DeviceIndex.empty(hi - lo, d, id_base=int(meta["id_base"]) + lo, dtype=self.dtype, device=self.device,
                                        with_ids=ids is not None)
        step = 1 << 18  # rows stream file -> pinned-size chunks -> HBM; the whole shard is never resident on the host
        for c0 in range(lo, hi, step):
            c1 = min(hi, c0 + step)
            self._index.set_rows(c0 - lo, np.ascontiguousarray(rows[c0:c1]),
                                 ids[c0 - lo:c1 - lo] if ids is not None else None)
        self._searcher = ShardedSearcher(self._local_search, None, self.group)
        self.n, self.d = n, d
        db = os.path.join(path, DOCS_FILE)
        self._docs_db = db if os.path.isfile(db) else None
        self._docs_mem = None
        self.weights = float(meta.get("weights", 0.5))
        self._sparse = BM25Index.load(path) if self.hybrid else None
        if self._host_ids is not None:
            self._row_ids = np.arange(n, dtype=np.int64)
        elif meta["has_ids"]:
            self._row_ids = np.fromfile(os.path.join(path, IDS_FILE), dtype=np.int64)
        else:
            self._row_ids = np.arange(n, dtype=np.int64) + int(meta["id_base"])
        return self
