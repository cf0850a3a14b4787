"""Device-resident flat inner-product index: the host-side owner of one ``vqa_index`` handle.

Counterpart of the ANN backend txtai keeps behind ``Embeddings`` (faiss ``IndexFlatIP`` + ``IDMap``; reference
call sites ``inference_pipeline/db_utils/heavy_ranker.py:86-94`` build/load, ``:98-101`` search).  All arithmetic
runs in ``libvqa_retrieval.so`` (hand-written HIP, gfx950); torch only owns device memory and the stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as N

RESCORE_COPY_MAX_BYTES = 32 << 30  # rows of up to 32 GiB get the row-major re-scoring copy by default (DeviceIndex)
_SRC_DTYPE = {torch.float32: N.VQA_F32, torch.float16: N.VQA_F16}
_STORE_NP = {N.VQA_F32: np.float32, N.VQA_F16: np.float16, N.VQA_FP8_E4M3: np.uint8}
FP8_SCALE = 16.0  # an fp8 index stores e4m3(16 * x) (and scores 16 * q the same way); scores come back divided by 256

# Process-wide option defaults laid UNDER ``DeviceIndex(options=...)`` (vqa_index_options of include/vqa_retrieval.h by field name,
# plus "sketch": keep an int8 sketch where the library's plan uses one).  Empty in production: the library's own defaults are the
# production plan.  Tests and A/B scripts put entries here -- explicit Python state instead of the VQA_* environment variables the
# library read up to round 4 (which could change a deployed process's behaviour from the caller's environment).
DEFAULT_OPTIONS: dict = {}


def _require_gpu(device: int) -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("vietnamese_qa_system_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; "
                           "there is no CPU fallback for the retrieval path")
    if device < 0 or device >= torch.cuda.device_count():
        raise ValueError(f"device {device} outside [0, {torch.cuda.device_count()})")


def resolve_dtype(dtype) -> int:
    if isinstance(dtype, int):
        if dtype not in N.DTYPE_BYTES:
            raise ValueError(f"unknown index dtype code {dtype}")
        return dtype
    try:
        return N.DTYPE_NAMES[str(dtype).lower()]
    except KeyError:
        raise ValueError(f"unknown index dtype {dtype!r}; expected one of {sorted(N.DTYPE_NAMES)}") from None


def _as_tensor(a, what: str) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{what} must be a torch tensor or numpy array")
    return t


class DeviceIndex:
    """One row shard of the corpus in HBM (tiled layout; fp16, fp8-e4m3 or fp32 storage), searched by the fused MFMA
    scoring + top-k kernel.

    ``DeviceIndex(vectors, ids)`` builds a shard from ``vectors`` [n, d] (torch tensor on any device or numpy array,
    ``float32`` or ``float16``); ``normalize=True`` L2-normalises float32 rows on the device first (txtai's behaviour
    at index time).  ``ids`` [n] int64 external ids or ``None`` (id = ``id_base`` + row position; sqlite
    AUTOINCREMENT rowids start at 1, ``setup_db.py:14``).  ``DeviceIndex.empty(n, d)`` + :meth:`set_rows` fills a
    large shard chunk by chunk without a second full copy.  ``options``: fields of ``vqa_index_options`` by name (A/B runs and
    tests; production passes none).
    """

    def __init__(self, vectors=None, ids=None, *, id_base: int = 0, dtype="fp16", device: int = 0, normalize: bool = False,
                 n: Optional[int] = None, d: Optional[int] = None, with_ids: Optional[bool] = None, sketch: Optional[bool] = None,
                 rescore_copy: Optional[bool] = None, options: Optional[dict] = None):
        self._handle = ctypes.c_void_p()
        self.device = int(device)
        _require_gpu(self.device)
        self._lib = N.load()
        self.dtype = resolve_dtype(dtype)
        if vectors is not None:
            v = _as_tensor(vectors, "vectors")
            if v.dim() != 2:
                raise ValueError("vectors must be 2-D [n, d]")
            n, d = int(v.shape[0]), int(v.shape[1])
        elif n is None or d is None:
            raise ValueError("either vectors or (n, d) must be given")
        has_ids = bool(with_ids) if with_ids is not None else ids is not None
        # int8 sketch beside the fp16 rows of a large shard (include/vqa_retrieval.h VQA_INDEX_SKETCH): on by default, the
        # library keeps one only where its two-stage search applies; VQA_SKETCH=0 or sketch=False: exact fp16 scan everywhere
        opts = dict(DEFAULT_OPTIONS)
        opts.update(options or {})
        if sketch is None:
            sketch = bool(opts.pop("sketch", True))
        opts.pop("sketch", None)
        flags = (N.VQA_INDEX_HAS_IDS if has_ids else 0) | (N.VQA_INDEX_SKETCH if sketch and self.dtype in (N.VQA_F16, N.VQA_F32) else 0)
        # row-major copy of the rows for the sketch search's exact re-scoring (VQA_INDEX_RESCORE_ROWS: +100 % of the rows' memory,
        # re-scoring 3x faster, bit-equal results; kept only where the library keeps a sketch).  Default: shards whose rows take
        # up to RESCORE_COPY_MAX_BYTES (an 80M x 768 fp16 shard on one device does without); VQA_RESCORE_COPY=0 / 1 overrides.
        # (the argument wins over options["rescore_copy"]; True / 1 forces the copy past the library's free-memory rule, False / 0 forbids it)
        if rescore_copy is None and int(opts.get("rescore_copy", -1)) in (0, 1):
            rescore_copy = bool(opts["rescore_copy"])
        elif rescore_copy is not None:
            opts["rescore_copy"] = 1 if rescore_copy else 0
        if rescore_copy is None:
            rescore_copy = int(n) * int(d) * N.DTYPE_BYTES[self.dtype] <= RESCORE_COPY_MAX_BYTES
        if rescore_copy and (flags & N.VQA_INDEX_SKETCH):
            flags |= N.VQA_INDEX_RESCORE_ROWS
        self.options = N.index_options(flags, **opts)
        with torch.cuda.device(self.device):
            N.check(self._lib.vqa_index_create_ex(ctypes.byref(self._handle), self.device, int(n), int(d), self.dtype, None,
                                                  N.VQA_F16, None, int(id_base), ctypes.byref(self.options)),
                    "vqa_index_create_ex")
        self.n, self.d, self.id_base, self.has_ids = int(n), int(d), int(id_base), has_ids
        if vectors is not None and n:
            self.set_rows(0, v, ids, normalize=normalize)

    @classmethod
    def empty(cls, n: int, d: int, *, id_base: int = 0, dtype="fp16", device: int = 0, with_ids: bool = False,
              sketch: Optional[bool] = None, rescore_copy: Optional[bool] = None, options: Optional[dict] = None) -> "DeviceIndex":
        """A shard of ``n`` zero rows to be filled with :meth:`set_rows`."""
        return cls(None, None, id_base=id_base, dtype=dtype, device=device, n=n, d=d, with_ids=with_ids, sketch=sketch,
                   rescore_copy=rescore_copy, options=options)

    # -- filling -------------------------------------------------------------------------------------------------
    def set_rows(self, first: int, vectors, ids=None, *, normalize: bool = False, chunk_rows: int = 1 << 20) -> None:
        """Write ``vectors`` [count, d] (float32/float16, host or device) to rows [first, first + count)."""
        if not self._handle.value:
            raise RuntimeError("index is closed")
        v = _as_tensor(vectors, "vectors")
        if v.dim() != 2 or int(v.shape[1]) != self.d:
            raise ValueError(f"vectors must be [count, {self.d}]")
        if v.dtype not in _SRC_DTYPE:
            raise ValueError(f"vectors must be float32 or float16, got {v.dtype}")
        count = int(v.shape[0])
        if first < 0 or first + count > self.n:
            raise ValueError(f"rows [{first}, {first + count}) outside the shard of {self.n} rows")
        ids_t = None
        if ids is not None:
            ids_t = _as_tensor(ids, "ids").to(torch.int64).contiguous()
            if ids_t.numel() != count:
                raise ValueError(f"{ids_t.numel()} ids for {count} rows")
        if (ids_t is not None) != self.has_ids and count:
            raise ValueError("ids must be given exactly when the index was created with ids")
        dev = torch.device("cuda", self.device)
        with torch.cuda.device(dev):
            for c0 in range(0, count, chunk_rows):
                c1 = min(count, c0 + chunk_rows)
                rows = v[c0:c1].contiguous()
                if normalize:
                    # x / ||x|| in fp32 on the device, then the library rounds to the storage type
                    src = rows.to(dev, dtype=torch.float32)
                    out = torch.empty_like(src)
                    stream = torch.cuda.current_stream(dev)
                    N.check(self._lib.vqa_normalize_convert(src.data_ptr(), c1 - c0, self.d, 1, N.VQA_F32, out.data_ptr(),
                                                            stream.cuda_stream), "vqa_normalize_convert")
                    stream.synchronize()
                    rows = out
                chunk_ids = ids_t[c0:c1].contiguous() if ids_t is not None else None
                if rows.is_cuda:
                    torch.cuda.current_stream(rows.device).synchronize()  # set_rows runs on the null stream
                N.check(self._lib.vqa_index_set_rows(self._handle, first + c0, c1 - c0, rows.data_ptr(), _SRC_DTYPE[rows.dtype],
                                                     chunk_ids.data_ptr() if chunk_ids is not None else None),
                        "vqa_index_set_rows")

    def get_rows(self, first: int = 0, count: Optional[int] = None,
                 out: Optional[np.ndarray] = None) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Stored rows [first, first + count) as a host array [count, d] in the storage type (``float16``, ``float32`` or
        ``uint8`` e4m3 codes of ``16 * x``) and their ids, if the index has an id vector -- the export used by
        ``Embeddings.save``.  ``out``: a C-contiguous host array of that shape and type to fill (a pinned one makes the
        device -> host copy a DMA)."""
        if not self._handle.value:
            raise RuntimeError("index is closed")
        count = self.n - first if count is None else int(count)
        if out is not None:
            if out.shape != (count, self.d) or out.dtype != _STORE_NP[self.dtype] or not out.flags["C_CONTIGUOUS"]:
                raise ValueError(f"out must be a C-contiguous {_STORE_NP[self.dtype]} array of shape {(count, self.d)}")
            rows = out
        else:
            rows = np.empty((count, self.d), dtype=_STORE_NP[self.dtype])
        ids = np.empty((count,), dtype=np.int64) if self.has_ids else None
        N.check(self._lib.vqa_index_get_rows(self._handle, int(first), count, rows.ctypes.data if count else None,
                                             ids.ctypes.data if ids is not None and count else None), "vqa_index_get_rows")
        return rows, ids

    # -- lifetime ------------------------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.vqa_index_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return self.n

    # -- search --------------------------------------------------------------------------------------------------
    def search(self, queries: torch.Tensor, k: int, *, return_positions: bool = False,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
               ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """``queries`` [B, d] fp32/fp16 cuda tensor -> (scores [B, k] fp32, ids [B, k] int64[, positions]), best first,
        ties by row position; slots beyond the shard's row count hold (-inf, -1).  Asynchronous on the current stream.
        ``out`` = (scores, ids): contiguous [B, k] float32 / int64 tensors on this device to write into (the sharded
        searcher passes views of its all-gather send buffer)."""
        if not self._handle.value:
            raise RuntimeError("index is closed")
        dev = torch.device("cuda", self.device)
        if not isinstance(queries, torch.Tensor) or queries.dim() != 2 or queries.shape[1] != self.d:
            raise ValueError(f"queries must be a [B, {self.d}] tensor")
        if queries.dtype not in (torch.float32, torch.float16):
            raise ValueError(f"queries must be float32 or float16, got {queries.dtype}")
        q = queries.to(dev).contiguous()
        b = int(q.shape[0])
        if b == 0:
            raise ValueError("empty query batch")
        if out is not None:
            scores, ids = out
            for t, dt in ((scores, torch.float32), (ids, torch.int64)):
                if t.dtype != dt or tuple(t.shape) != (b, k) or t.device != dev or not t.is_contiguous():
                    raise ValueError(f"out tensors must be contiguous [{b}, {k}] float32 / int64 on {dev}")
        with torch.cuda.device(dev):
            if out is None:
                scores = torch.empty((b, k), dtype=torch.float32, device=dev)
                ids = torch.empty((b, k), dtype=torch.int64, device=dev)
            pos = torch.empty((b, k), dtype=torch.int64, device=dev) if return_positions else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(self._lib.vqa_index_search(self._handle, q.data_ptr(), _SRC_DTYPE[q.dtype], b, int(k), scores.data_ptr(),
                                               ids.data_ptr(), pos.data_ptr() if pos is not None else None, stream),
                    "vqa_index_search")
        return scores, ids, pos

    def search_host(self, queries, k: int, *, normalize: bool = False, return_positions: bool = False):
        """The latency form (``vqa_index_search_host``): ``queries`` [B, d] float32 / float16 HOST array -> (scores [B, k] float32,
        ids [B, k] int64[, positions]) as numpy arrays, synchronous -- one library call, no torch tensor, no per-call device
        allocation, no copy operation (pinned device-mapped staging inside the handle).  ``normalize``: L2-normalise float32
        queries on the device first (the same kernel, hence the same bits, as the batch path).  The reference's own pattern:
        one question per call, limit 1 (``heavy_ranker.py:97-101``)."""
        if not self._handle.value:
            raise RuntimeError("index is closed")
        if isinstance(queries, torch.Tensor):
            # device-resident queries (the question encoder's output): host results all the same -- polled completion, no torch op
            q = queries
            if not q.is_cuda or q.device.index != self.device or q.dim() != 2 or q.shape[1] != self.d or not q.is_contiguous():
                raise ValueError(f"a tensor of queries must be a contiguous [B, {self.d}] tensor on cuda:{self.device}")
            if q.dtype not in _SRC_DTYPE:
                raise ValueError(f"queries must be float32 or float16, got {q.dtype}")
            qd, qptr = _SRC_DTYPE[q.dtype], q.data_ptr()
        else:
            q = np.ascontiguousarray(queries)
            if q.ndim != 2 or q.shape[1] != self.d:
                raise ValueError(f"queries must be a [B, {self.d}] array")
            if q.dtype == np.float32:
                qd = N.VQA_F32
            elif q.dtype == np.float16:
                qd = N.VQA_F16
            else:
                raise ValueError(f"queries must be float32 or float16, got {q.dtype}")
            qptr = q.ctypes.data
        b = int(q.shape[0])
        if b == 0:
            raise ValueError("empty query batch")
        scores = np.empty((b, k), dtype=np.float32)
        ids = np.empty((b, k), dtype=np.int64)
        pos = np.empty((b, k), dtype=np.int64) if return_positions else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.vqa_index_search_host(self._handle, qptr, qd, b, int(k), int(bool(normalize)), scores.ctypes.data,
                                                ids.ctypes.data, pos.ctypes.data if pos is not None else None, stream),
                "vqa_index_search_host")
        return (scores, ids, pos) if return_positions else (scores, ids)

    def set_timing(self, enabled, *, resume: bool = False) -> None:
        """Bracket the main scoring launch of every search with an event pair (``resume``: keep the pairs recorded so far)."""
        N.check(self._lib.vqa_index_set_timing(self._handle, (2 if resume else 1) if enabled else 0), "vqa_index_set_timing")

    def get_timing(self) -> Tuple[float, int]:
        """(sum of main scoring-kernel time in ms, launches) since the last call; waits for the last launch."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        N.check(self._lib.vqa_index_get_timing(self._handle, ctypes.byref(ms), ctypes.byref(n)), "vqa_index_get_timing")
        return ms.value, n.value

    def device_bytes(self) -> int:
        """Device memory the shard holds: rows + id vector + int8 sketch + re-scoring copy."""
        return int(self._lib.vqa_index_device_bytes(self._handle))

    def sketch_state(self) -> int:
        """-1: no sketch; 0: searches take the sketch search; n > 0: an overflow sent the next n searches to the exact scan."""
        return int(self._lib.vqa_index_sketch_state(self._handle))

    def sketch_pause(self, searches: int) -> None:
        """The next ``searches`` searches of this handle take the exact scan (0 ends a pause); same bits as the sketch search on the
        shards that have both paths (``include/vqa_retrieval.h``: vqa_index_sketch_pause, options.final_rescore)."""
        N.check(self._lib.vqa_index_sketch_pause(self._handle, int(searches)), "vqa_index_sketch_pause")

    def sketch_stats(self) -> dict:
        """Diagnostics of the last sketch search of this handle (synchronises the device; ``include/vqa_retrieval.h``)."""
        out = (ctypes.c_int64 * 8)()
        N.check(self._lib.vqa_index_sketch_stats(self._handle, out), "vqa_index_sketch_stats")
        names = ("last_scan_pairs", "largest_region", "rescored_pairs", "longest_sublist", "overflow", "overflow_earlier_tiles",
                 "region_capacity", "sublist_capacity")
        return dict(zip(names, (int(v) for v in out)))

    def sketch_tile(self, tile: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(int8 codes [256, d8], tile_info [4] = (max ||x_hi||, max ||x_lo||, 1 / scale, scale), centre mu [d8]) of one tile of
        the sketch -- what its score bound is computed from (tests restate it independently)."""
        d8 = (self.d + 127) // 128 * 128
        codes = np.empty((256, d8), dtype=np.int8)
        info = np.empty((4,), dtype=np.float32)
        mu = np.empty((d8,), dtype=np.float32)
        N.check(self._lib.vqa_index_get_sketch_tile(self._handle, int(tile), codes.ctypes.data, info.ctypes.data, mu.ctypes.data),
                "vqa_index_get_sketch_tile")
        return codes, info, mu

    def sketch_split(self, tile: int) -> Tuple[float, np.ndarray, np.ndarray, bool]:
        """(c, w [d8], beta [256], per_row): the shard's rotated centre direction w and, split form, the tile's max |w . x_lo| -- or,
        per-row form (rows collapsed onto w), the tile's max |beta| and its rows' beta = w . y (``include/vqa_retrieval.h``)."""
        d8 = (self.d + 127) // 128 * 128
        c = np.zeros((1,), dtype=np.float32)
        w = np.empty((d8,), dtype=np.float32)
        beta = np.empty((256,), dtype=np.float32)
        per_row = ctypes.c_int32(0)
        N.check(self._lib.vqa_index_get_sketch_split(self._handle, int(tile), c.ctypes.data, w.ctypes.data, beta.ctypes.data,
                                                     ctypes.addressof(per_row)), "vqa_index_get_sketch_split")
        return float(c[0]), w, beta, bool(per_row.value)

    def launch_info(self, b: int, k: int) -> N.LaunchInfo:
        info = N.LaunchInfo()
        N.check(self._lib.vqa_index_launch_info(self._handle, int(b), int(k), ctypes.byref(info)), "vqa_index_launch_info")
        return info


def merge_topk(scores: torch.Tensor, ids: torch.Tensor, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Final merge of per-shard candidates ``[R, B, k]`` (after the RCCL all-gather) -> ``[B, k_out]``.  The rank
    dimension may be strided (views into one gathered buffer); each ``[B, k]`` block must be contiguous."""
    if scores.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("scores/ids must both be [R, B, k]")
    if not scores.is_cuda:
        raise RuntimeError("merge_topk runs on the GPU only (no CPU fallback)")
    lib = N.load()
    r, b, k = (int(x) for x in scores.shape)
    dev = scores.device

    def blocks(t, dt):  # keep a rank-strided view as it is; anything else becomes a plain [R, B, k] array
        if t.dtype == dt and t.stride(2) == 1 and t.stride(1) == k and (r == 1 or t.stride(0) >= b * k):
            return t
        return t.to(dt).contiguous()
    scores = blocks(scores, torch.float32)
    ids = blocks(ids, torch.int64)
    with torch.cuda.device(dev):
        out_s = torch.empty((b, k_out), dtype=torch.float32, device=dev)
        out_i = torch.empty((b, k_out), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.vqa_merge_topk(scores.data_ptr(), ids.data_ptr(), int(scores.stride(0)) if r > 1 else 0,
                                   int(ids.stride(0)) if r > 1 else 0, r, b, k, int(k_out), out_s.data_ptr(),
                                   out_i.data_ptr(), stream), "vqa_merge_topk")
    return out_s, out_i
