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

_TORCH_DTYPE = {N.VQA_F32: torch.float32, N.VQA_F16: torch.float16}


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


class DeviceIndex:
    """One row shard of the corpus in HBM, searched by the fused MFMA scoring + top-k kernel.

    ``vectors`` [n, d]: a torch tensor (cpu or cuda) or numpy array holding either values already in the storage
    type (``float16`` for ``dtype='fp16'``) or ``float32`` embeddings, which are L2-normalised (when
    ``normalize=True``, txtai's behaviour) and converted on the device.  ``ids`` [n] int64 external ids or ``None``
    (id = ``id_base`` + row position; sqlite AUTOINCREMENT rowids start at 1, ``setup_db.py:14``).
    """

    def __init__(self, vectors, ids=None, *, id_base: int = 0, dtype="fp16", device: int = 0, normalize: bool = False,
                 borrow: bool = False):
        self._handle = ctypes.c_void_p()
        self.device = int(device)
        _require_gpu(self.device)
        self._lib = N.load()
        self.dtype = resolve_dtype(dtype)
        dev = torch.device("cuda", self.device)
        v = torch.from_numpy(np.ascontiguousarray(vectors)) if isinstance(vectors, np.ndarray) else vectors
        if not isinstance(v, torch.Tensor) or v.dim() != 2:
            raise ValueError("vectors must be a 2-D torch tensor or numpy array [n, d]")
        n, d = int(v.shape[0]), int(v.shape[1])
        stored_t = _TORCH_DTYPE.get(self.dtype)
        if self.dtype == N.VQA_FP8_E4M3:
            stored_t = torch.uint8
        with torch.cuda.device(dev):
            if v.dtype == stored_t and not normalize:
                rows = v.contiguous()
            elif v.dtype == torch.float32:
                src = v.to(dev).contiguous()
                rows = torch.empty((n, d), dtype=stored_t, device=dev)
                if n:
                    stream = torch.cuda.current_stream(dev).cuda_stream
                    N.check(self._lib.vqa_normalize_convert(src.data_ptr(), n, d, int(bool(normalize)), self.dtype,
                                                            rows.data_ptr(), stream), "vqa_normalize_convert")
                    torch.cuda.current_stream(dev).synchronize()
                del src
            else:
                raise ValueError(f"vectors of dtype {v.dtype} cannot back a {dtype} index "
                                 f"(pass float32 embeddings or values already stored as {stored_t})")
            ids_t = None
            if ids is not None:
                ids_t = torch.as_tensor(np.asarray(ids) if not isinstance(ids, torch.Tensor) else ids).to(torch.int64)
                if ids_t.numel() != n:
                    raise ValueError(f"{ids_t.numel()} ids for {n} rows")
                ids_t = ids_t.contiguous()
            # by default the library copies the rows into its own allocation and torch's buffer is released;
            # borrow=True (device rows, d % 64 == 0) shares the caller's tensor instead: no second copy of a
            # 15-123 GB shard.  The tensor is kept alive by this object.
            flags = 0
            self._borrowed = None
            if borrow:
                if not rows.is_cuda or rows.device.index != self.device:
                    raise ValueError("borrow=True needs the rows on the index's device")
                flags = N.VQA_ROWS_BORROW
                self._borrowed = rows
            N.check(self._lib.vqa_index_create(ctypes.byref(self._handle), self.device, n, d, self.dtype,
                                               rows.data_ptr() if n else None,
                                               ids_t.data_ptr() if ids_t is not None and n else None, int(id_base), flags),
                    "vqa_index_create")
        self.n, self.d = n, d
        self.id_base = int(id_base)

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
    def search(self, queries: torch.Tensor, k: int, *, return_positions: bool = False
               ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """``queries`` [B, d] fp32/fp16 cuda tensor -> (scores [B, k] fp32, ids [B, k] int64[, positions]), best first,
        ties by row position; slots beyond the shard's row count hold (-inf, -1).  Asynchronous on the current stream."""
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
        with torch.cuda.device(dev):
            scores = torch.empty((b, k), dtype=torch.float32, device=dev)
            ids = torch.empty((b, k), dtype=torch.int64, device=dev)
            pos = torch.empty((b, k), dtype=torch.int64, device=dev) if return_positions else None
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(self._lib.vqa_index_search(self._handle, q.data_ptr(), N.VQA_F32 if q.dtype == torch.float32 else N.VQA_F16,
                                               b, int(k), scores.data_ptr(), ids.data_ptr(),
                                               pos.data_ptr() if pos is not None else None, stream), "vqa_index_search")
        return scores, ids, pos

    def set_timing(self, enabled: bool) -> None:
        N.check(self._lib.vqa_index_set_timing(self._handle, int(bool(enabled))), "vqa_index_set_timing")

    def get_timing(self) -> Tuple[float, int]:
        """(sum of main scoring-kernel time in ms, launches) since the last call; waits for the last launch."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        N.check(self._lib.vqa_index_get_timing(self._handle, ctypes.byref(ms), ctypes.byref(n)), "vqa_index_get_timing")
        return ms.value, n.value

    def launch_info(self, b: int, k: int) -> N.LaunchInfo:
        info = N.LaunchInfo()
        N.check(self._lib.vqa_index_launch_info(self._handle, int(b), int(k), ctypes.byref(info)), "vqa_index_launch_info")
        return info


def merge_topk(scores: torch.Tensor, ids: torch.Tensor, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Final merge of per-shard candidates ``[R, B, k]`` (after the RCCL all-gather) -> ``[B, k_out]``."""
    if scores.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("scores/ids must both be [R, B, k]")
    if not scores.is_cuda:
        raise RuntimeError("merge_topk runs on the GPU only (no CPU fallback)")
    lib = N.load()
    r, b, k = (int(x) for x in scores.shape)
    dev = scores.device
    scores = scores.contiguous().to(torch.float32)
    ids = ids.contiguous().to(torch.int64)
    with torch.cuda.device(dev):
        out_s = torch.empty((b, k_out), dtype=torch.float32, device=dev)
        out_i = torch.empty((b, k_out), dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        N.check(lib.vqa_merge_topk(scores.data_ptr(), ids.data_ptr(), r, b, k, int(k_out), out_s.data_ptr(), out_i.data_ptr(),
                                   stream), "vqa_merge_topk")
    return out_s, out_i
