"""ctypes binding of ``libvqa_retrieval.so`` (the C ABI declared in ``include/vqa_retrieval.h``).

There is no CPU fallback: if the shared library is missing or a call fails this module raises.  PyTorch is used
by the callers only for device memory and streams; no torch type crosses this boundary -- only raw device
pointers (``tensor.data_ptr()``) and the ``hipStream_t`` of ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VQA_LIB") or os.path.join(_HERE, "lib", "libvqa_retrieval.so")  # VQA_LIB: dev override

VQA_VERSION = 113  # include/vqa_retrieval.h: the ABI these bindings were written against
VQA_F32, VQA_F16, VQA_FP8_E4M3 = 0, 1, 2
VQA_INDEX_HAS_IDS = 1
VQA_INDEX_SKETCH = 2
VQA_INDEX_RESCORE_ROWS = 4
VQA_QUERY_TILE = 256
VQA_POOL_CLS, VQA_POOL_MEAN = 0, 1
VQA_POS_ROBERTA, VQA_POS_ABSOLUTE = 0, 1

DTYPE_NAMES = {"fp32": VQA_F32, "f32": VQA_F32, "float32": VQA_F32, "fp16": VQA_F16, "f16": VQA_F16,
               "float16": VQA_F16, "fp8": VQA_FP8_E4M3, "fp8_e4m3": VQA_FP8_E4M3, "e4m3": VQA_FP8_E4M3}
DTYPE_BYTES = {VQA_F32: 4, VQA_F16: 2, VQA_FP8_E4M3: 1}

# every symbol include/vqa_retrieval.h declares (tests/test_capi_symbols.py checks the two lists agree)
EXPORTS = (
    "vqa_version", "vqa_last_error", "vqa_index_create", "vqa_index_options_init", "vqa_index_create_ex", "vqa_index_set_rows", "vqa_index_get_rows", "vqa_index_destroy", "vqa_index_size", "vqa_index_dim",
    "vqa_index_dtype", "vqa_index_device_bytes", "vqa_index_sketch_state", "vqa_index_sketch_pause", "vqa_index_sketch_stats", "vqa_index_get_sketch_tile", "vqa_index_get_sketch_split", "vqa_index_search", "vqa_index_search_host", "vqa_merge_topk", "vqa_index_launch_info", "vqa_measure_read_stream", "vqa_index_set_timing",
    "vqa_index_get_timing", "vqa_encoder_create", "vqa_encoder_options_init", "vqa_encoder_create_ex",
    "vqa_encoder_destroy", "vqa_encoder_forward", "vqa_encoder_forward_host", "vqa_encoder_forward_hidden", "vqa_normalize_convert",
)


class VqaError(RuntimeError):
    """A libvqa_retrieval call returned a negative status."""


class LaunchInfo(ctypes.Structure):
    _fields_ = [("grid", ctypes.c_int32), ("block", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
                ("rows_per_tile", ctypes.c_int32), ("rows_per_launch", ctypes.c_int64),
                ("bytes_per_launch", ctypes.c_int64), ("flops_per_launch", ctypes.c_int64),
                ("seed_grid", ctypes.c_int32), ("seed_tiles", ctypes.c_int32), ("first_stage_rows", ctypes.c_int64),
                ("sketch_scan", ctypes.c_int32), ("levels", ctypes.c_int32), ("scan_kernel", ctypes.c_int32)]


class IndexOptions(ctypes.Structure):
    """``vqa_index_options`` (include/vqa_retrieval.h): every knob of an index handle; ``vqa_index_options_init`` fills the defaults."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in (
        "two_pass", "wide_k", "seed_mult", "seed_div", "stage_min_tiles", "stage_pct", "f16_loop", "sketch_cascade", "sketch_rotate",
        "sketch_center", "sketch_split", "sketch_per_row", "sketch_ring_stages", "sketch_mid_k", "sketch_mid_min_tiles", "sketch_mid_pct",
        "sketch_pre_k", "sketch_cooldown")] + [("sketch_profit", ctypes.c_float), ("rescore_copy", ctypes.c_int32),
                                               ("poison_workspace", ctypes.c_int32), ("one_launch", ctypes.c_int32),
                                               ("sketch_regq", ctypes.c_int32), ("final_rescore", ctypes.c_int32)]


class EncoderOptions(ctypes.Structure):
    """``vqa_encoder_options``."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("fold_layernorm", ctypes.c_int32), ("first_rows", ctypes.c_int32),
                ("graphs", ctypes.c_int32), ("latency_path", ctypes.c_int32), ("persistent", ctypes.c_int32), ("persistent_grid", ctypes.c_int32)]


class EncoderConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("vocab_size", "hidden", "layers", "heads", "ffn", "max_pos", "type_vocab",
                                              "pad_id")] + [("ln_eps", ctypes.c_float), ("position_ids", ctypes.c_int32)]


_F = ctypes.POINTER(ctypes.c_float)


class EncoderLayerWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b", "w1",
                                               "b1", "w2", "b2", "ln2_g", "ln2_b")]


class EncoderWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("word_emb", "pos_emb", "type_emb", "emb_ln_g", "emb_ln_b")] + [
        ("layer", ctypes.POINTER(EncoderLayerWeights))]


_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises ``FileNotFoundError`` with the build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -m vietnamese_qa_system_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for the retrieval path.")
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    lib.vqa_version.restype = c.c_int
    got = lib.vqa_version()
    if got != VQA_VERSION:
        # a stale library (git-ignored, copied from another checkout) would be called with shifted arguments: a GPU memory
        # fault instead of an error
        raise RuntimeError(f"{LIB_PATH} reports ABI version {got} but these bindings expect {VQA_VERSION}: rebuild it with "
                           "`python -m vietnamese_qa_system_amd.build --force`")
    lib.vqa_last_error.restype = c.c_char_p
    lib.vqa_index_create.argtypes = [c.POINTER(c.c_void_p), c.c_int, c.c_int64, c.c_int32, c.c_int32, c.c_void_p,
                                     c.c_int32, c.c_void_p, c.c_int64, c.c_uint32]
    lib.vqa_index_options_init.argtypes = [c.POINTER(IndexOptions)]
    lib.vqa_index_options_init.restype = None
    lib.vqa_index_create_ex.argtypes = [c.POINTER(c.c_void_p), c.c_int, c.c_int64, c.c_int32, c.c_int32, c.c_void_p,
                                        c.c_int32, c.c_void_p, c.c_int64, c.POINTER(IndexOptions)]
    lib.vqa_encoder_options_init.argtypes = [c.POINTER(EncoderOptions)]
    lib.vqa_encoder_options_init.restype = None
    lib.vqa_encoder_create_ex.argtypes = [c.POINTER(c.c_void_p), c.c_int, c.POINTER(EncoderConfig), c.POINTER(EncoderWeights), c.c_int32,
                                          c.POINTER(EncoderOptions)]
    lib.vqa_index_set_rows.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_void_p, c.c_int32, c.c_void_p]
    lib.vqa_index_get_rows.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_void_p, c.c_void_p]
    lib.vqa_index_destroy.argtypes = [c.c_void_p]
    lib.vqa_index_destroy.restype = None
    lib.vqa_index_size.argtypes = [c.c_void_p]
    lib.vqa_index_size.restype = c.c_int64
    lib.vqa_index_dim.argtypes = [c.c_void_p]
    lib.vqa_index_dim.restype = c.c_int32
    lib.vqa_index_dtype.argtypes = [c.c_void_p]
    lib.vqa_index_dtype.restype = c.c_int32
    lib.vqa_index_device_bytes.argtypes = [c.c_void_p]
    lib.vqa_index_device_bytes.restype = c.c_int64
    lib.vqa_index_sketch_state.argtypes = [c.c_void_p]
    lib.vqa_index_sketch_state.restype = c.c_int32
    lib.vqa_index_sketch_pause.argtypes = [c.c_void_p, c.c_int32]
    lib.vqa_index_sketch_pause.restype = c.c_int
    lib.vqa_index_sketch_stats.argtypes = [c.c_void_p, c.POINTER(c.c_int64)]
    lib.vqa_index_get_sketch_tile.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.vqa_index_get_sketch_split.argtypes = [c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.vqa_index_search.argtypes = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_void_p,
                                     c.c_void_p, c.c_void_p]
    lib.vqa_index_search_host.argtypes = [c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_void_p,
                                          c.c_void_p, c.c_void_p]
    lib.vqa_merge_topk.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.c_int64, c.c_int32, c.c_int32, c.c_int32, c.c_int32,
                                   c.c_void_p, c.c_void_p, c.c_void_p]
    lib.vqa_index_launch_info.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.POINTER(LaunchInfo)]
    lib.vqa_index_set_timing.argtypes = [c.c_void_p, c.c_int32]
    lib.vqa_index_get_timing.argtypes = [c.c_void_p, c.POINTER(c.c_double), c.POINTER(c.c_int64)]
    lib.vqa_measure_read_stream.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.POINTER(c.c_double), c.c_void_p]
    lib.vqa_encoder_create.argtypes = [c.POINTER(c.c_void_p), c.c_int, c.POINTER(EncoderConfig),
                                       c.POINTER(EncoderWeights), c.c_int32]
    lib.vqa_encoder_destroy.argtypes = [c.c_void_p]
    lib.vqa_encoder_destroy.restype = None
    lib.vqa_encoder_forward.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_int32,
                                        c.c_void_p, c.c_void_p]
    lib.vqa_encoder_forward_host.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_void_p]
    lib.vqa_encoder_forward_hidden.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32, c.c_void_p,
                                               c.c_void_p]
    lib.vqa_normalize_convert.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_int32, c.c_int32, c.c_void_p, c.c_void_p]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is c.c_int and name not in ("vqa_version", "vqa_index_options_init", "vqa_encoder_options_init"):
            fn.restype = c.c_int
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    """Raise ``VqaError`` (ValueError-like for bad arguments) when a call failed."""
    if status == 0:
        return
    msg = load().vqa_last_error().decode("utf-8", "replace")
    if status == -1:
        raise ValueError(f"{what}: {msg}")
    if status == -3:
        raise MemoryError(f"{what}: {msg}")
    raise VqaError(f"{what} failed ({status}): {msg}")


def index_options(flags: int = 0, **kw) -> IndexOptions:
    """The library's defaults with ``kw`` laid over them (unknown names raise)."""
    o = IndexOptions()
    load().vqa_index_options_init(ctypes.byref(o))
    o.flags = int(flags)
    known = {n for n, _ in IndexOptions._fields_} - {"struct_size", "flags"}
    for k, v in kw.items():
        if k not in known:
            raise ValueError(f"unknown index option {k!r} (known: {sorted(known)})")
        setattr(o, k, float(v) if k == "sketch_profit" else int(v))
    return o


def encoder_options(**kw) -> EncoderOptions:
    o = EncoderOptions()
    load().vqa_encoder_options_init(ctypes.byref(o))
    known = {n for n, _ in EncoderOptions._fields_} - {"struct_size"}
    for k, v in kw.items():
        if k not in known:
            raise ValueError(f"unknown encoder option {k!r} (known: {sorted(known)})")
        setattr(o, k, int(v))
    return o
