"""Host side of the question encoder: weight hand-off to ``vqa_encoder_create`` and the text-encoder hook of
:class:`~vietnamese_qa_system_amd.embeddings.Embeddings`.

Counterpart of the transformer forward txtai runs per query (reference: model chosen by ``path=`` at
``inference_pipeline/db_utils/heavy_ranker.py:80,83``; DPR form ``q_model(input_ids).pooler_output`` at
``src/test.py:84-86``).  Weights use HF ``RobertaModel`` state-dict names (no ``roberta.`` prefix), fp32; tokenisation
stays on the host and is pluggable (PhoBERT needs word segmentation + BPE files that are not available offline).
"""
from __future__ import annotations

import ctypes
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N

POOLING = {"cls": N.VQA_POOL_CLS, "mean": N.VQA_POOL_MEAN}

PHOBERT_BASE = dict(vocab_size=64001, hidden=768, layers=12, heads=12, ffn=3072, max_pos=258, type_vocab=1, pad_id=1,
                    ln_eps=1e-5)
# the two models the reference's retriever loads (inference_pipeline/db_utils/heavy_ranker.py:80,83), by their published configs:
# paraphrase-multilingual-mpnet-base-v2 is an XLM-RoBERTa base, paraphrase-multilingual-MiniLM-L12-v2 a BERT (absolute position
# ids, two token types) with 12 heads of 32; both are mean-pooled by sentence-transformers
XLMR_BASE = dict(vocab_size=250002, hidden=768, layers=12, heads=12, ffn=3072, max_pos=514, type_vocab=1, pad_id=1, ln_eps=1e-5)
MINILM_L12 = dict(vocab_size=250037, hidden=384, layers=12, heads=12, ffn=1536, max_pos=512, type_vocab=2, pad_id=0, ln_eps=1e-12,
                  position_ids="absolute")
POSITION_IDS = {"roberta": N.VQA_POS_ROBERTA, "absolute": N.VQA_POS_ABSOLUTE}
# process-wide defaults laid under QuestionEncoder(options=...): vqa_encoder_options by field name (fold_layernorm, first_rows, graphs);
# empty in production (tests / A/B scripts set entries: explicit Python state, no environment variable is read)
DEFAULT_OPTIONS: dict = {}

_LAYER_FIELDS = (
    ("wq", "attention.self.query.weight"), ("bq", "attention.self.query.bias"),
    ("wk", "attention.self.key.weight"), ("bk", "attention.self.key.bias"),
    ("wv", "attention.self.value.weight"), ("bv", "attention.self.value.bias"),
    ("wo", "attention.output.dense.weight"), ("bo", "attention.output.dense.bias"),
    ("ln1_g", "attention.output.LayerNorm.weight"), ("ln1_b", "attention.output.LayerNorm.bias"),
    ("w1", "intermediate.dense.weight"), ("b1", "intermediate.dense.bias"),
    ("w2", "output.dense.weight"), ("b2", "output.dense.bias"),
    ("ln2_g", "output.LayerNorm.weight"), ("ln2_b", "output.LayerNorm.bias"),
)
_TOP_FIELDS = (("word_emb", "embeddings.word_embeddings.weight"), ("pos_emb", "embeddings.position_embeddings.weight"),
               ("type_emb", "embeddings.token_type_embeddings.weight"), ("emb_ln_g", "embeddings.LayerNorm.weight"),
               ("emb_ln_b", "embeddings.LayerNorm.bias"))


class QuestionEncoder:
    """RoBERTa / PhoBERT-base-shaped encoder resident on one MI355X.

    ``weights``: mapping HF state-dict name -> float32 array (numpy or torch, host or device).  ``config`` keys:
    vocab_size, hidden, layers, heads, ffn, max_pos, type_vocab, pad_id, ln_eps and, optionally, position_ids ("roberta":
    pad-offset ids, the default; "absolute": BERT's 0 .. L - 1).  ``max_tokens`` bounds B * L of one call.
    """

    def __init__(self, weights: Dict[str, object], config: dict, *, device: int = 0, max_tokens: int = 1024 * 32,
                 options: Optional[dict] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("QuestionEncoder needs an MI355X (gfx950); there is no CPU fallback")
        self._lib = N.load()
        self._handle = ctypes.c_void_p()
        self.device = int(device)
        self.config = dict(config)
        self.max_tokens = int(max_tokens)
        self.pooling: Optional[str] = None      # set by from_pretrained (sentence-transformers modules.json)
        self.normalize: Optional[bool] = None
        self.model_dir: Optional[str] = None
        keep = []  # host copies kept alive until vqa_encoder_create returns

        def ptr(name: str, shape: Tuple[int, ...]) -> int:
            if name not in weights:
                raise KeyError(f"missing weight {name!r}")
            w = weights[name]
            if isinstance(w, torch.Tensor):
                w = w.detach().to(torch.float32).contiguous()
                if tuple(w.shape) != shape:
                    raise ValueError(f"{name}: shape {tuple(w.shape)}, expected {shape}")
                keep.append(w)
                return w.data_ptr()
            a = np.ascontiguousarray(w, dtype=np.float32)
            if a.shape != shape:
                raise ValueError(f"{name}: shape {a.shape}, expected {shape}")
            keep.append(a)
            return a.ctypes.data

        c = self.config
        h, f = int(c["hidden"]), int(c["ffn"])
        pos_mode = str(c.get("position_ids", "roberta"))
        if pos_mode not in POSITION_IDS:
            raise ValueError(f"position_ids must be one of {sorted(POSITION_IDS)}")
        cfg = N.EncoderConfig(int(c["vocab_size"]), h, int(c["layers"]), int(c["heads"]), f, int(c["max_pos"]),
                              int(c["type_vocab"]), int(c["pad_id"]), float(c["ln_eps"]), POSITION_IDS[pos_mode])
        shapes = {"word_emb": (c["vocab_size"], h), "pos_emb": (c["max_pos"], h), "type_emb": (c["type_vocab"], h),
                  "emb_ln_g": (h,), "emb_ln_b": (h,)}
        lshapes = {"wq": (h, h), "wk": (h, h), "wv": (h, h), "wo": (h, h), "w1": (f, h), "w2": (h, f), "bq": (h,), "bk": (h,),
                   "bv": (h,), "bo": (h,), "b1": (f,), "b2": (h,), "ln1_g": (h,), "ln1_b": (h,), "ln2_g": (h,), "ln2_b": (h,)}
        layers = (N.EncoderLayerWeights * int(c["layers"]))()
        for i in range(int(c["layers"])):
            for field, name in _LAYER_FIELDS:
                setattr(layers[i], field, ptr(f"encoder.layer.{i}.{name}", lshapes[field]))
        top = N.EncoderWeights()
        for field, name in _TOP_FIELDS:
            setattr(top, field, ptr(name, shapes[field]))
        top.layer = ctypes.cast(layers, ctypes.POINTER(N.EncoderLayerWeights))
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            opts = dict(DEFAULT_OPTIONS)
            opts.update(options or {})
            self.options = N.encoder_options(**opts)
            N.check(self._lib.vqa_encoder_create_ex(ctypes.byref(self._handle), self.device, ctypes.byref(cfg), ctypes.byref(top),
                                                    self.max_tokens, ctypes.byref(self.options)), "vqa_encoder_create_ex")
        del keep

    @classmethod
    def from_pretrained(cls, model_dir: str, *, device: int = 0, max_tokens: int = 1024 * 32) -> "QuestionEncoder":
        """The encoder of a LOCAL Hugging Face / sentence-transformers model directory (``config.json`` + ``model.safetensors`` or
        ``pytorch_model.bin``; what the reference names by ``path=`` at ``heavy_ranker.py:80,83``).  The returned object carries
        ``.pooling`` / ``.normalize`` as the directory's ``modules.json`` prescribes them (``None`` when it says nothing)."""
        from .pretrained import load_pretrained
        weights, cfg, pooling, normalize = load_pretrained(model_dir)
        if int(cfg["max_pos"]) < 2:
            raise ValueError("max_position_embeddings < 2")
        enc = cls(weights, cfg, device=device, max_tokens=max_tokens)
        enc.pooling, enc.normalize, enc.model_dir = pooling, normalize, model_dir
        return enc

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.vqa_encoder_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, input_ids, attention_mask, *, pooling: str = "cls", normalize: bool = True,
                real_tokens: Optional[int] = None) -> torch.Tensor:
        """``input_ids`` / ``attention_mask`` [B, L] integers -> [B, hidden] fp32 cuda tensor (pooled, L2-normalised).

        Ragged batches: with a right-padded mask only the real tokens are computed (sequence packing inside the library).
        Host-side masks (what a tokenizer returns) are checked and counted here; for a device-resident mask pass
        ``real_tokens`` = its number of set entries (``0`` / ``None``: compute every position)."""
        if not self._handle.value:
            raise RuntimeError("encoder is closed")
        if pooling not in POOLING:
            raise ValueError(f"pooling must be one of {sorted(POOLING)}")
        dev = torch.device("cuda", self.device)
        ids = torch.as_tensor(np.asarray(input_ids) if not isinstance(input_ids, torch.Tensor) else input_ids)
        mask = torch.as_tensor(np.asarray(attention_mask) if not isinstance(attention_mask, torch.Tensor) else attention_mask)
        if ids.dim() != 2 or mask.shape != ids.shape:
            raise ValueError("input_ids and attention_mask must both be [B, L]")
        if not ids.is_cuda and ids.numel():
            # host-side ids (what a tokenizer returns) are range-checked here, before anything is launched; device-resident
            # ids are clamped by the kernel and reported by the next forward call (include/vqa_retrieval.h)
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= int(self.config["vocab_size"]):
                raise ValueError(f"token ids span [{lo}, {hi}] but the embedding table has {self.config['vocab_size']} rows "
                                 "(tokenizer / vocabulary mismatch?)")
        if real_tokens is None:
            real_tokens = 0
            if not mask.is_cuda and mask.numel():
                m = mask != 0
                right_padded = bool(m[:, 0].all()) and bool((m[:, :-1] >= m[:, 1:]).all())
                real_tokens = int(m.sum()) if right_padded else 0
        ids = ids.to(dev, dtype=torch.int32).contiguous()
        mask = mask.to(dev, dtype=torch.int32).contiguous()
        b, l = int(ids.shape[0]), int(ids.shape[1])
        with torch.cuda.device(dev):
            out = torch.empty((b, int(self.config["hidden"])), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(self._lib.vqa_encoder_forward(self._handle, ids.data_ptr(), mask.data_ptr(), b, l, int(real_tokens),
                                                  POOLING[pooling], int(bool(normalize)), out.data_ptr(), stream),
                    "vqa_encoder_forward")
        return out

    __call__ = forward

    def forward_host(self, input_ids: np.ndarray, attention_mask: np.ndarray, out: torch.Tensor, *, pooling: str = "cls",
                     normalize: bool = True) -> torch.Tensor:
        """The forward for HOST token ids / masks (``vqa_encoder_forward_host``): [B, L] integer numpy arrays in, ``out`` [>= B, hidden]
        fp32 cuda tensor filled (its first B rows) -- one library call, no torch tensor created, no copy operation; asynchronous on the
        current stream.  The text form of the reference's one-question call (``heavy_ranker.py:98``) pairs it with
        ``DeviceIndex.search_host(out[:B], ...)``."""
        if not self._handle.value:
            raise RuntimeError("encoder is closed")
        if pooling not in POOLING:
            raise ValueError(f"pooling must be one of {sorted(POOLING)}")
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        mask = np.ascontiguousarray(attention_mask, dtype=np.int32)
        if ids.ndim != 2 or mask.shape != ids.shape:
            raise ValueError("input_ids and attention_mask must both be [B, L]")
        b, l = int(ids.shape[0]), int(ids.shape[1])
        h = int(self.config["hidden"])
        if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.device.index != self.device or out.dtype != torch.float32 or
                out.dim() != 2 or out.shape[0] < b or out.shape[1] != h or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous [>= {b}, {h}] float32 tensor on cuda:{self.device}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        N.check(self._lib.vqa_encoder_forward_host(self._handle, ids.ctypes.data, mask.ctypes.data, b, l, POOLING[pooling],
                                                   int(bool(normalize)), out.data_ptr(), stream), "vqa_encoder_forward_host")
        return out[:b]

    def hidden_states(self, input_ids, attention_mask, n_layers: Optional[int] = None, *,
                      real_tokens: Optional[int] = None) -> torch.Tensor:
        """Hidden state of every position after ``n_layers`` layers (``None``: all = HF ``last_hidden_state``; 0: the embedding
        output) as a [B, L, hidden] fp32 cuda tensor -- HF ``output_hidden_states[n_layers]``.  Same kernels as :meth:`forward`
        of this shape; with a right-padded host mask (or ``real_tokens``) the packed form runs and padding positions are zeros."""
        if not self._handle.value:
            raise RuntimeError("encoder is closed")
        dev = torch.device("cuda", self.device)
        ids = torch.as_tensor(np.asarray(input_ids) if not isinstance(input_ids, torch.Tensor) else input_ids)
        mask = torch.as_tensor(np.asarray(attention_mask) if not isinstance(attention_mask, torch.Tensor) else attention_mask)
        if ids.dim() != 2 or mask.shape != ids.shape:
            raise ValueError("input_ids and attention_mask must both be [B, L]")
        if real_tokens is None:
            real_tokens = 0
            if not mask.is_cuda and mask.numel():
                m = mask != 0
                right_padded = bool(m[:, 0].all()) and bool((m[:, :-1] >= m[:, 1:]).all())
                real_tokens = int(m.sum()) if right_padded else 0
        n_layers = int(self.config["layers"]) if n_layers is None else int(n_layers)
        ids = ids.to(dev, dtype=torch.int32).contiguous()
        mask = mask.to(dev, dtype=torch.int32).contiguous()
        b, l = int(ids.shape[0]), int(ids.shape[1])
        with torch.cuda.device(dev):
            out = torch.empty((b, l, int(self.config["hidden"])), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            N.check(self._lib.vqa_encoder_forward_hidden(self._handle, ids.data_ptr(), mask.data_ptr(), b, l, int(real_tokens),
                                                         n_layers, out.data_ptr(), stream), "vqa_encoder_forward_hidden")
        return out


class TextEncoder:
    """``list[str] -> [B, hidden]`` hook for ``Embeddings(encoder=...)``: a host tokenizer + the HIP encoder.

    ``tokenizer(texts) -> (input_ids [B, L], attention_mask [B, L])`` (lists, numpy or torch); any HF tokenizer wrapped
    as ``lambda t: (lambda e: (e["input_ids"], e["attention_mask"]))(tok(t, padding=True, truncation=True,
    max_length=128, return_tensors="np"))`` fits.  ``batch_size`` texts are tokenised together and go through the encoder
    in ONE forward when the workspace (``max_tokens``) holds them: 1024 questions of 32 tokens run the GEMMs at 1.5x the
    rate of four batches of 256 (``bench.py: end_to_end.super_batch``), so ``Embeddings.batchsearch`` with many text queries
    is a throughput path."""

    def __init__(self, tokenizer: Callable[[List[str]], Tuple[Sequence, Sequence]], encoder: QuestionEncoder, *,
                 pooling: str = "mean", normalize: bool = True, batch_size: int = 1024):
        self.tokenizer, self.encoder = tokenizer, encoder
        self.pooling, self.normalize, self.batch_size = pooling, normalize, batch_size

    def __call__(self, texts: List[str]) -> torch.Tensor:
        outs = []
        for c0 in range(0, len(texts), self.batch_size):
            ids, mask = self.tokenizer(list(texts[c0:c0 + self.batch_size]))
            ids = np.asarray(ids) if not isinstance(ids, torch.Tensor) else ids
            mask = np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask
            if ids.ndim != 2:
                raise ValueError("the tokenizer must return [B, L] input_ids")
            b, l = int(ids.shape[0]), int(ids.shape[1])
            # a tokenised batch of B x L tokens goes through the encoder in slices of at most max_tokens tokens
            rows = max(1, self.encoder.max_tokens // max(l, 1))
            if l > self.encoder.max_tokens:
                raise ValueError(f"sequence length {l} exceeds the encoder workspace of {self.encoder.max_tokens} tokens")
            for r0 in range(0, b, rows):
                outs.append(self.encoder.forward(ids[r0:r0 + rows], mask[r0:r0 + rows], pooling=self.pooling,
                                                 normalize=self.normalize))
        if not outs:
            return torch.zeros((0, int(self.encoder.config["hidden"])), dtype=torch.float32,
                               device=torch.device("cuda", self.encoder.device))
        return torch.cat(outs) if len(outs) != 1 else outs[0]
