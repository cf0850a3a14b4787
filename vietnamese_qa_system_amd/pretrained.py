"""Reads a LOCAL Hugging Face / sentence-transformers model directory into the HIP question encoder.

The reference selects its encoder by ``path=`` (``inference_pipeline/db_utils/heavy_ranker.py:78-83``:
``Embeddings(hybrid=True, content=True, path="sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2")``) and txtai /
sentence-transformers resolve that name to a directory of this layout (there is no network here: the directory must already be
on disk -- a hub snapshot, ``save_pretrained`` output, or ``SentenceTransformer.save``):

    config.json                      HF model config (model_type, hidden_size, num_hidden_layers, ...)
    model.safetensors | pytorch_model.bin
    modules.json                     sentence-transformers only: [Transformer, Pooling, (Normalize)]
    1_Pooling/config.json            sentence-transformers only: pooling_mode_cls_token / pooling_mode_mean_tokens

Nothing here imports ``transformers`` or ``safetensors``: the safetensors container is an 8-byte little-endian header length, a
JSON header {name: {dtype, shape, data_offsets}} and the raw tensors; ``pytorch_model.bin`` goes through ``torch.load``.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Optional, Tuple

import numpy as np

_ST_DTYPES = {"F64": np.float64, "F32": np.float32, "F16": np.float16, "I64": np.int64, "I32": np.int32, "I16": np.int16,
              "I8": np.int8, "U8": np.uint8, "BOOL": np.bool_}

# model_type -> position-id rule of include/vqa_retrieval.h (VQA_POS_*): RoBERTa-family checkpoints offset positions by the pad id
_ROBERTA_TYPES = {"roberta", "xlm-roberta", "camembert", "xlm_roberta"}
_BERT_TYPES = {"bert", "dpr"}
# leading components of state-dict names to drop, longest first: task heads wrap the encoder as `roberta.` / `bert.`, DPR as
# `question_encoder.bert_model.`, sentence-transformers' in-memory module as `0.auto_model.`
_PREFIXES = ("question_encoder.bert_model.", "ctx_encoder.bert_model.", "0.auto_model.", "auto_model.", "xlm_roberta.", "roberta.",
             "bert.", "model.")


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """name -> array of a ``.safetensors`` file (bf16 tensors come back as float32)."""
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
        (n,) = struct.unpack("<Q", head)
        if n <= 0 or n > 100 << 20:
            raise ValueError(f"{path}: implausible safetensors header length {n}")
        header = json.loads(f.read(n).decode("utf-8"))
        base = 8 + n
        out = {}
        for name, meta in header.items():
            if name == "__metadata__":
                continue
            lo, hi = meta["data_offsets"]
            shape = tuple(int(x) for x in meta["shape"])
            f.seek(base + lo)
            raw = f.read(hi - lo)
            if len(raw) != hi - lo:
                raise ValueError(f"{path}: tensor {name!r} runs past the end of the file")
            dt = meta["dtype"]
            if dt == "BF16":  # upper half of an fp32
                a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
            elif dt in _ST_DTYPES:
                a = np.frombuffer(raw, dtype=np.dtype(_ST_DTYPES[dt]).newbyteorder("<"))
            else:
                raise ValueError(f"{path}: tensor {name!r} has unsupported dtype {dt}")
            if a.size != int(np.prod(shape, dtype=np.int64)):
                raise ValueError(f"{path}: tensor {name!r}: {a.size} elements for shape {shape}")
            out[name] = a.reshape(shape)
        return out


def _read_state_dict(model_dir: str) -> Dict[str, np.ndarray]:
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        return read_safetensors(st)
    pt = os.path.join(model_dir, "pytorch_model.bin")
    if os.path.exists(pt):
        import torch
        sd = torch.load(pt, map_location="cpu", weights_only=True)
        return {k: v.to(torch.float32).numpy() if v.is_floating_point() else v.numpy() for k, v in sd.items()}
    raise FileNotFoundError(f"{model_dir}: neither model.safetensors nor pytorch_model.bin (sharded checkpoints are not read)")


def _strip(name: str) -> str:
    for p in _PREFIXES:
        if name.startswith(p):
            return name[len(p):]
    return name


def encoder_config_from_hf(cfg: dict) -> dict:
    """HF ``config.json`` -> the config dict of :class:`~vietnamese_qa_system_amd.encoder.QuestionEncoder`."""
    mt = str(cfg.get("model_type", "")).lower()
    if mt in _ROBERTA_TYPES:
        pos = "roberta"
    elif mt in _BERT_TYPES:
        pos = "absolute"
    else:
        raise ValueError(f"model_type {mt!r} is not a BERT / RoBERTa-family encoder (supported: {sorted(_ROBERTA_TYPES | _BERT_TYPES)})")
    act = cfg.get("hidden_act", "gelu")
    if act != "gelu":
        raise ValueError(f"hidden_act {act!r}: the encoder computes the erf GELU only")
    if cfg.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError("relative position embeddings are not supported")
    for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "max_position_embeddings"):
        if k not in cfg:
            raise ValueError(f"config.json lacks {k!r}")
    default_pad = 1 if pos == "roberta" else 0
    pad = cfg.get("pad_token_id")
    return dict(vocab_size=int(cfg["vocab_size"]), hidden=int(cfg["hidden_size"]), layers=int(cfg["num_hidden_layers"]),
                heads=int(cfg["num_attention_heads"]), ffn=int(cfg["intermediate_size"]), max_pos=int(cfg["max_position_embeddings"]),
                type_vocab=int(cfg.get("type_vocab_size", 1)), pad_id=int(default_pad if pad is None else pad),
                ln_eps=float(cfg.get("layer_norm_eps", 1e-12)), position_ids=pos)


def _sentence_transformers_layout(model_dir: str) -> Tuple[str, Optional[str], Optional[bool]]:
    """(directory of the transformer files, pooling or None, normalize or None) from ``modules.json`` when there is one."""
    mj = os.path.join(model_dir, "modules.json")
    if not os.path.exists(mj):
        return model_dir, None, None
    with open(mj) as f:
        modules = json.load(f)
    tdir, pooling, normalize = model_dir, None, False
    for m in modules:
        kind = str(m.get("type", "")).rsplit(".", 1)[-1]
        sub = os.path.join(model_dir, m.get("path", "") or "")
        if kind == "Transformer":
            tdir = sub
        elif kind == "Pooling":
            with open(os.path.join(sub, "config.json")) as f:
                pc = json.load(f)
            on = [k for k, v in pc.items() if k.startswith("pooling_mode_") and v is True]
            if on == ["pooling_mode_cls_token"]:
                pooling = "cls"
            elif on == ["pooling_mode_mean_tokens"]:
                pooling = "mean"
            else:
                raise ValueError(f"{sub}/config.json: pooling modes {on} (supported: cls token alone or mean tokens alone)")
        elif kind == "Normalize":
            normalize = True
        elif kind == "Dense":
            raise ValueError(f"{model_dir}: a sentence-transformers Dense module after pooling is not supported")
    return tdir, pooling, normalize


def load_pretrained(model_dir: str) -> Tuple[Dict[str, np.ndarray], dict, Optional[str], Optional[bool]]:
    """(weights by ``QuestionEncoder`` names, encoder config, pooling or None, normalize or None) of a local model directory.
    ``pooling`` / ``normalize`` are what a sentence-transformers ``modules.json`` prescribes (None for a plain HF directory; a DPR
    question encoder pools its CLS row: ``src/test.py:84-86`` ``.pooler_output``)."""
    if not os.path.isdir(model_dir):
        raise FileNotFoundError(f"{model_dir}: not a directory (model names are not fetched: there is no hub access here)")
    tdir, pooling, normalize = _sentence_transformers_layout(model_dir)
    cj = os.path.join(tdir, "config.json")
    if not os.path.exists(cj):
        raise FileNotFoundError(f"{cj} is missing")
    with open(cj) as f:
        hf = json.load(f)
    cfg = encoder_config_from_hf(hf)
    if str(hf.get("model_type", "")).lower() == "dpr":
        if int(hf.get("projection_dim", 0)) != 0:
            raise ValueError("a DPR encoder with a projection layer (projection_dim > 0) is not supported")
        pooling = pooling or "cls"
    raw = _read_state_dict(tdir)
    weights = {}
    for name, a in raw.items():
        short = _strip(name)
        if short.startswith(("embeddings.", "encoder.layer.")) and not short.endswith(("position_ids", "token_type_ids")):
            weights[short] = np.ascontiguousarray(a, dtype=np.float32)
    if "embeddings.token_type_embeddings.weight" not in weights:  # some RoBERTa exports drop the single-row table
        weights["embeddings.token_type_embeddings.weight"] = np.zeros((cfg["type_vocab"], cfg["hidden"]), np.float32)
    need = 5 + 16 * cfg["layers"]
    if len(weights) < need:
        raise KeyError(f"{tdir}: found {len(weights)} encoder tensors, a {cfg['layers']}-layer model has {need} "
                       f"(first names seen: {sorted(raw)[:4]})")
    return weights, cfg, pooling, normalize


# ---- hub names, offline -------------------------------------------------------------------------------------------------------
def _hub_cache_roots():
    """Directories a Hugging Face / sentence-transformers install keeps downloaded models in, most specific first."""
    env, home = os.environ, os.path.expanduser("~")
    roots = []
    for var in ("SENTENCE_TRANSFORMERS_HOME", "HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "TRANSFORMERS_CACHE"):
        if env.get(var):
            roots.append(env[var])
    if env.get("HF_HOME"):
        roots.append(os.path.join(env["HF_HOME"], "hub"))
    xdg = env.get("XDG_CACHE_HOME") or os.path.join(home, ".cache")
    roots += [os.path.join(xdg, "huggingface", "hub"), os.path.join(xdg, "torch", "sentence_transformers")]
    seen, out = set(), []
    for r in roots:
        if r and r not in seen:
            seen.add(r)
            out.append(r)
    return out


def resolve_model_path(path) -> Optional[str]:
    """``path=`` of the reference's constructor -> a local model directory, or None.

    ``heavy_ranker.py:80,83`` pass HUB NAMES (``"sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2"``).  There is no network
    here, but a model that was downloaded once sits in the local cache; this looks where the hub client and sentence-transformers
    put it, without importing either:

      <root>/models--{org}--{name}/snapshots/<revision>/config.json     huggingface_hub (``$HF_HUB_CACHE``, ``$HF_HOME/hub``,
                                                                        ``~/.cache/huggingface/hub``, ``$SENTENCE_TRANSFORMERS_HOME``);
                                                                        the revision ``refs/main`` names, else the newest snapshot
      <root>/{org}_{name}/config.json                                   sentence-transformers <= 2.2 (``~/.cache/torch/sentence_transformers``)

    A directory is returned as it is.  A bare name (no ``/``) is tried under the ``sentence-transformers`` organisation too, as that
    library does."""
    if path is None:
        return None
    p = str(path)
    if os.path.isdir(p):
        return p
    if not p or p.startswith((".", os.sep)) or p.count("/") > 1 or "\\" in p:
        return None  # a file-system path that does not exist, not a hub name
    names = [p] if "/" in p else [p, "sentence-transformers/" + p]
    for name in names:
        org, _, model = name.rpartition("/")
        for root in _hub_cache_roots():
            hub = os.path.join(root, "models--" + (org + "--" if org else "") + model)
            snaps = os.path.join(hub, "snapshots")
            if os.path.isdir(snaps):
                cands = []
                ref = os.path.join(hub, "refs", "main")
                if os.path.isfile(ref):
                    with open(ref) as f:
                        cands.append(os.path.join(snaps, f.read().strip()))
                cands += sorted((os.path.join(snaps, d) for d in os.listdir(snaps)), key=lambda d: -os.path.getmtime(d))
                for c in cands:
                    if os.path.isfile(os.path.join(c, "config.json")):
                        return c
            flat = os.path.join(root, (org + "_" if org else "") + model)
            if os.path.isfile(os.path.join(flat, "config.json")):
                return flat
    return None
