"""``Embeddings(path=<local model directory>)`` (heavy_ranker.py:78-83): the loader of Hugging Face / sentence-transformers
directories -- host logic here (safetensors container, prefix stripping, config mapping, pooling from modules.json); the
``-m gpu`` half runs the loaded model on the device and holds it to the HF outputs stored beside the same weights."""
import json
import os
import struct

import numpy as np
import pytest

from vietnamese_qa_system_amd import pretrained as P

TINY = dict(vocab_size=100, hidden=64, layers=2, heads=4, ffn=128, max_pos=40, type_vocab=1, pad_id=1, ln_eps=1e-5, position_ids="roberta")
BERT_TINY = dict(vocab_size=120, hidden=64, layers=2, heads=2, ffn=128, max_pos=48, type_vocab=2, pad_id=0, ln_eps=1e-12,
                 position_ids="absolute")


def _npz_weights(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}")
    return g, {k[2:]: g[k] for k in g.files if k.startswith("w.")}


def test_sentence_transformers_directory_with_safetensors(golden_dir):
    w, cfg, pooling, normalize = P.load_pretrained(f"{golden_dir}/hf_tiny_roberta_st")
    _, ref = _npz_weights(golden_dir, "enc_tiny.npz")
    assert cfg == TINY and pooling == "mean" and normalize is True
    assert set(w) == set(ref) and all(np.array_equal(w[k], ref[k]) and w[k].dtype == np.float32 for k in ref)


def test_plain_hf_directory_with_prefixed_pytorch_bin(golden_dir):
    w, cfg, pooling, normalize = P.load_pretrained(f"{golden_dir}/hf_tiny_bert_bin")
    _, ref = _npz_weights(golden_dir, "enc_bert_tiny.npz")
    assert cfg == BERT_TINY and pooling is None and normalize is None
    assert set(w) == set(ref) and all(np.array_equal(w[k], ref[k]) for k in ref)  # pooler / head / position_ids tensors skipped


def test_safetensors_reader_dtypes_and_errors(tmp_path):
    a32 = np.arange(6, dtype=np.float32).reshape(2, 3)
    a16 = np.array([1.5, -2.0], dtype=np.float16)
    bf = (np.array([1.0, -0.5, 3.140625], np.float32).view(np.uint32) >> 16).astype("<u2")  # exactly representable in bf16
    blobs = [("x", "F32", a32.shape, a32.tobytes()), ("y", "F16", a16.shape, a16.tobytes()), ("z", "BF16", (3,), bf.tobytes())]
    header, off = {"__metadata__": {"format": "pt"}}, 0
    for name, dt, shape, raw in blobs:
        header[name] = {"dtype": dt, "shape": list(shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
    hj = json.dumps(header).encode()
    path = tmp_path / "m.safetensors"
    path.write_bytes(struct.pack("<Q", len(hj)) + hj + b"".join(b[3] for b in blobs))
    t = P.read_safetensors(str(path))
    assert np.array_equal(t["x"], a32) and np.array_equal(t["y"], a16) and t["z"].dtype == np.float32
    assert t["z"].tolist() == [1.0, -0.5, 3.140625]
    path.write_bytes(struct.pack("<Q", len(hj)) + hj + b"\x00" * 5)  # truncated data
    with pytest.raises(ValueError, match="past the end"):
        P.read_safetensors(str(path))
    path.write_bytes(b"\x01\x02")
    with pytest.raises(ValueError, match="not a safetensors"):
        P.read_safetensors(str(path))


def test_config_mapping_and_refusals(tmp_path):
    base = {"model_type": "xlm-roberta", "vocab_size": 250002, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12,
            "intermediate_size": 3072, "max_position_embeddings": 514, "type_vocab_size": 1, "pad_token_id": 1, "layer_norm_eps": 1e-5}
    from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE
    assert P.encoder_config_from_hf(base) == dict(XLMR_BASE, position_ids="roberta")
    minilm = {"model_type": "bert", "vocab_size": 250037, "hidden_size": 384, "num_hidden_layers": 12, "num_attention_heads": 12,
              "intermediate_size": 1536, "max_position_embeddings": 512, "type_vocab_size": 2, "pad_token_id": 0, "layer_norm_eps": 1e-12}
    assert P.encoder_config_from_hf(minilm) == MINILM_L12  # paraphrase-multilingual-MiniLM-L12-v2, heavy_ranker.py:80
    with pytest.raises(ValueError, match="model_type"):
        P.encoder_config_from_hf(dict(base, model_type="t5"))
    with pytest.raises(ValueError, match="GELU"):
        P.encoder_config_from_hf(dict(base, hidden_act="relu"))
    with pytest.raises(ValueError, match="relative"):
        P.encoder_config_from_hf(dict(base, position_embedding_type="relative_key"))
    with pytest.raises(FileNotFoundError, match="not a directory"):
        P.load_pretrained("sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2")  # a hub name: nothing to fetch it with
    (tmp_path / "config.json").write_text(json.dumps(base))
    with pytest.raises(FileNotFoundError, match="safetensors"):
        P.load_pretrained(str(tmp_path))
    # a pooling mix the encoder does not implement is refused, not approximated
    os.makedirs(tmp_path / "1_Pooling")
    (tmp_path / "modules.json").write_text(json.dumps([{"path": "", "type": "sentence_transformers.models.Transformer"},
                                                       {"path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]))
    (tmp_path / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": True, "pooling_mode_max_tokens": True}))
    with pytest.raises(ValueError, match="pooling modes"):
        P.load_pretrained(str(tmp_path))


@pytest.mark.gpu
def test_embeddings_loads_the_model_named_by_path(native_lib, golden_dir):
    """heavy_ranker.py:78-83 in miniature: Embeddings(content=True, path=<directory>) -> index(texts) -> search(text).  The HIP
    encoder built from the directory reproduces HF's last_hidden_state of the same weights (enc_tiny.npz) and its mean-pooled,
    L2-normalised vectors drive the search."""
    import torch
    from oracle import encoder as E
    from vietnamese_qa_system_amd import Embeddings
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    g, w = _npz_weights(golden_dir, "enc_tiny.npz")
    ids, mask = g["input_ids"], g["attention_mask"]
    enc = QuestionEncoder.from_pretrained(f"{golden_dir}/hf_tiny_roberta_st", max_tokens=64)
    assert enc.pooling == "mean" and enc.normalize is True and enc.config == TINY
    real = mask.astype(bool)
    hs = enc.hidden_states(ids, mask).cpu().numpy()
    assert np.abs(hs[real] - g["last_hidden_state"][real]).max() < 2.4e-2  # tests/test_gpu_encoder.py BOUNDS["tiny_hidden"]
    enc.close()
    table = {f"text {i}": (ids[i], mask[i]) for i in range(ids.shape[0])}

    def tokenizer(texts):
        return np.stack([table[t][0] for t in texts]), np.stack([table[t][1] for t in texts])

    emb = Embeddings(content=True, path=f"{golden_dir}/hf_tiny_roberta_st", tokenizer=tokenizer, max_tokens=64, min_score=None)
    emb.index([{"id": 10 + i, "text": f"text {i}", "source": "s"} for i in range(4)])
    assert emb.pooling == "mean"
    ref = E.encode({k: np.asarray(v, np.float64) for k, v in w.items()}, TINY, ids, mask, pooling="mean")
    for i in range(4):
        hit = emb.search(f"text {i}", 1)[0]
        assert hit["id"] == 10 + i and hit["text"] == f"text {i}" and abs(hit["score"] - 1.0) < 2e-3
    res = emb.search("text 2", 4)
    want = np.argsort(-(ref @ ref[2]))
    assert [r["id"] for r in res] == [10 + int(j) for j in want]
    # heavy_ranker.py:87 / :91-94: save, then a FRESH object loads the directory; the model path travels in meta.json, so text queries work
    # again as soon as a tokenizer is there (the reference re-creates Embeddings() and calls .load(...) before it searches)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        emb.save(tmp)
        again = Embeddings(tokenizer=tokenizer, max_tokens=64, min_score=None).load(tmp)
        assert again.path == f"{golden_dir}/hf_tiny_roberta_st" and again.content is True
        assert again.search("text 2", 4) == res
        again._index.close()
    # the BERT directory (prefixed pytorch_model.bin, no modules.json): pooling stays what the caller asked for
    gb, _ = _npz_weights(golden_dir, "enc_bert_tiny.npz")
    encb = QuestionEncoder.from_pretrained(f"{golden_dir}/hf_tiny_bert_bin", max_tokens=64)
    assert encb.pooling is None and encb.config == BERT_TINY
    got = encb.forward(gb["input_ids"], gb["attention_mask"], pooling="mean", normalize=False).cpu().numpy()
    assert np.abs(got - gb["mean_pooled"]).max() < 1.2e-2  # BOUNDS["tiny_raw"]
    encb.close()
    with pytest.raises(RuntimeError, match="local Hugging Face cache"):  # (not in any local cache either: tests above plant one)
        Embeddings(path="sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2").index([{"id": 1, "text": "q"}])


def _write_safetensors(path, tensors):
    """A minimal safetensors WRITER for the tests (8-byte header length, JSON header, raw little-endian tensors)."""
    names = {np.dtype(np.float32): "F32", np.dtype(np.float16): "F16", np.dtype(np.int64): "I64"}
    header, blobs, off = {"__metadata__": {"format": "pt"}}, [], 0
    for name, a in tensors.items():
        raw = np.ascontiguousarray(a).tobytes()
        header[name] = {"dtype": names[a.dtype], "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * (-len(hj) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))


@pytest.mark.gpu
def test_minilm_shaped_directory_in_the_old_sentence_transformers_layout(native_lib, tmp_path):
    """The shape of paraphrase-multilingual-MiniLM-L12-v2 (heavy_ranker.py:80: BERT, hidden 384, 12 heads of 32, absolute positions;
    two layers here) as a directory in sentence-transformers' OLDER layout -- the transformer files under `0_Transformer/`, weights
    stored as fp16 safetensors with the `bert.` prefix, `1_Pooling` asking for the CLS token -- written by the test's own
    safetensors writer: from_pretrained reads it, and the HIP forward matches the oracle on the same (fp16-rounded) weights."""
    from oracle import encoder as E
    from vietnamese_qa_system_amd.encoder import QuestionEncoder
    cfg = dict(E.MINILM_L12, layers=2, vocab_size=3000)
    w = {k: v.astype(np.float16) for k, v in E.synthetic_weights(cfg, seed=33, layers=2).items()}
    root = tmp_path / "minilm"
    (root / "0_Transformer").mkdir(parents=True)
    (root / "1_Pooling").mkdir()
    _write_safetensors(str(root / "0_Transformer" / "model.safetensors"),
                       {**{"bert." + k: v for k, v in w.items()}, "bert.embeddings.position_ids": np.arange(512, dtype=np.int64)[None]})
    (root / "0_Transformer" / "config.json").write_text(json.dumps(
        {"model_type": "bert", "vocab_size": 3000, "hidden_size": 384, "num_hidden_layers": 2, "num_attention_heads": 12,
         "intermediate_size": 1536, "max_position_embeddings": 512, "type_vocab_size": 2, "pad_token_id": 0, "layer_norm_eps": 1e-12,
         "hidden_act": "gelu"}))
    (root / "modules.json").write_text(json.dumps(
        [{"idx": 0, "name": "0", "path": "0_Transformer", "type": "sentence_transformers.models.Transformer"},
         {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]))
    (root / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": True, "pooling_mode_mean_tokens": False}))
    enc = QuestionEncoder.from_pretrained(str(root), max_tokens=40 * 32)
    assert enc.pooling == "cls" and enc.normalize is False and enc.config == cfg
    ids, mask = E.synthetic_tokens(cfg, 40, 32, seed=2)
    got = enc.forward(ids, mask, pooling=enc.pooling).cpu().numpy()
    ref = E.encode({k: v.astype(np.float32) for k, v in w.items()}, cfg, ids[:6], mask[:6], pooling="cls")
    assert np.abs(got[:6] - ref).max() < 8.5e-4  # tests/test_gpu_encoder.py BOUNDS["minilm", 2]
    enc.close()


# ---- hub names, offline (heavy_ranker.py:80,83 pass "sentence-transformers/..." names) ------------------------
def _fake_hub_cache(root, golden_dir, name="sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2", revision="abc123", refs=True):
    """<root>/models--org--name/snapshots/<revision>/ = a copy of the tiny golden model directory, as huggingface_hub lays it out."""
    import shutil
    hub = os.path.join(root, "models--" + name.replace("/", "--"))
    snap = os.path.join(hub, "snapshots", revision)
    shutil.copytree(os.path.join(golden_dir, "hf_tiny_roberta_st"), snap)
    if refs:
        os.makedirs(os.path.join(hub, "refs"))
        with open(os.path.join(hub, "refs", "main"), "w") as f:
            f.write(revision)
    return snap


def test_hub_name_resolves_from_the_local_cache(golden_dir, tmp_path, monkeypatch):
    name = "sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2"
    for var in ("SENTENCE_TRANSFORMERS_HOME", "HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "TRANSFORMERS_CACHE", "HF_HOME", "XDG_CACHE_HOME"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HOME", str(tmp_path / "home"))
    assert P.resolve_model_path(name) is None  # nothing cached: the caller raises its clear error
    assert P.resolve_model_path(str(tmp_path / "no" / "such" / "dir")) is None and P.resolve_model_path(None) is None
    assert P.resolve_model_path(f"{golden_dir}/hf_tiny_roberta_st") == f"{golden_dir}/hf_tiny_roberta_st"  # a directory is itself
    # $HF_HOME/hub, revision named by refs/main (an older snapshot beside it is not taken)
    hub = tmp_path / "hf_home" / "hub"
    _fake_hub_cache(str(hub), golden_dir, name, "old000", refs=False)
    snap = _fake_hub_cache_second = os.path.join(str(hub), "models--" + name.replace("/", "--"), "snapshots", "new111")
    import shutil
    shutil.copytree(os.path.join(golden_dir, "hf_tiny_roberta_st"), snap)
    os.makedirs(os.path.join(str(hub), "models--" + name.replace("/", "--"), "refs"))
    with open(os.path.join(str(hub), "models--" + name.replace("/", "--"), "refs", "main"), "w") as f:
        f.write("new111")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf_home"))
    assert P.resolve_model_path(name) == snap
    # a bare name is tried under the sentence-transformers organisation, as that library does
    assert P.resolve_model_path("paraphrase-multilingual-MiniLM-L12-v2") == snap
    # the default location ~/.cache/huggingface/hub and sentence-transformers' older flat layout
    monkeypatch.delenv("HF_HOME")
    default = _fake_hub_cache(str(tmp_path / "home" / ".cache" / "huggingface" / "hub"), golden_dir, "org/model-x")
    assert P.resolve_model_path("org/model-x") == default
    flat = tmp_path / "st_home" / "sentence-transformers_all-tiny"
    shutil.copytree(os.path.join(golden_dir, "hf_tiny_roberta_st"), flat)
    monkeypatch.setenv("SENTENCE_TRANSFORMERS_HOME", str(tmp_path / "st_home"))
    assert P.resolve_model_path("sentence-transformers/all-tiny") == str(flat)
    # what loads from there is the model
    w, cfg, pooling, normalize = P.load_pretrained(P.resolve_model_path(name) or snap)
    assert cfg == TINY and pooling == "mean"


def test_a_hub_name_that_is_not_cached_is_a_clear_error(monkeypatch, tmp_path):
    """No GPU needed: the text path fails before any device work when the name resolves to nothing."""
    for var in ("SENTENCE_TRANSFORMERS_HOME", "HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "TRANSFORMERS_CACHE", "HF_HOME", "XDG_CACHE_HOME"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HOME", str(tmp_path))
    from vietnamese_qa_system_amd.embeddings import Embeddings
    emb = Embeddings.__new__(Embeddings)
    emb.encoder, emb.path = None, "sentence-transformers/paraphrase-multilingual-mpnet-base-v2"
    with pytest.raises(RuntimeError, match="already in the local Hugging Face cache"):
        emb._encode(["câu hỏi"])


@pytest.mark.gpu
def test_embeddings_with_the_references_hub_name(native_lib, golden_dir, tmp_path, monkeypatch):
    """heavy_ranker.py:78-80 verbatim -- Embeddings(hybrid=..., content=True, path="sentence-transformers/...") -- with the model in a
    local hub cache: index(texts), search(text) run through the HIP encoder built from the cached snapshot."""
    from vietnamese_qa_system_amd import Embeddings
    name = "sentence-transformers/paraphrase-multilingual-MiniLM-L12-v2"
    for var in ("SENTENCE_TRANSFORMERS_HOME", "HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "TRANSFORMERS_CACHE", "XDG_CACHE_HOME"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    _fake_hub_cache(str(tmp_path / "hf" / "hub"), golden_dir, name)
    g, _ = _npz_weights(golden_dir, "enc_tiny.npz")
    ids, mask = g["input_ids"], g["attention_mask"]
    table = {f"text {i}": (ids[i], mask[i]) for i in range(ids.shape[0])}

    def tokenizer(texts):
        return np.stack([table[t][0] for t in texts]), np.stack([table[t][1] for t in texts])

    emb = Embeddings(content=True, path=name, tokenizer=tokenizer, max_tokens=64, min_score=None)
    emb.index([{"id": 10 + i, "text": f"text {i}", "source": "s"} for i in range(4)])
    for i in range(4):
        hit = emb.search(f"text {i}", 1)[0]
        assert hit["id"] == 10 + i and hit["text"] == f"text {i}" and abs(hit["score"] - 1.0) < 2e-3
    assert emb.path == name  # what save() records is what the caller passed
