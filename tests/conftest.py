import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def native_lib():
    """Path of the in-tree shared library (built on demand; hipcc cross-compiles without a GPU)."""
    from vietnamese_qa_system_amd import build
    return build.build()


# ---- library options in tests.  Up to round 4 the library read VQA_* environment variables at create time and tests reached the
# plans they wanted with monkeypatch.setenv; the library now takes explicit options (include/vqa_retrieval.h vqa_index_options /
# vqa_encoder_options) and reads no environment.  set_option keeps the old names as test vocabulary and maps each onto the option
# field it became, through the Python-side default dictionaries (index.DEFAULT_OPTIONS / encoder.DEFAULT_OPTIONS).
_INDEX_OPTION = {"VQA_STAGE_MIN": "stage_min_tiles", "VQA_STAGE_PCT": "stage_pct", "VQA_WIDE_K": "wide_k", "VQA_TWO_PASS": "two_pass",
                 "VQA_SKETCH": "sketch", "VQA_SKETCH_CASCADE": "sketch_cascade", "VQA_SKETCH_MID_K": "sketch_mid_k",
                 "VQA_SKETCH_MID_MIN": "sketch_mid_min_tiles", "VQA_SKETCH_PRE_K": "sketch_pre_k", "VQA_POISON_WORKSPACE": "poison_workspace", "VQA_ONE_LAUNCH": "one_launch",
                 "VQA_SKETCH_CENTER": "sketch_center", "VQA_SKETCH_PER_ROW": "sketch_per_row", "VQA_SKETCH_ROTATE": "sketch_rotate",
                 "VQA_SKETCH_COOLDOWN": "sketch_cooldown", "VQA_SKETCH_PROFIT": "sketch_profit", "VQA_SKETCH_SPLIT": "sketch_split",
                 "VQA_F16_LOOP": "f16_loop", "VQA_RESCORE_COPY": "rescore_copy"}
_ENCODER_OPTION = {"VQA_ENC_FOLD": "fold_layernorm", "VQA_ENC_FIRST_ROWS": "first_rows", "VQA_ENCODER_GRAPH": "graphs", "VQA_ENC_TINY": "latency_path"}


def set_option(monkeypatch, name, value):
    """``value`` None: back to the library default."""
    from vietnamese_qa_system_amd import encoder as enc_mod
    from vietnamese_qa_system_amd import index as index_mod
    table, key = (index_mod.DEFAULT_OPTIONS, _INDEX_OPTION[name]) if name in _INDEX_OPTION else (enc_mod.DEFAULT_OPTIONS, _ENCODER_OPTION[name])
    if value is None:
        monkeypatch.delitem(table, key, raising=False)
        return
    v = float(value) if key == "sketch_profit" else int(str(value), 0)
    monkeypatch.setitem(table, key, v)
