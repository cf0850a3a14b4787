"""Host-side logic that needs no GPU: shard bounds, dtype names, the document store (mirror of the reference's
setup_db helpers), the agreement rule, argument validation, and the no-CPU-fallback guarantee."""
import os

import numpy as np
import pytest
import torch

from vietnamese_qa_system_amd import docstore, heavy_ranker
from vietnamese_qa_system_amd.index import DeviceIndex, merge_topk, resolve_dtype
from vietnamese_qa_system_amd.sharded import shard_bounds

needs_no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour WITHOUT a GPU")


def test_shard_bounds_cover_rows_contiguously():
    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            bounds = [shard_bounds(n, world, r) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in bounds]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_dtype_names():
    assert resolve_dtype("fp16") == 1 and resolve_dtype("float32") == 0 and resolve_dtype("fp8_e4m3") == 2
    with pytest.raises(ValueError):
        resolve_dtype("int4")


@needs_no_gpu
def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DeviceIndex(np.zeros((4, 64), np.float16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        merge_topk(torch.zeros(2, 1, 3), torch.zeros(2, 1, 3, dtype=torch.int64), 3)
    from vietnamese_qa_system_amd import Embeddings
    emb = Embeddings(min_score=None)
    with pytest.raises(RuntimeError):
        emb.index_vectors(None, np.zeros((4, 64), np.float32))
    with pytest.raises(RuntimeError, match="empty"):
        emb.search(np.zeros(64, np.float32), 1)


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vietnamese_qa_system_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libflat_ip" not in text, f


def test_docstore_roundtrip_matches_reference_helpers(tmp_path):
    """Pinned by the behaviour probed from the reference's setup_db.py: fetch_size=50000 returns a
    list of row tuples, fetch_size=1 returns one row tuple, ids are 1-based AUTOINCREMENT rowids."""
    path = docstore.setup_database("documents", database_dir=str(tmp_path), verbose=False)
    assert path.endswith("documents.db")
    docstore.insert_data(path, "documents", [{"doc": "a", "source": "s"}, {"doc": "b", "source": "s"}], verbose=False)
    assert docstore.query(path, "SELECT * FROM documents", fetch_size=50000) == [(1, "a", "s"), (2, "b", "s")]
    assert docstore.query(path, "SELECT doc FROM documents WHERE id = 2", fetch_size=1) == ("b",)
    assert docstore.query(path, "SELECT doc FROM documents", fetch_size="all") == [("a",), ("b",)]
    with pytest.raises(ValueError):
        docstore.query(path, "SELECT 1", fetch_size=0)
    assert docstore.load_documents(path) == [{"id": 1, "text": "a", "source": "s"}, {"id": 2, "text": "b", "source": "s"}]
    assert docstore.fetch_docs(path, [2, 2, 1, -1, 99]) == {1: "a", 2: "b"}
    docstore.drop_tables(path, ["documents"], verbose=False)
    with pytest.raises(Exception):
        docstore.query(path, "SELECT * FROM documents")
    with pytest.raises(AssertionError):
        docstore.connect_database(str(tmp_path / "missing.db"))


def test_docstore_import_of_reference_shape(tmp_path):
    db = str(tmp_path / "x" / "documents.db")
    docstore.write_documents(db, [{"id": 5, "text": "năm", "source": "wiki"}, {"id": 9, "text": "chín", "source": None}])
    assert docstore.fetch_docs(db, [9, 5]) == {5: "năm", 9: "chín"}


def test_fetch_docs_keeps_its_connection_and_follows_the_file(tmp_path):
    """The doc-text join runs once per search: it keeps one sqlite connection per (file, thread) and reopens when the file was
    written to, replaced or removed and re-created (what `Embeddings.save` over an old directory does)."""
    import os
    import threading
    db = str(tmp_path / "documents.db")
    docstore.write_documents(db, [{"id": i + 1, "text": f"doc {i}", "source": "s"} for i in range(100)])
    assert docstore.fetch_docs(db, [5]) == {5: "doc 4"}
    conn = docstore._read_connection(db)
    assert docstore._read_connection(db) is conn and docstore.fetch_docs(db, [7, 8]) == {7: "doc 6", 8: "doc 7"}
    docstore.write_documents(db, [{"id": 500, "text": "moi", "source": "s"}])           # written to: the new row is seen
    assert docstore.fetch_docs(db, [500, 5]) == {5: "doc 4", 500: "moi"}
    os.remove(db)
    docstore.write_documents(db, [{"id": 5, "text": "khac", "source": "s"}])            # re-created: the old rows are gone
    assert docstore.fetch_docs(db, [500, 5]) == {5: "khac"}
    seen = {}
    t = threading.Thread(target=lambda: seen.update(docstore.fetch_docs(db, [5])))      # another thread: a connection of its own
    t.start()
    t.join()
    assert seen == {5: "khac"}
    assert docstore.fetch_docs(db, []) == {} and docstore.fetch_docs(db, [-1]) == {}


def test_agreement_rule():
    # heavy_ranker.py:110: same id AND score sum > 0.4 (strict)
    assert heavy_ranker.agreement(7, 0.25, 7, 0.16)
    assert not heavy_ranker.agreement(7, 0.2, 7, 0.2)
    assert not heavy_ranker.agreement(7, 0.9, 8, 0.9)


def test_embeddings_document_normalisation_and_validation():
    from vietnamese_qa_system_amd import Embeddings
    docs = Embeddings._normalise_documents([{"id": 3, "text": "a", "source": "s"}, (4, "b", None), "c"])
    assert docs == [{"id": 3, "text": "a", "source": "s"}, {"id": 4, "text": "b"}, {"id": 2, "text": "c"}]
    with pytest.raises(ValueError):
        Embeddings(pooling="max")
    with pytest.raises(ValueError):
        Embeddings(dtype="int4")
    emb = Embeddings(hybrid=True, content=True, path="sentence-transformers/paraphrase-multilingual-mpnet-base-v2")  # heavy_ranker.py:78-83
    assert emb.hybrid and emb.content and emb.weights == 0.5
    with pytest.raises(FileNotFoundError):
        Embeddings().load("/nonexistent/index/dir")


def test_bench_bare_multi_gpu_launch_never_raises_systemexit_for_a_missing_launcher():
    """`python bench.py --gpus N` WITHOUT torch.distributed.run (the driver's 1-GPU command shape with N > 1): bench.py is
    its own launcher.  On a box with fewer devices than ranks (this CPU container shows none) and no device sharing asked for, it
    says so and exits 2 before any rank starts -- no traceback, nothing on stdout."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VQA_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    import torch
    if torch.cuda.device_count() < 8:
        assert out.returncode == 2 and "GPU(s)" in out.stderr and "Traceback" not in out.stderr and out.stdout.strip() == ""
