"""MI355X-native dense retrieval for the Vietnamese_QA_System retriever API (one hot path, nothing else).

``Embeddings`` is the txtai-shaped object the reference's ``inference_pipeline/db_utils/heavy_ranker.py`` drives;
everything below it (L2-normalise, fp16 index in HBM, fused MFMA scoring + top-k, shard merge, question encoder)
runs in ``libvqa_retrieval.so`` (hand-written HIP, gfx950).  There is no CPU fallback.
"""
from .sharded import ensure_multi_process_gpu_env as _ensure_env  # noqa: F401  (first: sets HSA_ENABLE_IPC_MODE_LEGACY=0 under a distributed launcher)
from .embeddings import Embeddings  # noqa: F401
from .index import DeviceIndex, merge_topk  # noqa: F401
from .sharded import ShardedSearcher, shard_bounds, sharded_index_searcher  # noqa: F401

__all__ = ["Embeddings", "DeviceIndex", "merge_topk", "ShardedSearcher", "shard_bounds", "sharded_index_searcher"]
