"""Row-sharded search across the GPUs of one node: one process per GPU, RCCL all-gather of per-shard candidates.

New design (the reference is single-process; its retriever call is ``heavy_ranker.py:98-101``): the corpus is split
into contiguous row ranges, rank r holds rows ``[r*n/R, (r+1)*n/R)`` as a :class:`DeviceIndex`; every rank searches
the same query batch on its shard (fused HIP kernel), the ``[B, k]`` (score, id) candidates are all-gathered with
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI; 12 bytes x B x k per rank -- latency bound, one
collective per batch), and every rank runs the same final merge kernel, so all ranks return identical results.
Ties resolve by (rank asc, slot asc) = global row position asc, identical to a single-GPU search of the whole corpus.

``local_search`` / ``merge`` are injectable so the collective plumbing can be exercised on CPU with ``gloo``
(tests only: there they are backed by the oracle); the defaults are the HIP paths and need a GPU.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``: the first ``n % world`` ranks hold one extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedSearcher:
    """All-gather + merge around a per-rank shard search."""

    def __init__(self, local_search: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                 merge: Optional[Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]] = None,
                 group=None):
        self.local_search = local_search
        if merge is None:
            from .index import merge_topk
            merge = merge_topk
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._gs = None
        self._gi = None

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        s, i = self.local_search(queries, k)
        if self.world == 1:
            return s, i
        b = s.shape[0]
        if self._gs is None or self._gs.shape[1:] != (b, k) or self._gs.device != s.device:
            self._gs = torch.empty((self.world, b, k), dtype=torch.float32, device=s.device)
            self._gi = torch.empty((self.world, b, k), dtype=torch.int64, device=s.device)
        # output viewed as the concatenation along dim 0 ([R * B, k]): same memory as [R, B, k], accepted by RCCL and gloo
        dist.all_gather_into_tensor(self._gs.view(self.world * b, k), s.contiguous(), group=self.group)
        dist.all_gather_into_tensor(self._gi.view(self.world * b, k), i.contiguous(), group=self.group)
        return self.merge(self._gs, self._gi, k)


def sharded_index_searcher(index, group=None) -> ShardedSearcher:
    """:class:`ShardedSearcher` over a :class:`~vietnamese_qa_system_amd.index.DeviceIndex` shard."""
    def local(q, k):
        s, i, _ = index.search(q, k)
        return s, i
    return ShardedSearcher(local, None, group)
