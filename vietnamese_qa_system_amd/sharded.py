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

import os
from typing import Callable, Optional, Tuple


def ensure_multi_process_gpu_env() -> None:
    """A multi-rank job on this driver needs dmabuf IPC (``HSA_ENABLE_IPC_MODE_LEGACY=0``): in the legacy mode RCCL's intra-node
    transport fails with ``hipIpcGetMemHandle: invalid argument``.  The variable is read when the HSA runtime initialises, so it is
    set here -- at import, before this process has made a GPU call -- whenever a distributed launcher started this process
    (RANK / WORLD_SIZE in the environment) and the caller has not decided otherwise.  (bench.py does the same for itself; an already initialised GPU is left alone.)"""
    if "WORLD_SIZE" in os.environ and "RANK" in os.environ:  # started by a distributed launcher (any world size: a group of one rank runs the same RCCL calls)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


ensure_multi_process_gpu_env()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``: the first ``n % world`` ranks hold one extra row."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedSearcher:
    """All-gather + merge around a per-rank shard search.

    ``local_search(queries, k, out_scores, out_ids)`` writes the shard's ``[B, k]`` candidates into the two given
    tensors -- views of ONE send buffer ``[ids int64 | scores float32]`` (12 * B * k bytes, padded to 8), so a batch
    costs a single collective; ``merge(scores [R, B, k], ids [R, B, k], k)`` gets rank-strided views of the gathered
    buffer.

    ``always_gather`` (default: the environment variable ``VQA_ALWAYS_GATHER=1``) runs the all-gather and the merge even in
    a process group of ONE rank, so that a single-GPU box takes every collective call of the N > 1 path through RCCL
    (``tests/test_gpu_sharded_exec.py::test_one_rank_over_rccl``); without it a lone rank returns its shard's result directly."""

    def __init__(self, local_search: Callable[[torch.Tensor, int, torch.Tensor, torch.Tensor], None],
                 merge: Optional[Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]] = None,
                 group=None, always_gather: Optional[bool] = None):
        self.local_search = local_search
        if merge is None:
            from .index import merge_topk
            merge = merge_topk
        self.merge = merge
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if always_gather is None:
            always_gather = os.environ.get("VQA_ALWAYS_GATHER") == "1"
        self.collective = self.world > 1 or (bool(always_gather) and dist.is_initialized())
        self._pools = {}  # (device, slot) -> [send bytes, recv bytes]: grown, never dropped (see _buffers)
        self.collectives = 0  # all-gathers issued (tests / bench read it)

    def _buffers(self, b: int, k: int, device: torch.device, slot: int = 0):
        """Views of this slot's byte pools for a [b, k] batch.  A pool only ever GROWS (to the next power of two): batches of
        any mix of sizes reuse the same two allocations per slot, so an asynchronous gather in flight on one slot never
        sees its buffers dropped or re-allocated under a stream of varied (B, k) -- the round-2 cache cleared itself at 8
        keys.  (A grown pool replaces the old one only for LATER batches; a pending gather keeps its views alive.)"""
        row = (12 * b * k + 7) // 8 * 8  # bytes per rank: ids [B, k] int64, scores [B, k] float32, pad to 8
        pool = self._pools.get((device, slot))
        if pool is None or pool[0].numel() < row:
            cap = 1 << max(12, (row - 1).bit_length())
            pool = [torch.zeros((cap,), dtype=torch.uint8, device=device),
                    torch.zeros((self.world * cap,), dtype=torch.uint8, device=device)]
            self._pools[(device, slot)] = pool
        send = pool[0][:row]
        recv = pool[1][:self.world * row].view(self.world, row)
        n = b * k
        send_i = send[:8 * n].view(torch.int64).view(b, k)
        send_s = send[8 * n:12 * n].view(torch.float32).view(b, k)
        recv_i = recv[:, :8 * n].view(torch.int64).unflatten(1, (b, k))
        recv_s = recv[:, 8 * n:12 * n].view(torch.float32).unflatten(1, (b, k))
        return send, recv, (send_s, send_i, recv_s, recv_i)

    def search(self, queries: torch.Tensor, k: int, stamps=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """One batch: shard search -> all-gather -> merge.  ``stamps`` (bench / profiling only): four ``torch.cuda.Event``s
        recorded on the current stream before the shard search, after it, after the all-gather (the current stream has
        waited for RCCL's stream by then) and after the merge -- the three phases of a step as the device saw them."""
        b = int(queries.shape[0])
        send, recv, (send_s, send_i, recv_s, recv_i) = self._buffers(b, k, self._device(queries))
        if stamps is not None:
            stamps[0].record()
        self.local_search(queries, k, send_s, send_i)
        if stamps is not None:
            stamps[1].record()
        if not self.collective:
            out = send_s.clone(), send_i.clone()
            if stamps is not None:
                stamps[2].record()
                stamps[3].record()
            return out
        self.collectives += 1
        dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)  # the one exchange step
        if stamps is not None:
            stamps[2].record()
        out = self.merge(recv_s, recv_i, k)
        if stamps is not None:
            stamps[3].record()
        return out

    def search_pipelined(self, batches, k: int):
        """Several query batches back to back with the exchange of batch i hidden under the scan of batch i + 1 (SURVEY.md
        section 8e): the all-gather is issued asynchronously (RCCL runs it on its own stream behind the scan that produced
        the send buffer), the next batch's scan is enqueued at once, and only then does the compute stream wait for the
        gather and merge batch i.  Two buffer slots alternate.  Returns ``[(scores, ids), ...]`` in batch order, identical
        to calling :meth:`search` per batch."""
        results, pending = [], None
        for n, queries in enumerate(batches):
            b = int(queries.shape[0])
            send, recv, (send_s, send_i, recv_s, recv_i) = self._buffers(b, k, self._device(queries), slot=n & 1)
            self.local_search(queries, k, send_s, send_i)
            work = None
            if self.collective:
                self.collectives += 1
                work = dist.all_gather_into_tensor(recv.view(-1), send, group=self.group, async_op=True)
            if pending is not None:
                results.append(self._finish(*pending, k))
            pending = (work, send_s, send_i, recv_s, recv_i)
        if pending is not None:
            results.append(self._finish(*pending, k))
        return results

    def _finish(self, work, send_s, send_i, recv_s, recv_i, k):
        if work is None:
            return send_s.clone(), send_i.clone()
        work.wait()  # the compute stream waits for the gather; the host does not block on RCCL
        return self.merge(recv_s, recv_i, k)

    def _device(self, queries: torch.Tensor) -> torch.device:
        return queries.device


def sharded_index_searcher(index, group=None) -> ShardedSearcher:
    """:class:`ShardedSearcher` over a :class:`~vietnamese_qa_system_amd.index.DeviceIndex` shard: the fused kernel
    writes its results straight into the all-gather send buffer."""
    dev = torch.device("cuda", index.device)

    def local(q, k, out_s, out_i):
        index.search(q, k, out=(out_s, out_i))

    searcher = ShardedSearcher(local, None, group)
    searcher._device = lambda queries: dev
    return searcher
