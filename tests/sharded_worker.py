"""One rank of the job tests/test_gpu_sharded_exec.py starts with ``python -m torch.distributed.run --nproc-per-node 2``.

Runs the REAL sharded path -- DeviceIndex.search (HIP) -> dist.all_gather_into_tensor -> vqa_merge_topk (HIP) -- and
asserts on every rank that the result equals, bit for bit, the single-shard search of the concatenated corpus on the
same device: cross-shard exact ties, a shard shorter than k, string ids, k > 12, the per-rank-slice build and the
sharded save -> load round trip.  ``--share`` puts both ranks on cuda:0 (gloo, since RCCL refuses two ranks on one
device); without it every rank takes its own GPU over nccl (= RCCL).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--share", action="store_true")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    from vietnamese_qa_system_amd import build
    assert build.is_fresh(), "build the library before starting the workers"

    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if args.share else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")

    from vietnamese_qa_system_amd.embeddings import Embeddings
    from vietnamese_qa_system_amd.index import DeviceIndex
    from vietnamese_qa_system_amd.sharded import shard_bounds

    checks = []

    def unit(rng, n, d):
        v = rng.standard_normal((n, d)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float16)

    def single(x16, ids, q, k, id_base=0):
        ix = DeviceIndex(x16, ids=ids, id_base=id_base, dtype="fp16", device=dev_index)
        s, i, _ = ix.search(q, k)
        torch.cuda.synchronize()
        out = s.clone(), i.clone()
        ix.close()
        return out

    def compare(name, emb, q, k, ref):
        s, i = emb._searcher.search(q, k)
        torch.cuda.synchronize()
        assert torch.equal(i, ref[1]), f"{name}: ids differ on rank {rank}"
        assert torch.equal(s, ref[0]), f"{name}: scores differ on rank {rank}"
        checks.append(name)

    rng = np.random.default_rng(5)
    n, d, b, k = 6001, 96, 37, 10
    x = unit(rng, n, d)
    lo1, _ = shard_bounds(n, world, world - 1)
    x[lo1 + 999] = x[10]           # exact duplicates on both sides of the shard boundary: equal scores, order by position
    x[n - 2] = x[10]
    x[lo1 - 1] = x[lo1] = x[77]    # ... and straddling it
    q16 = unit(rng, b, d)
    q16[0], q16[1] = x[10], x[77]  # the duplicate groups lead these queries' results
    q = torch.from_numpy(q16).to(device)
    ids = (np.arange(n, dtype=np.int64) * 3 + 11)

    # 1. explicit id vector, whole arrays on every rank
    emb = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb.index_vectors(ids.tolist(), x)
    ref = single(x, ids, q, k)
    compare("id-vector", emb, q, k, ref)
    dup = ref[1][0, :3].cpu().numpy().tolist()
    assert dup == [int(ids[10]), int(ids[lo1 + 999]), int(ids[n - 2])], dup  # the tie group, by global row position
    # the txtai-shaped call returns the same thing on every rank
    res = emb.batchsearch(q16[:4].astype(np.float32), 5)
    assert [r[0] for r in res[0][:3]] == dup
    # 1b. pipelined batches: the all-gather of batch i runs under the scan of batch i + 1; results equal per-batch search
    batches = [q, q[:5], torch.from_numpy(unit(rng, 64, d)).to(device), q]
    want = [tuple(t.clone() for t in emb._searcher.search(bq, k)) for bq in batches]
    got = emb._searcher.search_pipelined(batches, k)
    torch.cuda.synchronize()
    assert len(got) == len(want) and all(torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) for g, w in zip(got, want))
    checks.append("pipelined")
    # 2. k > 12 (one-pass wide search per shard + merge of R * k candidates)
    compare("k=300", emb, q, 300, single(x, ids, q, 300))
    # 3. sharded save (every rank writes its byte range) -> load (every rank reads its byte range)
    path = os.path.join(args.out, "saved")
    emb.save(path)
    emb2 = Embeddings(device=dev_index, min_score=None).load(path)
    compare("save-load", emb2, q, k, ref)
    assert emb2.load_stats["bytes"] == (shard_bounds(n, world, rank)[1] - shard_bounds(n, world, rank)[0]) * d * 2
    # 4. per-rank slices only (no rank holds the whole corpus), contiguous positions as ids
    lo, hi = shard_bounds(n, world, rank)
    emb3 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb3.index_vectors(None, x[lo:hi], local=True)
    compare("local-slice", emb3, q, k, single(x, None, q, k, id_base=0))
    # 5. row producer
    emb4 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb4.index_vectors(None, lambda a, c: x[a:c], total=n, chunk_rows=1000)
    compare("producer", emb4, q, k, single(x, None, q, k, id_base=0))
    # 6. a shard shorter than k: 15 rows over the ranks, k = 10 (padding candidates (-inf, -1) take part in the merge)
    xs = unit(rng, 15, d)
    emb5 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb5.index_vectors(list(range(1, 16)), xs)  # contiguous ids: id_base path
    compare("short-shard", emb5, q, k, single(xs, None, q, k, id_base=1))
    compare("short-shard-k12", emb5, q, 12, single(xs, None, q, 12, id_base=1))
    # 7. string ids map through the host-side list on every rank
    emb6 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb6.index_vectors([f"doc-{j}" for j in range(15)], xs)
    r6 = emb6.batchsearch(q16[:2].astype(np.float32), 3)
    pos = single(xs, None, q[:2], 3)[1].cpu().numpy()
    assert [[t[0] for t in row] for row in r6] == [[f"doc-{j}" for j in row] for row in pos.tolist()]
    checks.append("string-ids")
    # re-indexing the same object with integer ids must not keep the string ids
    emb6.index_vectors(list(range(100, 115)), xs)
    r7 = emb6.batchsearch(q16[:2].astype(np.float32), 3)
    assert [[t[0] for t in row] for row in r7] == (pos + 100).tolist()
    checks.append("reindex")

    dist.barrier()
    with open(os.path.join(args.out, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "backend": args.backend, "device": dev_index, "checks": checks}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
