"""One rank of the jobs tests/test_gpu_sharded_exec.py starts with
``python -m torch.distributed.run --nproc-per-node R`` (R = 1, 2 or 8).

Runs the REAL sharded path -- DeviceIndex.search (HIP) -> dist.all_gather_into_tensor -> vqa_merge_topk (HIP) -- and
asserts on every rank that the result equals, bit for bit, the single-shard search of the concatenated corpus on the
same device and in the same storage type (fp16, fp8 e4m3, fp32): exact ties on both sides of and straddling every shard
boundary, a row count the world size does not divide, ranks with fewer rows than k, an id vector, string ids, k > 12,
the R * k = 8192 limit of the merge (8 ranks x 1024 candidates: what a hybrid search at limit >= 103 asks for), the
per-rank-slice build, the row-producer build and the sharded save -> load round trip (fp16 and fp8).
``--share`` puts every rank on cuda:0 (gloo, since RCCL refuses two ranks on one device); without it every rank takes its
own GPU over nccl (= RCCL).  A world of ONE rank over nccl with VQA_ALWAYS_GATHER=1 takes every collective call of the
path through RCCL on a 1-GPU box.
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
    collective = world > 1 or os.environ.get("VQA_ALWAYS_GATHER") == "1"

    def unit(rng, n, d):
        v = rng.standard_normal((n, d)).astype(np.float32)
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float16)

    def single(x16, ids, q, k, id_base=0, dtype="fp16"):
        ix = DeviceIndex(x16, ids=ids, id_base=id_base, dtype=dtype, device=dev_index)
        s, i, _ = ix.search(q, k)
        torch.cuda.synchronize()
        out = s.clone(), i.clone()
        ix.close()
        return out

    def compare(name, emb, q, k, ref):
        before = emb._searcher.collectives
        s, i = emb._searcher.search(q, k)
        torch.cuda.synchronize()
        assert torch.equal(i, ref[1]), f"{name}: ids differ on rank {rank}"
        assert torch.equal(s, ref[0]), (f"{name}: scores differ on rank {rank}: max |diff| {float((s - ref[0]).abs().max()):.3g} at "
                                        f"{(s != ref[0]).nonzero()[:4].tolist()} ids {i[s != ref[0]][:12].tolist()}; shard sketch state {emb._index.sketch_state()}"
                                        + (f" stats {emb._index.sketch_stats()}" if emb._index.sketch_state() >= 0 else ""))
        assert emb._searcher.collectives == before + (1 if collective else 0), f"{name}: the all-gather did not run"
        checks.append(name)

    rng = np.random.default_rng(5)
    n, d, b, k = 6001, 96, 37, 10  # 6001 = 8 * 750 + 1: ragged over 2 and over 8 ranks
    x = unit(rng, n, d)
    bounds = [shard_bounds(n, world, r) for r in range(world)]
    lo1 = bounds[-1][0]
    x[(lo1 + 999) % n] = x[10]     # exact duplicates in the first and the last shard: equal scores, order by position
    x[n - 2] = x[10]
    # ... and a run of three equal rows straddling EVERY shard boundary (up to three boundaries with 4+ ranks): the tie
    # group of query 1 then spans every pair of neighbouring shards
    straddle = sorted({lo for lo, _ in bounds[1:4] if lo > 0})
    for lo in straddle:
        x[lo - 1] = x[lo] = x[lo + 1] = x[77]
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
    assert dup == sorted(int(ids[p]) for p in {10, (lo1 + 999) % n, n - 2}), dup  # the tie group, by global row position
    if straddle:
        rows77 = sorted({77} | {p for lo in straddle for p in (lo - 1, lo, lo + 1)})
        got77 = ref[1][1, :len(rows77)].cpu().numpy().tolist()
        assert got77 == [int(ids[p]) for p in rows77][:len(got77)], got77
    # the txtai-shaped call returns the same thing on every rank
    res = emb.batchsearch(q16[:4].astype(np.float32), 5)
    assert [r[0] for r in res[0][:3]] == dup
    # 1b. pipelined batches: the all-gather of batch i runs under the scan of batch i + 1; results equal per-batch search.
    #     Six distinct (B, k) shapes in a row: the buffer pools must survive them without dropping a slot in flight.
    batches = [q, q[:5], torch.from_numpy(unit(rng, 64, d)).to(device), q, q[:7], q[:9], q[:11], q[:13]]
    want = [tuple(t.clone() for t in emb._searcher.search(bq, k)) for bq in batches]
    got = emb._searcher.search_pipelined(batches, k)
    torch.cuda.synchronize()
    assert len(got) == len(want) and all(torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]) for g, w in zip(got, want))
    assert len(emb._searcher._pools) <= 2
    checks.append("pipelined")
    # 2. k > 12 (one-pass wide search per shard + merge of R * k candidates)
    compare("k=300", emb, q, 300, single(x, ids, q, 300))
    # 3. sharded save (every rank writes its byte range) -> load (every rank reads its byte range)
    path = os.path.join(args.out, "saved")
    emb.save(path)
    emb2 = Embeddings(device=dev_index, min_score=None).load(path)
    compare("save-load", emb2, q, k, ref)
    assert emb2.load_stats["bytes"] == (shard_bounds(n, world, rank)[1] - shard_bounds(n, world, rank)[0]) * d * 2
    # 4. per-rank slices only (no rank holds the whole corpus), contiguous positions as ids; total by all_reduce
    lo, hi = shard_bounds(n, world, rank)
    emb3 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb3.index_vectors(None, x[lo:hi], local=True)
    compare("local-slice", emb3, q, k, single(x, None, q, k, id_base=0))
    # 5. row producer
    emb4 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    emb4.index_vectors(None, lambda a, c: x[a:c], total=n, chunk_rows=1000)
    compare("producer", emb4, q, k, single(x, None, q, k, id_base=0))
    # 6. shards shorter than k: 15 rows over the ranks, k = 10 (padding candidates (-inf, -1) take part in the merge);
    #    with 8 ranks every shard holds one or two rows
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

    # 8. the other storage types of BASELINE configs[1] / configs[4] through shard -> gather -> merge: bit-equal to the
    #    single-shard search of the same storage type (a row's score does not depend on the shard it sits in)
    for dt in ("fp8", "fp32"):
        e = Embeddings(dtype=dt, device=dev_index, min_score=None)
        e.index_vectors(ids.tolist(), x)
        r_dt = single(x, ids, q, k, dtype=dt)
        compare(f"{dt}-shards", e, q, k, r_dt)
        compare(f"{dt}-k=40", e, q, 40, single(x, ids, q, 40, dtype=dt))
        if dt == "fp8":
            # scores of an fp8 index come back divided by 256 (rows and queries are stored as e4m3(16 x)); the duplicate
            # group still leads query 0, by position
            assert r_dt[1][0, :3].cpu().numpy().tolist() == dup
            # sharded save of an fp8 index (codes -> exactly representable fp16) -> load re-encodes the very same codes
            p8 = os.path.join(args.out, "saved_fp8")
            e.save(p8)
            e2 = Embeddings(device=dev_index, min_score=None).load(p8)
            assert e2.dtype == "fp8"
            compare("fp8-save-load", e2, q, k, r_dt)
        e._index.close()

    # 9. the merge at its R * k limit: 1024 candidates per rank (what hybrid search asks for from limit = 103 on) -- at 8
    #    ranks R * k = 8192 = the documented maximum of vqa_merge_topk; shards of 750 rows pad their lists with (-inf, -1)
    n2 = 20011
    x2 = unit(rng, n2, 64)
    q2 = torch.from_numpy(unit(rng, 5, 64)).to(device)
    e9 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    e9.index_vectors(None, x2)
    compare("k=1024", e9, q2, 1024, single(x2, None, q2, 1024))
    compare("k=1024-short-shards", emb, q[:3], 1024, single(x, ids, q[:3], 1024))
    e9._index.close()
    # ... and through the hybrid (dense + BM25) entry point that produces such a k: index(documents) with a text encoder
    # (every rank encodes only its own shard: the row-producer path), batchsearch(list[str], limit = 103)
    texts = [f"tok{j} grp{j % 13} fam{j % 101}" for j in range(n2)]
    vec_of = {t: x2[j].astype(np.float32) for j, t in enumerate(texts)}
    qtexts = [f"grp{3} fam{7}", f"tok{55} fam{55}"]
    for t_ in qtexts:
        vec_of[t_] = x2[len(vec_of) % n2].astype(np.float32)
    calls = []

    def fake_encoder(ts):
        calls.append(len(ts))
        return torch.from_numpy(np.stack([vec_of[t] for t in ts]))

    eh = Embeddings(hybrid=True, dtype="fp16", device=dev_index, encoder=fake_encoder)
    eh.index([{"id": j + 1, "text": t} for j, t in enumerate(texts)], batch_size=500)
    lo2, hi2 = shard_bounds(n2, world, rank)
    assert sum(calls) <= (hi2 - lo2) + 1, (sum(calls), hi2 - lo2)  # this rank encoded its own rows only (+ the shape probe)
    before = eh._searcher.collectives
    rh = eh.batchsearch(qtexts, 103)
    assert eh._searcher.collectives == before + (1 if collective else 0)
    assert all(0 < len(r) <= 103 for r in rh) and all(r[j][1] >= r[j + 1][1] for r in rh for j in range(len(r) - 1))
    gathered = [None] * world
    dist.all_gather_object(gathered, [[t[0] for t in r] for r in rh])
    assert all(g == gathered[0] for g in gathered), "hybrid results differ between ranks"
    checks.append("hybrid-limit-103")

    # 10. shards large enough for the int8 sketch pre-pass (VQA_STAGE_MIN=2 brings the two-stage / sketch switch-over down to
    #     131 072 rows per shard): every rank's main launch scans its sketch and re-scores the survivors; the merged result
    #     equals the single-shard sketch search of the concatenated corpus bit for bit -- a row's exact score does not depend on
    #     the shard it sits in, and ties astride shard AND stage boundaries resolve by global row position
    from vietnamese_qa_system_amd import index as index_mod
    index_mod.DEFAULT_OPTIONS["stage_min_tiles"] = 2  # (vqa_index_options.stage_min_tiles; rounds 1-4: VQA_STAGE_MIN=2 in the environment)
    n3 = 140_000 * world + 3
    x3 = unit(rng, n3, 64)
    b3 = [shard_bounds(n3, world, r) for r in range(world)]
    for lo, _ in b3[1:]:
        x3[lo - 1] = x3[lo] = x3[lo + 65_536] = x3[123]  # astride each shard boundary + astride the next shard's first stage
    q3 = unit(rng, 24, 64)
    q3[0] = x3[123]
    q3t = torch.from_numpy(q3).to(device)
    e10 = Embeddings(dtype="fp16", device=dev_index, min_score=None)
    e10.index_vectors(None, x3)
    assert e10._index.launch_info(24, k).sketch_scan == 1
    one = DeviceIndex(x3, dtype="fp16", device=dev_index)
    assert one.launch_info(24, k).sketch_scan == 1
    s1, i1, _ = one.search(q3t, k)
    torch.cuda.synchronize()
    assert one.sketch_state() == 0, (one.sketch_state(), one.sketch_stats())  # a fallback would return the exact scan's bits instead
    compare("sketch-shards", e10, q3t, k, (s1.clone(), i1.clone()))
    tie = sorted({123} | {p for lo, _ in b3[1:] for p in (lo - 1, lo, lo + 65_536)})
    assert i1[0, :min(k, len(tie))].cpu().numpy().tolist() == tie[:k], (i1[0].tolist(), tie)
    one.close()
    e10._index.close()
    del index_mod.DEFAULT_OPTIONS["stage_min_tiles"]

    dist.barrier()
    with open(os.path.join(args.out, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "backend": args.backend, "device": dev_index, "checks": checks,
                   "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                   "collectives": emb._searcher.collectives}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
