#!/usr/bin/env python3
"""Headline benchmark: queries/sec (+ recall@10) of the dense-retrieval hot path on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W      # no launcher: bench.py starts that same job itself as a child process

A "step" = one pass of the hot path over one batch: 256 L2-normalised query vectors scored against the rank's
row shard (10M x 768 fp16 per GPU by default = BASELINE.json configs[2]; N GPUs hold N x 10M rows = configs[3]
at N = 8), fused MFMA scoring + top-10, and for N > 1 the RCCL all-gather of per-shard (score, id) candidates and
the final merge.  Inputs (index, queries) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Protocol: W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize pairs
(``ms_per_step`` / ``value`` come from that wall-clock bracket, max over ranks); every timed step is also bracketed by
an event pair on its stream -> ``step_ms`` median / p10 / p90; the dominant kernel is timed by HIP events on its own
launch stream inside the library (``roofline``).  After the timed region, outside it: recall@10 against the oracle over
the full shard, an end-to-end leg (question encoder + search + merge) and the CPU baseline.

The oracle (``oracle/``) is used here only (a) as the ``cpu_baseline`` timed on the host cores over a bounded sample
and (b) as the checker for recall@10 -- never as the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy rate)
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide, Matrix cores: f32-input MFMA = the f32 vector rate (155 measured)
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16/fp16
FP8_SCALE = 16.0


def ensure_library():
    """Build the HIP library BEFORE anything touches the GPU (all ranks call this; a file lock lets one build).  Under a
    profiler the tool library initialises the GPU in every child it is preloaded into, so a stale library is refused
    there instead of being rebuilt: run `python -m vietnamese_qa_system_amd.build` first."""
    from vietnamese_qa_system_amd import build
    if build.is_fresh():
        return
    if build.under_profiler():
        raise SystemExit("libvqa_retrieval.so is stale and a profiler environment is active: build first with "
                         "`python -m vietnamese_qa_system_amd.build`, then profile")
    build.build()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` as a child process (never an exec: this
    process stays the parent and has not initialised the GPU -- counting devices does not), pass rank 0's single JSON line
    through to stdout, everything else to stderr, and return the job's exit code."""
    import socket
    import subprocess
    ensure_library()  # one build here, before N ranks would queue on the build lock
    share = os.environ.get("VQA_BENCH_SHARE_GPU") == "1"
    try:
        import torch
        have = torch.cuda.device_count()  # does not initialise the GPU on this image
    except Exception:  # noqa: BLE001 -- the ranks themselves report a missing GPU
        have = None
    if have is not None and have < n and not share:
        sys.stderr.write(f"bench.py: --gpus {n} but this node shows {have} GPU(s); one rank per GPU needs {n} "
                         "(VQA_BENCH_SHARE_GPU=1 puts every rank on the devices there are, over gloo: plumbing runs only)\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in proc.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def build_shard(torch, n, d, seed, device, dtype, chunk=1 << 18):
    """Synthetic corpus shard generated ON the device: i.i.d. standard normal rows, L2-normalised in fp32
    (SURVEY.md section 8d); kept as fp16 for an fp16 index (= the stored values), as fp32 for fp32 / fp8 indexes."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    buf = torch.empty((n, d), dtype=torch.float16 if dtype == "fp16" else torch.float32, device=device)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        x = torch.randn((c1 - c0, d), generator=gen, device=device, dtype=torch.float32)
        x /= x.norm(dim=1, keepdim=True)
        buf[c0:c1] = x.to(buf.dtype)
    return buf


def cpu_oracle_topk(np, shard, q, k, dtype, rows=None, chunk_rows=1 << 19):
    """Exact top-k of the oracle over the first ``rows`` rows of the device shard, streamed to the host in chunks.
    ``dtype``: 'fp16' / 'fp32' score the stored values as they are; 'fp8' scores the e4m3 codes of 16 * x (rows and
    queries, exactly what the index stores) and divides by 256; 'ref32' scores the un-quantised fp32 rows."""
    from oracle import retrieval as R
    n = shard.shape[0] if rows is None else min(rows, shard.shape[0])
    q = q.astype(np.float32)
    if dtype == "fp8":
        q = R.e4m3_decode(R.e4m3_encode_fast(q * FP8_SCALE)) / FP8_SCALE ** 2
    best_s = best_p = None
    for c0 in range(0, n, chunk_rows):
        c1 = min(n, c0 + chunk_rows)
        xc = shard[c0:c1].cpu().numpy()
        if dtype == "fp8":
            s, _, p = R.search(q, R.e4m3_encode_fast(xc.astype(np.float32) * FP8_SCALE), k, dtype=R.DTYPE_FP8_E4M3)
        else:
            s, _, p = R.search(q, xc, k, dtype=R.DTYPE_F16 if xc.dtype == np.float16 else R.DTYPE_F32)
        p = p + c0
        if best_s is None:
            best_s, best_p = s, p
        else:
            ms = np.concatenate([best_s, s], axis=1)
            mp = np.concatenate([best_p, p], axis=1)
            o = np.lexsort((mp, -ms.astype(np.float64)), axis=1)[:, :k]
            best_s = np.take_along_axis(ms, o, axis=1)
            best_p = np.take_along_axis(mp, o, axis=1)
    return best_s, best_p


def measure_copy(torch, device, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate of this box in GB/s of bytes read + written (what a plain HBM stream achieves here)."""
    src = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.zero_()
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    del src, dst
    torch.cuda.empty_cache()
    return round(2 * nbytes / (ms * 1e-3) / 1e9, 1)


def measure_read_stream(torch, device, nbytes=8 << 30):
    """GB/s of a READ-ONLY stream on this box with the scan kernels' own loads (LDS-DMA nt into an LDS ring, one persistent workgroup per
    CU; csrc/diag.hip, the library's measurement entry): the ceiling the scans' roofline fractions can be read against here
    (torch's read + write copy is not that ceiling)."""
    import ctypes
    from vietnamese_qa_system_amd import _native as N
    lib = N.load()
    buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
    buf.fill_(0x5A)
    torch.cuda.synchronize(device)
    out = ctypes.c_double()
    N.check(lib.vqa_measure_read_stream(ctypes.c_void_p(buf.data_ptr()), nbytes, 5, ctypes.byref(out), None), "vqa_measure_read_stream")
    del buf
    torch.cuda.empty_cache()
    return round(out.value, 1)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus() -> int:
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box reports 256
    logical CPUs but grants a quota of 16: 256 BLAS threads on it run 50x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def percentiles(np, xs):
    a = np.asarray(xs, dtype=np.float64)
    return {"median": round(float(np.median(a)), 4), "p10": round(float(np.percentile(a, 10)), 4),
            "p90": round(float(np.percentile(a, 90)), 4), "min": round(float(a.min()), 4), "max": round(float(a.max()), 4)}


def cpu_baseline(np, torch, shard, q_host, k, dtype, n, sample_rows, budget_s=25.0):
    """The oracle's blocked flat inner-product search (oracle/retrieval.py:search_blocked = fp32 GEMM on the host BLAS +
    exact top-k per block: faiss IndexFlatIP semantics) timed on the host cores over a BOUNDED sample of the same
    workload, host-resident fp32 rows (the device->host copy is outside the timing).  B = 256 and B = 1 (the
    reference's own calling pattern, heavy_ranker.py:97-98) at 1k / 100k / sample rows."""
    from oracle import retrieval as R
    cores = usable_cpus()
    rows = min(sample_rows, n)
    x = shard[:rows].cpu()
    if dtype == "fp8":  # the values an fp8 index scores: e4m3(16 x) / 16
        x = torch.from_numpy(R.e4m3_decode(R.e4m3_encode_fast(x.float().numpy() * FP8_SCALE)) / FP8_SCALE)
    x = x.float().contiguous()
    q32 = np.ascontiguousarray(q_host, dtype=np.float32)
    # thread count: the quota-derived number, unless a short calibration on one 64k-row block finds a smaller pool faster
    # (oversubscribed BLAS threads on a throttled container cost an order of magnitude)
    best_t, best_s = cores, float("inf")
    for t in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16), min(cores, 8)}, reverse=True):
        torch.set_num_threads(t)
        R.search_blocked(q32[:8], x[:4096], k)  # BLAS / thread pool warm-up
        t0 = time.perf_counter()
        R.search_blocked(q32, x[:min(rows, 65536)], k)
        dt = time.perf_counter() - t0
        if dt < 0.9 * best_s:
            best_t, best_s = t, dt
    cores = best_t
    torch.set_num_threads(cores)
    points, spent = [], 0.0
    for nrows in sorted({min(1000, rows), min(100_000, rows), rows}):
        for b in (q32.shape[0], 1):
            reps, best = 0, float("inf")
            t_begin = time.perf_counter()
            while reps < 5 and (reps < 2 or time.perf_counter() - t_begin < budget_s / 8):
                t0 = time.perf_counter()
                R.search_blocked(q32[:b], x[:nrows], k)
                best = min(best, time.perf_counter() - t0)
                reps += 1
            spent += time.perf_counter() - t_begin
            points.append({"rows": nrows, "batch": b, "seconds": round(best, 5), "queries_per_s": round(b / best, 2),
                           "gflops": round(2.0 * b * nrows * x.shape[1] / best / 1e9, 1)})
    full = next(p for p in points if p["rows"] == rows and p["batch"] == q32.shape[0])
    one = next(p for p in points if p["rows"] == rows and p["batch"] == 1)
    return {"value": round(q32.shape[0] / (full["seconds"] * (n / rows)), 3), "unit": "queries/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "logical_cpus": os.cpu_count(),
            "sample": (f"oracle/retrieval.py:search_blocked (fp32 torch.mm on the host BLAS + exact top-k per 16384-row block) on "
                       f"{q32.shape[0]} queries x the first {rows} rows of the shard, host-resident fp32, {cores} threads (cgroup quota / calibrated), best of 5 runs "
                       f"{full['seconds']:.3f} s; value = {q32.shape[0]} / (t * {n}/{rows}); whole leg {spent:.1f} s of CPU work"),
            "value_batch1": round(1.0 / (one["seconds"] * (n / rows)), 3),
            "points": points}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)  # the chip's clock management settles over the first ~8 launches
    ap.add_argument("--docs-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", choices=["fp16", "fp8", "fp32"], default="fp16", help="index storage type (headline: fp16)")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--verify-queries", type=int, default=16, help="queries checked against the CPU oracle over the FULL shard")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline and the recall check (profiling runs)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (encoder + search) leg")
    ap.add_argument("--e2e-steps", type=int, default=30)
    ap.add_argument("--no-sketch", action="store_true", help="no int8 sketch: the exact scan of the index type (profiling runs)")
    ap.add_argument("--no-other", action="store_true", help="skip the other_configs legs (fp8, fp32 + encoder, configs[0] API latency)")
    args = ap.parse_args()

    # RCCL's intra-node transport needs dmabuf IPC on this driver (hipIpcGetMemHandle fails in the legacy mode): set before torch is
    # imported / anything touches the GPU, for EVERY way this file is started -- bare, self-spawned or under an external launcher
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        # a bare `python bench.py --gpus N`: this process becomes the launcher of its own N ranks (a CHILD torch.distributed.run
        # job, started before anything here has touched the GPU) and relays rank 0's JSON line and the job's exit code
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a launcher started a different number of ranks than --gpus names: the job's own world size decides the sharding, the
        # JSON line reports it as n_gpus (and says so), nothing exits
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; running {world} ranks\n")
    ensure_library()  # before the first GPU call of this process

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the measured path")
    # one process per GPU; VQA_BENCH_SHARE_GPU=1 (dev / tests) lets several ranks share a device to exercise the N > 1 path
    # on a 1-GPU box (then over gloo, since RCCL refuses two ranks on one device)
    share = os.environ.get("VQA_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from vietnamese_qa_system_amd.index import DeviceIndex
    from vietnamese_qa_system_amd.sharded import sharded_index_searcher

    n, d, b, k = args.docs_per_gpu, args.dim, args.batch, args.k
    # (ranks that share a device -- VQA_BENCH_SHARE_GPU=1: the N-rank program on a 1-GPU box -- build their shards one after the other
    # and without the row-major re-scoring copy, so that eight full-size shards, configs[3] / configs[4], fit one MI355X)
    shard = index = None
    for turn in range(world if share and world > 1 else 1):
        if not (share and world > 1) or turn == rank:
            shard = build_shard(torch, n, d, 1234 + rank, device, args.dtype)
            index = DeviceIndex(shard, id_base=1 + rank * n, dtype=args.dtype, device=dev_index, sketch=False if args.no_sketch else None,
                                rescore_copy=False if share and world > 1 else None)
            if rank != 0:  # only rank 0 checks its shard against the oracle afterwards
                shard = shard[:1].clone()
                torch.cuda.empty_cache()
        if share and world > 1:
            torch.cuda.synchronize(device)
            dist.barrier()
    gq = torch.Generator(device=device)
    gq.manual_seed(99)
    q = torch.randn((b, d), generator=gq, device=device, dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).to(shard.dtype)
    searcher = sharded_index_searcher(index)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        out = searcher.search(q, k)
    # the dominant launch is bracketed by HIP events on its stream inside the library on every 4th step of the timed region (an
    # event record between two launches stalls the stream for ~6 us each side: bracketing every step costs the step 0.6 %)
    KERNEL_EVENT_EVERY = 4
    index.set_timing(True)
    index.set_timing(False)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for i, (e0, e1) in enumerate(ev):
        sampled = i % KERNEL_EVENT_EVERY == 0
        if sampled:
            index.set_timing(True, resume=True)
        e0.record()
        out = searcher.search(q, k)
        e1.record()
        if sampled:
            index.set_timing(False)
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = index.get_timing()
    index.set_timing(False)
    step_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    multi = None
    if world > 1:
        # max over ranks decides value / ms_per_step; every rank's own bracket is kept beside it
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in per_rank]
        elapsed = max(float(x.item()) for x in per_rank)
        # where a step's time goes, as the device saw it: events on the compute stream before / after the shard search, after the
        # all-gather (the compute stream has waited for the collective's stream by then) and after the merge -- a separate, untimed
        # run of min(steps, 20) steps on every rank (four more event records per step would perturb the headline bracket)
        ps = max(2, min(args.steps, 20))
        pev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(ps)]
        c0 = searcher.collectives
        sync()
        for st in pev:
            searcher.search(q, k, stamps=st)
        sync()
        ph = torch.tensor([[st[j].elapsed_time(st[j + 1]) for j in range(3)] for st in pev], dtype=torch.float64, device=device)
        med = ph.median(dim=0).values
        allmed = [torch.zeros_like(med) for _ in range(world)]
        dist.all_gather(allmed, med)
        allmed = torch.stack(allmed).cpu().numpy()  # [rank, phase]
        # proof of N ranks on N devices that does not depend on NCCL_DEBUG: every rank contributes (rank, local device index, the
        # device's uuid bytes) through the job's own backend
        props = torch.cuda.get_device_properties(device)
        uuid_hex = "".join(ch for ch in str(getattr(props, "uuid", "")) if ch in "0123456789abcdefABCDEF")[-32:].rjust(32, "0")
        uuid_bytes = bytes.fromhex(uuid_hex)
        ident = torch.tensor([rank, dev_index, *uuid_bytes], dtype=torch.int64, device=device)
        idents = [torch.zeros_like(ident) for _ in range(world)]
        dist.all_gather(idents, ident)
        idents = [[int(v) for v in t.cpu().tolist()] for t in idents]
        rank_devices = [{"rank": t[0], "device_index": t[1], "device_uuid": bytes(t[2:18]).hex()} for t in idents]
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:  # noqa: BLE001 -- diagnostics only
            rccl_version = None
        one_gpu = None
        p80 = next((f for f in (os.path.join(ROOT, "profiles", f"r0{r}_80M_one_gpu.json") for r in (6, 5)) if os.path.exists(f)), "")
        if p80:
            with open(p80) as f:
                j80 = json.load(f)
            one_gpu = {"file": "profiles/" + os.path.basename(p80), "rows": 80_000_000,
                       "ms_per_batch_sketch_path": j80["default_path_int8_sketch_scan_plus_exact_rescoring"]["ms_per_batch"],
                       "ms_per_batch_exact_scan": j80["exact_fp16_scan_VQA_SKETCH_0"]["ms_per_batch"],
                       "note": "the WHOLE 80M x 768 fp16 corpus of configs[3] on ONE MI355X (256 queries, top-10): the strong-scaling denominator; "
                               "this job's ms_per_step at N = 8 x 10M rows is the numerator's inverse"}
        multi = {"backend": backend + (" (RCCL)" if backend == "nccl" else " (ranks share a device: plumbing run, VQA_BENCH_SHARE_GPU=1)"),
                 "rccl_version": rccl_version, "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                 "ranks": rank_devices, "distinct_devices": len({(t["device_index"], t["device_uuid"]) for t in rank_devices}),
                 "corpus_80M_on_one_gpu": one_gpu,
                 "world_size": world, "device_count": torch.cuda.device_count(), "gpus_flag": args.gpus,
                 "devices": sorted({(r % torch.cuda.device_count()) if share else r for r in range(world)}),
                 "collectives_per_step": (searcher.collectives - c0) / ps,
                 "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 4), "max": round(max(per_rank_ms), 4),
                                          "by_rank": [round(x, 4) for x in per_rank_ms]},
                 "phase_ms": {"steps": ps, "how": "median over steps of event differences on each rank's compute stream; rank 0, then min / max over ranks",
                              **{name: {"rank0": round(float(allmed[0, j]), 4), "min": round(float(allmed[:, j].min()), 4),
                                        "max": round(float(allmed[:, j].max()), 4)}
                                 for j, name in enumerate(("local_search_ms", "gather_wait_ms", "merge_ms"))}}}
    info = index.launch_info(b, k)
    device_bytes = index.device_bytes()
    sketch_state = index.sketch_state()
    sketch_stats = index.sketch_stats() if info.sketch_scan else None  # pair counts of the last timed search (synchronises: after the timed region)
    # multi-batch steady state through search_pipelined (the all-gather of batch i hidden under the scan of batch i + 1;
    # at N = 1 the same four scans back to back): S = 4 resident query batches per call, outside the headline's timed region
    S = 4
    qs = [q] + [torch.roll(q, j, dims=0) for j in range(1, S)]
    reps = max(2, min(args.steps // S, 10))
    searcher.search_pipelined(qs, k)
    sync()
    tp0 = time.perf_counter()
    for _ in range(reps):
        searcher.search_pipelined(qs, k)
    sync()
    piped = time.perf_counter() - tp0
    if world > 1:
        t = torch.tensor([piped], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        piped = float(t.item())
    copy_gbs = measure_copy(torch, device) if rank == 0 else None
    read_gbs = measure_read_stream(torch, device) if rank == 0 else None

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = b * args.steps / elapsed  # every rank answers the same batch: whole-job queries/s over world*n rows
        kern_ms = kernel_ms / max(launches, 1)
        achieved = info.bytes_per_launch / (kern_ms * 1e-3) / 1e9 if launches else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        dt = {"fp16": "f16", "fp8": "fp8_e4m3", "fp32": "f32"}[args.dtype]
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(f"{n}x{d}_{dt}_b{b}_k{k}")
        mode = {"fp16": 1, "fp8": 2, "fp32": 0}[args.dtype]
        sketch = bool(info.sketch_scan)
        esize = {"fp16": 2, "fp8": 1, "fp32": 4}[args.dtype]
        kname = ((f"sketch_scan_regq_kernel<{(d + 127) // 128 * 2}, 0>" if info.scan_kernel == 1 else "score_topk_kernel<2, 3, 0, 1>") if sketch
                 else f"score_topk_kernel<1, {mode}, 0, 0>")
        traffic_key = f"{n}x{d}_{dt}_b{b}_k{k}" + ("_sketch" if sketch else "")
        traffic_note = None
        if os.path.exists(tpath):  # the PMC bytes of THAT launch (profiles/traffic.json: an earlier rocprofv3 run of this repository, not this run)
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get(traffic_key)
            rec_kernel = tj.get("_kernel_" + traffic_key)
            if traffic is not None and rec_kernel is not None and kname.split("<")[0] not in str(rec_kernel):
                traffic, traffic_note = None, f"profiles/traffic.json holds the bytes of {rec_kernel}, not of this run's dominant kernel: dropped"
            elif traffic is not None:
                traffic_note = f"PMC bytes per launch recorded by an earlier profiling run ({tj.get('_source_' + traffic_key)}, kernel {rec_kernel}); not measured by this run"
        result = {
            "metric": "queries_per_sec", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dt, "data": "synthetic",
            "config": {"workload": f"{world}x{n}x{d} {args.dtype} index row-sharded, batch={b} queries, top-{k}, "
                                   + ("int8 sketch scan (v_mfma_i32_16x16x64_i8, rigorous score bound) + exact " + args.dtype
                                      + " re-scoring of the surviving pairs + top-k" if sketch else "fused MFMA scoring + top-k")
                                   + (", RCCL all-gather + merge" if world > 1 else ""),
                       "docs_total": world * n, "docs_per_gpu": n, "dim": d, "batch": b, "k": k,
                       "parallelism": f"row-shard x{world}"},
            # per-step event times on rank 0's stream (the bracketed wall clock above is what value / ms_per_step report)
            "step_ms": percentiles(np, step_ms),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel_ms": round(kern_ms, 4), "launches": launches,
                         "kernel_events": f"HIP events on the launch's stream around every {KERNEL_EVENT_EVERY}th timed step's dominant launch",
                         "step_ms_of_event_steps": round(float(np.median(step_ms[::KERNEL_EVENT_EVERY])), 4),
                         "bytes_per_launch": info.bytes_per_launch, "flops_per_launch": info.flops_per_launch,
                         "mfma_tflops": round(info.flops_per_launch / (kern_ms * 1e-3) / 1e12, 1) if launches else None,
                         # what a step holds besides the dominant kernel: query staging, the seed pass (re-scores the first
                         # seed rows once: outside kernel_ms, inside ms_per_step), two list merges, launch gaps
                         "step_minus_kernel_ms": round(float(np.median(step_ms)) - kern_ms, 4) if launches else None,
                         "seed_pass_rows": int(info.seed_tiles) * int(info.rows_per_tile),
                         # two-stage search: the dominant launch covers rows_per_launch of the shard's rows, a first-stage
                         # launch of the same code (symbol score_topk_kernel<1, DT, 1, 0>) the first first_stage_rows; bytes,
                         # flops, achieved and kernel_ms above are the dominant launch's alone; step_frac_physical (below) prices
                         # the WHOLE step on the bytes all its launches move
                         "rows_per_launch": int(info.rows_per_launch), "first_stage_rows": int(info.first_stage_rows),
                         # sketch_scan: large fp16 shards run the dominant launch over an int8 sketch of the rows (one byte per
                         # element: bytes_per_launch / achieved / frac price THAT launch on the bytes it has to read; the
                         # multiply-adds are int8 x int8 -> int32, mfma_tflops counts them) and score exactly only the pairs
                         # its rigorous bound cannot exclude (index_bytes = rows x d x element size of the index type as stored)
                         "sketch_scan": sketch, "index_bytes": n * d * esize,
                         # what the shard holds on the device: the rows as stored + the int8 sketch (+50 %) + the row-major copy the
                         # sketch search re-scores its survivors from (+100 %; VQA_RESCORE_COPY=0 does without: re-scoring 3x slower)
                         "device_bytes": device_bytes,
                         # 0: every timed search took the sketch search (no overflow into its exact fallback); -1: no sketch kept
                         "sketch_state": sketch_state,
                         # queries/s over the queries/s of BASELINE.md's HBM roofline for this index (one read of the shard AS STORED
                         # per batch at 8 TB/s: 133 333 q/s for 10M x 768 fp16).  NOT a bandwidth fraction: a sketch search reads fewer
                         # bytes than the shard holds (round 3 printed this as "whole_step_frac")
                         "qps_over_hbm_roofline_qps": round(n * d * esize / (float(np.median(step_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         # what the step PHYSICALLY moves: algorithmic bytes of every launch of one search (below), and that sum over
                         # the median step time as a fraction of 8 TB/s
                         **step_bytes(info, sketch_stats, n, d, esize, b, float(np.median(step_ms))),
                         # the ceiling this box's HBM really gives a plain stream: a 1 GiB device-to-device copy (torch's copy
                         # kernel), bytes read + written over its event time, in this process
                         "hbm_copy_measured_gbs": copy_gbs,
                         # ... and a READ-ONLY stream with the scan's own loads and nothing else (csrc/diag.hip): the box's ceiling for a scan;
                         # frac_of_measured_read_stream = the dominant launch against it
                         "hbm_read_stream_measured_gbs": read_gbs,
                         "frac_of_measured_read_stream": round(achieved / read_gbs, 4) if achieved and read_gbs else None},
            **({"multi_gpu": multi} if multi is not None else {}),
            "pipelined": {"batches": S, "calls": reps, "ms_per_batch": round(piped / (reps * S) * 1e3, 4),
                          "value": round(b * S * reps / piped, 1), "unit": "queries/s",
                          "note": "ShardedSearcher.search_pipelined over 4 resident query batches per call (N > 1: asynchronous "
                                  "all-gather of batch i under the scan of batch i + 1), max over ranks"},
        }
        if args.dtype == "fp32" and launches:
            # an fp32 index at B = 256 is MFMA-bound (128 flop/B against a balance of ~20): price it against the f32 MFMA peak
            tf = info.flops_per_launch / (kern_ms * 1e-3) / 1e12
            result["roofline"].update({"bound": "mfma", "achieved": round(tf, 1), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                                       "hbm_gbs": round(achieved, 1)})

    # ---- outside the timed region: recall@10 against the CPU oracle, the CPU baseline (rank 0, N = 1 only) and the
    # end-to-end leg (every rank encodes the same queries: replicate-encode)
    if rank == 0 and not args.no_cpu:
        from oracle import retrieval as R
        s_gpu, i_gpu, p_gpu = index.search(q, k, return_positions=True)
        torch.cuda.synchronize(device)
        q16 = q.cpu().numpy()
        nv = min(args.verify_queries, b)
        t1 = time.perf_counter()
        # the fp8 oracle encodes every row on the CPU (slow): it checks a 2M-row prefix through a second, prefix-only index
        vrows = n if args.dtype != "fp8" else min(n, 2_000_000)
        if vrows < n:
            pre = DeviceIndex(shard[:vrows], id_base=1, dtype=args.dtype, device=dev_index)
            s_gpu, i_gpu, p_gpu = pre.search(q, k, return_positions=True)
            torch.cuda.synchronize(device)
            pre.close()
        ref_s, ref_p = cpu_oracle_topk(np, shard, q16[:nv], k, args.dtype, rows=vrows)
        verify_s = time.perf_counter() - t1
        recall = R.recall_at_k(p_gpu[:nv].cpu().numpy(), ref_p)
        score_err = float(np.abs(s_gpu[:nv].cpu().numpy() - ref_s).max())
        result["recall_at_10"] = recall
        result["recall_check"] = {"queries": nv, "rows": vrows, "max_abs_score_err": score_err, "oracle_seconds": round(verify_s, 1),
                                  "oracle": "same stored values"}
        if args.dtype == "fp8":  # configs[4]: recall of the fp8 index against the un-quantised fp32 rows
            _, ref32_p = cpu_oracle_topk(np, shard, q16[:nv], k, "ref32", rows=vrows)
            result["recall_at_10_vs_fp32"] = R.recall_at_k(p_gpu[:nv].cpu().numpy(), ref32_p)
            # ... and the TIMED index itself (all n rows, all b queries): oracle scores of the e4m3 codes of seeded 512-row blocks
            rng = np.random.default_rng(41)
            starts = np.sort(rng.choice(max(1, n - 512), size=min(120, max(1, n // 512)), replace=False))
            pos = np.concatenate([np.arange(s0, min(n, s0 + 512), dtype=np.int64) for s0 in starts])
            rows = shard[torch.from_numpy(pos).to(device)].float().cpu().numpy()
            codes = R.e4m3_encode_fast(rows * FP8_SCALE)
            q8 = R.e4m3_decode(R.e4m3_encode_fast(q16.astype(np.float32) * FP8_SCALE)) / FP8_SCALE ** 2
            s_all, _, p_all = index.search(q, k, return_positions=True)
            torch.cuda.synchronize(device)
            result["recall_check"]["timed_index_sampled_blocks"] = sampled_exactness(np, s_all, p_all, pos, R.full_scores(q8, codes, R.DTYPE_FP8_E4M3), 2e-3, 1e-4)
            result["recall_check"]["note"] = (f"full oracle over a {vrows}-row prefix index (the fp8 oracle encodes every row on the CPU); the timed {n}-row index "
                                              "is held to the oracle on sampled blocks: no sampled row beats a returned k-th score, returned rows in the sample carry the oracle's score")
        if world == 1:
            result["cpu_baseline"] = cpu_baseline(np, torch, shard, q16, k, args.dtype, n, args.cpu_sample_rows)

    if not args.no_e2e and d == 768:
        e2e = end_to_end(torch, np, searcher, device, dev_index, b, k, args.e2e_steps, sync)
        if rank == 0:
            result["end_to_end"] = e2e
    if rank == 0 and world == 1 and not args.no_other and bool(info.sketch_scan):
        # the same shard through the EXACT fp16 scan (no int8 sketch: what small shards, k > 12 and the overflow fallback run),
        # beside the headline: step and main-launch time of the fused MFMA scoring + top-k kernel reading every fp16 byte
        exact = DeviceIndex(shard, id_base=1 + rank * n, dtype=args.dtype, device=dev_index, sketch=False)
        st_ms, kn_ms = timed_search(torch, exact, q, k, 30)
        ei = exact.launch_info(b, k)
        s_ex, _, p_ex = exact.search(q, k, return_positions=True)
        s_sk, _, p_sk = index.search(q, k, return_positions=True)
        torch.cuda.synchronize(device)
        result["exact_scan"] = {"ms_per_step": round(st_ms, 4), "queries_per_s": round(b / (st_ms * 1e-3), 1), "kernel": f"score_topk_kernel<1, {mode}, 0, 0>",
                                "kernel_ms": round(kn_ms, 4), "bytes_per_launch": int(ei.bytes_per_launch),
                                "frac": round(ei.bytes_per_launch / (kn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "qps_over_hbm_roofline_qps": round(n * d * esize / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                **{kk: vv for kk, vv in step_bytes(ei, None, n, d, esize, b, st_ms).items() if kk != "step_bytes_by_launch"},
                                "same_rows_as_sketch_search": bool(torch.equal(p_ex, p_sk)),
                                "max_abs_score_diff": float((s_ex - s_sk).abs().max())}
        exact.close()
    if rank == 0 and world == 1 and not args.no_other and not args.no_cpu:
        # small and ragged batches on the headline index (the 256-query MFMA tile is paid whatever B is): B = 1 and B = 257
        big_latency = {}
        for bb in (1, b + 1):
            qb = torch.cat([q, q[:1]])[:bb].contiguous()
            st, _ = timed_search(torch, index, qb, k, 20)
            big_latency[f"batch{bb}_step_ms"] = round(st, 4)
        index.close()
        del shard
        torch.cuda.empty_cache()
        result["other_configs"] = other_configs(torch, np, device, dev_index, n, d, b, k)
        result["other_configs"]["latency"][f"{n}_rows_{args.dtype}"] = big_latency
    if rank == 0:
        print(json.dumps(result), flush=True)
    index.close()
    if world > 1:
        dist.barrier()  # rank 0's recall check (CPU oracle, tens of seconds) ends before any rank tears the group down
        dist.destroy_process_group()


def step_bytes(info, stats, n, d, esize, b, step_ms):
    """Algorithmic HBM bytes of every launch of ONE search step (cross-checked against PMC once per round: profiles/), their sum
    and that sum over the step time as a fraction of the 8 TB/s peak."""
    seed = int(info.seed_tiles) * int(info.rows_per_tile) * d * esize
    qbytes = 256 * d * (4 + esize)  # query staging (read fp32 / fp16 rows, write the tiled tile); L2-resident afterwards
    if stats is None:  # exact search: seed rows once more + every row of the shard once (first stage + main launch)
        parts = {"query_staging": qbytes, "exact_seed_rows": seed, "exact_scan_rows": n * d * esize}
    else:
        # the cascade (vqa_index_search): exact seeds over 2 tiles per workgroup, int8 scan of the early stages (first 10 %, then -- k >= 16
        # or a shard of >= 128 tiles per workgroup -- the next 20 %: info.first_stage_rows counts both), exact re-scoring of their
        # pairs, int8 scan of the rest, exact re-scoring of its pairs (one stored row of d x esize bytes per pair; the query rows come
        # from L2), selections (the candidate keys: 8 bytes per pair, written once and read twice)
        seed = min(2 * int(info.grid), int(info.first_stage_rows) // int(info.rows_per_tile)) * int(info.rows_per_tile) * d * esize
        pairs = int(stats["rescored_pairs"])
        parts = {"query_staging_and_sketch": qbytes + 256 * d, "exact_seed_rows": seed,
                 "sketch_scan_early_stages": int(info.first_stage_rows) * d, "sketch_scan_main": int(info.rows_per_launch) * d,
                 "rescored_rows": pairs * d * esize, "candidate_pairs_and_keys": pairs * (8 + 8 + 3 * 8)}
    total = sum(parts.values())
    out = {"step_bytes_moved": total, "step_bytes_by_launch": parts,
           "step_frac_physical": round(total / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if stats is not None:
        out["candidate_pairs"] = {"main_scan": int(stats["last_scan_pairs"]), "rescored_total": int(stats["rescored_pairs"]),
                                  "largest_region": int(stats["largest_region"]), "region_capacity": int(stats["region_capacity"]),
                                  "longest_sublist": int(stats["longest_sublist"]), "sublist_capacity": int(stats["sublist_capacity"])}
    return out


def sampled_exactness(np, s_gpu, p_gpu, pos, ref, score_tol, margin):
    """The full-size check of tests/test_gpu_fullsize.py on the index that was TIMED: `ref` [B, sample] = oracle scores of the
    sampled rows `pos`.  (1) no sampled row beats a query's returned k-th score by more than `margin` without being in its
    result; (2) every returned row that falls in the sample carries the oracle's score within `score_tol`."""
    s_h, p_h = s_gpu.cpu().numpy(), p_gpu.cpu().numpy()
    kth = s_h[:, -1][:, None].astype(np.float64)
    beat = ref > kth + margin
    missed = 0
    for bq in range(ref.shape[0]):
        if beat[bq].any():
            missed += len(set(pos[beat[bq]].tolist()) - set(p_h[bq].tolist()))
    where = {int(r): j for j, r in enumerate(pos.tolist())}
    hits, err = 0, 0.0
    for bq in range(ref.shape[0]):
        for j in range(p_h.shape[1]):
            col = where.get(int(p_h[bq, j]))
            if col is not None:
                hits += 1
                err = max(err, abs(float(s_h[bq, j]) - float(ref[bq, col])))
    return {"sampled_rows": int(pos.size), "queries": int(ref.shape[0]), "sampled_rows_beating_kth_but_not_returned": int(missed),
            "returned_rows_in_sample": hits, "max_abs_score_err_on_those": err, "ok": bool(missed == 0 and err <= score_tol)}


def timed_search(torch, index, q, k, steps, warmup=5):
    """(median step ms, main-launch kernel ms) of `steps` searches of one index, events on the current stream."""
    for _ in range(warmup):
        index.search(q, k)
    index.set_timing(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in ev:
        e0.record()
        index.search(q, k)
        e1.record()
    torch.cuda.synchronize()
    kernel_ms, launches = index.get_timing()
    index.set_timing(False)
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ms[len(ms) // 2], kernel_ms / max(launches, 1)


def other_configs(torch, np, device, dev_index, n, d, b, k):
    """The other BASELINE.json configs on this GPU, compact and bounded (~20 s), after and outside the headline's timed region:
    configs[4]'s storage type (fp8 e4m3, same row count as the headline shard), configs[1] (fp32 index of 1M rows + the
    PhoBERT-base-shaped encoder on all B x L positions) and configs[0] (1k x 768 through the Embeddings API, wall clock).
    Rows are regenerated chunk by chunk from the headline's seeded recipe, so no second copy of the corpus is held."""
    from oracle import retrieval as R
    from vietnamese_qa_system_amd.embeddings import Embeddings
    from vietnamese_qa_system_amd.index import DeviceIndex
    out = {}
    chunk = 1 << 18
    gq = torch.Generator(device=device)
    gq.manual_seed(99)
    q32 = torch.randn((b, d), generator=gq, device=device, dtype=torch.float32)
    q32 = q32 / q32.norm(dim=1, keepdim=True)

    def fill(index, rows, tap=None):
        """Rows of the headline's seeded recipe, written chunk by chunk; ``tap(c0, x)`` sees every chunk's fp32 rows on the device."""
        gen = torch.Generator(device=device)
        gen.manual_seed(1234)
        for c0 in range(0, rows, chunk):
            c1 = min(rows, c0 + chunk)
            x = torch.randn((c1 - c0, d), generator=gen, device=device, dtype=torch.float32)
            x /= x.norm(dim=1, keepdim=True)
            index.set_rows(c0, x)
            if tap is not None:
                tap(c0, x)

    class SampleTap:
        """Keeps the fp32 rows of a few 512-row blocks per chunk on the host (seeded offsets: ~60k rows of a 10M-row fill) for the
        oracle, and the top-k of the un-quantised fp32 rows over ALL rows for `nq` queries (torch on the device, chunk by chunk:
        what configs[4]'s "recall@10 vs the fp32 index" is measured against -- a recall reference, not the parity oracle)."""

        def __init__(self, nq, blocks_per_chunk=3, block=512):
            self.rng = np.random.default_rng(41)
            self.bpc, self.block, self.nq = blocks_per_chunk, block, nq
            self.pos, self.rows = [], []
            self.best_s = self.best_p = None

        def __call__(self, c0, x):
            m = x.shape[0]
            for off in sorted(self.rng.choice(max(1, m - self.block + 1), size=min(self.bpc, max(1, m // self.block)), replace=False).tolist()):
                cnt = min(self.block, m - off)
                self.pos.append(np.arange(c0 + off, c0 + off + cnt, dtype=np.int64))
                self.rows.append(x[off:off + cnt].cpu().numpy())
            sc = q32[:self.nq] @ x.T
            ts, tp = torch.topk(sc, min(k, m), dim=1)
            tp = tp + c0
            if self.best_s is None:
                self.best_s, self.best_p = ts, tp
            else:
                ms, mp = torch.cat([self.best_s, ts], 1), torch.cat([self.best_p, tp], 1)
                o = torch.topk(ms, k, dim=1).indices
                self.best_s, self.best_p = torch.gather(ms, 1, o), torch.gather(mp, 1, o)

    # ---- configs[4] storage type: fp8 e4m3 (rows and queries stored as e4m3(16 x), block-scaled MFMA at twice the fp16 rate)
    t0 = time.perf_counter()
    nv = min(16, b)
    ix8 = DeviceIndex.empty(n, d, id_base=1, dtype="fp8", device=dev_index)
    tap = SampleTap(nv)
    fill(ix8, n, tap)
    step_ms, kern_ms = timed_search(torch, ix8, q32, k, 30)
    info = ix8.launch_info(b, k)
    s8, _, p8 = ix8.search(q32, k, return_positions=True)
    torch.cuda.synchronize(device)
    # the oracle on the SAME index that was timed (all n rows resident, all b queries): e4m3 codes of the sampled rows computed on
    # the host from the fp32 rows (e4m3_encode_fast), held bit-equal to what the index stores there; scores of those codes in
    # fp64; then the sampled-exactness check.  And the returned rows themselves: their stored codes read back, scored by the oracle.
    pos = np.concatenate(tap.pos)
    codes = R.e4m3_encode_fast(np.concatenate(tap.rows) * FP8_SCALE)
    stored = np.concatenate([ix8.get_rows(int(blk[0]), int(blk.size))[0] for blk in tap.pos])
    qh = q32.cpu().numpy()
    q8 = R.e4m3_decode(R.e4m3_encode_fast(qh * FP8_SCALE)) / FP8_SCALE ** 2
    chk = sampled_exactness(np, s8, p8, pos, R.full_scores(q8, codes, R.DTYPE_FP8_E4M3), 2e-3, 1e-4)
    chk["stored_codes_equal_oracle_codes"] = bool(np.array_equal(stored, codes))
    p8h, s8h = p8.cpu().numpy(), s8.cpu().numpy()
    ret_err = 0.0
    for bq in range(nv):
        rows = np.concatenate([ix8.get_rows(int(r), 1)[0] for r in p8h[bq]])
        ret_err = max(ret_err, float(np.abs(R.full_scores(q8[bq:bq + 1], rows, R.DTYPE_FP8_E4M3)[0] - s8h[bq]).max()))
    chk["returned_rows_rescored_by_oracle"] = {"queries": nv, "max_abs_score_err": ret_err}
    chk["ok"] = bool(chk["ok"] and chk["stored_codes_equal_oracle_codes"] and ret_err <= 2e-3)
    gbs = info.bytes_per_launch / (kern_ms * 1e-3) / 1e9
    out["fp8_e4m3"] = {"rows": n, "queries_per_s": round(b / (step_ms * 1e-3), 1), "step_ms": round(step_ms, 4),
                       "kernel_ms": round(kern_ms, 4), "main_launch_frac": round(gbs / HBM_PEAK_GBS, 4),
                       **{kk: vv for kk, vv in step_bytes(info, None, n, d, 1, b, step_ms).items() if kk != "step_bytes_by_launch"},
                       "mfma_tflops": round(info.flops_per_launch / (kern_ms * 1e-3) / 1e12, 1),
                       "recall_at_10_vs_fp32_rows": R.recall_at_k(p8h[:nv], tap.best_p.cpu().numpy()),
                       "recall_check": {"index": f"the timed index: all {n} rows, all {b} queries",
                                        "what": "oracle (fp64 over the same e4m3 codes, computed on the host from the fp32 rows) on seeded 512-row blocks: "
                                                "no sampled row may beat a returned k-th score by > 1e-4, returned rows in the sample score within 2e-3; "
                                                "the stored codes of the blocks read back bit-equal; every returned row of the first queries re-scored by the oracle from its stored codes",
                                        "vs_fp32_rows": f"top-{k} of the un-quantised fp32 rows over all {n} rows for {nv} queries (torch fp32 on the device, chunk by chunk)",
                                        **chk},
                       "seconds": round(time.perf_counter() - t0, 1)}
    ix8.close()
    del tap, codes, stored
    if n >= 10_000_000:
        # configs[4]'s per-GPU share: 100M x 768 fp8 over 8 GPUs = 12.5M rows (9.6 GB) per shard
        t0 = time.perf_counter()
        n125 = 12_500_000
        ix8 = DeviceIndex.empty(n125, d, id_base=1, dtype="fp8", device=dev_index)
        fill(ix8, n125)
        step_ms, kern_ms = timed_search(torch, ix8, q32, k, 20)
        info = ix8.launch_info(b, k)
        ix8.close()
        out["fp8_e4m3_12p5M_rows_per_gpu_share_of_configs4"] = {
            "rows": n125, "queries_per_s": round(b / (step_ms * 1e-3), 1), "step_ms": round(step_ms, 4), "kernel_ms": round(kern_ms, 4),
            "main_launch_frac": round(info.bytes_per_launch / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            **{kk: vv for kk, vv in step_bytes(info, None, n125, d, 1, b, step_ms).items() if kk != "step_bytes_by_launch"},
            "seconds": round(time.perf_counter() - t0, 1)}
    # ---- configs[1]: fp32 index of 1M rows + the encoder over all B x L positions (padded form).  Default path: the int8 sketch
    # scan + exact fp32 re-scoring (shards of >= 524k rows); beside it the exact scan on the f32 MFMA (VQA_SKETCH=0)
    t0 = time.perf_counter()
    n32 = min(n, 1_000_000)
    ix32 = DeviceIndex.empty(n32, d, id_base=1, dtype="fp32", device=dev_index)
    host_rows = np.empty((n32, d), dtype=np.float32)  # every fp32 row, kept for the oracle (3.07 GB at 1M rows)

    def keep(c0, x):
        host_rows[c0:c0 + x.shape[0]] = x.cpu().numpy()

    fill(ix32, n32, keep)
    step_ms, kern_ms = timed_search(torch, ix32, q32, k, 20)
    info = ix32.launch_info(b, k)
    ex32 = DeviceIndex.empty(n32, d, id_base=1, dtype="fp32", device=dev_index, sketch=False)
    fill(ex32, n32)
    ex_step, ex_kern = timed_search(torch, ex32, q32, k, 20)
    ex_info = ex32.launch_info(b, k)
    tf = ex_info.flops_per_launch / (ex_kern * 1e-3) / 1e12
    s_a, _, p_a = ix32.search(q32, k, return_positions=True)
    s_b, _, p_b = ex32.search(q32, k, return_positions=True)
    torch.cuda.synchronize(device)
    same_rows = bool(torch.equal(p_a, p_b))
    ex32.close()
    # the oracle over ALL rows of the index that was timed (both paths: default = int8 sketch scan at this size, and the exact f32 scan)
    tv = time.perf_counter()
    ref_s, _, ref_p = R.search(q32[:nv].cpu().numpy(), host_rows, k, dtype=R.DTYPE_F32)
    verify_s = time.perf_counter() - tv
    del host_rows
    c1 = {"rows": n32, "queries_per_s": round(b / (step_ms * 1e-3), 1), "step_ms": round(step_ms, 4), "kernel_ms": round(kern_ms, 4),
          "sketch_scan": bool(info.sketch_scan),
          "exact_scan": {"queries_per_s": round(b / (ex_step * 1e-3), 1), "step_ms": round(ex_step, 4), "kernel_ms": round(ex_kern, 4),
                         "bound": "mfma", "mfma_tflops": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4),
                         "same_rows_as_default_path": same_rows,
                         "recall_at_10": R.recall_at_k(p_b[:nv].cpu().numpy(), ref_p)},
          "recall_at_10": R.recall_at_k(p_a[:nv].cpu().numpy(), ref_p),
          "recall_check": {"index": f"the timed index: all {n32} rows", "queries": nv, "rows": n32, "oracle": "same stored values (fp32 rows kept on the host during the fill)",
                           "max_abs_score_err": float(np.abs(s_a[:nv].cpu().numpy() - ref_s).max()), "oracle_seconds": round(verify_s, 1)}}
    if d == 768:
        enc, ids, mask, _, _ = make_encoder(torch, device, dev_index, b, 32)
        for _ in range(3):
            ix32.search(enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=0), k)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
              for _ in range(10)]
        for e0, e1, e2 in ev:
            e0.record()
            qv = enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=0)
            e1.record()
            ix32.search(qv, k)
            e2.record()
        torch.cuda.synchronize(device)
        enc.close()
        em = sorted(e0.elapsed_time(e1) for e0, e1, _ in ev)[5]
        tm = sorted(e0.elapsed_time(e2) for e0, _, e2 in ev)[5]
        c1.update({"encoder_padded_ms": round(em, 4), "end_to_end_ms": round(tm, 4), "end_to_end_queries_per_s": round(b / (tm * 1e-3), 1)})
    c1["seconds"] = round(time.perf_counter() - t0, 1)
    ix32.close()
    out["fp32_1M_plus_encoder"] = c1
    # ---- configs[0]: 1k x 768 random embeddings, cosine top-10 through the txtai-shaped API (wall clock, Python included)
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((1000, d)).astype(np.float32)
    q0 = rng.standard_normal((b, d)).astype(np.float32)
    emb = Embeddings(dtype="fp32", device=dev_index)
    emb.index_vectors(list(range(1, 1001)), x0)
    emb.batchsearch(q0, 10)
    t0 = time.perf_counter()
    for _ in range(20):
        res = emb.batchsearch(q0, 10)
    t_batch = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for j in range(50):
        emb.search(q0[j % b], 10)
    t_one = (time.perf_counter() - t0) / 50
    _, ref_ids, _ = R.search(R.l2_normalize(q0), R.l2_normalize(x0), 10, dtype=R.DTYPE_F32, id_base=1)
    got = np.array([[t[0] for t in r] + [-1] * (10 - len(r)) for r in res])
    out["config0_api_1k"] = {"batch256_ms": round(t_batch * 1e3, 3), "single_query_ms": round(t_one * 1e3, 3),
                             "recall_at_10": R.recall_at_k(got, ref_ids)}
    emb._index.close()
    out["latency"] = latency_legs(torch, np, device, dev_index, d)
    if d == 768:
        out["non_isotropic"] = non_isotropic_legs(torch, np, device, dev_index, b, k)
    out["reference_model_shapes"] = reference_model_shapes(torch, np, device, dev_index, b)
    return out


def latency_legs(torch, np, device, dev_index, d):
    """The reference's real calling pattern (heavy_ranker.py:97-101): ONE question per call, limit = 1, a corpus of thousands of
    documents.  Wall clock of `Embeddings.search(vector, 1)` (Python included; vqa_index_search_host underneath: no torch tensor, no
    copy operation, polled completion) at 1k / 5k / 50k documents, checked against the oracle; and the question encoder alone on one
    32-token question (event time of the replayed graph)."""
    from oracle import retrieval as R
    from vietnamese_qa_system_amd.embeddings import Embeddings
    rng = np.random.default_rng(3)
    out = {}
    for rows in (1000, 5000, 50_000):
        x = rng.standard_normal((rows, d)).astype(np.float32)
        qs = rng.standard_normal((64, d)).astype(np.float32)
        emb = Embeddings(dtype="fp16", device=dev_index)
        emb.index_vectors(list(range(1, rows + 1)), x)
        for j in range(10):
            emb.search(qs[j], 1)
        ts, got = [], []
        for j in range(200):
            t0 = time.perf_counter()
            r = emb.search(qs[j % 64], 1)
            ts.append(time.perf_counter() - t0)
            if j < 64:
                got.append(r[0][0] if r else -1)
        stored = emb._index.get_rows()[0]
        _, ref_ids, _ = R.search(R.l2_normalize(qs).astype(np.float32), stored, 1, dtype=R.DTYPE_F16, id_base=1)
        ts = np.sort(np.asarray(ts)) * 1e3
        out[f"search_one_vector_limit1_{rows}_docs_ms"] = {"median": round(float(np.median(ts)), 4), "p10": round(float(ts[20]), 4),
                                                          "p90": round(float(ts[180]), 4), "top1_equals_oracle": bool(got == ref_ids[:, 0].tolist())}
        emb._index.close()
    if d == 768:
        enc, ids, mask, _, _ = make_encoder(torch, device, dev_index, 1, 32, max_tokens=64)
        for _ in range(5):
            enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for e0, e1 in ev:
            e0.record()
            enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
            e1.record()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(30):
            enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0)
        torch.cuda.synchronize(device)
        wall = (time.perf_counter() - t0) / 30
        out["encoder_one_question_32_tokens_ms"] = {"event_median": round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4),
                                                    "wall_back_to_back": round(wall * 1e3, 4)}
        # ... and the whole text call of heavy_ranker.py:98: `search(question, 1)` = tokenizer (a stand-in: a table look-up) -> the
        # encoder's host entry -> the index's host-result entry, wall clock with Python, 5000 documents
        from vietnamese_qa_system_amd.encoder import TextEncoder
        ids_h, mask_h = ids.cpu().numpy(), mask.cpu().numpy()
        te = TextEncoder(lambda texts: (np.repeat(ids_h, len(texts), 0), np.repeat(mask_h, len(texts), 0)), enc, pooling="mean")
        x = rng.standard_normal((5000, d)).astype(np.float32)
        emb = Embeddings(dtype="fp16", device=dev_index, encoder=te)
        emb.index_vectors(list(range(1, 5001)), x)
        for _ in range(10):
            emb.search("cau hoi", 1)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            hit = emb.search("cau hoi", 1)
            ts.append(time.perf_counter() - t0)
        qv = enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=0).cpu().numpy()
        stored = emb._index.get_rows()[0]
        _, ref_ids, _ = R.search(R.l2_normalize(qv).astype(np.float32), stored, 1, dtype=R.DTYPE_F16, id_base=1)
        ts = np.sort(np.asarray(ts)) * 1e3
        out["search_one_text_question_limit1_5000_docs_ms"] = {"median": round(float(np.median(ts)), 4), "p10": round(float(ts[10]), 4),
                                                              "p90": round(float(ts[90]), 4),
                                                              "top1_equals_oracle_on_the_encoded_vector": bool(hit and hit[0][0] == int(ref_ids[0, 0]))}
        emb._index.close()
        enc.close()
        out["one_question_through_both_reference_models_ms"] = both_reference_models(torch, np, device, dev_index)
    return out


def both_reference_models(torch, np, device, dev_index, L=32):
    """The reference's loop body (heavy_ranker.py:98-101): ONE question through its two retrievers -- paraphrase-multilingual-MiniLM-L12-v2
    (d = 384) and paraphrase-multilingual-mpnet-base-v2 (XLM-R base, d = 768) by their shapes, random weights, a stand-in tokenizer
    (32 tokens), 5000 documents each, limit 1; wall clock with Python: the two `search(question, 1)` calls one after the other, and
    `heavy_ranker.rank_query` (the two forwards on a stream each)."""
    from vietnamese_qa_system_amd import heavy_ranker
    from vietnamese_qa_system_amd.embeddings import Embeddings
    from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE, TextEncoder
    rng = np.random.default_rng(5)
    embs, encs = [], []
    for cfg in (MINILM_L12, XLMR_BASE):
        enc, ids, mask, _, _ = make_encoder(torch, device, dev_index, 1, L, max_tokens=2 * L, cfg=cfg)
        ids_h = ids.cpu().numpy()
        ids_h[0, 1:L - 1] = rng.integers(3, cfg["vocab_size"], L - 2)
        ids_h[0, L - 1] = 2
        mask_h = np.ones_like(ids_h)
        te = TextEncoder(lambda texts, i=ids_h, m=mask_h: (np.repeat(i, len(texts), 0), np.repeat(m, len(texts), 0)), enc, pooling="mean")
        emb = Embeddings(dtype="fp16", device=dev_index, encoder=te)
        emb.index_vectors(list(range(1, 5001)), rng.standard_normal((5000, cfg["hidden"])).astype(np.float32))
        embs.append(emb)
        encs.append(enc)
    a, b = embs
    res = {}
    same = heavy_ranker.rank_query(a, b, "cau hoi", 1) == (a.search("cau hoi", 1), b.search("cau hoi", 1))
    for name, fn in (("two_searches_in_turn", lambda: (a.search("cau hoi", 1), b.search("cau hoi", 1))),
                     ("rank_query_two_streams", lambda: heavy_ranker.rank_query(a, b, "cau hoi", 1))):
        for _ in range(10):
            fn()
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        res[name] = round(float(np.median(ts)) * 1e3, 4)
    res["same_results"] = bool(same)
    for emb, enc in zip(embs, encs):
        emb._index.close()
        enc.close()
    torch.cuda.empty_cache()
    return res


def reference_model_shapes(torch, np, device, dev_index, b, L=32):
    """Forward time of the two encoders the reference's retriever loads (heavy_ranker.py:80,83), by their published shapes, random-init
    weights: paraphrase-multilingual-MiniLM-L12-v2 (BERT, hidden 384, 12 heads of 32, FFN 1536) and paraphrase-multilingual-mpnet-base-v2
    (XLM-R base, vocab 250 002).  B x L = 256 x 32 ragged questions (packed), mean pooling + L2 norm; and one 128-token batch."""
    from vietnamese_qa_system_amd.encoder import MINILM_L12, XLMR_BASE
    out = {}
    for name, cfg in (("minilm_l12_h384_dh32", MINILM_L12), ("xlmr_base_h768_dh64", XLMR_BASE)):
        enc, ids, mask, lens, g = make_encoder(torch, device, dev_index, b, L, max_tokens=64 * 128, cfg=cfg)
        h, f, layers = cfg["hidden"], cfg["ffn"], cfg["layers"]

        def timed(ids, mask, lens, reps=12):
            real = int(lens.sum())
            for _ in range(3):
                enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=real)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for e0, e1 in ev:
                e0.record()
                enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=real)
                e1.record()
            torch.cuda.synchronize(device)
            ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
            lf = lens.double()
            flops = layers * (float(lf.sum()) * 2 * (4 * h * h + 2 * h * f) + float((lf * lf).sum()) * 4 * h)
            return {"forward_ms": round(ms, 4), "real_tokens": real, "sequences": int(ids.shape[0]), "max_len": int(ids.shape[1]),
                    "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "frac_of_f16_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4)}

        out[name] = {"questions_256x32": timed(ids, mask, lens)}
        ids2, mask2, lens2 = make_tokens(torch, device, g, cfg, 64, 128)
        out[name]["passages_64x128"] = timed(ids2, mask2, lens2)
        # one 32-token question (heavy_ranker.py:98 asks one at a time): the latency form of the encoder, event time of the replayed graph
        ids1, mask1 = ids[:1].clone(), torch.ones_like(mask[:1])
        ids1[0, 1:L - 1] = torch.randint(3, cfg["vocab_size"], (L - 2,), generator=g, device=device, dtype=torch.int32)
        ids1[0, L - 1] = 2
        for _ in range(5):
            enc.forward(ids1, mask1, pooling="mean", normalize=True, real_tokens=0)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for e0, e1 in ev:
            e0.record()
            enc.forward(ids1, mask1, pooling="mean", normalize=True, real_tokens=0)
            e1.record()
        torch.cuda.synchronize(device)
        out[name]["one_question_32_tokens_ms"] = round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4)
        enc.close()
        torch.cuda.empty_cache()
    return out


def non_isotropic_legs(torch, np, device, dev_index, b, k):
    """The sketch search OFF the benchmark's isotropic Gaussian corpus (its best case), driver-visible and bounded (~15 s):
    (1) rows produced by this build's own PhoBERT-base-shaped encoder (random-init weights, mean pooling, L2 norm) from 1.57M synthetic
    token sequences -- the one anisotropic embedding source available offline --, queries encoded the same way;
    (2) 3M rows in 1000 clusters (within-cluster sigma 0.3 of the centre norm), queries drawn like the rows;
    (3) 10M rows that share one common component (mean cosine 0.9): collapsed embeddings at the size the metric is quoted on.
    Per leg: candidate pairs, whether the search stayed on the sketch (sketch_state 0), sketch search vs exact scan of the same shard
    (queries/s), same rows as the exact scan."""
    from vietnamese_qa_system_amd.index import DeviceIndex
    d = 768
    legs = {}

    def compare(x, q, extra):
        t0 = time.perf_counter()
        ske = DeviceIndex(x, id_base=1, dtype="fp16", device=dev_index, sketch=True)
        build_s = time.perf_counter() - t0
        s1, _, p1 = ske.search(q, k, return_positions=True)
        torch.cuda.synchronize(device)
        stats, state_first = ske.sketch_stats(), ske.sketch_state()
        per_row = bool(ske.sketch_split(0)[3])
        sk_ms, sk_kern = timed_search(torch, ske, q, k, 20)
        state = ske.sketch_state()
        li = ske.launch_info(b, k)
        ske.close()
        ref = DeviceIndex(x, id_base=1, dtype="fp16", device=dev_index, sketch=False)
        s0, _, p0 = ref.search(q, k, return_positions=True)
        ex_ms, _ = timed_search(torch, ref, q, k, 20)
        ref.close()
        diff = p0 != p1
        n = int(x.shape[0])
        xs = x[:: max(1, n // 4096)].float()
        cen = xs.mean(0)
        return {"rows": n, "sketch_scan": bool(li.sketch_scan),
                # "per-row": rows collapsed onto their centre direction (||centroid||^2 >= 0.6 of the mean ||x||^2) -- rows and queries are projected off it before
                # they are sketched and the scan adds the rank-one term per (query, row); "centre-split": every other centred shard
                "bound_form": "per-row" if per_row else "centre-split",
                "candidate_pairs_main_scan": stats["last_scan_pairs"],
                "rescored_pairs": stats["rescored_pairs"], "pairs_per_query": round(stats["rescored_pairs"] / b, 1),
                "overflow_first_search": stats["overflow"], "sketch_state_after_first_search": state_first,
                "sketch_state_after_25_searches": state,
                "sketch_search": {"step_ms": round(sk_ms, 4), "queries_per_s": round(b / (sk_ms * 1e-3), 1), "main_launch_ms": round(sk_kern, 4)},
                "exact_scan": {"step_ms": round(ex_ms, 4), "queries_per_s": round(b / (ex_ms * 1e-3), 1)},
                "speedup_over_exact": round(ex_ms / sk_ms, 3),
                "same_rows_as_exact": bool(not diff.any()), "rows_differing": int(diff.sum()),
                "max_abs_score_diff": float((s0 - s1).abs().max()),
                "max_score_gap_where_rows_differ": float((s0 - s1).abs()[diff].max()) if diff.any() else 0.0,
                # how far from isotropic: mean cosine of a row to the corpus centroid direction, norm of the centroid of unit rows
                "mean_cosine_to_centroid": round(float((xs @ (cen / cen.norm())).mean()), 4), "centroid_norm": round(float(cen.norm()), 4),
                "index_build_s": round(build_s, 2), **extra}

    # ---- (1) the build's own encoder
    t0 = time.perf_counter()
    S, L, n_enc = 1024, 32, 1_572_864
    enc, _, _, _, g = make_encoder(torch, device, dev_index, S, L, max_tokens=S * L)
    from vietnamese_qa_system_amd.encoder import PHOBERT_BASE
    x = torch.empty((n_enc, d), dtype=torch.float16, device=device)
    for c0 in range(0, n_enc, S):
        ids, mask, lens = make_tokens(torch, device, g, PHOBERT_BASE, S, L)
        x[c0:c0 + S] = enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=int(lens.sum())).to(torch.float16)
    ids, mask, lens = make_tokens(torch, device, g, PHOBERT_BASE, b, L)
    q = enc.forward(ids, mask, pooling="mean", normalize=True, real_tokens=int(lens.sum())).to(torch.float16)
    torch.cuda.synchronize(device)
    enc.close()
    enc_s = time.perf_counter() - t0
    legs["own_encoder_outputs"] = compare(x, q, {"source": f"{n_enc} synthetic token sequences (lengths 8-{L}) through this build's PhoBERT-base-shaped "
                                                          "encoder (random init), mean pooling, L2 norm, fp16 rows; queries encoded alike",
                                                "encode_s": round(enc_s, 1), "encode_docs_per_s": round(n_enc / enc_s, 1)})
    del x
    torch.cuda.empty_cache()
    # ---- (2) 1000 clusters
    t0 = time.perf_counter()
    gen = torch.Generator(device=device)
    gen.manual_seed(11)
    n_cl = 3_000_000
    centres = torch.randn((1000, d), generator=gen, device=device)
    centres /= centres.norm(dim=1, keepdim=True)

    def draw(m):
        v = centres[torch.randint(0, 1000, (m,), generator=gen, device=device)] + 0.3 * torch.randn((m, d), generator=gen, device=device) / d ** 0.5
        return (v / v.norm(dim=1, keepdim=True)).to(torch.float16)

    x = torch.cat([draw(1 << 19) for _ in range(0, n_cl, 1 << 19)])[:n_cl]
    q = draw(b)
    legs["clusters_1000"] = compare(x, q, {"source": "1000 cluster centres on the unit sphere + 0.3 / sqrt(d) N(0, 1) noise per component, L2-normalised; "
                                                     "queries drawn like the rows", "generate_s": round(time.perf_counter() - t0, 1)})
    del x
    torch.cuda.empty_cache()
    # ---- (3) collapsed embeddings at the headline size: one common component of mean cosine 0.9 (what the encoder leg shows at 1.57M rows,
    # synthetic here so that 10M rows take seconds): the per-row form of the bound at the size the metric is quoted on
    t0 = time.perf_counter()
    n_cc = 10_000_000
    common = torch.randn((1, d), generator=gen, device=device)
    common /= common.norm()

    def draw_cc(m):
        v = torch.randn((m, d), generator=gen, device=device)
        v = 3.0 * common + v / v.norm(dim=1, keepdim=True)
        return (v / v.norm(dim=1, keepdim=True)).to(torch.float16)

    x = torch.empty((n_cc, d), dtype=torch.float16, device=device)
    for c0 in range(0, n_cc, 1 << 19):
        c1 = min(n_cc, c0 + (1 << 19))
        x[c0:c1] = draw_cc(c1 - c0)
    q = draw_cc(b)
    legs["common_component_cos0p9_10M"] = compare(x, q, {"source": "3 x one fixed unit direction + a uniform unit vector, L2-normalised (mean cosine 0.9 between any two rows); "
                                                                   "queries drawn like the rows", "generate_s": round(time.perf_counter() - t0, 1)})
    return legs


def make_encoder(torch, device, dev_index, b, L, max_tokens=None, cfg=None):
    """PhoBERT-base-shaped encoder (or `cfg`) with seeded random weights + one synthetic ragged token batch (lengths uniform 8..L)."""
    from vietnamese_qa_system_amd.encoder import PHOBERT_BASE, QuestionEncoder
    cfg = cfg or PHOBERT_BASE
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    h, f = cfg["hidden"], cfg["ffn"]

    def w(*shape):
        return torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02

    weights = {"embeddings.word_embeddings.weight": w(cfg["vocab_size"], h), "embeddings.position_embeddings.weight": w(cfg["max_pos"], h),
               "embeddings.token_type_embeddings.weight": w(cfg["type_vocab"], h),
               "embeddings.LayerNorm.weight": torch.ones(h, device=device), "embeddings.LayerNorm.bias": torch.zeros(h, device=device)}
    for i in range(cfg["layers"]):
        p = f"encoder.layer.{i}."
        for name, shape in (("attention.self.query", (h, h)), ("attention.self.key", (h, h)), ("attention.self.value", (h, h)),
                            ("attention.output.dense", (h, h)), ("intermediate.dense", (f, h)), ("output.dense", (h, f))):
            weights[p + name + ".weight"] = w(*shape)
            weights[p + name + ".bias"] = w(shape[0])
        for ln in ("attention.output.LayerNorm", "output.LayerNorm"):
            weights[p + ln + ".weight"] = torch.ones(h, device=device)
            weights[p + ln + ".bias"] = torch.zeros(h, device=device)
    enc = QuestionEncoder(weights, cfg, device=dev_index, max_tokens=max_tokens or b * L)
    del weights
    ids, mask, lens = make_tokens(torch, device, g, cfg, b, L)
    return enc, ids, mask, lens, g


def make_tokens(torch, device, g, cfg, b, L):
    ids = torch.randint(3, cfg["vocab_size"], (b, L), generator=g, device=device, dtype=torch.int32)
    lens = torch.randint(8, L + 1, (b,), generator=g, device=device)
    pos = torch.arange(L, device=device)[None, :]
    mask = (pos < lens[:, None]).to(torch.int32)
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, cfg["pad_id"]))
    ids[:, 0] = 0
    ids[torch.arange(b, device=device), (lens - 1).to(torch.int64)] = 2
    return ids, mask, lens


def end_to_end(torch, np, searcher, device, dev_index, b, k, steps, sync, L=32, S=4):
    """Question encoder (PhoBERT-base shape, random-init weights, B x L = 256 x 32 synthetic token ids, CLS pooling +
    L2 normalisation on the device) -> search of the resident shard -> merge: the whole path of `embeddings.search`
    (heavy_ranker.py:98-101) per batch; then the same as a SUPER BATCH of S x B questions: one encoder forward over all
    of them (larger GEMMs quantise better on 256 CUs) and S scans back to back inside one search call."""
    from vietnamese_qa_system_amd.encoder import PHOBERT_BASE
    cfg = PHOBERT_BASE
    h, f = cfg["hidden"], cfg["ffn"]
    enc, ids, mask, lens, g = make_encoder(torch, device, dev_index, b, L, max_tokens=S * b * L)
    real = int(mask.sum())  # the mask is right-padded: the encoder computes only these rows (sequence packing)
    for _ in range(5):
        qv = enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=real)
        searcher.search(qv, k)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(steps)]
    sync()
    t0 = time.perf_counter()
    for e0, e1, e2 in ev:
        e0.record()
        qv = enc.forward(ids, mask, pooling="cls", normalize=True, real_tokens=real)
        e1.record()
        searcher.search(qv, k)
        e2.record()
    sync()
    el = time.perf_counter() - t0
    enc_ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
    tot_ms = [e0.elapsed_time(e2) for e0, _, e2 in ev]

    def flops_of(lens_t, nseq):
        # algorithmic flops of the REAL tokens (padding rows are not computed): GEMMs per token + attention over each sequence's own length
        lens_f = lens_t.double()
        rt = float(lens_f.sum())
        att = float((lens_f * lens_f).sum()) * 4 * h
        return int(cfg["layers"] * (rt * 2 * h * 3 * h + att) + (cfg["layers"] - 1) * rt * 2 * (h * h + 2 * h * f)
                   + nseq * 2 * (h * h + 2 * h * f))  # CLS pooling: the last layer's out-projection + FFN run on the first rows only

    # ---- super batch: S x B questions per call
    ids_s, mask_s, lens_s = [ids], [mask], [lens]
    for _ in range(S - 1):
        i2, m2, l2 = make_tokens(torch, device, g, cfg, b, L)
        ids_s.append(i2), mask_s.append(m2), lens_s.append(l2)
    ids_s, mask_s, lens_s = torch.cat(ids_s), torch.cat(mask_s), torch.cat(lens_s)
    real_s = int(mask_s.sum())
    ssteps = max(3, steps // S)
    for _ in range(3):
        searcher.search(enc.forward(ids_s, mask_s, pooling="cls", normalize=True, real_tokens=real_s), k)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ssteps)]
    sync()
    t1 = time.perf_counter()
    for e0, e1 in evs:
        e0.record()
        qv = enc.forward(ids_s, mask_s, pooling="cls", normalize=True, real_tokens=real_s)
        e1.record()
        searcher.search(qv, k)
    sync()
    el_s = time.perf_counter() - t1
    enc_s_ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in evs]))
    enc.close()
    flops = flops_of(lens, b)
    flops_s = flops_of(lens_s, S * b)
    enc_med = float(np.median(enc_ms))
    return {"workload": f"PhoBERT-base-shape question encoder (random init, B={b}, L={L}, {real} real tokens of {b * L}: lengths uniform "
                        f"8-{L}, right-padded, packed; CLS pooling, L2 norm) + search + merge",
            "steps": steps, "ms_per_batch": round(el / steps * 1e3, 4), "value": round(b * steps / el, 1), "unit": "queries/s",
            "batch_ms": percentiles(np, tot_ms), "encoder_ms": percentiles(np, enc_ms),
            "encoder_roofline": {"bound": "mfma", "achieved": round(flops / (enc_med * 1e-3) / 1e12, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(flops / (enc_med * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4),
                                 "flops_per_forward": flops, "note": "whole forward (GEMMs + attention + LayerNorm + pooling) over its median event time"},
            "super_batch": {"batches": S, "questions": S * b, "real_tokens": real_s, "calls": ssteps,
                            "ms_per_call": round(el_s / ssteps * 1e3, 4), "value": round(S * b * ssteps / el_s, 1), "unit": "queries/s",
                            "encoder_ms": round(enc_s_ms, 4),
                            "encoder_frac": round(flops_s / (enc_s_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, 4),
                            "note": f"ONE encoder forward over {S} x {b} questions + {S} scans back to back in one search call "
                                    "(what Embeddings.batchsearch does with more than 256 text queries)"}}


if __name__ == "__main__":
    main()
