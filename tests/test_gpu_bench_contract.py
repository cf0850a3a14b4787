"""bench.py's output contract on a small shard: one JSON line with the driver's keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys(native_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--docs-per-gpu", "300000", "--steps", "3", "--warmup", "1",
                          "--cpu-sample-rows", "100000", "--verify-queries", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "recall_at_10"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["dtype"] == "f16" and r["data"] == "synthetic" and "workload" in r["config"]
    assert r["value"] > 0 and abs(r["value"] - 256 / r["ms_per_step"] * 1e3) / r["value"] < 0.01
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches"] == 1  # the dominant launch is bracketed on every 4th step
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["cpu_model"]
    assert {(p["rows"], p["batch"]) for p in cb["points"]} >= {(1000, 256), (1000, 1), (100000, 256), (100000, 1)}
    assert r["recall_at_10"] == 1.0
    assert rf["step_bytes_moved"] >= 300000 * 768 * 2 and 0 < rf["step_frac_physical"] < 1 and "qps_over_hbm_roofline_qps" in rf
    oc = r["other_configs"]
    assert {"own_encoder_outputs", "clusters_1000"} <= set(oc["non_isotropic"]) and {"minilm_l12_h384_dh32", "xlmr_base_h768_dh64"} <= set(oc["reference_model_shapes"])
    # the other BASELINE configs are oracle-checked on the index that is timed, over all of its rows
    f8, f32 = oc["fp8_e4m3"], oc["fp32_1M_plus_encoder"]
    assert f8["recall_check"]["ok"] and f8["recall_check"]["stored_codes_equal_oracle_codes"] and f8["rows"] == 300000
    assert f8["recall_check"]["sampled_rows"] >= 1024 and f8["recall_check"]["queries"] == 256
    assert f32["recall_at_10"] == 1.0 and f32["recall_check"]["rows"] == f32["rows"] == 300000 and f32["recall_check"]["max_abs_score_err"] < 1e-5
    lat = oc["latency"]
    for rows in (1000, 5000, 50000):
        one = lat[f"search_one_vector_limit1_{rows}_docs_ms"]
        assert one["top1_equals_oracle"] and 0 < one["p10"] <= one["median"] <= one["p90"] < 5.0
    assert lat["encoder_one_question_32_tokens_ms"]["event_median"] > 0
    text = lat["search_one_text_question_limit1_5000_docs_ms"]
    assert text["top1_equals_oracle_on_the_encoded_vector"] and 0 < text["median"] < 5.0
    both = lat["one_question_through_both_reference_models_ms"]
    assert both["same_results"] and 0 < both["rank_query_two_streams"] < 5.0 and 0 < both["two_searches_in_turn"] < 5.0
    assert all(v["one_question_32_tokens_ms"] > 0 for v in oc["reference_model_shapes"].values())
    assert set(lat["300000_rows_fp16"]) == {"batch1_step_ms", "batch257_step_ms"}
    sm = r["step_ms"]
    assert sm["p10"] <= sm["median"] <= sm["p90"]
    e2e = r["end_to_end"]
    assert e2e["value"] > 0 and e2e["encoder_roofline"]["bound"] == "mfma" and 0 < e2e["encoder_roofline"]["frac"] < 1


def test_bench_two_ranks_started_by_bench_itself(native_lib):
    """`python bench.py --gpus 2` with NO launcher (how the driver starts the 1-GPU run, with N = 2): bench.py spawns its own
    torch.distributed.run child before touching the GPU, relays rank 0's one JSON line and the exit code.  On the 1-GPU box
    VQA_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 over gloo.  The row-sharded searcher, the all-gather, the merge, the
    max-over-ranks timing, the per-phase event times and the end-to-end leg on every rank all run."""
    # (HSA_ENABLE_IPC_MODE_LEGACY is NOT preset: bench.py sets it for itself and its ranks before anything touches the GPU)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(VQA_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--docs-per-gpu",
                          "200000", "--steps", "4", "--warmup", "2", "--verify-queries", "4", "--e2e-steps", "3"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(out.stdout.splitlines()) == 1  # stdout carries the JSON line and nothing else
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["docs_total"] == 400000 and r["scaling"] == "weak"
    assert r["recall_at_10"] == 1.0 and "cpu_baseline" not in r  # the CPU baseline is an N = 1 leg
    assert r["roofline"]["launches"] == 1 and r["end_to_end"]["value"] > 0
    mg = r["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"].startswith("gloo") and mg["device_count"] >= 1 and mg["collectives_per_step"] == 1
    assert mg["hsa_enable_ipc_mode_legacy"] == "0" and [t["rank"] for t in mg["ranks"]] == [0, 1] and mg["distinct_devices"] == 1
    assert mg["corpus_80M_on_one_gpu"]["ms_per_batch_sketch_path"] > 0
    pr = mg["per_rank_ms_per_step"]
    assert len(pr["by_rank"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - r["ms_per_step"]) < 1e-3
    for phase in ("local_search_ms", "gather_wait_ms", "merge_ms"):
        assert 0 <= mg["phase_ms"][phase]["min"] <= mg["phase_ms"][phase]["max"]
    assert mg["phase_ms"]["local_search_ms"]["rank0"] > 0


def test_bench_more_gpus_than_the_node_has_is_a_clear_error(native_lib):
    """A bare `--gpus 64` on a box with fewer devices (and no share flag) ends with a message and exit code 2 before any rank starts."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "VQA_BENCH_SHARE_GPU")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 2 and "GPU(s)" in out.stderr and out.stdout.strip() == ""


def test_bench_eight_ranks_sharing_the_device(native_lib):
    """The world size of BASELINE configs[3] / configs[4] as the driver would launch it on an 8-GPU node, with all eight ranks
    on cuda:0 (gloo): 8 x 200 000 rows, rank-strided merge of 8 candidate blocks, max-over-ranks timing, one JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # an EXTERNAL launcher, as the driver's, and no HSA_ENABLE_IPC_MODE_LEGACY in its environment: bench.py's own first lines set it
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env.update(VQA_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--docs-per-gpu",
                          "200000", "--steps", "4", "--warmup", "2", "--verify-queries", "4", "--e2e-steps", "2"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["docs_total"] == 1600000 and r["scaling"] == "weak"
    assert r["config"]["parallelism"] == "row-shard x8" and r["recall_at_10"] == 1.0 and "cpu_baseline" not in r
    assert r["multi_gpu"]["hsa_enable_ipc_mode_legacy"] == "0" and len(r["multi_gpu"]["ranks"]) == 8
    assert r["roofline"]["launches"] == 1 and r["end_to_end"]["value"] > 0
    assert r["pipelined"]["value"] > 0 and r["pipelined"]["batches"] >= 2
    mg = r["multi_gpu"]
    assert mg["world_size"] == 8 and len(mg["per_rank_ms_per_step"]["by_rank"]) == 8 and mg["collectives_per_step"] == 1
