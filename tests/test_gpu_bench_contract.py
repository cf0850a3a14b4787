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
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["launches"] == 3
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert r["recall_at_10"] == 1.0
