"""The real N > 1 path on hardware: HIP shard search -> all-gather -> HIP merge, as ONE program.

The job is started as a fresh child process (``python -m torch.distributed.run``), never by re-executing this one.
* two and EIGHT ranks sharing cuda:0 over gloo -- run on the 1-GPU box (8 ranks = the world size of BASELINE configs[3]
  and configs[4]: ragged row counts, shards shorter than k, ties across three shard boundaries, fp8 / fp32 shards, the
  R * k = 8192 limit of the merge);
* ONE rank over nccl with VQA_ALWAYS_GATHER=1 -- every collective call of the path (all_gather_into_tensor of the packed
  candidate buffer, the asynchronous gather of search_pipelined, the all_reduce of a local-slice build, barriers of a
  sharded save) goes through RCCL on the 1-GPU box;
* two ranks on two GPUs over nccl (= RCCL over xGMI) -- lights up on a multi-GPU lease.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPECTED = {"id-vector", "pipelined", "k=300", "save-load", "local-slice", "producer", "short-shard", "short-shard-k12", "string-ids",
            "reindex", "fp8-shards", "fp8-k=40", "fp8-save-load", "fp32-shards", "fp32-k=40", "k=1024", "k=1024-short-shards",
            "hybrid-limit-103", "sketch-shards"}


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, extra, world=2, env_extra=None, preset_ipc=True):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sharded_worker.py"), "--out", str(tmp_path), *extra]
    env = dict(os.environ, PYTHONPATH=ROOT, **(env_extra or {}))
    if preset_ipc:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    else:
        env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-4000:] + "\n" + r.stderr[-6000:])
    recs = []
    for rank in range(world):
        with open(tmp_path / f"rank{rank}.json") as f:
            rec = json.load(f)
        assert set(rec["checks"]) == EXPECTED, rec
        assert rec["world"] == world
        recs.append(rec)
    return recs


def test_two_ranks_sharing_one_device(native_lib, tmp_path):
    _run(tmp_path, ["--backend", "gloo", "--share"])


def test_eight_ranks_sharing_one_device(native_lib, tmp_path):
    """BASELINE configs[3] / configs[4] run 8 ranks: the product program at that world size (one device, gloo)."""
    _run(tmp_path, ["--backend", "gloo", "--share"], world=8)


def test_one_rank_over_rccl_without_the_ipc_variable_preset(native_lib, tmp_path):
    """An external `torch.distributed.run` whose environment does not carry HSA_ENABLE_IPC_MODE_LEGACY: importing the
    package under a launcher sets it to 0 before the first GPU call, and the RCCL group of one rank runs every collective of the path."""
    recs = _run(tmp_path, ["--backend", "nccl"], world=1, env_extra={"VQA_ALWAYS_GATHER": "1"}, preset_ipc=False)
    assert recs[0]["backend"] == "nccl" and recs[0]["collectives"] >= 10 and recs[0]["ipc_mode_legacy"] == "0"


def test_one_rank_over_rccl(native_lib, tmp_path):
    """backend nccl = RCCL: a group of one rank executes every collective call site of the sharded path for real."""
    recs = _run(tmp_path, ["--backend", "nccl"], world=1, env_extra={"VQA_ALWAYS_GATHER": "1"})
    assert recs[0]["backend"] == "nccl" and recs[0]["collectives"] >= 10


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_two_devices_rccl(native_lib, tmp_path):
    _run(tmp_path, ["--backend", "nccl"])
