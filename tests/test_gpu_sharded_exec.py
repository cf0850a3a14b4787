"""The real N > 1 path on hardware: HIP shard search -> all-gather -> HIP merge, as ONE program.

The job is started as a fresh child process (``python -m torch.distributed.run``), never by re-executing this one.
* two ranks sharing cuda:0 over gloo -- runs on the 1-GPU box;
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
EXPECTED = {"id-vector", "pipelined", "k=300", "save-load", "local-slice", "producer", "short-shard", "short-shard-k12", "string-ids", "reindex"}


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(tmp_path, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "sharded_worker.py"), "--out", str(tmp_path), *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-4000:] + "\n" + r.stderr[-6000:])
    for rank in (0, 1):
        with open(tmp_path / f"rank{rank}.json") as f:
            rec = json.load(f)
        assert set(rec["checks"]) == EXPECTED, rec


def test_two_ranks_sharing_one_device(native_lib, tmp_path):
    _run(tmp_path, ["--backend", "gloo", "--share"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_two_devices_rccl(native_lib, tmp_path):
    _run(tmp_path, ["--backend", "nccl"])
