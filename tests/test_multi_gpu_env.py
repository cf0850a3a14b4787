"""The multi-process GPU environment rule (vietnamese_qa_system_amd/sharded.py: ensure_multi_process_gpu_env): a process started by a
distributed launcher gets HSA_ENABLE_IPC_MODE_LEGACY=0 at package import -- RCCL's intra-node transport needs dmabuf IPC on this driver --
unless the caller set the variable; a plain process is left alone.  Runs on CPU (no GPU call is made)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import os, vietnamese_qa_system_amd; print(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset'))"


def _run(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY")}
    env.update(PYTHONPATH=ROOT, **extra)
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().splitlines()[-1]


def test_a_launched_rank_gets_dmabuf_ipc():
    assert _run(RANK="3", WORLD_SIZE="8") == "0"
    assert _run(RANK="0", WORLD_SIZE="1") == "0"  # a group of one rank makes the same RCCL calls


def test_the_callers_choice_stands():
    assert _run(RANK="0", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="1") == "1"


def test_a_plain_process_is_left_alone():
    assert _run() == "unset"


def test_bench_sets_it_before_torch_is_imported():
    """bench.py's main() sets the variable in its first lines, whatever started it (bare, self-spawned ranks, an external launcher)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < main.index("import torch")
