"""Builds libvqa_retrieval.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m vietnamese_qa_system_amd.build [--force]

The shared library is written to ``vietnamese_qa_system_amd/lib/`` so it travels with the source tree to the
GPU box; it is git-ignored (``*.so``).  hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import fcntl
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
LIB_PATH = os.path.join(LIB_DIR, "libvqa_retrieval.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++20", "-I", INCLUDE, "-I", CSRC, "-Wall", "-Wno-unused-function"]


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    files = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    files += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return max(os.path.getmtime(f) for f in files)


_TOOL_ENV_PREFIXES = ("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTX", "RPD_")


def _clean_env() -> dict:
    """Environment for the hipcc child: a profiler's preload / tool variables are dropped.  Under ``rocprofv3`` the
    preloaded tool library initialises the GPU in every process it is loaded into, and hipcc then ``exec``s clang --
    an exec from a GPU-initialised process, which takes a machine of this pool down.  The compiler needs none of them."""
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(_TOOL_ENV_PREFIXES)}
    return env


def under_profiler() -> bool:
    return "LD_PRELOAD" in os.environ or any(k.startswith(_TOOL_ENV_PREFIXES) for k in os.environ)


def is_fresh() -> bool:
    return os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _deps_mtime()


def _compile(src: str, extra=(), tag: str = "") -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + tag + ".o")
    cmd = [HIPCC, *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env())
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    """Compile every .hip under csrc/ for gfx950 and link the C-ABI shared library.  Returns its path."""
    if not force and is_fresh():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time (the ranks of a torch.distributed.run job all call this before they touch the GPU)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_fresh():
                return LIB_PATH
            with ThreadPoolExecutor(max_workers=4) as ex:
                objs = list(ex.map(_compile, sources()))
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp, *objs]
            r = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env())
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
            os.replace(tmp, LIB_PATH)  # a reader never sees a half-written library
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def build_variant(tag: str, defines: list[str], only: tuple[str, ...] = ()) -> str:
    """Dev helper: a second copy of the library with extra -D flags (timing ablations), lib/libvqa_retrieval_<tag>.so.
    ``only``: source basenames the flags apply to; the other objects are the default build's (built first if stale)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    # a "define" that starts with '-' is passed to hipcc as it is (compiler-flag experiments)
    extra = []
    for d in defines:
        extra += d.split() if d.startswith("-") else [f"-D{d}"]
    if only:
        build()
        objs = [_compile(src, tuple(extra), "_" + tag) if os.path.basename(src) in only
                else os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o") for src in sources()]
    else:
        objs = [_compile(src, tuple(extra), "_" + tag) for src in sources()]
    out = os.path.join(LIB_DIR, f"libvqa_retrieval_{tag}.so")
    r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out, *objs], capture_output=True, text=True,
                       env=_clean_env())
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
