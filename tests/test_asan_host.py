"""AddressSanitizer + UBSan over the HOST side of the C ABI (argument checks, error plumbing, cleanup on failed creates).

The library's .hip sources are compiled with ``-fsanitize=address,undefined -fno-gpu-sanitize`` (host code instrumented, device
code as in the product) and linked with tests/asan/capi_errors.cpp; the program exercises every entry point's error
path and exits 0 only if each returned the documented code and the sanitizers reported nothing.  CPU only: GPU sanitizers
are not available on the pool, and without a device every ``*_create`` ends in ``VQA_ENODEV`` after its checks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_capi_error_paths_under_asan_ubsan(tmp_path):
    from concurrent.futures import ThreadPoolExecutor

    from vietnamese_qa_system_amd import build as B
    # host code instrumented, device code built as usual (-fno-gpu-sanitize): the same translation units as the product
    flags = ("-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g")
    os.makedirs(B.OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda src: B._compile(src, flags, "_asan"), B.sources()))
    exe = str(tmp_path / "capi_errors")
    cmd = [HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-g", "-std=c++20", "-I",
           os.path.join(ROOT, "include"), *objs, os.path.join(ROOT, "tests", "asan", "capi_errors.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=B._clean_env())
    assert r.returncode == 0, r.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")  # the error paths are the subject: keep the run off any GPU
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:] + r.stderr[-6000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
