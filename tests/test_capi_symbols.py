"""The C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports exactly the
symbols include/vqa_retrieval.h declares.  No compute call is made here."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vqa_retrieval.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqa_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(native_lib):
    lib = ctypes.CDLL(native_lib)
    decl = declared_symbols()
    assert "vqa_index_search" in decl and "vqa_merge_topk" in decl and "vqa_encoder_forward" in decl
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_python_binding_lists_every_symbol(native_lib):
    from vietnamese_qa_system_amd import _native
    assert sorted(_native.EXPORTS) == declared_symbols()
    lib = _native.load()
    assert lib.vqa_version() == _native.VQA_VERSION == 113
    assert isinstance(lib.vqa_last_error(), bytes)


def test_exported_dynamic_symbols_are_only_ours(native_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", native_lib], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert set(declared_symbols()) <= exported


def test_code_object_targets_gfx950_only(native_lib, tmp_path):
    # llvm-objdump --offloading drops the extracted bundles next to its input: work on a copy outside the tree
    import shutil
    copy = shutil.copy(native_lib, tmp_path / "lib.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(copy)], capture_output=True, text=True,
                         cwd=tmp_path)
    archs = set(re.findall(r"gfx[0-9a-f]+", out.stdout + out.stderr))
    assert archs == {"gfx950"}, archs


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from vietnamese_qa_system_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _native.load()
    except FileNotFoundError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP library is missing")
