"""CPU-side checks of the boundary: the HIP library loads without a GPU, exports every symbol that
include/*.h declare, and the ctypes mirror matches the compiled struct layouts.  No compute calls."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
abi = importlib.import_module("marl-ctf-development_amd._abi")


def declared_symbols():
    names = set()
    for header in ("ctf_env.h", "ctf_policy.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ctf_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return abi.load_library()


def test_header_and_binding_name_the_same_entry_points():
    assert declared_symbols() == sorted(abi.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(abi.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), f"libctf_hip.so does not export {name}"


def test_struct_mirrors_match_compiled_layout(lib):
    assert lib.ctf_abi_version() == abi.ABI_VERSION
    assert lib.ctf_sizeof_config() == ctypes.sizeof(abi.CtfConfig)
    assert lib.ctf_sizeof_state_view() == ctypes.sizeof(abi.CtfStateView)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(abi.CtfLibraryError):
        abi.load_library(str(tmp_path / "nope.so"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "marl-ctf-development_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "ctf_oracle" not in src, f


def test_only_tests_smoke_and_the_cpu_baseline_leg_touch_the_oracle():
    """oracle/ is test infrastructure: outside tests/ only __graft_entry__.smoke() and bench.py's cpu_baseline may use it."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    for dirpath, dirnames, files in os.walk(ROOT):
        dirnames[:] = [d for d in dirnames if d not in (".git", "gpurun_out", "tests", "oracle", "__pycache__", ".pytest_cache")]
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith((".py", ".sh", ".hip", ".h", ".cpp", ".c")) and path not in allowed:
                src = open(path, errors="replace").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M) and "ctf_oracle" not in src, path
