"""The product library: loads, exports every symbol include/loro_merge.h declares, and refuses to run
without a HIP device (no CPU fallback).  No compute calls here — the GPU tests do that."""
import ctypes, os, re
import pytest

import loro_amd

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "loro_merge.h")


@pytest.fixture(scope="module")
def lib():
    loro_amd.build_library()  # hipcc cross-compiles for gfx950 without a GPU
    return ctypes.CDLL(loro_amd.LIB_PATH)


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_[a-z_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ("lm_create", "lm_destroy", "lm_merge_batch", "lm_stage", "lm_run", "lm_fetch", "lm_result_meta", "lm_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    for s in declared_symbols():
        assert hasattr(lib, s), f"libloromerge.so does not export {s}"


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    lib.lm_create.restype = ctypes.c_void_p
    lib.lm_create.argtypes = [ctypes.c_int]
    assert not lib.lm_create(0), "lm_create must fail without a HIP device"
    with pytest.raises(RuntimeError):
        loro_amd.MergeEngine(0)


def test_python_binding_covers_the_abi():
    from loro_amd import _cabi
    for s in declared_symbols():
        assert s[3:] in _cabi.SYMBOLS, s
