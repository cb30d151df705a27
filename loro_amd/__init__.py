"""loro_amd — MI355X-native batched CRDT merge engine for Loro's import → diff_calc → state path.

The product is the C-ABI shared library built from loro_amd/csrc (HIP, gfx950): `libloromerge.so`,
declared in include/loro_merge.h.  This package is only the Python binding used by tests and bench.py;
it mirrors the reference's host-side calls for this path:

    LoroDoc::import_batch(blobs) ; doc.get_deep_value().to_json_value() ; doc.oplog_vv().encode()
    (crates/loro/src/lib.rs:710,887,937; crates/loro-internal/src/loro.rs:1432-1523)

There is no CPU fallback: `MergeEngine()` raises when the library or a HIP device is missing.
"""
import os
import subprocess

from ._cabi import Binding, Context

_ROOT = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_ROOT, "csrc")
LIB_PATH = os.environ.get("LORO_AMD_LIB") or os.path.join(_CSRC, "libloromerge.so")   # (LORO_AMD_LIB: an experiment build, tests/tools only)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

__all__ = ["build_library", "MergeEngine", "LIB_PATH", "merge_batch"]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(_CSRC, f)) > t for f in os.listdir(_CSRC) if f.endswith((".h", ".cpp")))


def build_library(force=False, defines=(), out=None):
    """hipcc --offload-arch=gfx950 → loro_amd/csrc/libloromerge.so (cross-compiles without a GPU).
    `defines` / `out`: an experiment build beside the product library, e.g. build_library(defines=["LM_LOC16"],
    out="tests/tools/ab/lib_loc16.so") for `tests/tools/gpu_ab.py ... --so` (NEXT.md §5)."""
    if out is None and not defines and not force and not _stale():
        return LIB_PATH
    target = os.path.abspath(out) if out else LIB_PATH
    os.makedirs(os.path.dirname(target), exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + ["-D" + d for d in defines] + [
           "-o", target, os.path.join(_CSRC, "lm_hip.cpp")]
    subprocess.check_call(cmd, cwd=_CSRC)
    return target


_BINDING = None


def _binding():
    global _BINDING
    if _BINDING is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is the only implementation; there is no CPU fallback)")
        _BINDING = Binding(LIB_PATH, "lm_")
    return _BINDING


class MergeEngine(Context):
    """One lm_ctx on HIP device `device`.  merge_batch(docs) takes a list of documents, each a list of
    update blobs imported in order, and returns [(status, json_bytes, vv_bytes, pending_ops)]."""

    def __init__(self, device=0):
        super().__init__(_binding(), device)


def merge_batch(docs, device=0):
    with MergeEngine(device) as e:
        return e.merge_batch(docs)
