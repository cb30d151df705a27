"""Kernel-logic harness: the HIP kernels compiled for the host (tests/emu/lm_emu.cpp).  Tests only."""
import os, subprocess
from loro_amd._cabi import Binding, Context

_HERE = os.path.dirname(os.path.abspath(__file__))
_B = None


def binding():
    global _B
    if _B is None:
        so = os.path.join(_HERE, "emu", "libloroemu.so")
        src = os.path.join(_HERE, "emu", "lm_emu.cpp")
        csrc = os.path.join(os.path.dirname(_HERE), "loro_amd", "csrc")
        newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".h")])
        if not os.path.exists(so) or os.path.getmtime(so) < newest:
            tmp = f"{so}.{os.getpid()}.tmp"   # (built beside the target and renamed: another pytest-xdist worker never loads a half-written file)
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DLM_EMU_TRACE", "-o", tmp, src])
            os.replace(tmp, so)
        _B = Binding(so, "lmemu_")
    return _B


def variant(defines):
    """The harness built with extra -D defines (experiment builds of the kernels, e.g. LM_LOC16), cached beside the default one."""
    tag = "_".join(d.lower() for d in defines)
    so = os.path.join(_HERE, "emu", f"libloroemu_{tag}.so")
    src = os.path.join(_HERE, "emu", "lm_emu.cpp")
    csrc = os.path.join(os.path.dirname(_HERE), "loro_amd", "csrc")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc) if f.endswith(".h")])
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        tmp = f"{so}.{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-DLM_EMU_TRACE"] + ["-D" + d for d in defines] + ["-o", tmp, src])
        os.replace(tmp, so)
    return Binding(so, "lmemu_")


def merge_batch(docs, frontiers=None):
    with Context(binding()) as c:
        return c.merge_batch(docs, frontiers)
