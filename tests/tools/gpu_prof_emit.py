"""Ad-hoc: where k_emit_any / k_emit_text spend their time — ticks per kind of container from an LM_PROF_EMIT build
(python -c "import loro_amd; loro_amd.build_library(defines=['LM_PROF_EMIT'], out='tests/tools/ab/lib_prof_emit.so')")."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
import _cases
b = Binding(os.path.join(ROOT, "tests", "tools", "ab", "lib_prof_emit.so"), "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
if which == "cfg3":
    base = [workload.cfg3_doc(d, combined=(d % 2 == 0)) for d in range(4)]
elif which == "cfg4":
    base = _cases.cfg4_docs(96)
else:
    tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
    base = [tpl.stamp(d) for d in range(64)]
docs = [base[i % len(base)] for i in range(n_docs)]
names = ["set-up (roots, order)", "text", "list: everything around the values", "map: key sort", "map: entries", "version vector + close", "root names", "list: the items' values (sink_value)"]
with Context(b, 0) as e:
    e.stage(docs)
    e.run(); e.run()
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    tot = sum(out[8 + i] for i in range(8))
    print("%s, %d docs: renderer time by part (s_memtime ticks summed over waves)" % (which, n_docs))
    vt = ["null / bool", "i64", "f64", "string", "binary", "list", "map", "other"]
    tv = sum(out[i] for i in range(8))
    print("  top-level values by kind (sink_value, ticks):")
    for i, n in enumerate(vt):
        print("    %-22s %14d  %5.1f%%" % (n, out[i], 100.0 * out[i] / max(tv, 1)))
    for i, n in enumerate(names):
        print("  %-24s %14d  %5.1f%%" % (n, out[8 + i], 100.0 * out[8 + i] / max(tot, 1)))
