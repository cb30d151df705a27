"""Ad-hoc: cycle accounting (LM_PROF build) of the integrate stage of an IMPORT run — base + A's branch resident, B's concurrent
branch imported (configs[1]) — next to the from-empty replay of all three blobs.   python tests/tools/gpu_prof_res.py [n_docs]"""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loro_amd
from loro_amd._cabi import Binding, Context
from loro_amd import workload

so = os.path.join(ROOT, "loro_amd", "csrc", "libloromerge_prof.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(loro_amd.LIB_PATH):
    loro_amd.build_library(defines=["LM_PROF"], out=so)
b = Binding(so, "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
names = ["row", "find", "leaf", "oright", "between", "place", "delete", "checkout", "n_ins", "n_del", "n_leaf_loads", "n_between_items", "total", "n_ins_inside_run", "n_ins_merged", "n_upd_via_loc"]


def show(e, what, dt):
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    tot = out[12]
    print("%s: run %.1f ms; per-document averages (cycle counter ticks)" % (what, dt * 1e3))
    for i, n in enumerate(names):
        v = out[i] / n_docs
        if n == "leaf":
            print("  %-16s fast inserts %8.0f   fast updates %8.0f" % (n, (out[i] >> 20) / n_docs, (out[i] & 0xfffff) / n_docs))
        else:
            print("  %-16s %12.0f  %s" % (n, v, ("%.1f%%" % (100.0 * out[i] / tot)) if i < 8 else ""))


with Context(b, 0) as e:
    e.stage(docs); e.run(); e.run()
    t = time.time(); e.run(); show(e, "batch (from the empty version, k_integrate_span_plain_sweep)", time.time() - t)
    for rep in range(2):
        e.stage([x[:2] for x in docs]); e.import_more([[] for _ in docs])
        t = time.time(); e.run(); dt0 = time.time() - t
        if rep: show(e, "resident, base + A from the empty version", dt0)
        e.import_more([x[2:] for x in docs])
        t = time.time(); e.run(); dt1 = time.time() - t
        if rep: show(e, "resident, import of B", dt1)
