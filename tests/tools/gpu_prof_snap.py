"""Ad-hoc: cycle accounting of the integrate stage (LM_PROF build, loro_amd/csrc/libloromerge_prof.so) on snapshot + updates documents."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loro_amd._cabi import Binding, Context
from loro_amd import workload

so = os.path.join(ROOT, "loro_amd", "csrc", "libloromerge_prof.so")
b = Binding(so, "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [workload.cfg2_snapshot_doc(s) for s in range(4)]
names = ["row", "find", "leaf", "oright", "between", "place", "delete", "checkout", "n_ins", "n_del", "n_leaf_loads", "n_between_items", "total", "n_ins_inside_run", "n_ins_merged", "n_upd_via_loc"]
for label, env in (("state", {}), ("history", {"LM_SNAPSHOT_STATE": "0"})):
    os.environ.update(env)
    docs = [list(base[i % 4]) for i in range(n_docs)]
    with Context(b, 0) as e:
        e.stage(docs)
        e.run(); e.run()
        t = time.time(); e.run(); dt = time.time() - t
        out = (ctypes.c_uint64 * 16)()
        b.lib.lm_prof_sum(e.h, out)
        tot = out[12]
        print("%s: run %.1f ms; per-doc avg ticks" % (label, dt * 1e3))
        for i, n in enumerate(names):
            v = out[i] / n_docs
            print("  %-16s %12.0f  %s" % (n, v, ("%.1f%%" % (100.0 * out[i] / max(tot, 1))) if i < 8 else ""))
    for k in env:
        del os.environ[k]
