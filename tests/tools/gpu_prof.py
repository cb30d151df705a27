"""Ad-hoc: cycle accounting of k_integrate from the LM_PROF build (loro_amd/csrc/libloromerge_prof.so)."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loro_amd._cabi import Binding, Context
from loro_amd import workload

so = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "loro_amd", "csrc", "libloromerge_prof.so")
b = Binding(so, "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for kv in sys.argv[2:]:
    k, _, v = kv.partition('=')
    os.environ[k] = v
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
if os.environ.get("GPU_PROF_BLOBS"):   # e.g. 02 = base + B's branch only (round 5: what the THIRD node — A against B's future items — costs = the difference)
    docs = [[b_[int(c)] for c in os.environ["GPU_PROF_BLOBS"]] for b_ in docs]
names = ["row", "find", "leaf", "oright", "between", "place", "delete", "checkout", "n_ins", "n_del", "n_leaf_loads", "n_between_items", "total", "n_ins_inside_run", "n_ins_merged", "n_upd_via_loc"]
with Context(b, 0) as e:
    print("selftest mismatches:", b.selftest(e.h))
    e.stage(docs)
    e.run(); e.run()
    t = time.time(); e.run(); dt = time.time() - t
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    tot = out[12]
    print("run %.1f ms; per-doc avg cycles (100MHz refclk?)" % (dt * 1e3))
    for i, n in enumerate(names):
        v = out[i] / n_docs
        print("  %-14s %12.0f  %s" % (n, v, ("%.1f%%" % (100.0 * out[i] / tot)) if i < 8 else ""))
