"""Ad-hoc: LM_PROF cycle accounting of k_integrate_span on configs[4]-shaped documents (deep history, hand-over peers, checkouts)."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
so = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "loro_amd", "csrc", "libloromerge_prof.so")
b = Binding(so, "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
blobs, fr = workload.cfg5_doc(0, n_ops=n_ops, turn=1000, n_checkouts=16)
docs = [blobs] * (len(fr) + 1)
fronts = fr + [None]
names = ["row", "find", "leaf(fast upd | fast ins<<20)", "oright", "between", "place", "delete", "checkout", "n_ins", "n_del", "n_leaf_loads", "n_between_items", "total", "n_ins_inside_run", "n_ins_merged", "n_upd_via_loc"]
with Context(b, 0) as e:
    e.stage(docs, fronts)
    e.run()
    t = time.time(); e.run(); dt = time.time() - t
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    print("run %.1f ms for %d renderings of a %d-op document; sizing %s" % (dt * 1e3, len(docs), n_ops, e.sizing()))
    for i, n in enumerate(names):
        print("  %-30s %14.0f  %s" % (n, out[i] / len(docs), ("%.1f%%" % (100.0 * out[i] / out[12])) if i < 8 else ""))
