"""GPU box: where k_map_fused's time goes — the kernel stopped after each phase (LM_MF_STOP; the results of such runs are not the
documents', only the stage time is read).   python tests/tools/gpu_mf_phases.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import loro_amd
from loro_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = [workload.cfg3_doc(d, combined=(d % 2 == 0)) for d in range(8)]
docs = [base[i % 8] for i in range(n)]
for stop in ("1", "2", "3", "4", "0"):
    os.environ["LM_MF_STOP"] = stop
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs); e.run(); e.set_profiling(True); e.run()
        agg = {}
        for name, ms in e.kernel_times(): agg[name] = agg.get(name, 0) + ms
        print("stop after phase", stop, "k_map_lww stage %.2f ms" % agg.get("k_map_lww", 0), "fused", e.b.fused_documents(e.h), flush=True)
if len(sys.argv) > 2:   # an -DLM_PROF_MF build: ticks per part of the kernel
    import ctypes
    from loro_amd._cabi import Binding, Context
    os.environ["LM_MF_STOP"] = "0"
    b = Binding(sys.argv[2], "lm_")
    with Context(b, 0) as e:
        e.stage(docs); e.run(); e.run()
        out = (ctypes.c_uint64 * 16)()
        b.lib.lm_prof_sum(ctypes.c_void_p(e.h), out)
        names = ["block set-up", "keys", "op header + small columns", "prop: staging", "prop: heads", "prop: runs", "prop: literals", "prop: prefix", "values", "row columns / changes", "table"]
        tot = sum(out[i] for i in range(11)) or 1
        for i, nm in enumerate(names): print("  %-28s %6.2f %%" % (nm, 100.0 * out[i] / tot))
