"""Ad-hoc: where k_block_decode_wave spends its time — cycle accounting per phase from an LM_PROF_DEC build
(python -c "import loro_amd; loro_amd.build_library(defines=['LM_PROF_DEC'], out='tests/tools/ab/lib_prof_dec.so')")."""
import sys, os, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loro_amd._cabi import Binding, Context
from loro_amd import workload
b = Binding(os.path.join(ROOT, "tests", "tools", "ab", "lib_prof_dec.so"), "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
which = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
if which == "cfg3":
    base = [workload.cfg3_doc(d, combined=(d % 2 == 0)) for d in range(4)]
    docs = [base[i % 4] for i in range(n_docs)]
elif which == "key":   # one change per keystroke: ~200 changes per block
    tpl = workload.Cfg2Template(20000, 10000, seed=0, commit_every=1, fuse=False)
    docs = [tpl.stamp(d) for d in range(n_docs)]
else:
    tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
    docs = [tpl.stamp(d) for d in range(n_docs)]
names = ["stage (HBM->LDS)", "head (role 0)", "cursors", "A1 op columns", "T + A2 delete columns", "W value walker", "B rows out", "close"]
with Context(b, 0) as e:
    e.stage(docs)
    e.run(); e.run()
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    tot = sum(out[i] for i in range(8))
    print("%s, %d docs: decoder phase shares (s_memtime ticks summed over waves)" % (which, n_docs))
    for i, n in enumerate(names):
        print("  %-24s %14d  %5.1f%%" % (n, out[i], 100.0 * out[i] / max(tot, 1)))
