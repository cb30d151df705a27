"""Ad-hoc: where k_dag_a spends its time on the heterogeneous configs[1] batch (two of seven shapes hold 10k / 40k one-row changes) —
ticks per pass from an LM_PROF_DAG build
(python -c "import loro_amd; loro_amd.build_library(defines=['LM_PROF_DAG'], out='tests/tools/ab/lib_prof_dag.so')")."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
from loro_amd._cabi import Binding, Context
from loro_amd import workload

def gen(sh):
    n_base, n_branch, every, fuse = sh
    return workload.Cfg2Template(n_base, n_branch, seed=n_base % 97, commit_every=every, fuse=fuse)

b = Binding(os.path.join(ROOT, "tests", "tools", "ab", "lib_prof_dag.so"), "lm_")
b.lib.lm_prof_sum.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
only_big = len(sys.argv) > 2 and sys.argv[2] == "keystroke"
shapes = [(2000, 1000, 10, True), (10000, 5000, 10, True), (25000, 12500, 10, True), (50000, 25000, 10, True),
          (100000, 50000, 10, True), (5000, 2500, 1, False), (20000, 10000, 1, False)]
if only_big:
    shapes = shapes[-1:]
with mp.get_context("fork").Pool(7) as pool:
    tpls = pool.map(gen, shapes)
docs = [tpls[(d * 7919) % len(tpls)].stamp(d) for d in range(n_docs)]
names = ["block order", "coverage walk (drop known, slice, park)", "dependency fixpoint", "compaction to applied changes", "per-peer ranges, element bases",
         "dependencies -> changes (dep_ci)", "node numbering", "row counts, flags"]
with Context(b, 0) as e:
    e.stage(docs)
    e.run(); e.run()
    out = (ctypes.c_uint64 * 16)()
    b.lib.lm_prof_sum(e.h, out)
    tot = sum(out[i] for i in range(8))
    print("%d documents%s: k_dag_a ticks by pass (s_memtime, summed over waves)" % (n_docs, " (40k one-row changes each)" if only_big else " (heterogeneous batch)"))
    for i, n in enumerate(names):
        print("  %-44s %14d  %5.1f%%" % (n, out[i], 100.0 * out[i] / max(tot, 1)))
