"""GPU box: BASELINE configs[4] (1M-op documents x 16 checkouts), batch path: ms per batch of renderings, per library build.
   python tests/tools/gpu_cfg5.py [n_renderings] [lib.so ...]   ('-' = the product library)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import loro_amd
from loro_amd import workload
from loro_amd._cabi import Binding, Context
def gen(d): return workload.cfg5_doc(d, n_ops=1000000, turn=1000, n_checkouts=16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
libs = sys.argv[2:] or ["-"]
with mp.get_context("fork").Pool(4) as pool:
    g5 = pool.map(gen, range(4))
docs, fr = [], []
for blobs, f in g5:
    docs += [blobs] * 16; fr += f
docs = [docs[i % 64] for i in range(n)]; fr = [fr[i % 64] for i in range(n)]
import _oracle
want = _oracle.merge_batch(docs[:64], threads=16, frontiers=fr[:64])
ref = None
for lib in libs:
    b = Binding(loro_amd.LIB_PATH if lib == "-" else lib, "lm_")
    with Context(b, 0) as e:
        e.stage(docs, fr); e.run()
        best = 1e9
        for _ in range(2):
            t = time.perf_counter(); e.run(); best = min(best, time.perf_counter() - t)
        got = e.fetch()
        if ref is None: ref = got
        e.set_profiling(1); e.run()
        kt = {}
        for name, ms in e.kernel_times(): kt[name] = round(kt.get(name, 0.0) + ms, 2)
        print(os.path.basename(lib), "renderings", n, "ms %.1f" % (best * 1e3), "per s %.0f" % (n / best), "same as first:", got == ref, "first 64 equal to the oracle:", got[:64] == want, "mismatches vs oracle:", [i for i in range(64) if got[i] != want[i]][:8], kt, flush=True)
