"""Ad-hoc GPU probe: bench.py's heterogeneous configs[1] batch (seven shapes interleaved, two of them one change per keystroke) under
several environment settings, one process.  Usage: python tests/tools/gpu_hetero.py 10000 base: nofuse:LM_FUSE_ROWS=0"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import loro_amd, _oracle
from loro_amd import workload

def gen(sh):
    n_base, n_branch, every, fuse = sh
    return workload.Cfg2Template(n_base, n_branch, seed=n_base % 97, commit_every=every, fuse=fuse)

n_docs = int(sys.argv[1])
variants = sys.argv[2:] or ["base:"]
shapes = [(2000, 1000, 10, True), (10000, 5000, 10, True), (25000, 12500, 10, True), (50000, 25000, 10, True),
          (100000, 50000, 10, True), (5000, 2500, 1, False), (20000, 10000, 1, False)]
with mp.get_context("fork").Pool(7) as pool:
    tpls = pool.map(gen, shapes)
docs = [tpls[(d * 7919) % len(tpls)].stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:64], threads=16)
touched = set()
for var in variants:
    name, _, kv = var.partition(":")
    for k in touched:
        os.environ.pop(k, None)
    for item in filter(None, kv.split(",")):
        k, _, v = item.partition("=")
        os.environ[k] = v; touched.add(k)
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs); e.run()
        e.set_profiling(1); e.run()
        agg = {}
        for kn, ms in e.kernel_times():
            agg[kn] = agg.get(kn, 0) + ms
        e.set_profiling(0)
        best = 1e9
        for _ in range(3):
            t = time.time(); e.run(); best = min(best, time.time() - t)
        got = e.fetch()
        ok = got[:64] == want and all(g[0] == 0 for g in got)
    print("[%s] %.1f ms (%.0f docs/s) parity %s | %s" % (name, best * 1e3, n_docs / best, ok, "  ".join("%s=%.2f" % (k.replace("k_", ""), v) for k, v in agg.items())), flush=True)
