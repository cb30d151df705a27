"""Ad-hoc GPU probe: bench.py's end_to_end leg with other ring shapes (contexts in rotation x helper threads x batch split).
   python tests/tools/gpu_e2e.py [docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, loro_amd
from loro_amd._cabi import Context
from concurrent.futures import ThreadPoolExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tpl, docs = bench.build_docs(N, 0, 50000, 25000, 10, seed=0)


def leg(engs, packs, n_docs_each, steps, helpers):
    n = len(engs)
    pending = [None] * n
    pool = ThreadPoolExecutor(max_workers=helpers)
    t_stage = 0.0
    ev = []
    def _fetch(k, i):
        engs[k].wait(); t1 = time.perf_counter()
        engs[k].fetch_raw(); t2 = time.perf_counter()
        ev.append((i, "run_done", t1)); ev.append((i, "fetch_done", t2))
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % n
        if pending[k] is not None:
            pending[k].result(); pending[k] = None
        t = time.perf_counter(); engs[k].stage_packed(packs[k % len(packs)]); t_stage += time.perf_counter() - t
        ev.append((i, "staged", time.perf_counter()))
        engs[k].run_async()
        pending[k] = pool.submit(_fetch, k, i)
    for k in range(n):
        if pending[k] is not None:
            pending[k].result()
    pool.shutdown()
    dt = time.perf_counter() - t0
    return n_docs_each * steps / dt, dt / steps * 1e3, t_stage / steps * 1e3, sorted((i, w, round((t - t0) * 1e3, 1)) for i, w, t in ev)


for n_ctx, helpers, split in ((3, 1, 1), (4, 1, 1), (4, 2, 1), (5, 2, 1)):
    per = N // split
    engs = [loro_amd.MergeEngine(0) for _ in range(n_ctx)]
    try:
        pinned = [engs[0].pack_pinned(docs[s * per:(s + 1) * per]) for s in range(split)]
        for k, e in enumerate(engs):
            e.stage_packed(pinned[k % split]); e.run(); e.fetch_raw()
        rate, ms, st, ev = leg(engs, pinned, per, 12 * split, helpers)
        print(f"contexts {n_ctx} helpers {helpers} split {split}: {rate:.0f} docs/s, {ms * split:.2f} ms per {N} docs, stage {st:.2f} ms per call, direct {bool(engs[0].b.staged_direct(engs[0].h))}", flush=True)
        if (n_ctx, helpers, split) == (3, 1, 1):
            print("  timeline:", ev[:30])
    finally:
        for p in pinned:
            engs[0].free_pinned(p)
        for e in engs:
            e.close()
