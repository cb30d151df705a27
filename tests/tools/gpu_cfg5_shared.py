"""GPU box: BASELINE configs[4] through lm_stage / lm_run with the 16 entries of a document sharing their blobs (shared replay,
include/loro_merge.h) against LM_SHARE_REPLAY=0 (one replay per entry).   python tests/tools/gpu_cfg5_shared.py [documents] [share,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import loro_amd
from loro_amd import workload
def gen(d): return workload.cfg5_doc(d, n_ops=1000000, turn=1000, n_checkouts=16)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "0"]
with mp.get_context("fork").Pool(4) as pool:
    g5 = pool.map(gen, range(4))
import _oracle
want = _oracle.merge_batch([g5[d][0] for d in range(4) for _ in range(16)], threads=16, frontiers=[f for d in range(4) for f in g5[d][1]])
inst = [[bytes(bytearray(b)) for b in g5[d % 4][0]] for d in range(U)]
docs = [inst[i // 16] for i in range(16 * U)]
fr = [g5[(i // 16) % 4][1][i % 16] for i in range(16 * U)]
for share in modes:
    os.environ["LM_SHARE_REPLAY"] = share
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs, fr)
        ns = e.b.shared_documents(e.h)
        e.run()
        best = 1e9
        for _ in range(2):
            t = time.perf_counter(); e.run(); best = min(best, time.perf_counter() - t)
        got = e.fetch()
        ok = all(got[i] == want[i % 64] for i in range(len(docs)))
        e.set_profiling(1); e.run()
        kt = {}
        for name, ms in e.kernel_times(): kt[name] = round(kt.get(name, 0.0) + ms, 1)
        print("LM_SHARE_REPLAY=%s: %d renderings of %d documents (folded into %d), %.1f ms per lm_run, %.0f renderings/s, equal to the oracle: %s" % (share, len(docs), U, ns, best * 1e3, len(docs) / best, ok), kt, flush=True)
