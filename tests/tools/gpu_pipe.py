"""Ad-hoc GPU probe: two contexts in flight (double buffering) vs one, configs[1]."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loro_amd
from loro_amd import workload
import _oracle
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:8], threads=8)
for depth in (1, 2, 3):
    engs = [loro_amd.MergeEngine(0) for _ in range(depth)]
    for e in engs:
        e.stage(docs); e.run()
    K = 12
    t = time.time()
    for i in range(K):
        e = engs[i % depth]
        e.wait()
        e.run_async()
    for e in engs:
        e.wait()
    dt = (time.time() - t) / K
    ok = all(e.fetch()[:8] == want for e in engs)
    print("contexts in flight %d: %.1f ms per step -> %.0f docs/s  parity %s" % (depth, dt * 1e3, n_docs / dt, ok), flush=True)
    for e in engs:
        e.close()
