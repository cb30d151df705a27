"""Ad-hoc GPU probe: throughput of configs[1] vs the number of streams a context splits the batch into."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loro_amd
from loro_amd import workload
import _oracle

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:8], threads=8) + _oracle.merge_batch(docs[-8:], threads=8)
for ns in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]:
    os.environ["LM_STREAMS"] = str(ns)
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs)
        e.run()
        best = 1e9
        for it in range(4):
            t = time.time(); e.run(); best = min(best, time.time() - t)
        res = e.fetch()
        print("streams %d (split %d): best %.1f ms -> %.0f docs/s  parity %s" % (
            ns, e.b.n_streams(e.h), best * 1e3, n_docs / best, (res[:8] + res[-8:]) == want), flush=True)
