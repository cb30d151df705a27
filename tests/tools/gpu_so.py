"""Ad-hoc GPU probe: configs[1] step time with a given build of the library (register-budget experiments)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
import _oracle
n_docs = int(sys.argv[1])
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:8], threads=8)
for so in sys.argv[2:]:
    b = Binding(so, "lm_")
    with Context(b, 0) as e:
        e.stage(docs); e.run()
        best = 1e9
        for it in range(4):
            t = time.time(); e.run(); best = min(best, time.time() - t)
        ok = e.fetch()[:8] == want
    print("%s: best %.1f ms -> %.0f docs/s parity %s" % (os.path.basename(so), best * 1e3, n_docs / best, ok), flush=True)
