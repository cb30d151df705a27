"""Ad-hoc: run a batch of configs[1]-shaped documents of a given size (n_docs n_base n_branch) a few times; prints op-run counts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loro_amd
from loro_amd import workload
n, nb, nr = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tpl = workload.Cfg2Template(nb, nr, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n)]
with loro_amd.MergeEngine(0) as e:
    e.stage(docs); e.run()
    t = time.time(); e.run(); dt = time.time() - t
    print("docs %d ops %d op runs %d changes %d: run %.1f ms" % (n, tpl.n_ops, tpl.n_runs, tpl.n_changes, dt * 1e3))
