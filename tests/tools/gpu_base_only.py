"""GPU box: per-stage times of base + A's branch only (configs[1] without B's concurrent branch), batch path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loro_amd
from loro_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n)]
for what, sel in (("base only", slice(0, 1)), ("base + A", slice(0, 2)), ("base + A + B", slice(0, 3)), ("base + B", (0, 2))):
    with loro_amd.MergeEngine(0) as e:
        dd = [[b[i] for i in sel] if isinstance(sel, tuple) else b[sel] for b in docs]
        e.stage(dd); e.run(); e.set_profiling(1); e.run()
        kt = {}
        for name, ms in e.kernel_times():
            kt[name] = round(kt.get(name, 0.0) + ms, 3)
        print(what, kt)
