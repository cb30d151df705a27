"""Ad-hoc GPU A/B probe: configs[1] batch under several environment settings of the engine (kernel variants are selected
by env vars read at lm_run time), one process, same staged batch.  Usage:
    python tests/tools/gpu_ab.py 10000 base: lane:LM_DECODE=0 elem:LM_SPAN=0 [--so path/to/lib.so ...]
Prints per-stage times with the context's streams run one after the other (a stage's own duration), the step time with
streams overlapped, and a parity check of the first documents against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from loro_amd._cabi import Binding, Context
from loro_amd import workload
import loro_amd
import _oracle

n_docs = int(sys.argv[1])
variants = [a for a in sys.argv[2:] if not a.startswith("--")]
so = loro_amd.LIB_PATH
if "--so" in sys.argv:
    so = sys.argv[sys.argv.index("--so") + 1]
    variants = [v for v in variants if v != so]
tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
want = _oracle.merge_batch(docs[:8], threads=8)
b = Binding(so, "lm_")
touched = set()
for var in variants or ["base:"]:
    name, _, kv = var.partition(":")
    for k in touched:
        os.environ.pop(k, None)
    for item in filter(None, kv.split(",")):
        k, _, v = item.partition("=")
        os.environ[k] = v
        touched.add(k)
    with Context(b, 0) as e:
        e.stage(docs)
        e.run()
        e.set_profiling(1)
        agg = {}
        for _ in range(2):
            e.run()
            for kn, ms in e.kernel_times():
                agg.setdefault(kn, []).append(ms)
        e.set_profiling(0)
        best = 1e9
        for _ in range(4):
            t = time.time(); e.run(); best = min(best, time.time() - t)
        ok = e.fetch()[:8] == want
        ns = e.b.n_streams(e.h)
    stages = "  ".join("%s=%.2f" % (k.replace("k_", ""), sum(v) / len(v)) for k, v in agg.items())
    print("[%s] step %.1f ms (%.0f docs/s, %d streams) parity %s | per launch alone: %s" % (name, best * 1e3, n_docs / best, ns, ok, stages), flush=True)
