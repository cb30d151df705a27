"""Ad-hoc GPU probe: bench.py's configs[3] (mixed containers) and MovableList batches under several environment settings / library
builds, one process each.  Usage: python tests/tools/gpu_other.py cfg4|movable [name:ENV=V,...]... [--so lib.so]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import loro_amd, _oracle, _cases, _fuzz
from loro_amd._cabi import Binding, Context

which = sys.argv[1]
args = sys.argv[2:]
so = loro_amd.LIB_PATH
if "--so" in args:
    so = args[args.index("--so") + 1]
    args = [a for a in args if a not in ("--so", so)]
variants = args or ["base:"]
if which == "cfg4":
    base = _cases.cfg4_docs(96)
    docs = [base[i % 96] for i in range(12500)]
    nd = 96
else:
    base = [_fuzz.blobs_of(_fuzz.movable_session(7000 + d, n_peers=3, n_steps=500, sync_prob=0.08, nested=True, bulk=300), random.Random(d)) for d in range(16)]
    docs = [base[i % 16] for i in range(4096)]
    nd = 16
want = _oracle.merge_batch(base, threads=16)
b = Binding(so, "lm_")
touched = set()
for var in variants:
    name, _, kv = var.partition(":")
    for k in touched:
        os.environ.pop(k, None)
    for item in filter(None, kv.split(",")):
        k, _, v = item.partition("=")
        os.environ[k] = v; touched.add(k)
    with Context(b, 0) as e:
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
        ok = all(got[i] == want[i % nd] for i in range(len(docs)))
    print("[%s %s] %.1f ms (%.0f docs/s) parity %s | %s" % (which, name, best * 1e3, len(docs) / best, ok, "  ".join("%s=%.2f" % (k.replace("k_", ""), v) for k, v in agg.items())), flush=True)
