"""Ad-hoc GPU probe: where the element-granular integrate kernel stops paying against the span-granular one — rich-text documents
(two peers alternating, marks: the common kernel's kind) of several sizes, each size a batch of its own under LM_SPAN=1 / 0.
   python tests/tools/gpu_small_docs.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import loro_amd, _oracle
from loro_amd import workload
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
os.environ["LM_SPAN_AUTO"] = "0"
for n_ops, turn in ((2000, 100), (8000, 200), (32000, 500), (100000, 1000)):
    base = [workload.cfg5_doc(d, n_ops=n_ops, turn=turn, n_checkouts=1, commit_every=10)[0] for d in range(8)]
    docs = [base[i % 8] for i in range(n_docs)]
    want = _oracle.merge_batch(base, threads=8)
    for span in ("1", "0"):
        os.environ["LM_SPAN"] = span
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
            st = e.stats()
        ok = all(got[i] == want[i % 8] for i in range(n_docs))
        integ = {k: round(v, 2) for k, v in agg.items() if "integrate" in k}
        print("[%d ops/doc, LM_SPAN=%s] %.1f ms (%.0f docs/s) parity %s %s" % (n_ops, span, best * 1e3, n_docs / best, ok, integ), flush=True)
