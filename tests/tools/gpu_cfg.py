"""Ad-hoc GPU probe: BASELINE.json configs[2] (LWW map, 16 peers x 10k writes) and configs[3] (mixed containers)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loro_amd
from loro_amd import workload
import _oracle, _fuzz

which = sys.argv[1]
n_docs = int(sys.argv[2])
n_distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t = time.time()
if which == "cfg3":
    base = [workload.cfg3_doc(d, combined=(d % 2 == 0)) for d in range(n_distinct)]
else:
    base = [_fuzz.blobs_of(_fuzz.random_session(1000 + d, n_peers=4, n_steps=1000, kinds=("text", "list", "map"), sync_prob=0.02, styles=True))
            for d in range(n_distinct)]
docs = [base[i % n_distinct] for i in range(n_docs)]
print(which, "docs", n_docs, "gen %.1fs" % (time.time() - t), "bytes/doc", sum(len(b) for b in base[0]), flush=True)
want = _oracle.merge_batch(base, threads=8)
t = time.time(); _oracle.merge_batch(base[:1]); print("oracle single-thread %.1f docs/s" % (1 / (time.time() - t)))
with loro_amd.MergeEngine(0) as e:
    t = time.time(); e.stage(docs); print("stage %.2fs" % (time.time() - t), flush=True)
    e.run()
    e.set_profiling(True)
    e.run()
    agg = {}
    for name, ms in e.kernel_times():
        agg[name] = agg.get(name, 0) + ms
    for name, ms in agg.items():
        print("   %-28s %9.3f ms (sum over streams)" % (name, ms))
    e.set_profiling(False)
    best = 1e9
    for it in range(3):
        t = time.time(); e.run(); best = min(best, time.time() - t)
    print("best %.1f ms -> %.0f docs/s" % (best * 1e3, n_docs / best))
    res = e.fetch()
    st = e.stats()
    print("in_bytes %d out_bytes %d device_alloc %.2f GB" % (st.in_bytes, st.out_bytes, st.device_bytes_allocated / 1e9))
    print("parity:", all(res[i] == want[i % n_distinct] for i in range(n_docs)), "statuses", sorted(set(r[0] for r in res)), "oracle", sorted(set(w[0] for w in want)))
