"""GPU box: per-stage times (streams serialized) of BASELINE configs[2] — LWW Map, 16 peers x 10,000 writes on 1,024 keys —
for the combined-blob variant and the 16-blob variant.   python tests/tools/gpu_cfg3_prof.py [n_docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loro_amd
from loro_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for combined in (True, False):
    base = [workload.cfg3_doc(d, combined=combined) for d in range(4)]
    docs = [base[i % 4] for i in range(n)]
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs); e.run()
        t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
        e.set_profiling(1); e.run()
        kt = {}
        for name, ms in e.kernel_times():
            kt[name] = round(kt.get(name, 0.0) + ms, 3)
        st = e.stats()
        print("combined" if combined else "16 blobs", "docs", n, "run ms %.1f" % (dt * 1e3), "in GB %.2f out MB %.1f" % (st.in_bytes / 1e9, st.out_bytes / 1e6), kt)
