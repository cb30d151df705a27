"""Ad-hoc GPU probe: per-stage timings of the device pipeline on config-2 shaped documents."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loro_amd
from loro_amd import workload
import _oracle

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_base = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n_branch = n_base // 2
ce = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tpl = workload.Cfg2Template(n_base, n_branch, seed=0, commit_every=ce, fuse=True)
docs = [tpl.stamp(d) for d in range(n_docs)]
print("docs", n_docs, "blob bytes/doc", sum(len(b) for b in docs[0]), "runs", tpl.n_runs, "changes", tpl.n_changes, flush=True)
with loro_amd.MergeEngine(0) as e:
    t = time.time(); e.stage(docs); print("stage %.3fs" % (time.time() - t), flush=True)
    e.set_profiling(True)
    for it in range(3):
        t = time.time(); e.run(); dt = time.time() - t
        print("run %d: %.1f ms  -> %.0f docs/s" % (it, dt * 1e3, n_docs / dt), flush=True)
        if it == 2:
            for name, ms in e.kernel_times():
                print("   %-28s %9.3f ms" % (name, ms))
    e.set_profiling(False)
    for it in range(3):
        t = time.time(); e.run(); dt = time.time() - t
        print("run(noprof) %d: %.1f ms  -> %.0f docs/s" % (it, dt * 1e3, n_docs / dt), flush=True)
    import ctypes
    out3 = (ctypes.c_uint32 * 5)()
    e.b.lib.lm_sizing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    e.b.lib.lm_sizing(e.h, out3)
    print("sizing: leaves used max %d, leaf_cap max %d, n_elems max %d, retried docs %d" % (out3[0], out3[1], out3[2], out3[3]))
    res = e.fetch()
    st = e.stats()
    print("in_bytes %d out_bytes %d device_alloc %.2f GB" % (st.in_bytes, st.out_bytes, st.device_bytes_allocated / 1e9))
    want = _oracle.merge_batch(docs[:16], threads=8)
    print("parity first 16:", res[:16] == want, "statuses", sorted(set(r[0] for r in res)))
