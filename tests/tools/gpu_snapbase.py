"""Ad-hoc GPU probe: snapshot + updates on a state base (lm_snapshot_base.h) — where the integrate stage's time goes.
   python tests/tools/gpu_snapbase.py [docs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import loro_amd
from loro_amd import workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
base = [workload.cfg2_snapshot_doc(s) for s in range(4)]


def go(name, docs, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        with loro_amd.MergeEngine(0) as e:
            e.stage(docs); e.run()
            st = e.result_meta()[0]
            t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
            e.set_profiling(1); e.run()
            agg = {}
            for kn, ms in e.kernel_times():
                agg[kn] = agg.get(kn, 0) + ms
            print(f"{name}: {dt * 1e3:.1f} ms per {len(docs)} docs, failed {int((st != 0).sum())}, state docs {e.b.state_documents(e.h)}, redo {e.b.redo_documents(e.h)}, posdel {e.b.last_posdel(e.h) if hasattr(e.b, 'last_posdel') else '?'} | " +
                  "  ".join(f"{k.replace('k_', '')}={v:.1f}" for k, v in agg.items() if v > 0.5), flush=True)
    finally:
        for k in (env or {}):
            del os.environ[k]


rep = lambda f: [f(base[i % 4]) for i in range(N)]
go("snapshot alone", rep(lambda d: [d[0]]))
go("snapshot + A", rep(lambda d: [d[0], d[1]]))
go("snapshot + B", rep(lambda d: [d[0], d[2]]))
go("snapshot + A + B", rep(lambda d: list(d)))
go("snapshot + A + B, history, LM_PLAIN=0 (the general kernel)", rep(lambda d: list(d)), {"LM_SNAPSHOT_STATE": "0", "LM_PLAIN": "0"})
go("snapshot + A + B, history, LM_LINEAR=0", rep(lambda d: list(d)), {"LM_SNAPSHOT_STATE": "0", "LM_LINEAR": "0"})
go("snapshot + A + B, LM_PD_STATE_PIECES=0 (every document replayed from its history by the side engine)", rep(lambda d: list(d)), {"LM_PD_STATE_PIECES": "0"})
go("snapshot + A, history", rep(lambda d: [d[0], d[1]]), {"LM_SNAPSHOT_STATE": "0"})
go("snapshot + A + B, history", rep(lambda d: list(d)), {"LM_SNAPSHOT_STATE": "0"})
