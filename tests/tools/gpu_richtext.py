"""k_richtext on configs[4]-shaped documents (workload.cfg5_doc: 2 peers alternating, ~1 % of the actions are bold marks): wall time of
lm_richtext behind lm_run, parity against the oracle.  usage: python tests/tools/gpu_richtext.py [n_ops] [instances]
(profiles: rocprofv3 --kernel-trace --stats -- python tests/tools/gpu_richtext.py)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import loro_amd, _oracle
from loro_amd import workload


def gen(d):
    return workload.cfg5_doc(d, n_ops=N_OPS, n_checkouts=1)[0]


N_OPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
INST = int(sys.argv[2]) if len(sys.argv) > 2 else 512
if __name__ == "__main__":
    base = [gen(d) for d in range(8)]   # (no fork pool: the script is also run under rocprofv3)
    print("generated", flush=True)
    docs = [[bytes(bytearray(b)) for b in base[i % 8]] for i in range(INST)]
    t = time.perf_counter(); want = _oracle.richtext_batch(base); t_cpu = time.perf_counter() - t
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs, None); e.run()
        print("imported", flush=True)
        t = time.perf_counter(); got = e.richtext(); print("first lm_richtext", round(time.perf_counter() - t, 3), "s", flush=True)
        assert all(g[0] == 0 for g in got)
        assert all(got[i][1] == want[i % 8][1] for i in range(INST)), "richtext differs from the oracle"
        ts = []
        for _ in range(5):
            t = time.perf_counter(); e.richtext(); ts.append(time.perf_counter() - t)
        t = time.perf_counter(); e.run(); t_run = time.perf_counter() - t
    nb = sum(len(g[1]) for g in got)
    print(json.dumps({"docs": INST, "ops_per_doc": N_OPS, "lm_richtext_ms": round(min(ts) * 1e3, 2), "lm_run_ms": round(t_run * 1e3, 2), "richtext_bytes": nb,
                      "attributed_spans_doc0": got[0][1].count(b'"attributes"'), "oracle_import_plus_richtext_s_per_doc_1_thread": round(t_cpu / 8, 3), "parity": "all equal"}))
