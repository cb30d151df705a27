"""Run the pre-generated corpus (tests/tools/gen_fuzz_cache.py) through the HIP path and compare with the stored oracle answers."""
import sys, os, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loro_amd
docs, fronts, want = pickle.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_fuzz_cache.pkl"), "rb"))
# (entries of one history share their blob objects in the pickle: the checkouts go through the shared replay; the variants below
# re-run the corpus with the flags / kernel choices documents of this size do not get by default)
for env in ({}, {"LM_CUT_MIN_ROWS": "0"}, {"LM_SPAN_AUTO": "1"}, {"LM_SHARE_REPLAY": "0", "LM_CUT_MIN_ROWS": "0", "LM_PLAIN": "0"}):
    for k, v in env.items():
        os.environ[k] = v
    with loro_amd.MergeEngine(0) as e:
        e.stage(docs, fronts)
        shared = e.b.shared_documents(e.h)
        e.run()
        got = e.fetch()
    for k in env:
        del os.environ[k]
    bad = [i for i in range(len(docs)) if (got[i] != want[i] if want[i][0] == 0 else got[i][0] != want[i][0])]
    print(env, "cases", len(docs), "folded into", shared, "oracle statuses", sorted(set(w[0] for w in want)), "mismatches", len(bad), flush=True)
    for i in bad[:10]:
        print("  case", i, "status", got[i][0], want[i][0], "checkout" if fronts[i] else "latest", len(got[i][1]), len(want[i][1]))
