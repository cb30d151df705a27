"""Run the pre-generated corpus (tests/tools/gen_fuzz_cache.py) through the HIP path and compare with the stored oracle answers."""
import sys, os, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loro_amd
docs, fronts, want = pickle.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_fuzz_cache.pkl"), "rb"))
with loro_amd.MergeEngine(0) as e:
    got = e.merge_batch(docs, fronts)
bad = [i for i in range(len(docs)) if (got[i] != want[i] if want[i][0] == 0 else got[i][0] != want[i][0])]
print("cases", len(docs), "oracle statuses", sorted(set(w[0] for w in want)), "mismatches", len(bad))
for i in bad[:10]:
    print("  case", i, "status", got[i][0], want[i][0], "checkout" if fronts[i] else "latest", len(got[i][1]), len(want[i][1]))
