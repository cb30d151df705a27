"""Run the pre-generated corpus (tests/tools/gen_fuzz_cache.py) through the kernel-logic harness (CPU) and compare with the
stored oracle answers."""
import sys, os, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _emu
docs, fronts, want = pickle.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_fuzz_cache.pkl"), "rb"))
got = _emu.merge_batch(docs, fronts)
bad = [i for i in range(len(docs)) if (got[i] != want[i] if want[i][0] == 0 else got[i][0] != want[i][0])]
print("cases", len(docs), "mismatches", len(bad), bad[:10])
