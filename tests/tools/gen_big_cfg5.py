"""Ad-hoc: one large configs[4]-shaped document (hand-over history of N trace actions, marks, checkouts) with oracle
answers, cached for a GPU run (tests/tools/gpu_fuzz_cache.py reads the same cache format)."""
import sys, os, pickle, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle
from loro_amd import workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
t = time.time()
docs, fronts = [], []
for d in range(2):
    blobs, fr = workload.cfg5_doc(d, n_ops=n, turn=1000, n_checkouts=6)
    docs += [blobs] * (len(fr) + 1); fronts += fr + [None]
print("generated in %.0fs, blob bytes %d" % (time.time() - t, len(docs[0][0])), flush=True)
t = time.time()
want = _oracle.merge_batch(docs, threads=8, frontiers=fronts)
print("oracle %.1fs statuses %s json lens %s" % (time.time() - t, [w[0] for w in want], [len(w[1]) for w in want]))
pickle.dump((docs, fronts, want), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_fuzz_cache.pkl"), "wb"))
