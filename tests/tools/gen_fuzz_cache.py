"""Pre-generate a large fuzz corpus (docs + optional checkout frontiers + oracle answers) into tests/_fuzz_cache.pkl so a
GPU box only has to run the engine (generation is CPU-bound Python).  Ad-hoc tool; the cache is not committed."""
import sys, os, pickle, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle, _fuzz
from loro_amd import wire

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
rng = random.Random(seed0)
docs, fronts = [], []
t = time.time()
for i in range(n):
    mode = rng.random()
    snaps = []
    if mode < 0.35:
        reps = _fuzz.nested_session(seed0 + i, n_peers=rng.choice([2, 3, 5]), n_steps=rng.choice([60, 200, 500]), sync_prob=rng.choice([0.02, 0.1, 0.3]),
                                    max_depth=rng.choice([2, 4, 7]))
    else:
        reps = _fuzz.random_session(seed0 + i, n_peers=rng.choice([2, 3, 4, 6, 9]), n_steps=rng.choice([100, 300, 800, 1500]),
                                    kinds=rng.choice([("text",), ("text", "list"), ("text", "list", "map"), ("map",), ("list",)]),
                                    sync_prob=rng.choice([0.005, 0.02, 0.08, 0.2]), max_ins=rng.choice([2, 6, 30, 90]), styles=rng.random() < 0.5,
                                    commit_prob=rng.choice([0.1, 0.4, 0.9]), snapshots=snaps,
                                    solo_steps=rng.choice([0, 0, 50, 400, 1200]), max_del=rng.choice([4, 4, 40, 300]))   # (round 5: histories with a linear prefix, long deletes)
    blobs = _fuzz.blobs_of(reps, rng=rng if rng.random() < 0.5 else None)
    docs.append(blobs); fronts.append(None)
    for fr, _ in snaps[:: max(1, len(snaps) // 3)][:3]:      # a few checkouts of the same history
        docs.append(blobs); fronts.append(wire.encode_frontiers(fr))
        if len(fr) == 1 and fr[0][1] > 0:
            docs.append(blobs); fronts.append(wire.encode_frontiers([(fr[0][0], fr[0][1] - 1)]))
print("generated %d sessions → %d cases in %.0fs" % (n, len(docs), time.time() - t), flush=True)
want = _oracle.merge_batch(docs, threads=8, frontiers=fronts)
print("oracle statuses", sorted(set(w[0] for w in want)))
pickle.dump((docs, fronts, want), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "_fuzz_cache.pkl"), "wb"))
