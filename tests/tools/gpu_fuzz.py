"""Ad-hoc GPU fuzz campaign: random multi-peer sessions of varied shapes, HIP path vs oracle."""
import sys, time, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loro_amd
import _oracle, _fuzz

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = random.Random(seed0)
docs, meta = [], []
t = time.time()
for i in range(n):
    kw = dict(n_peers=rng.choice([2, 3, 4, 6, 9]), n_steps=rng.choice([150, 400, 1000, 2000]),
              kinds=rng.choice([("text",), ("text",), ("text", "list"), ("text", "list", "map")]),
              sync_prob=rng.choice([0.005, 0.02, 0.08, 0.2]), max_ins=rng.choice([2, 6, 30, 90]), styles=rng.random() < 0.5,
              commit_prob=rng.choice([0.1, 0.4, 0.9]))
    reps = _fuzz.random_session(seed0 + i, **kw)
    blobs = _fuzz.blobs_of(reps, rng=rng if rng.random() < 0.5 else None)
    docs.append(blobs); meta.append(kw)
print("generated %d sessions in %.1fs" % (n, time.time() - t), flush=True)
want = _oracle.merge_batch(docs, threads=16)
with loro_amd.MergeEngine(0) as e:
    got = e.merge_batch(docs)
bad = [i for i in range(n) if got[i] != want[i]]
print("statuses", sorted(set(w[0] for w in want)), "mismatches", len(bad))
for i in bad[:10]:
    print("  seed", seed0 + i, meta[i], "status", got[i][0], want[i][0])
