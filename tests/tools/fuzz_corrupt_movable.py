"""Ad-hoc robustness fuzz for MovableList documents: damaged blobs (byte flips, truncation, splices; checksum re-fitted)
through the kernel-logic harness (optionally an ASan build: pass its path) and the oracle.
    g++ -O1 -g -fsanitize=address -std=c++17 -fPIC -shared -Wno-unknown-pragmas -DLM_EMU_TRACE -o /tmp/libloroemu_asan.so tests/emu/lm_emu.cpp
    LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tests/tools/fuzz_corrupt_movable.py 600 /tmp/libloroemu_asan.so
Requirement: no crash / out-of-bounds access, and a document the oracle accepts must never come back different."""
import sys, os, random, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle, _fuzz
from loro_amd._cabi import Binding, Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
so = sys.argv[2] if len(sys.argv) > 2 else None
rng = random.Random(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
base = [_fuzz.blobs_of(_fuzz.movable_session(9000 + i, nested=i % 2 == 0, n_steps=90)) for i in range(14)]


def refit(blob):
    body = blob[20:]
    return blob[:16] + struct.pack("<I", _oracle.xxh32(body)) + body


def corrupt(blob):
    b = bytearray(blob)
    k = rng.random()
    if k < 0.55:
        for _ in range(rng.choice([1, 1, 2, 5])):
            i = rng.randrange(22, len(b))
            b[i] = rng.choice([b[i] ^ (1 << rng.randrange(8)), rng.randrange(256), 0xFF, 0x80, 0, (b[i] + 1) & 0xff])
    elif k < 0.7:
        del b[rng.randrange(22, len(b)):]
    elif k < 0.85:
        i = rng.randrange(22, len(b)); j = min(len(b), i + rng.randrange(1, 40))
        b[i:j] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 50)))
    else:
        i = rng.randrange(22, len(b))
        b[i:i] = b[rng.randrange(22, len(b)):][: rng.randrange(1, 64)]
    return refit(bytes(b)) if len(b) > 22 and rng.random() < 0.92 else bytes(b)


docs = []
for i in range(n):
    d = list(rng.choice(base))
    j = rng.randrange(len(d))
    d[j] = corrupt(d[j])
    docs.append(d)
want = _oracle.merge_batch(docs, threads=4)
b = Binding(so, "lmemu_") if so else __import__("_emu").binding()
with Context(b) as c:
    got = c.merge_batch(docs)
ok_o = sum(1 for w in want if w[0] == 0)
diff_ok = [i for i in range(n) if want[i][0] == 0 and got[i][0] == 0 and got[i] != want[i]]
dev_ok_oracle_bad = [i for i in range(n) if want[i][0] != 0 and got[i][0] == 0]
dev_bad_oracle_ok = [i for i in range(n) if want[i][0] == 0 and got[i][0] != 0]
cls = [i for i in range(n) if want[i][0] != 0 and got[i][0] != 0 and want[i][0] != got[i][0]]
print("docs", n, "oracle ok", ok_o, "| both ok but different:", len(diff_ok), "| device ok / oracle error:", len(dev_ok_oracle_bad),
      "| device error / oracle ok:", len(dev_bad_oracle_ok), "| both error, different class:", len(cls))
print(" oracle statuses", sorted(set(w[0] for w in want)), "device statuses", sorted(set(g[0] for g in got)))
for i in (diff_ok + dev_ok_oracle_bad + dev_bad_oracle_ok)[:8]:
    print("  case", i, "dev", got[i][0], got[i][1][:90], "| oracle", want[i][0], want[i][1][:90])
from collections import Counter
print(" (oracle, device) error classes that differ:", sorted(Counter((want[i][0], got[i][0]) for i in cls).items()))
sys.exit(1 if diff_ok else 0)
