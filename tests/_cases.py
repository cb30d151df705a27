"""Hand-built edge cases shared by the CPU (kernel-logic harness) and GPU parity tests."""
import os, random
from loro_amd import wire, workload
import _fuzz, _oracle


def edge_case_docs():
    docs, names = [], []

    def add(name, blobs):
        names.append(name)
        docs.append(list(blobs))

    add("no blobs", [])
    add("empty updates blob", [wire.encode_updates([])])
    a = wire.Replica(1); a.text_insert("text", 0, "ab"); a.commit()
    good = a.export()
    bad = bytearray(good); bad[-1] ^= 1
    add("checksum mismatch", [bytes(bad)])
    add("bad magic", [b"lor0" + good[4:]])
    add("truncated", [good[:10]])
    # a FastSnapshot whose third (shallow-root state) section is not empty: history below the root is gone → unsupported
    add("shallow snapshot", [wire.envelope(b"\x00" * 8 + b"\x01\x00\x00\x00" + b"E", mode=3)])
    add("good next to bad docs", [good])
    # pending: second export without the first
    a.text_insert("text", 2, "cd"); a.commit()
    second = a.export({1: 2})
    add("pending only", [second])
    add("pending resolved later", [second, good])
    add("duplicate blob", [good, good, second, second])
    # overlapping exports of the same history with different change boundaries (sliced on import)
    s = "abcdefghijklmnopqrstuvwxyz0123"
    cid = wire.root_cid("text", wire.KIND_TEXT)
    c1 = wire.Change(7, 0, 0, [], [wire.Op(cid, 0, "text_insert", pos=0, text=s[:20])])
    c2 = wire.Change(7, 10, 10, [(7, 9)], [wire.Op(cid, 10, "text_insert", pos=10, text=s[10:])])
    add("overlapping changes", [wire.encode_updates([[c1]]), wire.encode_updates([[c2]])])
    add("overlapping changes reversed", [wire.encode_updates([[c2]]), wire.encode_updates([[c1]])])
    # overlapping delete: forward and reversed delete spans sliced in the middle
    base = wire.Change(3, 0, 0, [], [wire.Op(cid, 0, "text_insert", pos=0, text="0123456789")])
    d_full = wire.Change(3, 10, 10, [(3, 9)], [wire.Op(cid, 10, "delete", pos=2, del_id=(3, 2), signed_len=4)])
    d_tail = wire.Change(3, 12, 12, [(3, 11)], [wire.Op(cid, 12, "delete", pos=2, del_id=(3, 4), signed_len=2)])
    add("sliced forward delete", [wire.encode_updates([[base]]), wire.encode_updates([[d_tail]]), wire.encode_updates([[d_full]])])
    r_full = wire.Change(3, 10, 10, [(3, 9)], [wire.Op(cid, 10, "delete", pos=5, del_id=(3, 2), signed_len=-4)])
    add("reversed delete", [wire.encode_updates([[base]]), wire.encode_updates([[r_full]])])
    # long pastes (> 64, > 128 elements), multi-byte text crossing the 61-byte chunking of the payload kernel
    p = wire.Replica(11)
    p.text_insert("text", 0, "x" * 300); p.text_insert("text", 150, "é中😀λ" * 60); p.text_delete("text", 100, 250); p.commit()
    q = wire.Replica(12); q.text_insert("text", 0, "z" * 129); q.commit()
    add("long pastes", [p.export(), q.export()])
    # many leaves: > 64 leaves so directory chunks span several lanes
    big = wire.Replica(21)
    rng = random.Random(5)
    n = 0
    for i in range(900):
        pos = rng.randint(0, n)
        big.text_insert("text", pos, "abcdefgh"[: 1 + i % 8]); n += 1 + i % 8
        if i % 7 == 0 and n > 10:
            dp = rng.randint(0, n - 5); big.text_delete("text", dp, 3); n -= 3
        if i % 50 == 0:
            big.commit()
    big.commit()
    add("many leaves", [big.export()])
    # rich text anchors occupy positions but not the string
    m = wire.Replica(31); m.text_insert("text", 0, "hello world"); m.text_mark("text", 0, 5, "bold", True); m.text_insert("text", 3, "XY"); m.commit()
    add("style anchors", [m.export()])
    # list values and map values of every renderable scalar kind
    l = wire.Replica(41)
    l.list_insert("list", 0, [None, True, False, 0, -1, 2 ** 62, -2 ** 63, "q\"uo\\te\n\t\x01é", b"\x00\xff", [], [1, [2, [3]]]])
    l.list_delete("list", 1, 2)
    l.map_set("map", "b", "x"); l.map_set("map", "a", [1, 2]); l.map_set("map", "", 7); l.map_delete("map", "b"); l.commit()
    add("list and map values", [l.export()])
    # concurrent map writers with equal lamports
    w1 = wire.Replica(5); w1.map_set("map", "k", "from5"); w1.map_set("map", "gone", 1); w1.commit()
    w2 = wire.Replica(9); w2.map_set("map", "k", "from9"); w2.map_delete("map", "gone"); w2.commit()
    add("map tie break", [w1.export(), w2.export()])
    # f64 and map-typed plain values (rendered canonically: shortest round-trip digits, keys in bytewise order)
    u = wire.Replica(51); u.map_set("map", "f", 1.5); u.map_set("map", "g", [0.1, -2.5e-7, 1e21, 3.0, float("inf")]); u.commit()
    add("f64 value", [u.export()])
    u2 = wire.Replica(52)
    u2.map_set("map", "nested", {"b": 1, "a": {"z": [1, {"y": None, "x": 2.25}], "k": "v"}, "": True})
    u2.list_insert("list", 0, [{"q": 1, "p": [{"b": 2, "a": 1}]}, "tail"]); u2.commit()
    u3 = wire.Replica(53); u3.map_set("map", "other", {"k2": 2, "k1": 1}); u3.commit()
    add("nested map value", [u2.export(), u3.export()])
    # map keys that tie on their first eight bytes (the emit stage sorts by an 8-byte prefix and reads the strings on a tie)
    u4 = wire.Replica(55)
    for i, k in enumerate(["abcdefgh1", "abcdefgh0", "abcdefgh", "abcdefg", "ab", "a", "", "a\x00", "a\x00b", "abcdefgh\x00", "abcdefghij", "b",
                           "\u00e9t\u00e9", "\u00e9", "zzzzzzzzzzzzzzzz", "zzzzzzzzzzzzzzzy", "zzzzzzzz"]):
        u4.map_set("map", k, i)
    u4.commit()
    add("map keys with common prefixes", [u4.export()])
    # still outside the device scope: flagged, never guessed
    t = wire.Replica(54); t.map_set_container("map", "tree", 3); t.commit()
    add("tree child container", [t.export()])
    return names, docs


def container_existence_cases():
    """(name, blobs, encoded frontiers or None, expected JSON) — which root containers the reference's state store holds
    (diff_calc.rs:299 `!diff.is_empty() || bring_back`, state.rs:1352-1391): the batch is imported like import_batch
    (loro.rs:1432-1523: one diff empty → latest), a checkout is a second diff latest → version.  A root Text / List
    gets a state when something is visible at the latest version or at the checked-out one; a root Map as soon as a key
    was written (a deleted key is still an entry of the diff, diff_calc.rs:553-605)."""
    out = []
    a = wire.Replica(1)
    a.text_insert("text", 0, "ab"); a.list_insert("list", 0, [1, 2]); a.map_set("map", "k", 1); a.commit(); v1 = list(a.frontiers)
    first = a.export()
    a.text_delete("text", 0, 2); a.list_delete("list", 0, 2); a.map_delete("map", "k"); a.commit(); v2 = list(a.frontiers)
    one = a.export()
    two = [first, a.export({1: a.changes[1][0].ctr_end})]
    out.append(("inserted and fully deleted, one blob", [one], None, b'{"map":{}}'))
    out.append(("inserted and fully deleted, two blobs", two, None, b'{"map":{}}'))
    out.append(("…checked out where it was visible", [one], wire.encode_frontiers(v1), b'{"list":[1,2],"map":{"k":1},"text":"ab"}'))
    out.append(("…checked out at the latest (empty) version", [one], wire.encode_frontiers(v2), b'{"map":{}}'))
    out.append(("…checked out at the empty version", [one], wire.encode_frontiers([]), b'{"map":{}}'))
    a.text_insert("text", 0, "c"); a.commit()
    three = [a.export()]
    out.append(("visible again at the latest version", three, None, b'{"map":{},"text":"c"}'))
    out.append(("…checked out where it was empty: the state exists", three, wire.encode_frontiers(v2), b'{"map":{},"text":""}'))
    out.append(("…checked out at the empty version", three, wire.encode_frontiers([]), b'{"map":{},"text":""}'))
    # more deleted than inserted (two peers delete the same run concurrently) and still something visible at the latest version
    p, q, r = wire.Replica(11), wire.Replica(12), wire.Replica(13)
    p.text_insert("text", 0, "abc"); p.commit(); vp = list(p.frontiers)
    for x in (q, r):
        x.merge_from(p)
        x.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids([p.export()], "text", wire.KIND_TEXT))
        x.text_delete("text", 0, 2); x.commit()
    q.merge_from(r)
    both = [q.export()]
    out.append(("overlapping concurrent deletes", both, None, b'{"text":"c"}'))
    out.append(("…checked out before the insert", both, wire.encode_frontiers([]), b'{"text":""}'))
    q.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids(both, "text", wire.KIND_TEXT))
    q.text_delete("text", 0, 1); q.commit()
    gone = [q.export()]
    out.append(("…and the rest deleted later", gone, None, b'{}'))
    out.append(("…checked out before the insert", gone, wire.encode_frontiers([]), b'{}'))
    out.append(("…checked out at the insert", gone, wire.encode_frontiers(vp), b'{"text":"abc"}'))
    return out


def snapshot_cases():
    """(docs, check(got)): FastSnapshot (mode 3) ingest (lm_snapshot.h): a document that is one snapshot from its state section,
    any other from the history in the ChangeStore section + the set of root containers of the state section.  Pinned on the reference
    fixtures alone: `snapshot.blob` (Rust-written, LZ4-framed SSTable blocks) and `snapshot.ts.blob` hold the history of
    `updates.blob`, `runtime-snapshot.ts.blob` that of `runtime-updates.ts.blob` (crates/loro/tests/loro_js_interop.rs:58-90):
    the in-scope values must be those of snapshot.deep.json / the updates import, the ROOTS those of snapshot.deep.json."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))
    b = {k: bytes.fromhex(v) for k, v in fx["blobs"].items()}
    flipped = bytearray(b["snapshot.blob"]); flipped[300] ^= 0x40
    docs = [[b["snapshot.blob"]], [b["updates.blob"]], [b["snapshot.ts.blob"]], [b["runtime-snapshot.ts.blob"]], [b["runtime-updates.ts.blob"]],
            [b["shallow.ts.blob"]], [bytes(flipped)], [b["snapshot.blob"][:200]], [b["snapshot.blob"], b["updates.blob"], b["snapshot.ts.blob"]],
            [b["fugue-left.ts.blob"], b["fugue-right.ts.blob"]]]

    def check(got):
        snap, upd, snap_ts, rsnap, rupd, shallow, bad_sum, cut, both, plain = got
        assert snap[0] == upd[0] == 4 and snap[1] and snap_ts[1:] == snap[1:] and snap[2] == upd[2]   # out-of-scope containers ride along as null
        assert rsnap[2] == rupd[2] and rsnap[0] == rupd[0]
        deep = fx["json"]["snapshot.deep.json"]
        v, u = json.loads(snap[1]), json.loads(upd[1])
        # an empty document takes its state store from the snapshot's state section: every root of snapshot.deep.json is there,
        # "list" too, in which nothing is visible (an UPDATES import of the same history has no state for it)
        assert sorted(v) == sorted(deep) and v["list"] == deep["list"] == [] and v["text"] == deep["text"] == "" and "list" not in u
        assert {k: x for k, x in v.items() if k != "list"} == u
        for k, x in deep["map"].items():
            if k != "child_tree":
                assert v["map"][k] == x, k
        rv, ru = json.loads(rsnap[1]), json.loads(rupd[1])
        assert all(rv[k] == ru[k] for k in ru) and sorted(k for k in rv if ":$" not in k) == sorted(fx["json"]["runtime.expected.json"])
        # history below a shallow root is gone: rendered from the state sections (round 6; loro_js_interop.rs:129-139) —
        # LM_SNAPSHOT_STATE=0 leaves only the history path, which reports it LM_UNSUPPORTED
        assert shallow[:2] == (0, b'{"text":"0123456789"}') or (shallow[0] == 4 and not shallow[1] and os.environ.get("LM_SNAPSHOT_STATE") == "0")
        assert bad_sum[0] == 2 and cut[0] in (1, 2)           # checksum mismatch / truncated
        assert both[1:] == snap[1:]                           # the same history three times over, one of them a snapshot
        assert plain[:2] == (0, b'{"text":"Hello World!"}')
    return docs, check


def limit_docs():
    """(name, blobs): one document beyond each documented limit of the device path (include/loro_merge.h, "Limits"); each
    must come back LM_UNSUPPORTED (4), never a guessed value, and must not disturb its neighbours in the batch."""
    out = []
    r = wire.Replica(11)
    for i in range(300):
        c = r.map_set_container("root", "k%d" % i, wire.KIND_MAP)
        r.map_set(c, "v", i)
    r.commit()
    out.append(("more than 256 containers", [r.export()]))
    out.append(("more than 255 peers", [b for p in range(1, 258) for b in [_one_key(p)]]))
    r = wire.Replica(12)
    for i in range(65):
        r.map_set("root%d" % i, "k", i)
    r.commit()
    out.append(("more than 64 root containers", [r.export()]))
    r = wire.Replica(13)
    c = "root"
    for i in range(18):
        c = r.map_set_container(c, "down", wire.KIND_MAP)
    r.map_set(c, "bottom", 1); r.commit()
    out.append(("nesting deeper than 16", [r.export()]))
    r = wire.Replica(14)
    r.next_counter = (1 << 24) - 2
    r.text_insert("text", 0, "abcd"); r.commit()
    out.append(("counters beyond 2^24", [r.export()]))
    return out


def _one_key(p):
    r = wire.Replica(p)
    r.map_set("root", "k%d" % p, p); r.commit()
    return r.export()


def fuzz_docs(n, base=0, steps=40, peers=None, **kw):
    docs = []
    for seed in range(base, base + n):
        kinds = [("text",), ("text", "list"), ("text", "list", "map"), ("map",)][seed % 4]
        reps = _fuzz.random_session(seed, n_peers=peers or (2 + seed % 3), n_steps=steps + seed % 50, kinds=kinds, **kw)
        docs.append(_fuzz.blobs_of(reps, random.Random(seed)))
    return docs


def linear_prefix_docs(n, base=90000):
    """Histories that begin as ONE chain (a replica working alone: inserts at random places, typing that continues an item,
    deletes of a few elements up to whole leaves), which every other replica imports before the concurrent part begins: the
    hand-over is a critical version and everything before it the batch replay's linear prefix (lm_k_integrate_linear.h) —
    leaf splits, items cut by inserts and deletes, emptied leaves, prefixes of every length incl. whole documents (1 peer)."""
    docs = []
    for seed in range(n):
        kinds = [("text",), ("text", "list"), ("text", "list", "map")][seed % 3]
        n_peers = 1 + seed % 4
        reps = _fuzz.random_session(base + seed, n_peers=n_peers, n_steps=20 + seed % 40, kinds=kinds, solo_steps=40 + (seed * 37) % 400,
                                    max_del=[4, 40, 200][seed % 3], max_ins=[6, 30, 3][(seed // 3) % 3], solo_peer=(seed // 2) % n_peers)
        docs.append(_fuzz.blobs_of(reps, random.Random(seed)))
    return docs


def misnamed_delete_docs(n, base=97000):
    """Concurrent sessions (some with a solo prefix) in which delete ops NAME other elements than the ones at their positions: the
    target id of a few delete ops (up to five per document) is moved to another counter or another peer after the session was recorded.  The
    reference applies a delete by position and only remembers what it met (crdt_rope.rs:256-335, tracker.rs:193-252): the value
    of such a document is the value of the unharmed session.  Returns (damaged documents, unharmed documents)."""
    import copy
    bad, good = [], []
    for seed in range(n):
        kinds = [("text",), ("text", "list")][seed % 2]
        reps = _fuzz.random_session(base + seed, n_peers=2 + seed % 3, n_steps=60 + seed % 60, kinds=kinds, sync_prob=0.12, max_del=[4, 12][seed % 2],
                                    solo_steps=[0, 60][(seed // 2) % 2], max_ins=8)
        good.append(_fuzz.blobs_of(reps, random.Random(seed)))
        rng = random.Random(seed * 7 + 1)
        peers = [r.peer for r in reps]
        reps2 = copy.deepcopy(reps)
        hit = 0
        for r in reps2:
            for ch in r.changes.get(r.peer, []):
                for op in ch.ops:
                    if op.kind == "delete" and hit < 5 and rng.random() < 0.3:
                        p, c = op.del_id
                        if rng.random() < 0.5:
                            op.del_id = (rng.choice(peers), max(0, c + rng.randint(-5, 40)))
                        else:
                            op.del_id = (p, max(0, c + rng.choice([-3, -1, 1, 2, 7, 1000])))
                        hit += 1
        if hit == 0:
            good.pop()
            continue
        bad.append(_fuzz.blobs_of(reps2, random.Random(seed)))
    return bad, good


def trace_docs(n_base, variants=((10, True), (10, False), (0, True)), n_docs=2, seed=2):
    docs = []
    for ce, fuse in variants:
        tpl = workload.Cfg2Template(n_base, n_base // 2, seed=seed, commit_every=ce, fuse=fuse)
        for d in range(n_docs):
            s = tpl.stamp(d)
            docs.append(s)
            docs.append([s[0], s[2], s[1]])
            docs.append([s[0], s[1]])
    return docs


def cfg4_docs(n, first=1000, n_steps=1000):
    """BASELINE.json configs[3] shape: root List + Map + Text, 4 peers, ≈1k ops mixed, pairwise syncs every ≈50 ops
    (real DAG merges), bold marks on the text."""
    return [_fuzz.blobs_of(_fuzz.random_session(first + d, n_peers=4, n_steps=n_steps, kinds=("text", "list", "map"), sync_prob=0.02, styles=True))
            for d in range(n)]


def corrupted_docs(n, seed=1):
    """Valid documents with one blob damaged (byte flips, truncation, splices) and the envelope checksum re-fitted so the
    decoder is reached.  For robustness tests: nothing may crash, and a damaged document must not take the batch down."""
    import random, struct
    rng = random.Random(seed)
    base = fuzz_docs(12, base=5000) + [_fuzz.blobs_of(_fuzz.nested_session(8000 + i, n_steps=80)) for i in range(6)] + cfg4_docs(4, first=3000, n_steps=200)

    def refit(blob):
        body = blob[20:]
        return blob[:16] + struct.pack("<I", _oracle.xxh32(body)) + body

    def corrupt(blob):
        b = bytearray(blob)
        k = rng.random()
        if k < 0.5:
            for _ in range(rng.choice([1, 1, 2, 5])):
                i = rng.randrange(22, len(b))
                b[i] = rng.choice([b[i] ^ (1 << rng.randrange(8)), rng.randrange(256), 0xFF, 0x80, 0])
        elif k < 0.7:
            del b[rng.randrange(22, len(b)):]
        elif k < 0.85:
            i = rng.randrange(22, len(b)); j = min(len(b), i + rng.randrange(1, 40))
            b[i:j] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 50)))
        else:
            i = rng.randrange(22, len(b))
            b[i:i] = b[rng.randrange(22, len(b)):][: rng.randrange(1, 64)]
        return refit(bytes(b)) if len(b) > 22 and rng.random() < 0.9 else bytes(b)

    docs = []
    for _ in range(n):
        d = list(rng.choice(base))
        j = rng.randrange(len(d))
        d[j] = corrupt(d[j])
        docs.append(d)
    return docs


def ascii_paste_docs():
    """Plain-ASCII pastes whose length prefixes take 1, 2, 3 and 4 bytes (the decoder's arithmetic walk of plain-text chunks and
    k_elem_fill's flat copy see every prefix width), next to ordinary typing; two peers, both import orders."""
    from loro_amd import wire
    docs = []
    for n in (100, 200, 20000, 2200000):
        a, b = wire.Replica(21), wire.Replica(22)
        a.text_insert("text", 0, "head "); a.commit()
        b.merge_from(a); b.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids([a.export()], "text", wire.KIND_TEXT))
        a.text_insert("text", 2, "".join("abcdefghijklmnopqrstuvwxyz \n"[(i * 7 + i // 31) % 28] for i in range(n))); a.text_insert("text", 1, "xy"); a.text_delete("text", 4, 3); a.commit()
        b.text_insert("text", 5, "tail"); b.text_delete("text", 0, 1); b.commit()
        docs.append([a.export(), b.export()]); docs.append([b.export(), a.export()])
    return docs


def map_render_docs():
    """Maps of hundreds of entries for the renderer's two paths (lm_k_emit.h): 64 entries at once, one per lane, when every entry of
    the group is plain (a short key without escapes, an integer / bool / null value), entry by entry otherwise.  Integers at the
    edges of i64, groups that are plain throughout, groups with one string / float / nested / escaped-key / long-key / child entry in
    the middle, deleted keys, two peers writing the same keys."""
    from loro_amd import wire
    docs = []
    edge = [0, 1, -1, 9, 10, -10, 999999999, 1000000000, -1000000000, 10**18, -(10**18), 2**63 - 1, -(2**63), 123456789012345678, 42]
    a = wire.Replica(71)
    for i in range(300):
        a.map_set("plain", "k%03d" % i, edge[i % len(edge)] if i % 7 else (None if i % 14 else (i % 4 == 0)))
        if i % 40 == 39:
            a.commit()
    a.commit()
    docs.append([a.export()])
    b = wire.Replica(72)
    for i in range(260):
        if i in (5, 70, 130, 200, 259):
            v = ["s", 1.5, {"x": [1, 2]}, 'str"q', b"\x01\x02"][(i // 60) % 5]
        else:
            v = i * 1000003 - 77
        key = "key-%d" % i
        if i == 100:
            key = "quote\"d"
        if i == 150:
            key = "a-key-that-is-longer-than-twenty-four-bytes-%d" % i
        if i == 180:
            key = "tab\tkey"
        b.map_set("mixed", key, v)
        if i % 50 == 49:
            b.commit()
    b.map_set_container("mixed", "child-190", 2)
    b.commit()
    for i in range(0, 260, 9):
        b.map_delete("mixed", "key-%d" % i)
    b.commit()
    c = wire.Replica(70)
    c.merge_from(b)
    for i in range(0, 260, 5):
        c.map_set("mixed", "key-%d" % i, -i)
    c.commit()
    docs.append([b.export(), c.export()])
    # integers of every LEB128 width (the decoder's integer-values fast path finds eight values at once in a 128-byte window), chunks
    # with a string / null among them, change boundaries every 100 rows
    vals = [0, 1, -1, 63, 64, -64, -65, 8191, 8192, -8192, -8193, 2**20, -(2**20) - 1, 2**27 - 1, 2**27, 2**34, -(2**34) - 1, 2**41, 2**48,
            -(2**48) - 1, 2**55, 2**62, -(2**62), 2**63 - 1, -(2**63)]
    for rep in range(3):
        r = wire.Replica(50 + rep)
        for i in range(400):
            v = vals[(i * 7 + rep) % len(vals)]
            if rep == 1 and i % 37 == 0:
                v = "s%d" % i
            if rep == 2 and i % 41 == 0:
                v = None
            r.map_set("m", "k%d" % (i % 97), v)
            if i % 100 == 99:
                r.commit()
        r.commit()
        docs.append([r.export()])
    return docs


def nested_key_docs():
    """(good, bad): single-blob documents whose op values hold a nested map — a Map set, a List insert, a MovableList insert and
    set, a Text mark — and the same blobs with that nested map's key index patched beyond the block's key table (checksum
    redone).  The reference decodes every value in full with the block (value.rs read_value: keys.get(idx).ok_or(DataCorruption)):
    all of `bad` are rejected, whether or not the value ever reaches the state."""
    import struct
    from loro_amd import wire
    marker = b"\x03" + wire.sleb(7777)

    def patch(blob):
        at = blob.index(marker)            # ... 0x08 0x01 <key idx> 0x03 sleb(7777)
        assert blob[at - 3:at - 1] == b"\x08\x01" and blob[at - 1] < 0x60 and blob.count(marker) == 1
        tail = bytearray(blob[20:])
        tail[at - 1 - 20] = 0x63
        return b"loro" + b"\x00" * 12 + struct.pack("<I", wire.xxh32(bytes(tail))) + bytes(tail)

    good = []
    for which in range(6):
        r = wire.Replica(900 + which)
        r.map_set("m", "first", 1)
        r.text_insert("t", 0, "hello world")
        r.list_insert("l", 0, [1, 2, 3])
        r.mlist_insert("ml", 0, ["a", "b"])
        r.commit()
        v = {"inner": 7777}
        if which == 0:
            r.map_set("m", "second", v)
        elif which == 1:
            r.list_insert("l", 1, ["x", [v], "y"])
        elif which == 2:
            r.mlist_insert("ml", 1, [v])
        elif which == 3:
            r.mlist_set("ml", 0, v)
        elif which == 4:
            r.text_mark("t", 0, 5, "bold", v)
        else:
            r.map_set("m", "second", v)
            r.map_set("m", "second", 5)      # (the corrupt value loses: still rejected)
        r.commit()
        good.append([r.export()])
    return good, [[patch(d[0])] for d in good]


def nested_map_order_docs():
    """List items and Map values that are MAPS: key order (bytewise, not insertion), keys that share their first eight bytes, keys
    that are prefixes of one another, an empty key, DUPLICATE keys inside one encoded map (the last occurrence wins — what a map
    built by successive inserts holds, value.rs read_value), a 70-entry map (beyond the 64 entries the renderer orders in one
    pass), maps nested in maps in lists, and enough of them in one value to exhaust the renderer's pool."""
    from loro_amd import wire

    class PairMap(dict):   # a map value whose encoded entries are exactly these pairs, duplicates included
        def __init__(self, pairs):
            super().__init__()
            self.pairs = list(pairs)

        def items(self):
            return self.pairs

        def __len__(self):
            return len(self.pairs)

    r = wire.Replica(77)
    long = ["prefix__%s" % x for x in ("b", "a", "", "aa", "ab", "zzzzzzzzzz", "a\u00e9")]
    vals = [
        PairMap([("k", 1), ("a", 2), ("k", 3), ("", 4), ("a", None)]),
        PairMap([(k, i) for i, k in enumerate(long)] + [("prefix__", "short"), ("prefix_", 0.5), ("prefix__a", "again")]),
        {"key%02d" % (i * 37 % 70): i for i in range(70)},
        PairMap([("m", PairMap([("y", [1, {"q": 1, "p": 2}]), ("x", PairMap([("b", 1), ("b", 2), ("a", 3)])), ("y", "last")])), ("l", [PairMap([]), {}])]),
        [{"n%d" % j: {"i%d" % i: i for i in range(60)} for j in range(6)}],     # 6 x 60 entries under one frame: the pool runs out
        {},
    ]
    r.list_insert("list", 0, vals)
    r.map_set("map", "m", PairMap([("z", 1), ("y", PairMap([("d", 1), ("c", 2), ("d", 3)])), ("z", 2)]))
    r.commit()
    return [[r.export()]]


def big_blob_checksum_docs(n=40, seed=3):
    """Blobs large enough for the wave-per-four-blobs envelope checksum (k_hash_big_blobs: >= 32 KB), of many different lengths —
    every tail length 0..15, stripe counts that are not multiples of 64, neighbours in a wave of very different lengths — some with
    one bit of the stored checksum flipped (LM_CHECKSUM_MISMATCH for that document only)."""
    import random, struct
    from loro_amd import wire
    rng = random.Random(seed)
    docs = []
    for k in range(n):
        r = wire.Replica(1000 + k)
        size = 33000 + rng.randrange(0, 70000) if k % 5 else 33000 + 1024 * rng.randrange(0, 40) + k % 16
        r.text_insert("text", 0, "".join(rng.choice("abcdefgh") for _ in range(size)))
        r.commit()
        b = r.export()
        if k % 7 == 3:
            b = b[:16] + struct.pack("<I", struct.unpack("<I", b[16:20])[0] ^ (1 << (k % 32))) + b[20:]
        docs.append([b])
    return docs


def huge_run_column_docs():
    """columns of a few bytes whose AnyRle run count is near 2^28 (a decoder that steps through a run value by value
    spins 2^28 times per column).  Each document is a small valid history whose one block has one column replaced: the value-type
    column as ONE run of 2^28 - 3 values, the len column likewise, a delete-start column with a giant run appended, and a prop
    column (DeltaRle) with a giant run of a delta that overflows i64.  All of them are DecodeError for the reference's reader
    (serde_columnar decodes a column in full: more values than rows / unequal delete-start columns)."""
    from loro_amd import wire
    docs, names = [], []
    giant = (1 << 28) - 3

    def build(patch_name, nth, make):
        orig = getattr(wire, patch_name)
        calls = {"n": 0}

        def patched(vals):
            calls["n"] += 1
            out = orig(vals)
            return make(out) if calls["n"] == nth else out
        setattr(wire, patch_name, patched)
        try:
            r = wire.Replica(7)
            r.text_insert("text", 0, "hello world")
            r.text_delete("text", 2, 3)
            r.text_insert("text", 1, "xy")
            r.text_delete("text", 0, 1)
            r.commit()
            return [r.export()]
        finally:
            setattr(wire, patch_name, orig)

    docs.append(build("enc_rle_u8", 1, lambda b: wire.zigzag(giant) + b"\x05")); names.append("value-type column: one giant run")
    # enc_any_rle_uvar calls per block: dep counts, dep peer idx, msg lens, then the len column (4th)
    docs.append(build("enc_any_rle_uvar", 4, lambda b: wire.zigzag(giant) + wire.uleb(1))); names.append("len column: one giant run")
    # enc_delta_rle calls per block: container idx, prop, then the three delete-start columns (3rd..5th)
    docs.append(build("enc_delta_rle", 4, lambda b: b + wire.zigzag(giant) + wire.zigzag(0))); names.append("delete-start counter column: giant run appended")
    docs.append(build("enc_delta_rle", 2, lambda b: b + wire.zigzag(giant) + wire.zigzag((1 << 62)))); names.append("prop column: giant run of an overflowing delta")
    docs.append(build("enc_delta_rle", 5, lambda b: b + wire.zigzag(giant) + wire.zigzag(-(1 << 40)))); names.append("delete-start len column: giant run appended")
    return names, docs


def damaged_change_meta_docs():
    """the two change_meta columns — timestamps (DeltaOfDelta) and commit-message lengths (AnyRle) — fail with
    LoroError::DecodeDataCorruptionError whatever the decoder's complaint is (block_encode.rs:563-571: both `take_n_finalize` calls
    and `DeltaOfDeltaDecoder::new` are `.map_err(|_| LoroError::DecodeDataCorruptionError)`), unlike the header columns of
    block_meta_encode.rs (DecodeError).  (names, documents): a small valid history whose one block has one of the two columns
    replaced — every one of them is LM_DATA_CORRUPTION (3)."""
    from loro_amd import wire
    names, docs = [], []

    def build(patch_name, nth, make):
        orig = getattr(wire, patch_name)
        calls = {"n": 0}

        def patched(vals):
            calls["n"] += 1
            out = orig(vals)
            return make(out) if calls["n"] == nth else out
        setattr(wire, patch_name, patched)
        try:
            r = wire.Replica(9)
            r.text_insert("text", 0, "hello"); r.commit()
            r.map_set("m", "k", 1); r.commit()
            r.text_insert("text", 2, "xy"); r.commit()
            return [r.export()]
        finally:
            setattr(wire, patch_name, orig)
    # enc_any_rle_uvar calls per block: dep counts, dep peer idx, MESSAGE LENGTHS (3rd), the len column
    docs.append(build("enc_any_rle_uvar", 3, lambda b: b"")); names.append("message lengths: no value at all")
    docs.append(build("enc_any_rle_uvar", 3, lambda b: wire.zigzag(2) + wire.uleb(0))); names.append("message lengths: two values for three changes")
    docs.append(build("enc_any_rle_uvar", 3, lambda b: wire.zigzag(7) + wire.uleb(0))); names.append("message lengths: a run that announces seven values")
    docs.append(build("enc_any_rle_uvar", 3, lambda b: wire.zigzag(3) + wire.uleb(9))); names.append("message lengths beyond the message bytes")
    # enc_delta_of_delta calls per block: dep counters, lamports, TIMESTAMPS (3rd)
    docs.append(build("enc_delta_of_delta", 3, lambda b: b"\x07" + b[1:])); names.append("timestamps: an option tag that is neither 0 nor 1")
    docs.append(build("enc_delta_of_delta", 3, lambda b: b[:-1] if len(b) > 10 else b"\x01")); names.append("timestamps: the stream is cut short")
    return names, docs


# a 129-byte blob (one change of a map / list / text session, no foreign dependency) whose EMPTY dependency-counter column carries the
# option tag 3a instead of 00: DecodeError in the reference (test_a_damaged_option_tag_in_front_of_an_empty_delta_of_delta_column)
DOD_TAG_BLOB_HEX = "6c6f726f0000000000000000000000006abb1f0600046a0f050f05011101f9456d4caa000000000101003a0000000501000001001003040102000004010000020401010006110474657874036d6170026b32046c6973740016010404010004020405000400040105040b04010304010010036a687807030301030205017a070101"


def map_sessions(n, base=61000, scalar_only=True):
    """LWW Map documents for the fused decode -> LWW kernel (lm_k_map_fused.h): 1-4 peers writing / deleting keys of one or two root
    Maps — integers of every width, strings (empty, long), floats, bools, null, binary; keys that are empty, non-ASCII, quoted, with a
    tab, 20-43 bytes long (a key table the kernel's candidate test cannot take: the document is replayed through the row tables) —
    commits of a few rows each, pairwise syncs, a duplicated blob, a missing peer (pending changes).  scalar_only=False adds nested
    list / map values (the kernel leaves those documents to the row tables too)."""
    import random
    docs = []
    for s in range(n):
        seed = base + s
        rng = random.Random(seed)
        n_peers = rng.randint(1, 4)
        reps = [wire.Replica(1000 + seed * 10 + i) for i in range(n_peers)]
        keys = ["k%d" % i for i in range(rng.randint(1, 40))] + ["", "ключ", "a\"b", "long-key-%s" % ("x" * rng.choice([3, 20, 30, 40])), "tab\tkey"][: rng.randint(0, 5)]
        names = ["map"] + (["m2"] if rng.random() < 0.3 else [])
        for step in range(rng.randint(5, 400)):
            r = rng.choice(reps)
            nm = rng.choice(names); key = rng.choice(keys)
            if rng.random() < 0.15:
                r.map_delete(nm, key)
            else:
                vals = [None, True, False, rng.randint(-10**14, 10**14), rng.randint(-3, 3), "s%d" % rng.randint(0, 999), b"\x00\xff",
                        rng.random() * 10 ** rng.randint(-5, 9), "x" * rng.randint(0, 200)]
                if not scalar_only and s % 3 == 0:
                    vals += [[1, 2, "z"], {"a": 1, "b": [2]}]
                r.map_set(nm, key, rng.choice(vals))
            if rng.random() < 0.3:
                r.commit()
            if rng.random() < 0.1 and n_peers > 1:
                a, b = rng.sample(reps, 2); a.commit(); b.commit(); a.merge_from(b)
        for r in reps:
            r.commit()
        blobs = _fuzz.blobs_of(reps, rng)
        if rng.random() < 0.2 and blobs:
            blobs.append(blobs[0])
        if rng.random() < 0.15 and len(blobs) > 1:
            blobs = blobs[1:]
        if blobs:
            docs.append(blobs)
    return docs


def damaged_map_docs(n, seed=1):
    """map_sessions() documents with one blob damaged (byte flips, truncation, splices; envelope checksum re-fitted)"""
    import random, struct
    import _oracle
    rng = random.Random(seed)

    def refit(blob):
        body = blob[20:]
        return blob[:16] + struct.pack("<I", _oracle.xxh32(body)) + body

    def corrupt(blob):
        b = bytearray(blob)
        k = rng.random()
        if k < 0.6:
            for _ in range(rng.choice([1, 1, 2, 5])):
                i = rng.randrange(22, len(b))
                b[i] = rng.choice([b[i] ^ (1 << rng.randrange(8)), rng.randrange(256), 0xFF, 0x80, 0])
        elif k < 0.75:
            del b[rng.randrange(22, len(b)):]
        elif k < 0.9:
            i = rng.randrange(22, len(b)); j = min(len(b), i + rng.randrange(1, 40))
            b[i:j] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 50)))
        else:
            i = rng.randrange(22, len(b))
            b[i:i] = b[rng.randrange(22, len(b)):][: rng.randrange(1, 64)]
        return refit(bytes(b)) if len(b) > 22 else bytes(b)
    # (no duplicated blobs here: a damaged copy beside the intact one is a corner of the pending-change bookkeeping of its own, NEXT.md)
    base = [d for d in map_sessions(24, base=62000 + seed * 100) if len(set(d)) == len(d)]
    docs = []
    for i in range(n):
        d = list(rng.choice(base)); j = rng.randrange(len(d)); d[j] = corrupt(d[j]); docs.append(d)
    return docs
