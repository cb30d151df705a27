"""Richtext values (lm_richtext): cases shared by the kernel-logic (CPU) and the GPU tests.  The checker is the
oracle's Doc::to_richtext (oracle/lo_doc.hpp), itself pinned on the reference's known answers below."""
import json
import random

import _fuzz, _oracle, _resident
from loro_amd import wire


def known_answers():
    """[(name, blobs, expected richtext value of root Text "text" as Python data)] — reference tests restated through the blob
    writer (wire.Replica places the anchors where TextHandler::mark places them for a plain range: StyleStart in front of the
    first scalar, StyleEnd behind the last, handler.rs mark_with_transaction):
      crates/loro/tests/loro_rust_test.rs:448-474 richtext_test (mark, then unmark 3..5 = a mark with value null)
      crates/loro/tests/loro_rust_test.rs:476-498 sync (the mark arrives as an update of another peer)
      crates/loro/src/lib.rs:2750-2772 get_richtext_value doc example"""
    out = []
    d = wire.Replica(1)
    d.text_insert("text", 0, "Hello world!"); d.text_mark("text", 0, 5, "bold", True); d.commit()
    out.append(("richtext_test: mark", [d.export()], [{"insert": "Hello", "attributes": {"bold": True}}, {"insert": " world!"}]))
    # unmark(3..5): scalars 3..5 sit behind the first Start anchor → entities 4..6
    d.text_mark("text", 4, 6, "bold", None); d.commit()
    out.append(("richtext_test: unmark", [d.export()], [{"insert": "Hel", "attributes": {"bold": True}}, {"insert": "lo world!"}]))
    a, b = wire.Replica(1), wire.Replica(2)
    a.text_insert("text", 0, "Hello world!"); a.commit()
    b.merge_from(a)
    b.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids([a.export()], "text", wire.KIND_TEXT))
    b.text_mark("text", 0, 5, "bold", True); b.commit()
    own = wire.Replica(2); own.changes = {2: b.changes[2]}
    out.append(("sync", [a.export(), own.export()], [{"insert": "Hello", "attributes": {"bold": True}}, {"insert": " world!"}]))
    e = wire.Replica(3)
    e.text_insert("text", 0, "Hello world!"); e.text_mark("text", 0, 5, "bold", True); e.commit()
    out.append(("doc example", [e.export()], [{"insert": "Hello", "attributes": {"bold": True}}, {"insert": " world!"}]))
    return out


def reference_held():
    """[(name, blobs, expected richtext value of root Text "text")] — answers the REFERENCE's own tests hold for blobs it ships
    (unlike known_answers(), nothing here goes through this repo's writer): crates/loro/tests/loro_js_interop.rs:86-94 asserts
    `doc.get_text("text").get_richtext_value()` of runtime-snapshot.ts.blob == [{"insert":"b","attributes":{"bold":true}}] and
    (`to_delta()` equality, :86-89) the same spans for runtime-updates.ts.blob.  The blobs come from
    tests/golden/reference_fixtures.json (make_reference_fixtures.py).  Both documents also hold Tree / Counter containers, so
    their status is LM_UNSUPPORTED (4) with everything in scope rendered."""
    import os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))["blobs"]
    want = [{"insert": "b", "attributes": {"bold": True}}]
    return [(n, [bytes.fromhex(fx[n])], want) for n in ("runtime-snapshot.ts.blob", "runtime-updates.ts.blob")]


def hand_cases():
    """[(name, blobs)] — shapes the rule has to get right: two peers mark the same range concurrently (greater (lamport, peer)
    decides), overlapping marks of different keys, a mark whose End anchor was deleted / whose Start anchor was deleted,
    text typed inside and at both edges of a range, an unmark over part of a range, equal values from different ops (one span),
    several Text containers (a child Text in a Map, a Text without any visible scalar, a Text that holds only anchors)."""
    out = []
    a, b = wire.Replica(10), wire.Replica(20)
    a.text_insert("text", 0, "0123456789"); a.commit()
    b.merge_from(a); b.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids([a.export()], "text", wire.KIND_TEXT))
    a.text_mark("text", 2, 6, "color", "red"); a.commit()
    b.text_mark("text", 4, 8, "color", "blue"); b.text_mark("text", 0, 3, "bold", True); b.commit()
    out.append(("concurrent marks of one key", _fuzz.blobs_of([a, b])))
    c = wire.Replica(5)
    c.text_insert("text", 0, "abcdefgh"); c.text_mark("text", 1, 5, "bold", True); c.commit()
    c.text_delete("text", 6, 1); c.commit()      # the End anchor (entity 6: a b c d e | End) … entity positions: 0 a,1 S,2 b..5 e,6 E
    out.append(("end anchor deleted", [c.export()]))
    c2 = wire.Replica(6)
    c2.text_insert("text", 0, "abcdefgh"); c2.text_mark("text", 1, 5, "bold", True); c2.commit()
    c2.text_delete("text", 1, 1); c2.commit()    # the Start anchor
    out.append(("start anchor deleted", [c2.export()]))
    t = wire.Replica(7)
    t.text_insert("text", 0, "abcd"); t.text_mark("text", 1, 3, "bold", True); t.commit()   # a S b c E d
    t.text_insert("text", 3, "X"); t.text_insert("text", 1, "L"); t.text_insert("text", 7, "R"); t.commit()
    out.append(("typing inside and at the edges", [t.export()]))
    u = wire.Replica(8)
    u.text_insert("text", 0, "abcdef"); u.text_mark("text", 0, 6, "link", "u1"); u.text_mark("text", 3, 5, "link", "u1"); u.text_mark("text", 1, 2, "em", 1); u.commit()
    out.append(("equal values from different ops", [u.export()]))
    n = wire.Replica(9)
    n.map_set("m", "k", 1); n.text_insert("empty", 0, "zz"); n.text_delete("empty", 0, 2); n.text_insert("t2", 0, "plain \"text\"\n"); n.commit()
    n.text_insert("only_anchors", 0, "q"); n.text_mark("only_anchors", 0, 1, "b", True); n.text_delete("only_anchors", 1, 1); n.commit()
    out.append(("several text containers", [n.export()]))
    return out


def fuzz_docs(n, base=5000, n_steps=90, **kw):
    return [_fuzz.blobs_of(_fuzz.random_session(base + s, n_peers=2 + s % 3, n_steps=n_steps, kinds=("text",) if s % 3 else ("text", "list", "map"),
                                                styles="rich", sync_prob=0.1, **kw)) for s in range(n)]


def nested_docs(n, base=5200):
    return [_fuzz.blobs_of(_fuzz.nested_session(base + s, n_peers=3, n_steps=120)) for s in range(n)]


def checkout_cases(n=6, base=5400):
    """(docs, frontiers): every few recorded versions of rich sessions, incl. versions that cut a StyleStart from its StyleEnd"""
    docs, fronts = [], []
    for s in range(n):
        snaps = []
        reps = _fuzz.random_session(base + s, n_peers=3, n_steps=70, kinds=("text",), styles="rich", snapshots=snaps)
        full = _fuzz.blobs_of(reps)
        for fr, _ in snaps[:: max(1, len(snaps) // 8)]:
            docs.append(list(full)); fronts.append(wire.encode_frontiers(fr))
    r = wire.Replica(77)
    r.text_insert("text", 0, "abcdef"); r.text_mark("text", 1, 4, "bold", True); r.text_insert("text", 0, "Z"); r.commit()
    blob = [r.export()]
    for c in range(0, 9):   # frontiers at every op of the change: between the two anchors as well
        docs.append(list(blob)); fronts.append(wire.encode_frontiers([(77, c)]))
    return docs, fronts


def resident_sessions(seeds, n_steps=5):
    out = []
    for seed in seeds:
        rng = random.Random(seed * 11 + 3)
        snaps = []
        reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 4), n_steps=rng.randint(60, 140), kinds=("text",), snapshots=snaps, styles="rich")
        out.append(_resident.plan_steps(_resident.chunked_blobs(reps, rng), rng, n_steps, versions=[v for v, _ in snaps]))
    return out


def run_resident(ctx, sessions):
    """per step: [(status, richtext bytes)] from lm_richtext behind every lm_run"""
    got = []
    for k in range(len(sessions[0])):
        docs = [s[k][0] for s in sessions]
        fr = [s[k][1] for s in sessions]
        if k == 0:
            ctx.stage(docs, fr)
            ctx.import_more([[] for _ in docs], fr)
        else:
            ctx.import_more(docs, fr)
        ctx.run()
        res = ctx.fetch()
        rt = ctx.richtext()
        got.append([(r[0], t[0], t[1]) for r, t in zip(res, rt)])
    return got


def oracle_resident(sessions):
    out = [[] for _ in sessions[0]]
    for s in sessions:
        o = _oracle.Session()
        o.want_richtext = True
        for k, (blobs, f) in enumerate(s):
            st = o.step(blobs, f)
            out[k].append((st[0], o.richtext()))
        o.close()
    return out


def same(got, want, what=""):
    """got / want: [(status, bytes)].  Equal statuses, equal bytes (members in the bytewise order of their JSON-encoded keys on both sides)"""
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g[0] == w[0], (what, i, g[0], w[0])
        if w[0] == 0:
            assert g[1] == w[1], (what, i, g[1][:400], w[1][:400])


import functools


@functools.lru_cache(maxsize=None)
def _base_rich():
    return [_fuzz.blobs_of(_fuzz.random_session(9000 + s, n_peers=3, n_steps=80, kinds=("text",), styles="rich")) for s in range(12)]


@functools.lru_cache(maxsize=None)
def _base_mixed():
    base = [_fuzz.blobs_of(_fuzz.random_session(9100 + s, n_peers=3, n_steps=90, kinds=("text", "list", "map"), styles="rich")) for s in range(8)]
    base += [_fuzz.blobs_of(_fuzz.nested_session(9200 + s, n_steps=100)) for s in range(6)]
    base += [_fuzz.blobs_of(_fuzz.movable_session(9300 + s, n_peers=3, n_steps=90, nested=s % 2 == 0)) for s in range(6)]
    return base


def damaged_docs(n=400, seed=5):
    """rich-text sessions (marks of several keys, multi-byte scalars) with one blob damaged by byte flips and the envelope checksum re-fitted"""
    import struct
    rng = random.Random(seed)
    base = _base_rich()

    def refit(blob):
        body = blob[20:]
        return blob[:16] + struct.pack("<I", _oracle.xxh32(body)) + body
    docs = []
    for _ in range(n):
        d = list(rng.choice(base)); j = rng.randrange(len(d)); b = bytearray(d[j])
        for _ in range(rng.choice([1, 1, 2, 4])):
            k = rng.randrange(22, len(b))
            b[k] = rng.choice([b[k] ^ (1 << rng.randrange(8)), rng.randrange(256), 0xFF, 0x80, 0])
        d[j] = refit(bytes(b))
        docs.append(d)
    return docs


def check_damaged(run, docs):
    """run(docs) -> (merge results, richtext results).  What both sides accept is rendered alike (JSON, version vector, richtext), and the
    device never renders a document the oracle rejects.  Returns (both accept, only the oracle accepts)."""
    want, want_j = _oracle.richtext_batch(docs, threads=16, with_merge=True)
    got_j, got = run(docs)
    n_both = n_oracle_only = 0
    for i in range(len(docs)):
        g, w = got_j[i], want_j[i]
        assert not (g[0] == 0 and w[0] not in (0, 4)), (i, "the device rendered a document the oracle rejects", w[0])
        if g[0] == 0 and w[0] == 0:
            n_both += 1
            assert g == w and got[i] == want[i], (i, g[1][:200], w[1][:200])
        n_oracle_only += g[0] != 0 and w[0] == 0
    return n_both, n_oracle_only


def damaged_mixed_docs(n=600, seed=1):
    """rich-text + list + map sessions, nested containers and MovableLists, one blob damaged per document by byte flips, truncation, a
    spliced range or a duplicated range, checksum re-fitted (the corpus that turned up the last-lamport rule, the surplus run in the
    message-length column and the insert beyond the end)"""
    import struct
    rng = random.Random(seed)
    base = _base_mixed()

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
    docs = []
    for _ in range(n):
        d = list(rng.choice(base)); j = rng.randrange(len(d)); d[j] = corrupt(d[j])
        docs.append(d)
    return docs
