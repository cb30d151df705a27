"""lm_export: the updates a staged document holds beyond a version, re-encoded from its own blobs — through the
kernel-logic harness (the export itself is host code of the product library; the version it exports up to comes from the run)."""
import json, os, random, struct
import pytest

import _emu, _fuzz, _oracle, _resident
from loro_amd._cabi import Context
from loro_amd import wire, workload

FX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))
BLOB = {k: bytes.fromhex(v) for k, v in FX["blobs"].items()}


def _ctx():
    return Context(_emu.binding())


def check_export_roundtrips(make_ctx):
    # the Rust-written fixture comes back byte for byte from the empty version (its blocks already are in (peer, counter) order)
    with make_ctx() as c:
        c.stage([[BLOB["updates.blob"]], [BLOB["updates.blob"], BLOB["updates.ts.blob"]], [BLOB["fugue-left.ts.blob"], BLOB["fugue-right.ts.blob"]]])
        c.run()
        assert c.export(0) == BLOB["updates.blob"]
        assert c.export(1) == BLOB["updates.blob"]              # the same history twice: every block once
        both = c.export(2)
        assert _oracle.merge([both])[1] == b'{"text":"Hello World!"}'
    # random sessions: blobs in any order, duplicated, overlapping exports — one blob with everything, equal to the whole
    docs = []
    for seed in range(40):
        rng = random.Random(seed)
        reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 4), n_steps=rng.randint(40, 160), kinds=("text", "list", "map"), styles=True) if seed % 3 else _fuzz.movable_session(seed, n_peers=3, n_steps=80, nested=True)
        blobs = _resident.chunked_blobs(reps, rng) + [reps[0].export()]
        rng.shuffle(blobs)
        docs.append(blobs)
    want = _oracle.merge_batch(docs, threads=4)
    with make_ctx() as c:
        c.stage(docs); c.run()
        got = c.fetch()
        assert got == want
        exported = [[c.export(i)] for i in range(len(docs))]
    again = _oracle.merge_batch(exported, threads=4)
    for i, (a, b) in enumerate(zip(again, want)):
        assert a[:3] == b[:3] and a[3] == 0, i              # the same value and version; nothing pending inside an export
    return docs, exported


def test_export_from_the_empty_version():
    check_export_roundtrips(_ctx)


def check_export_from_versions(make_ctx):
    """a document that holds version V imports export(from = V) of the full document: the full document.  V = what some replica
    knew at some point (change boundaries) and, for single-writer histories, any counter (cuts inside changes and op runs)."""
    sessions = []
    fulls = []
    for seed in range(24):
        rng = random.Random(1000 + seed)
        snaps = []
        reps = _fuzz.random_session(1000 + seed, n_peers=3, n_steps=150, kinds=("text", "list", "map"), snapshots=snaps, styles=True)
        full = _fuzz.blobs_of(reps, rng)
        if not snaps:
            continue
        v_front, v_blob = rng.choice(snaps)
        fulls.append(full)
        sessions.append((v_front, v_blob))
    with make_ctx() as c:
        c.stage(fulls); c.run()
        want_full = c.fetch()
        assert want_full == _oracle.merge_batch(fulls, threads=4)
        deltas = []
        for i, (v_front, v_blob) in enumerate(sessions):
            vv = _oracle.merge([v_blob])[2]                 # the version vector of V (VersionVector::encode bytes)
            deltas.append(c.export(i, vv))
    with make_ctx() as c:
        c.stage([[v_blob] for _, v_blob in sessions])
        c.import_more([[] for _ in sessions]); c.run()
        c.import_more([[d] for d in deltas]); c.run()
        got = c.fetch()
    for i, (g, w) in enumerate(zip(got, want_full)):
        assert g == w, (i, g[:2], w[:2])
    # the oracle agrees about what a delta is: prefix + delta == full
    for i, (v_front, v_blob) in enumerate(sessions):
        assert _oracle.merge([v_blob, deltas[i]]) == want_full[i], i


def test_export_from_a_version():
    check_export_from_versions(_ctx)


def test_cuts_inside_changes_and_runs():
    """one writer, one change of several runs: export from every counter — the sliced first change must carry exactly the rest
    (Text insert sliced by unicode scalars, List insert by items, forward and backward deletes by DeleteSpan::slice)"""
    a = wire.Replica(7)
    a.text_insert("text", 0, "héllo wörld"); a.list_insert("list", 0, [1, "two", [3], {"k": 4}, None])
    a.text_delete("text", 2, 3)                # forward delete of 3
    a.commit()
    a.text_insert("text", 8, "!!"); a.commit()
    for _ in range(3):                         # three backspaces merge into one backward delete span
        a.text_delete("text", len(a.seq[wire.root_cid("text", wire.KIND_TEXT)]) - 1, 1)
    a.commit()
    full = a.export()
    total = a.vv[7]
    with _ctx() as c:
        c.stage([[full]]); c.run()
        want = c.fetch()[0]
        assert want == _oracle.merge([full])
        for k in range(total + 1):
            d = c.export(0, wire.encode_vv({7: k}) if k else None)
            # imported next to the full history the slice changes nothing; and its first change starts exactly at k
            assert _oracle.merge([full, d]) == want, k
            with _ctx() as c2:
                c2.stage([[d]]); c2.run()
                r = c2.fetch()[0]
                assert r[0] == 0 and r[3] == (total - k if k else 0), (k, r[0], r[3])   # everything waits for the missing prefix [0, k)
    # and cut + prefix: a replica that typed the prefix itself (same peer, same ops up to a change boundary) imports the rest
    b = wire.Replica(7)
    b.text_insert("text", 0, "héllo wörld"); b.list_insert("list", 0, [1, "two", [3], {"k": 4}, None]); b.text_delete("text", 2, 3); b.commit()
    with _ctx() as c:
        c.stage([[full]]); c.run()
        d = c.export(0, wire.encode_vv(dict(b.vv)))
    assert _oracle.merge([b.export(), d]) == want


def test_a_sliced_run_carries_exactly_the_rest():
    """the prefix typed independently (same peer, same first k atoms), then export(from = {peer: k}) of the full history imported:
    k inside a Text insert, inside a List insert, inside a forward delete"""
    def full_replica():
        a = wire.Replica(7)
        a.text_insert("text", 0, "héllo wörld"); a.list_insert("list", 0, [1, "two", [3], {"k": 4}, None]); a.text_delete("text", 2, 3)
        a.commit()
        return a
    full = full_replica().export()
    want = _oracle.merge([full])
    prefixes = []
    p = wire.Replica(7); p.text_insert("text", 0, "hél"); p.commit(); prefixes.append(p)                                   # k = 3
    p = wire.Replica(7); p.text_insert("text", 0, "héllo wörld"); p.list_insert("list", 0, [1, "two"]); p.commit(); prefixes.append(p)   # k = 13
    p = wire.Replica(7); p.text_insert("text", 0, "héllo wörld"); p.list_insert("list", 0, [1, "two", [3], {"k": 4}, None]); p.text_delete("text", 2, 1); p.commit(); prefixes.append(p)   # k = 17
    with _ctx() as c:
        c.stage([[full]]); c.run()
        deltas = [c.export(0, wire.encode_vv(dict(p.vv))) for p in prefixes]
    for p, d in zip(prefixes, deltas):
        assert _oracle.merge([p.export(), d]) == want, dict(p.vv)
    with _ctx() as c:
        c.stage([[p.export()] for p in prefixes]); c.import_more([[] for _ in prefixes]); c.run()
        c.import_more([[d] for d in deltas]); c.run()
        assert all(g == want for g in c.fetch())
