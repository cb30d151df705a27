"""MovableList on the device path, checked in the kernel-logic harness (the HIP kernels compiled for the host): decode of
ListMove / ListSet rows (both block decoders), the move as delete + insert in both integrate kernels with its retreat /
forward, the per-element LWW of position and value (k_mlist_post) and the rendering — against the oracle, which
test_oracle_movable.py pins to the reference's known answers."""
import json, os, random
import pytest

import _oracle, _emu, _fuzz
from loro_amd import wire

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_fixtures.json")))
BLOB = {k: bytes.fromhex(v) for k, v in FX["blobs"].items()}
ML = wire.KIND_MOVABLE


def _check(docs, frontiers=None, run=None):
    got = (run or _emu.merge_batch)(docs, frontiers)
    want = _oracle.merge_batch(docs, frontiers=frontiers)
    for i, (g, w) in enumerate(zip(got, want)):
        if w[0] == 0:
            assert g == w, f"doc {i}: device={g[:3]!r} oracle={w[:3]!r}"
        else:
            assert g[0] == w[0] and (w[0] != 4 or g[1] == w[1]), f"doc {i}: device={g[:2]!r} oracle={w[:2]!r}"
    return got


def session_docs(seeds, **kw):
    docs = []
    for seed in seeds:
        reps = _fuzz.movable_session(seed, **kw)
        rng = random.Random(seed)
        docs.append(_fuzz.blobs_of(reps, rng))
        docs.append([r.export() for r in reps])          # overlapping histories: drop / slice known changes
    return docs


def checkout_docs(seeds, **kw):
    docs, fronts = [], []
    for seed in seeds:
        snaps = []
        reps = _fuzz.movable_session(seed, snapshots=snaps, **kw)
        blobs = _fuzz.blobs_of(reps)
        rng = random.Random(seed)
        for version, _ in rng.sample(snaps, min(4, len(snaps))):
            docs.append(blobs)
            fronts.append(wire.encode_frontiers(version))
    return docs, fronts


def known_answer_docs():
    """(docs, check): the reference's fixtures and the mov.rs / movable_list_state.rs scripts."""
    d1 = wire.Replica(1)
    for i, v in enumerate((1, 2, 3)):
        d1.mlist_insert("list", i, [v])
    d1.commit()
    d2 = wire.Replica(2)
    d2.merge_from(d1)
    d2.set_visible("list", ML, list(d1.seq[wire.root_cid("list", ML)]))
    d1.mlist_move("list", 0, 2); d2.mlist_move("list", 0, 1)
    d1.commit(); d2.commit()
    d = wire.Replica(8)
    d.mlist_insert("list", 0, [1]); d.mlist_insert("list", 1, [0]); d.mlist_move("list", 0, 1); d.mlist_move("list", 1, 0)
    d.mlist_move("list", 0, 1); d.mlist_insert("list", 2, [3]); d.mlist_set("list", 2, 2)
    d.commit()
    e = wire.Replica(9)
    e.mlist_insert("gone", 0, ["a", "b"]); e.mlist_delete("gone", 0, 2); e.list_insert("l", 0, [1]); e.list_delete("l", 0, 1)
    e.commit()
    docs = [[BLOB["updates.blob"]], [BLOB["updates.ts.blob"]], [BLOB["runtime-updates.ts.blob"]],
            [d1.export(), d2.export()], [d2.export(), d1.export()], [d.export()], [e.export()]]

    def check(got):
        want = FX["json"]["snapshot.deep.json"]
        for g in got[:2]:
            v = json.loads(g[1])
            assert g[0] == 4 and v["mlist"] == want["mlist"] == [] and v["map"]["child_mlist"] == want["map"]["child_mlist"]
        assert json.loads(got[2][1])["movable"] == FX["json"]["runtime.expected.json"]["movable"] == ["z", "x"]
        assert got[3][:2] == got[4][:2] == (0, b'{"list":[2,1,3]}')                    # crates/loro/tests/mov.rs:13-62
        assert got[5][:2] == (0, b'{"list":[0,1,2]}')                                  # movable_list_state.rs:1960-2031
        assert got[6][:2] == (0, b'{"gone":[]}')     # a MovableList exists once an element was made; an emptied List does not
    return docs, check


def test_known_answers():
    docs, check = known_answer_docs()
    check(_check(docs))


def existence_docs():
    """(docs, frontiers, expected JSON): which MovableList roots the state store holds."""
    r = wire.Replica(31)
    r.map_set("m", "k", 1); r.commit()
    v1 = list(r.frontiers)
    r.mlist_insert("ml", 0, ["a", "b"]); r.commit()
    v2 = list(r.frontiers)
    r.mlist_move("ml", 0, 1); r.mlist_set("ml", 0, "B"); r.commit()
    v3 = list(r.frontiers)
    r.mlist_delete("ml", 0, 2); r.commit()
    blob = [r.export()]
    f = wire.encode_frontiers
    docs = [blob] * 5
    fronts = [None, f(v1), f(v2), f(v3), f([])]
    want = [b'{"m":{"k":1},"ml":[]}',              # latest: everything deleted, the list is known
            b'{"m":{"k":1},"ml":[]}',              # checked out before the first insert: the import already created the state
            b'{"m":{"k":1},"ml":["a","b"]}', b'{"m":{"k":1},"ml":["B","a"]}',
            b'{"m":{},"ml":[]}']                   # the empty version
    return docs, fronts, want


@pytest.mark.parametrize("span", ["1", "0"])
def test_state_existence_and_checkout_of_moves(monkeypatch, span):
    monkeypatch.setenv("LM_SPAN", span)
    docs, fronts, want = existence_docs()
    got = _check(docs, fronts)
    assert [g[1] for g in got] == want


@pytest.mark.parametrize("variant", ["span", "element", "lane-decoder", "retry"])
def test_random_sessions(monkeypatch, variant):
    if variant == "element":
        monkeypatch.setenv("LM_SPAN", "0")
    if variant == "lane-decoder":
        monkeypatch.setenv("LM_DECODE", "0")
    if variant == "retry":
        monkeypatch.setenv("LM_DIR_OPT_MAX", "4")    # the optimistic directory overflows: documents are re-run
    docs = session_docs(range(100, 110)) + session_docs(range(200, 206), nested=True, n_steps=140, n_peers=4)
    _check(docs)


def test_multi_leaf_lists_and_many_moves():
    docs = session_docs(range(300, 303), bulk=400, n_steps=260, sync_prob=0.1)
    got = _check(docs)
    assert all(len(json.loads(g[1])["ml"]) > 200 for g in got)


@pytest.mark.parametrize("span", ["1", "0"])
def test_checkouts(monkeypatch, span):
    monkeypatch.setenv("LM_SPAN", span)
    docs, fronts = checkout_docs(range(400, 408), nested=True, n_steps=100)
    assert len(docs) >= 16
    _check(docs, fronts)


def sliced_docs():
    """One peer's history exported twice with different change boundaries: the second change is sliced on import, and the
    slice cuts away a move and a set row (or, delivered the other way round, waits as pending)."""
    r = wire.Replica(7)
    r.mlist_insert("ml", 0, ["a", "b", "c"])            # counters 0-2
    r.mlist_move("ml", 0, 2)                            # 3
    r.mlist_set("ml", 0, "B")                           # 4
    r.mlist_insert("ml", 1, ["d"])                      # 5
    r.mlist_delete("ml", 0, 1)                          # 6
    r.mlist_move("ml", 2, 0)                            # 7
    r.mlist_set("ml", 1, "D")                           # 8
    r.commit()
    whole = r.changes[7][0]
    ops = whole.ops
    assert [o.counter for o in ops] == [0, 3, 4, 5, 6, 7, 8]
    c1 = wire.Change(7, 0, 0, [], [o for o in ops if o.counter < 5])
    c2 = wire.Change(7, 3, 3, [(7, 2)], [o for o in ops if o.counter >= 3])
    b1, b2 = wire.encode_updates([[c1]]), wire.encode_updates([[c2]])
    return [[r.export()], [b1, b2], [b2, b1], [b1, b2, r.export()]]


def test_sliced_changes_with_move_and_set_rows():
    got = _check(sliced_docs())
    assert got[0][0] == 0 and all(g == got[0] for g in got[1:]) and json.loads(got[0][1]) == {"ml": ["a", "D", "c"]}


def nesting_docs():
    """(docs, expected): MovableLists as children of a Map, a List and another MovableList; a child that never received an op;
    a pending change made of move rows."""
    r = wire.Replica(5)
    never = r.map_set_container("m", "never", ML)
    inl = r.list_insert_container("l", 0, ML)
    r.mlist_insert(inl, 0, [1, 2, 3]); r.mlist_move(inl, 2, 0)
    inner = r.mlist_insert_container(inl, 1, ML)             # a MovableList inside a MovableList
    r.mlist_insert(inner, 0, ["x", "y"]); r.mlist_move(inner, 0, 1); r.mlist_set(inner, 0, "Y")
    setc = r.mlist_set_container(inl, 3, wire.KIND_TEXT)     # element 2 (value 2) becomes a Text container
    r.text_insert(setc, 0, "hi")
    r.commit()
    _, _, b2 = sliced_parts()
    docs = [[r.export()], [b2]]
    want = [(0, b'{"l":[[3,["Y","x"],1,"hi"]],"m":{"never":[]}}'), (0, b"{}")]
    return docs, want


def sliced_parts():
    r = wire.Replica(7)
    r.mlist_insert("ml", 0, ["a", "b", "c"]); r.mlist_move("ml", 0, 2); r.mlist_set("ml", 0, "B")
    r.mlist_insert("ml", 1, ["d"]); r.mlist_delete("ml", 0, 1); r.mlist_move("ml", 2, 0); r.mlist_set("ml", 1, "D")
    r.commit()
    ops = r.changes[7][0].ops
    c1 = wire.Change(7, 0, 0, [], [o for o in ops if o.counter < 5])
    c2 = wire.Change(7, 3, 3, [(7, 2)], [o for o in ops if o.counter >= 3])
    return r, wire.encode_updates([[c1]]), wire.encode_updates([[c2]])


def test_nesting_and_pending():
    docs, want = nesting_docs()
    got = _check(docs)
    assert [g[:2] for g in got] == want
    assert got[1][3] == 6        # the six atoms of the change whose dependency is missing


def damaged_docs():
    """Valid envelopes around impossible MovableList rows: every one is LM_DATA_CORRUPTION on both sides."""
    out = []
    def rep():
        r = wire.Replica(21)
        r.mlist_insert("ml", 0, ["a", "b", "c"])
        return r
    r = rep(); r._push(wire.Op(wire.root_cid("ml", ML), r._alloc(1), "list_move", pos=1, move_from=7, elem=(21, 0))); r.commit(); out.append([r.export()])      # source beyond the end
    r = rep(); r._push(wire.Op(wire.root_cid("ml", ML), r._alloc(1), "list_move", pos=9, move_from=0, elem=(21, 0))); r.commit(); out.append([r.export()])      # destination beyond the end
    r = rep(); r._push(wire.Op(wire.root_cid("ml", ML), r._alloc(1), "list_move", pos=1, move_from=0, elem=(21, 77))); r.commit(); out.append([r.export()])     # unknown element
    r = rep(); r._push(wire.Op(wire.root_cid("ml", ML), r._alloc(1), "list_set", elem=(5, 0), value=1)); r.commit(); out.append([r.export()])                     # unknown element (foreign peer)
    r = rep(); r.text_insert("t", 0, "xy"); r._push(wire.Op(wire.root_cid("ml", ML), r._alloc(1), "list_set", elem=(21, 3), value=1)); r.commit(); out.append([r.export()])   # a text element is no list element
    return out


def test_damaged_rows():
    got = _check(damaged_docs())
    assert [g[0] for g in got] == [3] * 5
