"""MovableList — the CPU oracle pinned by the reference's own known answers: the Rust-written
`updates.blob` / TS-written fixtures (snapshot.deep.json, runtime.expected.json) and the edit scripts of
crates/loro/tests/mov.rs and state/movable_list_state.rs:1936-2032 replayed through the FastUpdates writer."""
import itertools, json, os, random
import pytest

import _oracle, _fuzz
from loro_amd import wire

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "reference_fixtures.json")))
BLOB = {k: bytes.fromhex(v) for k, v in FX["blobs"].items()}
ML = wire.KIND_MOVABLE


def deep(blobs):
    st, js, _, pend = _oracle.merge(blobs)
    assert pend == 0
    return st, json.loads(js)


def test_rust_and_ts_written_fixtures():
    # crates/loro/tests/loro_js_interop.rs:42-73: the fixture's moves / sets / deletes over a root and a child MovableList
    want = FX["json"]["snapshot.deep.json"]
    for name in ("updates.blob", "updates.ts.blob"):
        st, v = deep([BLOB[name]])
        assert st == 4   # a Tree container rides along
        assert v["mlist"] == want["mlist"] == []
        assert v["map"]["child_mlist"] == want["map"]["child_mlist"] == ["", 166163940, 1551498871, 20, 10]
    st, v = deep([BLOB["runtime-updates.ts.blob"]])
    assert v["movable"] == FX["json"]["runtime.expected.json"]["movable"] == ["z", "x"]


def _all_orders(*reps):
    out = set()
    for perm in itertools.permutations([r.export() for r in reps]):
        st, js, _, pend = _oracle.merge(list(perm))
        assert st == 0 and pend == 0
        out.add(js)
    assert len(out) == 1
    return json.loads(out.pop())


def _sync(a, b):
    a.commit(); b.commit()
    for x, y in ((a, b), (b, a)):
        if x.merge_from(y):
            blob = x.export()
            for cid in {o.cid for chs in x.changes.values() for c in chs for o in c.ops}:
                if cid.kind in (wire.KIND_TEXT, wire.KIND_LIST, ML):
                    x.set_visible(cid, cid.kind, _oracle.visible_ids([blob], cid, cid.kind))


def test_conflict_moves():
    # crates/loro/tests/mov.rs:13-62
    d1 = wire.Replica(1)
    for i, v in enumerate((1, 2, 3)):
        d1.mlist_insert("list", i, [v])
    d1.commit()
    assert _all_orders(d1) == {"list": [1, 2, 3]}
    d2 = wire.Replica(2)
    _sync(d2, d1)
    d1.mlist_move("list", 0, 2)
    d2.mlist_move("list", 0, 1)
    d1.commit(); d2.commit()
    assert _all_orders(d1, d2) == {"list": [2, 1, 3]}
    _sync(d1, d2)
    assert [e for _, _, e in d1.mlist_elements("list")] == [e for _, _, e in d2.mlist_elements("list")]
    assert len(d1.seq[wire.root_cid("list", ML)]) == 4   # the losing move's item stays in the list, pointed at by nothing


def test_basic_handler_ops_and_sync():
    # state/movable_list_state.rs:1940-1958
    d = wire.Replica(7)
    val = lambda: _all_orders(d)["list"] if d.commit() is None else None
    for i in range(3):
        d.mlist_insert("list", i, [i])
    assert val() == [0, 1, 2]
    d.mlist_move("list", 0, 1); assert val() == [1, 0, 2]
    d.mlist_move("list", 2, 0); assert val() == [2, 1, 0]
    d.mlist_delete("list", 0, 2); assert val() == [0]
    d.mlist_insert("list", 0, [9]); assert val() == [9, 0]
    d.mlist_delete("list", 0, 2); assert val() == []
    # :1960-2031
    d = wire.Replica(8)
    d.mlist_insert("list", 0, [1]); d.mlist_insert("list", 1, [0]); d.mlist_move("list", 0, 1)
    assert val() == [0, 1]
    d.mlist_move("list", 1, 0); assert val() == [1, 0]
    d.mlist_move("list", 0, 1); d.mlist_insert("list", 2, [3]); d.mlist_set("list", 2, 2)
    assert val() == [0, 1, 2]


def test_move_set_delete_races():
    # a move and a delete of the same element race: the moved item survives (the delete hit the old item);
    # concurrent sets: the greater (lamport, peer) wins; a set on a concurrently deleted element is invisible
    a, b, c = wire.Replica(11), wire.Replica(12), wire.Replica(13)
    a.mlist_insert("l", 0, ["x", "y", "z"])
    _sync(b, a); _sync(c, a)
    a.mlist_move("l", 0, 2)
    b.mlist_delete("l", 0, 1)
    c.mlist_set("l", 1, "Y")
    b.mlist_set("l", 0, "yb")      # b's view after its delete: [y, z]
    a.commit(); b.commit(); c.commit()
    v = _all_orders(a, b, c)["l"]
    assert v[1:] == ["z", "x"] and v[0] in ("Y", "yb")
    lam_b = b.changes[12][-1].lamport + 1          # the set is b's second op in that change
    lam_c = c.changes[13][-1].lamport
    assert v[0] == ("yb" if (lam_b, 12) > (lam_c, 13) else "Y")


movable_session = _fuzz.movable_session


@pytest.mark.parametrize("seed", range(12))
def test_random_sessions_converge_and_match_local_views(seed):
    reps = movable_session(seed)
    blobs = _fuzz.blobs_of(reps)
    outs = set()
    rng = random.Random(seed)
    for _ in range(4):
        rng.shuffle(blobs)
        st, js, _, pend = _oracle.merge(blobs)
        assert st == 0 and pend == 0
        outs.add(js)
    assert len(outs) == 1
    # every replica's own export renders what the replica's local view (writer-side simulation) shows
    for r in reps:
        st, js, _, _ = _oracle.merge([r.export()])
        assert st == 0
        got = json.loads(js).get("ml", [])
        assert len(got) == r.mlist_len("ml")
