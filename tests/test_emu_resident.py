"""Resident documents (lm_import): the kernels' logic through the host harness (tests/emu) against the oracle's
Session — a document's history delivered in steps, every step rendered (latest or a checkout), trackers kept between the steps.
Reference: LoroDoc::import on a document that holds history (loro.rs:568-649,720-851), LoroDoc::checkout (loro.rs:1625-1760),
DiffCalculatorRetainMode::Persist (diff_calc.rs:62-68,1371-1376), Tracker::checkout / forward (tracker.rs:350-546)."""
import os
import random

import pytest

import _emu, _fuzz, _oracle, _resident
from loro_amd._cabi import Context
from loro_amd import wire, workload


def _check(sessions, binding=None, expect_incremental=True):
    want = _resident.oracle_sessions(sessions)
    fresh = []
    with Context(binding or _emu.binding()) as c:
        run0 = c.run

        def run_counting():
            run0()
            fresh.append(c.resident_fresh())
        c.run = run_counting
        got = _resident.run_sessions(c, sessions)
    for k, (g, w) in enumerate(zip(got, want)):
        for i, (a, b) in enumerate(zip(g, w)):
            assert a == b, f"step {k}, document {i}: harness {str(a)[:200]} != oracle {str(b)[:200]}"
    if expect_incremental:   # the first run replays everything; later runs must mostly continue from the resident trackers
        assert fresh[0] == len(sessions) and sum(fresh[1:]) < len(sessions) * (len(fresh) - 1), fresh
    return got, fresh


def _sessions(mode, seeds, n_steps=6):
    out = []
    for seed in seeds:
        rng = random.Random(seed * 7 + 1)
        snaps = []
        if mode == "flat":
            reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 5), n_steps=rng.randint(30, 160), kinds=("text", "list", "map"), snapshots=snaps, styles=True)
        elif mode == "text":
            reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 4), n_steps=rng.randint(100, 300), kinds=("text",), snapshots=snaps, styles=seed % 2 == 0, max_ins=12)
        elif mode == "nested":
            reps = _fuzz.nested_session(seed, n_peers=3, n_steps=120)
        else:
            reps = _fuzz.movable_session(seed, n_peers=3, n_steps=90, nested=seed % 2 == 0, snapshots=snaps)
        out.append(_resident.plan_steps(_resident.chunked_blobs(reps, rng), rng, n_steps, versions=[v for v, _ in snaps]))
    return out


def test_two_imports_are_not_one_import_batch():
    """insert, then delete everything, in two blobs: one import_batch (lm_stage of both) never sees the text — the root is
    absent; two LoroDoc::import calls create the state at the first one (diff_calc.rs:299) and it stays: "text":"" """
    a = wire.Replica(1)
    a.text_insert("text", 0, "ab"); a.list_insert("list", 0, [1, 2]); a.map_set("map", "k", 1); a.commit()
    v1 = list(a.frontiers)
    first = a.export()
    a.text_delete("text", 0, 2); a.list_delete("list", 0, 2); a.map_delete("map", "k"); a.commit()
    second = a.export({1: a.changes[1][0].ctr_end})
    assert _emu.merge_batch([[first, second]])[0][1] == b'{"map":{}}' == _oracle.merge([first, second])[1]
    sess = [[([first], None), ([second], None), ([], wire.encode_frontiers(v1)), ([], None), ([], wire.encode_frontiers([]))]]
    got, _ = _check(sess, expect_incremental=False)
    assert [g[0][1] for g in got] == [b'{"list":[1,2],"map":{"k":1},"text":"ab"}', b'{"list":[],"map":{},"text":""}',
                                      b'{"list":[1,2],"map":{"k":1},"text":"ab"}', b'{"list":[],"map":{},"text":""}', b'{"list":[],"map":{},"text":""}']


@pytest.mark.parametrize("mode", ["flat", "text", "nested", "movable"])
def test_random_sessions_delivered_in_steps(mode):
    # (every peer owns a region of the document's element slice with room to grow, k_res_layout: loc[], payload slots and the
    # slots MovableList moves keep the id of the item they deleted in stay where they are — all kinds continue incrementally)
    _check(_sessions(mode, range(100, 116)))


def test_several_streams_keep_their_documents():
    os.environ["LM_PART_MIN_DOCS"] = "4"
    try:
        _check(_sessions("flat", range(400, 416)))
    finally:
        del os.environ["LM_PART_MIN_DOCS"]


def test_lm_import_unfolds_a_batch_that_was_staged_folded():
    """entries of the staged batch that name the same blobs and ask for checkouts are uploaded ONCE (shared replay, lm_capi_impl.h);
    lm_import on such a batch makes every entry a resident document of its own over those bytes (Engine::expand) — with one stream
    and with the entries of a document spread over two — and the sessions go on like any others; also after an lm_run in between"""
    from loro_amd._cabi import Context
    base = _sessions("flat", range(700, 712))
    v = wire.encode_frontiers([])
    sessions = []
    for s in base:   # every history three times, the first step rendered at a checkout by two of them
        sessions += [[(s[0][0], v)] + s[1:], [(s[0][0], None)] + s[1:], [(s[0][0], s[-1][1] or v)] + s[1:]]
    want = _resident.oracle_sessions(sessions)
    for env, run_first in (({}, False), ({"LM_PART_MIN_DOCS": "4"}, False), ({"LM_PART_MIN_DOCS": "4"}, True)):
        os.environ.update(env)
        try:
            with Context(_emu.binding()) as c:
                docs = [s[0][0] for s in sessions]; fr = [s[0][1] for s in sessions]
                c.stage(docs, fr)
                assert 0 < c.b.shared_documents(c.h) < len(docs)
                if run_first:
                    c.run()
                    first = c.fetch()
                    assert [f[:3] for f in first] == [w[:3] for w in _oracle.merge_batch(docs, frontiers=fr)]
                got = []
                for k in range(len(sessions[0])):
                    c.import_more([[] if k == 0 else s[k][0] for s in sessions], [s[k][1] for s in sessions])
                    assert c.b.shared_documents(c.h) == 0
                    c.run()
                    got.append(c.fetch())
            for k, (g, w) in enumerate(zip(got, want)):
                assert [x[:3] for x in g] == [x[:3] for x in w], (env, run_first, k)
        finally:
            for k in env:
                del os.environ[k]


@pytest.mark.parametrize("defines,mode", [(["LM_SWEEP_EAGER", "LM_EMU_CHECK"], "text"), (["LM_SWEEP_EAGER", "LM_EMU_CHECK"], "movable"),
                                           (["LM_LOC_FULL", "LM_EMU_CHECK"], "flat")])
def test_structural_checker_builds(defines, mode):
    """every retreat / forward of three or more ids through the leaf sweep, the directory / leaf / loc[] checker after every op
    and after the closing checkout; and the per-element loc[] layout"""
    _check(_sessions(mode, range(300, 308)), binding=_emu.variant(defines), expect_incremental=mode != "movable")


def test_checkouts_move_the_tracker_back_and_forth():
    """one history, many versions, no new blobs between them: the tables are reused, the trackers move (tracker.rs:354-546)"""
    docs, fronts = [], []
    for d in range(3):
        blobs, fr = workload.cfg5_doc(d, n_ops=3000, turn=300, n_checkouts=12, commit_every=7)
        docs.append(blobs); fronts.append(fr)
    sessions = [[(docs[d], None)] + [([], fronts[d][k]) for k in range(12)] + [([], None)] for d in range(3)]
    got, fresh = _check(sessions, expect_incremental=False)
    assert fresh[0] == 3 and sum(fresh[1:]) == 0, fresh
    for d in range(3):   # and every version equals what a batch renders for it
        want = _oracle.merge_batch([docs[d]] * 12, frontiers=fronts[d])
        assert [got[1 + k][d] for k in range(12)] == want


def test_base_resident_then_the_concurrent_branch():
    """configs[1] in small: base + A's branch resident, B's concurrent branch imported — equal to the batch of all three"""
    tpl = workload.Cfg2Template(3000, 1500, seed=3, commit_every=10, fuse=True)
    docs = [tpl.stamp(d) for d in range(6)]
    sessions = [[(b[:2], None), (b[2:], None)] for b in docs]
    got, fresh = _check(sessions)
    assert fresh == [6, 0]
    assert got[1] == _oracle.merge_batch(docs)
    # and in the other order (B resident, base + A arriving: B's changes wait as pending changes until their base is there)
    sessions = [[(b[2:], None), (b[:2], None)] for b in docs]
    got, _ = _check(sessions, expect_incremental=False)
    assert got[1] == _oracle.merge_batch(docs) and all(g[3] > 0 and g[1] == b"{}" for g in got[0])


def test_a_failed_step_leaves_the_document_as_it_was():
    a = wire.Replica(7)
    a.text_insert("text", 0, "hello"); a.commit()
    b1 = a.export()
    a.text_insert("text", 5, " world"); a.commit()
    b2 = a.export({7: a.changes[7][0].ctr_end})
    bad = bytearray(b2); bad[-1] ^= 1
    a.text_delete("text", 0, 1); a.commit()
    b3 = a.export({7: a.changes[7][1].ctr_end})
    other = wire.Replica(9); other.map_set("map", "k", 1); other.commit()
    sess = [[([b1], None), ([bytes(bad)], None), ([b2], None), ([b3], wire.encode_frontiers([(99, 0)])), ([], None)],
            [([other.export()], None), ([], None), ([], None), ([], None), ([], None)]]
    got, _ = _check(sess, expect_incremental=False)
    assert [g[0][0] for g in got] == [0, 2, 0, 6, 0]
    assert got[2][0][1] == b'{"text":"hello world"}' and got[4][0][1] == b'{"text":"ello world"}'   # the refused checkout did not undo its import
    assert all(g[1][1] == b'{"map":{"k":1}}' for g in got)


def test_a_new_peer_that_sorts_in_front_renumbers_the_stored_leaves():
    rng = random.Random(1)
    sessions = []
    for seed in range(8):
        reps = _fuzz.random_session(500 + seed, n_peers=3, n_steps=90, kinds=("text", "list"), peer_base=1000)
        by_peer = sorted(reps, key=lambda r: r.peer)
        # the peer with the SMALLEST id is delivered last: its index 0 shifts every stored id
        order = [by_peer[2], by_peer[1], by_peer[0]]
        steps = []
        for r in order:
            own = r.changes.get(r.peer, [])
            steps.append(([wire.encode_updates(wire.split_blocks(own))] if own else [], None))
        sessions.append(steps)
    _, fresh = _check(sessions, expect_incremental=False)
    assert fresh == [8, 0, 0], fresh   # steps 2 and 3 continued from the stored trackers, renumbered


def test_directory_overflow_and_output_overflow_in_a_resident_run():
    os.environ["LM_DIR_OPT_MAX"] = "4"     # the optimistic LDS directory overflows: retry launch, replay from the empty version
    os.environ["LM_SLAB_CAP"] = "16"       # the optimistic output slab overflows: re-emit pass
    try:
        _check(_sessions("text", range(700, 706)), expect_incremental=False)
    finally:
        del os.environ["LM_DIR_OPT_MAX"], os.environ["LM_SLAB_CAP"]


def _import_info_matches(make_ctx, sessions):
    """LCA + DiffMode of every step's import, computed on the device (k_import_lca), against the oracle's find_common_ancestor"""
    oss = [_oracle.Session() for _ in sessions]
    modes = set()
    with make_ctx() as c:
        for k in range(len(sessions[0])):
            docs = [s[k][0] for s in sessions]
            fr = [s[k][1] for s in sessions]
            if k == 0:
                c.stage(docs, fr); c.import_more([[] for _ in docs], fr)
            else:
                c.import_more(docs, fr)
            c.run()
            got, info = c.fetch(), c.import_info()
            for i, o in enumerate(oss):
                w = o.step(docs[i], fr[i])
                assert w == got[i], (k, i)
                if w[0] in (0, 4, 6):       # the import went through (6: only the checkout behind it was refused)
                    assert info[i] == o.import_info(), (k, i, info[i], o.import_info())
                    modes.add(info[i][0])
    return modes


def test_import_mode_and_common_ancestors_on_the_device():
    """SURVEY §8 a9 on the device: dag.rs:487-765 (_find_common_ancestor_new) + the DiffMode it implies (oplog.rs:591-615)"""
    modes = set()
    for mode in ("flat", "text", "movable"):
        modes |= _import_info_matches(lambda: Context(_emu.binding()), _sessions(mode, range(2000, 2016)))
    assert {"Linear", "Import", "ImportGreaterUpdates"} <= modes
    # two peers fork after A's first change; B's branch is imported into base + A's branch.  The mode is Import (a Checkout promoted
    # because the target is greater, oplog.rs:610-615); the walk reaches the root on B's side without a match, so the reference
    # falls back to the empty version as the replay base (dag.rs:727-747) — the oracle, pinned to dag.rs:1108-1340, says the same
    a = wire.Replica(10); a.text_insert("text", 0, "base"); a.commit()
    b = wire.Replica(20); b.merge_from(a); b.seq = {k: list(v) for k, v in a.seq.items()}
    a.text_insert("text", 4, " A"); a.commit()
    b.text_insert("text", 0, "B "); b.commit()
    own_b = wire.Replica(20); own_b.changes = {20: b.changes[20]}
    with Context(_emu.binding()) as c:
        c.stage([[a.export()]]); c.import_more([[]]); c.run()
        assert c.import_info() == [("Linear", wire.encode_frontiers([]))]            # the empty document grew along one chain
        c.import_more([[own_b.export()]]); c.run()
        assert c.fetch()[0][1] == b'{"text":"B base A"}'
        o = _oracle.Session(); o.step([a.export()]); o.step([own_b.export()])
        assert c.import_info() == [o.import_info()] == [("Import", wire.encode_frontiers([]))]
        c.import_more([[]]); c.run()
        assert c.import_info()[0][0] == "Linear"                                   # nothing imported: diff_calc.rs:150-152


def test_stage_run_import_run_is_import_batch_then_import():
    """The call order include/loro_merge.h documents: lm_stage([[a]]) + lm_run, then lm_import([[b]]) + lm_run = import_batch([a])
    followed by import(b) — two diffs for the state store (the batch run records no tracker and no `exists` words, so the
    staged blobs are run once more as the resident documents' first step when lm_import arrives)."""
    a = wire.Replica(1)
    a.text_insert("text", 0, "ab"); a.list_insert("list", 0, [1, 2]); a.map_set("map", "k", 1); a.commit()
    first = a.export()
    a.text_delete("text", 0, 2); a.list_delete("list", 0, 2); a.map_delete("map", "k"); a.commit()
    second = a.export({1: a.changes[1][0].ctr_end})
    b = wire.Replica(2)
    b.text_insert("text", 0, "xyz"); b.commit()
    other = b.export()
    sess = _oracle.Session()
    want = [sess.step([first], None), sess.step([second], None), sess.step([other], None)]
    sess.close()
    with Context(_emu.binding()) as c:
        c.stage([[first], [first]])
        c.run()
        assert c.fetch()[0] == want[0]
        c.import_more([[second], [second]])
        c.run()
        got1 = c.fetch()
        c.import_more([[other], []])
        c.run()
        got2 = c.fetch()
    assert got1[0] == want[1] and got1[1] == want[1]
    assert got1[0][1] == b'{"list":[],"map":{},"text":""}'
    assert got2[0] == want[2] and got2[1] == want[1]


def test_rejected_import_leaves_the_documents_as_they_were():
    """lm_import validates everything before it records anything: after a refused call (zero-length checkout frontiers for the LAST
    document) the blob lists, checkouts and arena offsets are those of the last accepted call (the refused call's
    BlobRefs stayed behind and the next import's bytes landed under them)."""
    docs = []
    for seed in (3, 4, 5):
        r = wire.Replica(10 + seed)
        r.text_insert("text", 0, "doc%d" % seed); r.commit()
        docs.append(r)
    firsts = [[r.export()] for r in docs]
    for r in docs:
        r.text_insert("text", 0, "more "); r.commit()
    seconds = [[r.export({r.peer: r.changes[r.peer][0].ctr_end})] for r in docs]
    want = []
    for f, s in zip(firsts, seconds):
        o = _oracle.Session(); o.step(f, None); want.append(o.step(s, None)); o.close()
    with Context(_emu.binding()) as c:
        c.stage(firsts)
        c.run()
        with pytest.raises(RuntimeError):
            c.import_more(seconds, [None, None, b""])
        c.import_more([[], seconds[1], []])
        c.run()
        mid = c.fetch()
        assert mid[1] == want[1] and mid[0][3] == 0 and mid[2][3] == 0 and all(m[0] == 0 for m in mid)
        c.import_more([seconds[0], [], seconds[2]])
        c.run()
        assert c.fetch() == want
