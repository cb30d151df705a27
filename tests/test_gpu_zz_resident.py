"""GPU parity of resident documents (lm_import) through the C ABI: a document's history delivered in steps,
every step rendered, the trackers kept in HBM between the steps — against the oracle's Session (and, where the steps add up to
one batch, against the oracle's batch result)."""
import random
import pytest

import _oracle, _resident, _fuzz
import test_emu_resident as T
from loro_amd import workload, wire

pytestmark = pytest.mark.gpu


def _engine():
    import loro_amd
    return loro_amd.MergeEngine(0)


def _check(sessions, expect_incremental=True):
    want = _resident.oracle_sessions(sessions)
    fresh = []
    with _engine() as c:
        run0 = c.run

        def run_counting():
            run0()
            fresh.append(c.resident_fresh())
        c.run = run_counting
        got = _resident.run_sessions(c, sessions)
    for k, (g, w) in enumerate(zip(got, want)):
        for i, (a, b) in enumerate(zip(g, w)):
            assert a == b, f"step {k}, document {i}: device {str(a)[:200]} != oracle {str(b)[:200]}"
    if expect_incremental:
        assert fresh[0] == len(sessions) and sum(fresh[1:]) < len(sessions) * (len(fresh) - 1), fresh
    return got, fresh


def test_sequential_imports_and_checkouts_known_answers():
    a = wire.Replica(1)
    a.text_insert("text", 0, "ab"); a.list_insert("list", 0, [1, 2]); a.map_set("map", "k", 1); a.commit()
    v1 = list(a.frontiers)
    first = a.export()
    a.text_delete("text", 0, 2); a.list_delete("list", 0, 2); a.map_delete("map", "k"); a.commit()
    second = a.export({1: a.changes[1][0].ctr_end})
    sess = [[([first], None), ([second], None), ([], wire.encode_frontiers(v1)), ([], None), ([], wire.encode_frontiers([]))]]
    got, _ = _check(sess, expect_incremental=False)
    assert [g[0][1] for g in got] == [b'{"list":[1,2],"map":{"k":1},"text":"ab"}', b'{"list":[],"map":{},"text":""}',
                                      b'{"list":[1,2],"map":{"k":1},"text":"ab"}', b'{"list":[],"map":{},"text":""}', b'{"list":[],"map":{},"text":""}']


@pytest.mark.parametrize("mode,n", [("flat", 192), ("text", 96), ("nested", 96), ("movable", 96)])
def test_random_sessions_delivered_in_steps(mode, n):
    # enough documents for both streams of the context (128 per stream)
    base = 2000 + {"flat": 0, "text": 1000, "nested": 2000, "movable": 3000}[mode]
    sessions = T._sessions(mode, range(base, base + n))
    sessions = sessions + sessions + sessions if n < 128 else sessions + sessions   # (the same histories again: ≥ 256 documents)
    _check(sessions, expect_incremental=mode != "movable")


def test_failed_steps_renumbering_overflows():
    T.test_a_failed_step_leaves_the_document_as_it_was.__globals__["_check"] = _check
    try:
        T.test_a_failed_step_leaves_the_document_as_it_was()
        T.test_a_new_peer_that_sorts_in_front_renumbers_the_stored_leaves()
        T.test_directory_overflow_and_output_overflow_in_a_resident_run()
    finally:
        T.test_a_failed_step_leaves_the_document_as_it_was.__globals__["_check"] = T._check


def test_config2_full_size_base_resident_then_the_concurrent_branch():
    """BASELINE configs[1] at its stated document size: base + A's branch (75k ops) resident, B's 25k-op concurrent branch
    imported — 64 distinct documents (512 in the batch), bit-exact with the oracle's batch of all three blobs, and the import
    continued from the resident trackers for every document."""
    tpl = workload.Cfg2Template(50000, 25000, seed=0, commit_every=10, fuse=True)
    distinct = [tpl.stamp(d) for d in range(64)]
    docs = [distinct[i % 64] for i in range(512)]
    want = _oracle.merge_batch(distinct, threads=16)
    assert all(w[0] == 0 for w in want)
    with _engine() as c:
        c.stage([b[:2] for b in docs])
        c.import_more([[] for _ in docs])
        c.run()
        first = c.fetch()
        assert c.resident_fresh() == len(docs)
        base_want = _oracle.merge_batch([b[:2] for b in distinct], threads=16)
        assert all(first[i] == base_want[i % 64] for i in range(len(docs)))
        c.import_more([b[2:] for b in docs])
        c.run()
        got = c.fetch()
        assert c.resident_fresh() == 0
        assert all(got[i] == want[i % 64] for i in range(len(docs)))
        # the same version again: nothing is decoded, the trackers stay where they are
        c.import_more([[] for _ in docs])
        c.run()
        assert c.fetch() == got and c.resident_fresh() == 0


def test_config5_checkouts_as_one_replay_and_sixteen_moves():
    """BASELINE configs[4]: 1M-op rich-text documents rendered at 16 versions each — one replay, then the resident tracker moves
    from version to version (tracker.rs:354-546) — 4 distinct documents (in a batch of 256), every rendering bit-exact with the
    oracle's replay of that version."""
    import multiprocessing
    from test_gpu_parity import _gen_cfg5
    with multiprocessing.get_context("fork").Pool(4) as pool:
        gens = pool.map(_gen_cfg5, [(d, 1000000) for d in range(4)])
    n = 256
    docs = [gens[i % 4][0] for i in range(n)]
    flat_docs, flat_fr = [], []
    for d in range(4):
        flat_docs += [gens[d][0]] * 17
        flat_fr += gens[d][1] + [None]
    res = _oracle.merge_batch(flat_docs, threads=16, frontiers=flat_fr)
    assert all(r[0] == 0 for r in res)
    with _engine() as c:
        c.stage(docs)
        c.import_more([[] for _ in docs])
        c.run()
        got = c.fetch()
        assert all(got[i] == res[(i % 4) * 17 + 16] for i in range(n))
        for k in range(16):
            c.import_more([[] for _ in docs], [gens[i % 4][1][k] for i in range(n)])
            c.run()
            got = c.fetch()
            assert c.resident_fresh() == 0
            for i in range(n):
                assert got[i] == res[(i % 4) * 17 + k], (k, i)


def test_import_mode_and_common_ancestors_on_the_device():
    """k_import_lca (dag.rs:487-765 on the device) against the oracle's find_common_ancestor, step by step"""
    modes = set()
    for mode in ("flat", "text", "movable"):
        modes |= T._import_info_matches(_engine, T._sessions(mode, range(6000, 6064)))
    assert {"Linear", "Import", "ImportGreaterUpdates"} <= modes


def test_stage_run_import_run_and_a_rejected_import():
    """the documented call order lm_stage + lm_run, lm_import + lm_run (= import_batch, then import) and a rejected lm_import that
    must leave nothing behind — the device twins of tests/test_emu_resident.py's two tests"""
    a = wire.Replica(1)
    a.text_insert("text", 0, "ab"); a.list_insert("list", 0, [1, 2]); a.map_set("map", "k", 1); a.commit()
    first = a.export()
    a.text_delete("text", 0, 2); a.list_delete("list", 0, 2); a.map_delete("map", "k"); a.commit()
    second = a.export({1: a.changes[1][0].ctr_end})
    sess = _oracle.Session()
    want = [sess.step([first], None), sess.step([second], None)]
    sess.close()
    with _engine() as c:
        c.stage([[first]] * 300)
        c.run()
        assert all(g == want[0] for g in c.fetch())
        with pytest.raises(RuntimeError):
            c.import_more([[second]] * 300, [None] * 299 + [b""])
        c.import_more([[second]] * 150 + [[]] * 150)
        c.run()
        got = c.fetch()
        assert all(g == want[1] for g in got[:150]) and all(g == want[0] for g in got[150:])
        assert got[0][1] == b'{"list":[],"map":{},"text":""}'
        c.import_more([[]] * 150 + [[second]] * 150)
        c.run()
        assert all(g == want[1] for g in c.fetch())
