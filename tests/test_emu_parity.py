"""Kernel logic on CPU: the HIP kernels compiled for the host (tests/emu, every lane a fiber) must agree
bit-for-bit with the oracle.  This exercises the same kernel source and host pipeline the GPU runs; the
GPU parity tests proper are in test_gpu_parity.py (-m gpu)."""
import pytest

import os
import _oracle, _emu, _cases
from loro_amd._cabi import Context
from loro_amd import wire, workload


DEVICE_SCOPE_GAPS = set()   # every edge-case document is rendered by the device path


def _check(docs, names=None):
    got = _emu.merge_batch(docs)
    want = _oracle.merge_batch(docs)
    for i, (g, w) in enumerate(zip(got, want)):
        label = names[i] if names else f"doc {i}"
        if label in DEVICE_SCOPE_GAPS:
            assert g[0] == 4
            continue
        if w[0] in (0,):
            assert g == w, f"{label}: emu={g[:3]!r} oracle={w[:3]!r}"
        else:
            assert g[0] == w[0] or (w[0] == 4 and g[0] == 4), f"{label}: status emu={g[0]} oracle={w[0]}"
    return got, want


def test_edge_cases():
    names, docs = _cases.edge_case_docs()
    got, want = _check(docs, names)
    by = dict(zip(names, got))
    assert by["no blobs"][:2] == (0, b"{}")
    assert by["checksum mismatch"][0] == 2 and by["bad magic"][0] == 1 and by["truncated"][0] == 1
    assert by["shallow snapshot"][0] == 4
    assert by["good next to bad docs"][:2] == (0, b'{"text":"ab"}')
    assert by["pending only"][1] == b"{}" and by["pending only"][3] == 2
    assert by["pending resolved later"][1] == b'{"text":"abcd"}'
    assert by["duplicate blob"][1] == b'{"text":"abcd"}'
    assert by["overlapping changes"][1] == by["overlapping changes reversed"][1] == b'{"text":"abcdefghijklmnopqrstuvwxyz0123"}'
    assert by["sliced forward delete"][1] == b'{"text":"016789"}'
    assert by["reversed delete"][1] == b'{"text":"016789"}'
    assert by["f64 value"][1] == b'{"map":{"f":1.5,"g":[0.1,-2.5e-7,1e21,3.0,null]}}'
    assert by["nested map value"][1] == (b'{"list":[{"p":[{"a":1,"b":2}],"q":1},"tail"],'
                                         b'"map":{"nested":{"":true,"a":{"k":"v","z":[1,{"x":2.25,"y":null}]},"b":1},"other":{"k1":1,"k2":2}}}')
    assert by["tree child container"][0] == 4   # device scope: reported, never guessed


def test_fuzz_sessions():
    _check(_cases.fuzz_docs(48))


@pytest.mark.parametrize("mode", ["0", "1"])
def test_other_integrate_instantiations_stay_correct(monkeypatch, mode):
    """Documents without sliced changes / style anchors / checkouts (DF_PLAIN) are replayed by k_integrate_span_plain_sweep by
    default (measured fastest, profiles/r02_ab_prepared.log); LM_PLAIN=0 sends them to the common kernel, LM_PLAIN=1 to
    k_integrate_span_plain: same results."""
    monkeypatch.setenv("LM_PLAIN", mode)
    _check(_cases.fuzz_docs(16, base=4200) + _cases.fuzz_docs(4, base=4300, steps=120, peers=4, max_ins=30, sync_prob=0.08) + _cases.trace_docs(3000, n_docs=1))


@pytest.mark.parametrize("defines", [["LM_EMU_CHECK"], ["LM_LOC_FULL", "LM_EMU_CHECK"]], ids=["loc16+checker", "loc-full+checker"])
def test_loc_layouts_with_structural_checker(monkeypatch, defines):
    """loc[] kept for item heads and multiples of 16 only (default) and the per-element layout (-DLM_LOC_FULL), each with the
    structural checker compiled in (it verifies the kept / NONE pattern, the directory sums and the cached leaf after every
    op), under every integrate instantiation of the span kernel."""
    from loro_amd._cabi import Context
    b = _emu.variant(defines)
    docs = _cases.fuzz_docs(12, base=4400) + _cases.fuzz_docs(3, base=4500, steps=120, peers=4, max_ins=30, sync_prob=0.08) + _cases.trace_docs(3000, n_docs=1)
    want = _oracle.merge_batch(docs)
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("LM_PLAIN", mode)
        with Context(b) as c:
            assert c.merge_batch(docs) == want


def test_leaf_sweep_on_every_range(monkeypatch):
    """k_integrate_span_plain_sweep toggles a long retreat / forward range in one pass over the leaves (range longer than
    8 x leaves + 64 ids); the fuzz corpora's ranges are short, so this build (-DLM_SWEEP_EAGER) sweeps every range of three or
    more ids, with the structural checker compiled in."""
    from loro_amd._cabi import Context
    b = _emu.variant(["LM_SWEEP_EAGER", "LM_EMU_CHECK"])
    docs = (_cases.fuzz_docs(16, base=4600) + _cases.fuzz_docs(4, base=4700, steps=120, peers=4, max_ins=30, sync_prob=0.08)
            + _cases.fuzz_docs(2, base=7000, steps=300, peers=3, max_ins=40, sync_prob=0.03) + _cases.trace_docs(3000, n_docs=1))
    with Context(b) as c:
        assert c.merge_batch(docs) == _oracle.merge_batch(docs)


def test_batch_trackers_move_by_version_passes():
    """The common batch kernel (documents with style anchors / sliced changes) moves its trackers with ts_sweep_version — delete
    counts at the target version + one pass over the leaves — when the move is long enough; this build (-DLM_SWEEP_EAGER) sends
    EVERY move through the pass, with the structural checker after every op, on configs[3]-shaped documents (4 peers, pairwise
    syncs, marks), styled fuzz sessions, nested containers and checked-out versions."""
    import _fuzz
    from loro_amd._cabi import Context
    b = _emu.variant(["LM_SWEEP_EAGER", "LM_EMU_CHECK", "LM_BATCH_VSWEEP=1"])   # (off in the product build: measured slower on configs[3])
    docs = _cases.cfg4_docs(10, first=9100, n_steps=300)
    docs += [_fuzz.blobs_of(_fuzz.random_session(9300 + d, n_peers=3 + d % 3, n_steps=150, kinds=("text", "list"), sync_prob=0.05, styles=True)) for d in range(12)]
    docs += _nested_docs(4, n_peers=3, n_steps=120)
    fronts = [None] * len(docs)
    for d in range(3):
        blobs, fr = workload.cfg5_doc(40 + d, n_ops=1500, turn=150, n_checkouts=6, commit_every=7)
        docs += [blobs] * len(fr); fronts += fr
    with Context(b) as c:
        got = c.merge_batch(docs, fronts)
    assert got == _oracle.merge_batch(docs, frontiers=fronts)
    assert all(g[0] == 0 for g in got)


def test_envelope_checksum_of_large_blobs():
    """k_hash_big_blobs (four blobs per wave, words through the row's lane permutes, three rotating register banks) on blobs of many
    lengths, good and damaged checksums, next to small blobs hashed by k_frame_count's lanes"""
    docs = _cases.big_blob_checksum_docs() + _cases.fuzz_docs(6, base=77)
    got, want = _check(docs)
    assert sorted({g[0] for g in got}) == [0, 2]


def test_map_typed_values_are_rendered_in_key_order():
    """nested map values: the one-pass ordering of up to 64 entries (lane-parallel rank, LDS pool), its fallback, duplicate keys"""
    docs = _cases.nested_map_order_docs()
    got, want = _check(docs)
    assert got[0][0] == 0 and b'"k":3' in got[0][1] and b'"k":1' not in got[0][1]


def test_map_rendering_plain_groups_and_entry_by_entry():
    _check(_cases.map_render_docs())


def test_nested_map_key_index_beyond_the_key_table_is_rejected(monkeypatch):
    """The decoders flag rows whose value holds a list / map (OPF_NESTED) and k_remap walks those values again against the
    block's key table (lm_k_dag.h) — both decoders, every op shape that carries a nested value."""
    good, bad = _cases.nested_key_docs()
    want_good, want_bad = _oracle.merge_batch(good), _oracle.merge_batch(bad)
    assert all(w[0] == 0 for w in want_good) and all(w[0] != 0 for w in want_bad)
    for lane in ("1", "0"):
        monkeypatch.setenv("LM_DECODE", lane)
        got = _emu.merge_batch(good + bad)
        assert got[:len(good)] == want_good
        assert [g[0] for g in got[len(good):]] == [w[0] for w in want_bad]


def test_node_cut_replay_order_and_tracker_base_on_small_documents(monkeypatch):
    """LM_CUT_MIN_ROWS=0: every document — not only those of 2,048 op rows and more — gets its nodes cut at cross-peer dependency
    targets, replayed one node per pass with the largest ready peer first, and its trackers a base version at critical versions
    (conversion + one-pass reset; LM_SWEEP_EAGER: every move back to the base takes the pass), under the structural checker."""
    monkeypatch.setenv("LM_CUT_MIN_ROWS", "0")
    b = _emu.variant(["LM_SWEEP_EAGER", "LM_EMU_CHECK"])
    import _fuzz
    from loro_amd._cabi import Context
    docs = _cases.fuzz_docs(40, base=21000) + _cases.cfg4_docs(8, first=7300, n_steps=250) + _nested_docs(6, n_peers=4, n_steps=160)
    for seed in range(4):
        tpl = workload.Cfg2Template(1200 + 300 * seed, 600, seed=seed, commit_every=(1 if seed % 2 else 10), fuse=bool(seed % 2 == 0))
        docs += [tpl.stamp(seed), list(reversed(tpl.stamp(seed + 20)))]
    docs += [_fuzz.blobs_of(_fuzz.movable_session(8800 + d, n_peers=3, n_steps=120, nested=True)) for d in range(6)]
    with Context(b) as c:
        got = c.merge_batch(docs)
    want = _oracle.merge_batch(docs)
    assert got == want


def test_linear_prefix_of_a_batch_replay(monkeypatch):
    """The nodes in front of the first critical version that opens a concurrent section are replayed as a positional rope
    (lm_k_integrate_linear.h: no origins, no tombstones, no loc[]) and handed to the tracker — under the structural checker, for
    documents of every size (LM_CUT_MIN_ROWS=0: k_dag_b's flags), against the oracle's full replay; LM_LINEAR=0 (the tracker from
    the first node on) gives the same bytes."""
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_CUT_MIN_ROWS", "0")
    b = _emu.variant(["LM_SWEEP_EAGER", "LM_EMU_CHECK"])
    docs = _cases.linear_prefix_docs(60) + _cases.trace_docs(2000, n_docs=1)
    want = _oracle.merge_batch(docs)
    with Context(b) as c:
        got = c.merge_batch(docs)
    assert got == want
    monkeypatch.setenv("LM_LINEAR", "0")
    with Context(b) as c:
        assert c.merge_batch(docs[:24]) == want[:24]


def test_concurrent_sibling_scans():
    # many peers typing long runs at the same spots: stresses the run-head sibling scan
    _check(_cases.fuzz_docs(12, base=1000, steps=120, peers=4, max_ins=30, sync_prob=0.08))


def test_trace_shaped_documents_both_import_orders():
    docs = _cases.trace_docs(3000, n_docs=1)
    got, _ = _check(docs)
    for k in range(0, len(docs), 3):
        assert got[k][1] == got[k + 1][1] and got[k][2] == got[k + 1][2]


def test_optimistic_directory_overflow_is_retried(monkeypatch):
    """Sequential appends leave every leaf half full (the worst case the optimistic LDS directory does not cover):
    the document must be re-run with the worst-case directory and still match the oracle."""
    monkeypatch.setenv("LM_SPAN", "0")   # the element-granular kernel's directory sizing
    from loro_amd import wire
    from loro_amd._cabi import Context
    r = wire.Replica(77)
    n = 0
    for i in range(3500):
        r.text_insert("text", n, "abcdefg"[: 1 + i % 7]); n += 1 + i % 7
        if i % 3 == 0:
            r.text_delete("text", n - 1, 1); n -= 1     # keeps the runs from merging into one op
        if i % 100 == 0:
            r.commit()
    r.commit()
    docs = [[r.export()], _cases.fuzz_docs(1)[0]]
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        sizing = c.sizing()
    assert got == _oracle.merge_batch(docs)
    assert sizing[3] >= 1, f"expected a directory retry, sizing={sizing}"


def test_batch_split_over_several_streams(monkeypatch):
    """lm_ctx splits a batch into contiguous document ranges, one engine (HIP stream) each; results must come back in
    document order whatever the split."""
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_STREAMS", "3")
    monkeypatch.setenv("LM_PART_MIN_DOCS", "2")
    names, docs = _cases.edge_case_docs()
    docs = docs + _cases.fuzz_docs(10, base=4000)
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        assert c.b.n_streams(c.h) == 3
        got2 = c.merge_batch(docs[:3])          # shrinking batch re-uses the context
        assert c.b.n_streams(c.h) == 1
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g == w) if w[0] == 0 else (g[0] == w[0]), i
    assert got2 == got[:3]


def _checkout_cases():
    """(docs, frontiers): reference known answers (test.rs:518-603,659-693) + every recorded version of random
    concurrent sessions + malformed / unknown frontiers."""
    import _fuzz
    from loro_amd import wire
    docs, fronts = [], []
    r = wire.Replica(1)
    r.text_insert("text", 0, "你界"); r.text_insert("text", 1, "好世"); r.commit()
    r.text_delete("text", 3, 1); r.text_delete("text", 2, 1); r.commit()
    tb = [r.export()]
    for ids in [[(1, c)] for c in range(6)] + [[], [(1, 6)], [(2, 0)], [(1, -1)]]:
        docs.append(tb); fronts.append(wire.encode_frontiers(ids))
    docs.append(tb); fronts.append(b"\x05\x01")            # truncated Vec<ID>
    docs.append(tb); fronts.append(wire.encode_frontiers([(1, 2)]) + b"\x00")   # trailing byte
    docs.append(tb); fronts.append(None)
    a, b = wire.Replica(1), wire.Replica(2)
    a.map_set("meta", "key", 0); a.commit(); va = list(a.frontiers)
    b.map_set("meta", "s", 1); b.commit(); vb0 = list(b.frontiers)
    b.map_set("meta", "key", 1); b.commit(); vb1 = list(b.frontiers)
    a.merge_from(b)
    a.map_set("meta", "key", 2); a.commit(); vm = list(a.frontiers)
    mb = [a.export()]
    for v in (va, vb0, vb1, vm, va + vb1, []):
        docs.append(mb); fronts.append(wire.encode_frontiers(v))
    import test_oracle_golden as tog
    for blobs, ids in ((tog._two_peer_delete_doc(True), [(2, 4)]), (tog._two_peer_delete_doc(True), [(2, 0)]),
                       (tog._two_peer_delete_doc(False), [(2, 3)]), (tog._two_peer_delete_doc(False), [(1, 9)])):
        docs.append(blobs); fronts.append(wire.encode_frontiers(ids))     # tracker.rs:734-773 known answers
    for s in range(6):
        snaps = []
        reps = _fuzz.random_session(900 + s, n_peers=3, n_steps=60, kinds=("text", "list", "map"), styles=True, snapshots=snaps)
        full = _fuzz.blobs_of(reps)
        for fr, _ in snaps:
            docs.append(full); fronts.append(wire.encode_frontiers(fr))
            if len(fr) == 1 and fr[0][1] > 0:   # a version cutting through the middle of a change / op run
                docs.append(full); fronts.append(wire.encode_frontiers([(fr[0][0], fr[0][1] - 1)]))
    return docs, fronts


def test_checkout_versions():
    docs, fronts = _checkout_cases()
    want = _oracle.merge_batch(docs, frontiers=fronts)
    got = _emu.merge_batch(docs, fronts)
    assert [w[0] for w in want[:13]] == [0] * 7 + [6, 6, 6, 1, 1, 0]
    n_ok = 0
    for i, (g, w) in enumerate(zip(got, want)):
        if w[0] == 0:
            assert g == w, (i, g[:3], w[:3])
            n_ok += 1
        else:
            assert g[0] == w[0], (i, g[0], w[0])
    assert n_ok > 60


def _shared_replay_cases():
    """(docs, frontiers) whose entries share blob objects: every recorded version of random sessions (all container kinds, styles,
    MovableLists) against ONE list of blobs per session, the latest version among them, versions that differ in which root
    containers are visible (the state store's view is per rendering), and a document whose import fails."""
    import _fuzz
    docs, fronts = _checkout_cases()
    for s in range(5):
        snaps = []
        reps = _fuzz.random_session(4400 + s, n_peers=3, n_steps=90, kinds=("text", "list", "map"), styles=bool(s % 2), snapshots=snaps)
        full = _fuzz.blobs_of(reps)
        for fr, _ in snaps[:: max(1, len(snaps) // 12)]:
            docs.append(full); fronts.append(wire.encode_frontiers(fr))
        docs.append(full); fronts.append(None)
        docs.append(full); fronts.append(None)
    # text visible at an early version only / at a late version only: absent from the renderings where nothing shows
    r = wire.Replica(7)
    r.text_insert("early", 0, "abc"); r.commit(); v1 = list(r.frontiers)
    r.text_delete("early", 0, 3); r.commit(); v2 = list(r.frontiers)
    r.text_insert("late", 0, "xyz"); r.map_set("m", "k", 1); r.commit(); v3 = list(r.frontiers)
    r.text_delete("late", 0, 3); r.commit(); v4 = list(r.frontiers)
    tb = [r.export()]
    for v in (v1, v3, v2, v4, v1, v2, []):
        docs.append(tb); fronts.append(wire.encode_frontiers(v))
    docs.append(tb); fronts.append(None)
    bad = [tb[0][:-3] + b"\x00\x01\x02"]   # checksum mismatch: every entry of the document fails
    for v in (v1, v3):
        docs.append(bad); fronts.append(wire.encode_frontiers(v))
    return docs, fronts


def test_entries_that_share_their_blobs_are_replayed_once(monkeypatch):
    """lm_stage folds entries with the same blobs into one document (lm_shared_documents), lm_run imports it once and renders every
    entry by a move of the resident trackers: per entry the result of import_batch + checkout on a document of its own — the
    oracle's — and what LM_SHARE_REPLAY=0 (one replay per entry) gives."""
    from loro_amd._cabi import Context
    docs, fronts = _shared_replay_cases()
    want = _oracle.merge_batch(docs, frontiers=fronts)
    with Context(_emu.binding()) as c:
        c.stage(docs, fronts)
        n_shared = c.b.shared_documents(c.h)
        assert 0 < n_shared < len(docs) // 3
        c.run()
        got = c.fetch()
        c.run()                       # every lm_run is the whole job again
        assert c.fetch() == got
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")
    with Context(_emu.binding()) as c:
        c.stage(docs, fronts)
        assert c.b.shared_documents(c.h) == 0
        c.run()
        plain = c.fetch()
    for i, (g, p, w) in enumerate(zip(got, plain, want)):
        if w[0] == 0:
            assert g == w and p == w, (i, g[:2], p[:2], w[:2])
        else:
            assert g[0] == w[0] and p[0] == w[0], (i, g[0], p[0], w[0])


def test_snapshot_blobs_are_ingested_through_their_change_store():
    docs, check = _cases.snapshot_cases()
    check(_emu.merge_batch(docs))


def test_damaged_snapshots_never_take_the_batch_down():
    """Byte flips, truncations and splices in the Rust-written snapshot (envelope checksum re-fitted, so the SSTable / LZ4
    reader of lm_snapshot.h sees the damage): no crash, every document gets a status, the healthy neighbours are intact."""
    import json, os, random, struct
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))
    snap = bytes.fromhex(fx["blobs"]["snapshot.blob"])
    good = wire.Replica(99); good.map_set("root", "a", 1); good.commit()
    rng = random.Random(7)
    docs = []
    for i in range(300):
        b = bytearray(snap)
        k = rng.random()
        if k < 0.5:
            for _ in range(rng.randint(1, 4)):
                b[rng.randrange(22, len(b))] ^= 1 << rng.randrange(8)
        elif k < 0.75:
            del b[rng.randrange(22, len(b)):]
        else:
            a = rng.randrange(22, len(b)); c = rng.randrange(a, min(len(b), a + 64))
            b[a:c] = bytes(rng.randrange(256) for _ in range(rng.randint(0, 80)))
        if len(b) > 22:
            b[16:20] = struct.pack("<I", wire.xxh32(bytes(b[20:])))
        docs += [[bytes(b)], [good.export()]]
    got = _emu.merge_batch(docs)
    ref = _emu.merge_batch([[snap]])[0]
    n_same = 0
    for i in range(0, len(docs), 2):
        assert got[i][0] in (0, 1, 2, 3, 4), got[i][0]
        n_same += got[i][1:] == ref[1:]
        assert got[i + 1][:2] == (0, b'{"root":{"a":1}}')
    assert n_same < 150     # most mutations are noticed (a flip inside a value payload or an ignored section is not)


def test_documented_limits_are_reported_not_guessed():
    lim = _cases.limit_docs()
    good = wire.Replica(99); good.map_set("root", "a", 1); good.commit()
    docs = []
    for _, blobs in lim:
        docs += [blobs, [good.export()]]
    got = _emu.merge_batch(docs)
    for i, (name, _) in enumerate(lim):
        assert got[2 * i][0] == 4, (name, got[2 * i][:2])
        assert got[2 * i + 1][:2] == (0, b'{"root":{"a":1}}'), name


def test_root_containers_the_state_store_holds():
    cases = _cases.container_existence_cases()
    docs = [c[1] for c in cases]
    fronts = [c[2] for c in cases]
    want = _oracle.merge_batch(docs, frontiers=fronts)
    got = _emu.merge_batch(docs, fronts)
    for c, w, g in zip(cases, want, got):
        assert w[0] == 0 and w[1] == c[3], (c[0], w[1])
        assert g == w, (c[0], g[:2], w[:2])


def test_config5_alternating_peers_marks_and_checkouts():
    from loro_amd import workload
    docs, fronts = [], []
    for d in range(2):
        blobs, fr = workload.cfg5_doc(d, n_ops=3000, turn=400, n_checkouts=8)
        docs += [blobs] * len(fr)
        fronts += fr
    want = _oracle.merge_batch(docs, frontiers=fronts)
    assert all(w[0] == 0 for w in want) and len({w[1] for w in want}) > 8
    assert _emu.merge_batch(docs, fronts) == want


def test_sibling_in_the_first_slot_of_the_next_leaf():
    """Regression: a concurrent sibling (same origin_left) that sits in slot 0 of the leaf after the cursor's leaf was
    taken for a continuation of the previous leaf's last element and skipped by the run-head scan."""
    import _fuzz
    reps = _fuzz.random_session(1032, n_peers=4, n_steps=525, kinds=("text",), sync_prob=0.02, styles=True)
    _check([_fuzz.blobs_of(reps)])


def test_config4_mixed_containers_with_dag_merges():
    _check(_cases.cfg4_docs(10, first=1016))


def test_huge_paste_next_to_small_blocks():
    """A single 30 KB paste: one change block far above the usual 4 KiB and
    a string far above the 64-byte per-lane limit of k_elem_fill (cooperative routine); small blocks ride beside it."""
    from loro_amd import wire
    a, b = wire.Replica(7), wire.Replica(9)
    a.text_insert("text", 0, "start "); a.commit()
    b.merge_from(a); b.set_visible("text", wire.KIND_TEXT, _oracle.visible_ids([a.export()], "text", wire.KIND_TEXT))
    a.text_insert("text", 3, "".join(chr(0x4E00 + (i * 7) % 500) if i % 5 == 0 else "abcdefghij"[i % 10] for i in range(12000))); a.commit()
    b.text_insert("text", 6, "tail"); b.text_delete("text", 0, 2); b.commit()
    a.text_insert("text", 100, "xyz"); a.commit()
    _check([[a.export(), b.export()], [b.export(), a.export()]])


def _nested_docs(n, first=7000, **kw):
    import _fuzz
    return [_fuzz.blobs_of(_fuzz.nested_session(first + s, **kw)) for s in range(n)]


def test_child_containers_hand_built():
    """Child Map / List / Text containers (insert_container): nested rendering, a child that never gets an op (empty
    value of its kind), a child made unreachable by a later write to its slot, edits to a child from two peers."""
    from loro_amd import wire
    K = wire
    a = wire.Replica(3)
    m1 = a.map_set_container("root", "profile", K.KIND_MAP)
    a.map_set(m1, "name", "Ada"); a.map_set(m1, "age", 36)
    t1 = a.map_set_container(m1, "bio", K.KIND_TEXT)
    a.text_insert(t1, 0, "hello")
    l1 = a.map_set_container("root", "items", K.KIND_LIST)
    a.list_insert(l1, 0, [1, "two"])
    c1 = a.list_insert_container(l1, 1, K.KIND_MAP)
    a.map_set(c1, "k", [1, 2, 3])
    a.list_insert_container(l1, 3, K.KIND_TEXT)            # never receives an op → ""
    c3 = a.list_insert_container("rl", 0, K.KIND_LIST)
    a.list_insert(c3, 0, ["x"])
    a.map_set_container("root", "empty", K.KIND_LIST)      # no ops → []
    a.commit()
    first = a.export()
    b = wire.Replica(5)
    b.merge_from(a)
    b.set_visible(t1, K.KIND_TEXT, _oracle.visible_ids([first], t1, K.KIND_TEXT))
    b.text_insert(t1, 5, " world"); b.map_set(m1, "age", 37); b.commit()
    a.text_insert(t1, 0, ">> "); a.map_set_container("root", "old", K.KIND_MAP); a.map_set("root", "old", 1)
    a.commit()
    docs = [[first], [a.export(), b.export()], [b.export(), a.export()]]
    got, want = _check(docs)
    import json
    assert json.loads(got[0][1]) == {"rl": [["x"]], "root": {"empty": [], "items": [1, {"k": [1, 2, 3]}, "two", ""],
                                                             "profile": {"age": 36, "bio": "hello", "name": "Ada"}}}
    assert json.loads(got[1][1])["root"]["profile"] == {"age": 37, "bio": ">> hello world", "name": "Ada"}
    assert json.loads(got[1][1])["root"]["old"] == 1 and got[1][1] == got[2][1]


def test_child_containers_random_sessions():
    docs = _nested_docs(24, n_peers=3, n_steps=150)
    got, want = _check(docs)
    assert all(w[0] == 0 for w in want) and max(w[1].count(b"{") for w in want) > 4


def test_damaged_blobs_never_take_the_batch_down():
    """150 documents with one damaged blob each (checksum re-fitted), interleaved with healthy documents: every healthy
    document still comes back exact, no damaged document crashes the batch, and a damaged document is either rejected or
    — when the oracle accepts it too — rendered alike (a delete row whose position and target ids disagree, or whose span
    length differs from its op length, is LM_DATA_CORRUPTION: the reference deletes by position, the kernel by id —
    ts_del_pos_ok, lm_k_integrate_span.h)."""
    bad = _cases.corrupted_docs(150, seed=7) + _cases.corrupted_docs(100, seed=8) + _cases.corrupted_docs(100, seed=11)
    good = _cases.fuzz_docs(8, base=6000)
    docs = []
    for i, b in enumerate(bad):
        docs.append(b)
        if i % 8 == 0:
            docs.append(good[(i // 8) % len(good)])
    want = _oracle.merge_batch(docs, threads=8)
    got = _emu.merge_batch(docs)
    n_same = n_both_ok = n_dev_only = 0
    k = 0
    for i in range(len(bad)):
        g, w = got[k], want[k]
        if g[0] == 0 and w[0] == 0:
            n_both_ok += 1
            n_same += g == w
        n_dev_only += g[0] == 0 and w[0] not in (0, 4)
        k += 1
        if i % 8 == 0:
            assert got[k] == want[k] and want[k][0] == 0      # the healthy neighbour
            k += 1
    assert n_both_ok > 0 and n_same == n_both_ok
    assert n_dev_only == 0   # the device never renders a document the oracle rejects (surplus column values, nested key indices, ops across change boundaries)


def test_delete_rows_that_name_elements_nobody_inserted(monkeypatch):
    """The reference applies a delete BY POSITION and only remembers the ids it met (crdt_rope.rs:256-335, tracker.rs:193-252); the
    span-granular batch kernels apply rows by id, compare with the position, and on a mismatch apply the rest of the row by
    position too (ts_del_positional) — retreats / forwards of such a row use what was deleted, not what the row names.
    tests/golden/damaged_peer_table.json: a flipped PeerID byte in one blob's peer table leaves the other blobs' delete rows
    pointing at elements of a peer without any; sessions whose delete ops were re-pointed at other counters / peers after the fact
    render like the unharmed sessions (oracle == device == the unharmed value), under the structural checker, with the linear
    prefix and the tracker's base on documents of every size, through the common kernel; LM_POSDEL=0: LM_DATA_CORRUPTION."""
    import json, os
    from loro_amd._cabi import Context
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "damaged_peer_table.json")))
    doc = [bytes.fromhex(h) for h in fx["blobs_hex"]]
    want = _oracle.merge_batch([doc])[0]
    assert want[0] == 0 and _emu.merge_batch([doc])[0] == want
    bad, good = _cases.misnamed_delete_docs(48)
    want = _oracle.merge_batch(bad, threads=8)
    assert want == _oracle.merge_batch(good, threads=8) and all(w[0] == 0 for w in want)
    b = _emu.variant(["LM_SWEEP_EAGER", "LM_EMU_CHECK"])
    for env in ({}, {"LM_CUT_MIN_ROWS": "0"}, {"LM_PLAIN": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with Context(b) as c:
            assert c.merge_batch(bad) == want, env
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("LM_POSDEL", "0")
    got = _emu.merge_batch(bad)
    assert all(g[0] in (0, 3) for g in got) and sum(g[0] == 3 for g in got) > len(bad) // 2
    assert all(g == w for g, w in zip(got, want) if g[0] == 0)


def test_optimistic_lww_table_overflow_takes_the_second_pass(monkeypatch):
    """LM_HT_OPT=64: every document with more than 32 distinct (container, key) pairs — or more Map rows than 32 — fills its
    optimistic LWW table, is flagged DF_LWW_RETRY and resolved again in a table sized for its rows; documents next to it keep
    their small tables."""
    monkeypatch.setenv("LM_HT_OPT", "64")
    docs = [workload.cfg3_doc(d, n_peers=4, n_writes=300, n_keys=k, combined=(d % 2 == 0), per_change=50) for d, k in enumerate([8, 200, 31, 33, 120, 16])]
    docs += _cases.cfg4_docs(6, first=4100, n_steps=150)
    _check(docs, ["lww %d" % i for i in range(len(docs))])


def test_ascii_pastes_with_every_length_prefix_width():
    docs = _cases.ascii_paste_docs()
    got, _ = _check(docs, ["paste %d" % i for i in range(len(docs))])
    assert all(g[0] == 0 for g in got) and got[0][1] == got[1][1] and got[6][1] == got[7][1] and len(got[6][1]) > 2200000


def test_run_async_and_wait_with_two_contexts():
    """lm_run_async / lm_wait: two contexts alternate (the double-buffered serving loop of bench.py)."""
    from loro_amd._cabi import Context
    docs_a = _cases.fuzz_docs(6, base=7100)
    docs_b = _cases.fuzz_docs(6, base=7200)
    want_a, want_b = _oracle.merge_batch(docs_a), _oracle.merge_batch(docs_b)
    with Context(_emu.binding()) as a, Context(_emu.binding()) as b:
        a.stage(docs_a); b.stage(docs_b)
        for _ in range(3):
            a.run_async(); b.run_async()
            a.wait(); b.wait()
        a.wait()                                   # waiting with nothing in flight is a no-op
        assert a.fetch() == want_a and b.fetch() == want_b
        a.run_async()
        with pytest.raises(RuntimeError):
            a.run_async()                          # one run in flight per context
        a.wait()


def test_retry_launch_keeps_map_containers(monkeypatch):
    """Regression: the second (worst-case directory) launch reset the `touched` flag of every container, so the root Map
    of a document whose Text overflowed the optimistic directory vanished from the JSON."""
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_DIR_OPT_MAX", "4")
    monkeypatch.setenv("LM_SPAN", "0")
    docs = _cases.cfg4_docs(3, first=1016, n_steps=400) + _cases.fuzz_docs(6, base=100, steps=120)
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        assert c.sizing()[3] >= 1
    assert got == want


@pytest.mark.parametrize("plain", ["2", "1", "0"])
def test_retry_launch_of_every_span_instantiation(monkeypatch, plain):
    """The worst-case-directory launch re-runs each overflowed document with the instantiation that owns it (plain sweep /
    plain / common) — a batch of plain and other documents with the optimistic directory forced down to four entries."""
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_DIR_OPT_MAX", "4")
    monkeypatch.setenv("LM_PLAIN", plain)
    docs = _cases.cfg4_docs(3, first=1016, n_steps=400) + _cases.fuzz_docs(6, base=100, steps=120) + _cases.trace_docs(3000, n_docs=1)
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        assert c.sizing()[3] >= 1
    assert got == want
    # … also when the overflow happens inside the linear prefix (tl_dir_insert_after; flags for documents of every size), and for
    # documents that are then replayed a THIRD time by k_integrate_span_pos (a delete row that does not match its position)
    monkeypatch.setenv("LM_CUT_MIN_ROWS", "0")
    bad, good = _cases.misnamed_delete_docs(12)
    docs = _cases.linear_prefix_docs(18) + bad
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        assert c.sizing()[3] >= 1
    assert got == want


def test_small_document_batches_pick_the_element_granular_kernel(monkeypatch):
    """LM_SPAN_AUTO (the product default): a batch whose documents are small and mostly the common kernel's — configs[3]-shaped
    documents, MovableLists, checkouts — is replayed by the element-granular kernel; plain documents, or one large document in the
    batch, keep the span-granular kernels.  Same bytes either way."""
    import _fuzz
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_SPAN_AUTO", "1")
    small = _cases.cfg4_docs(6, first=8100, n_steps=200) + [_fuzz.blobs_of(_fuzz.movable_session(8200 + d, n_peers=3, n_steps=80, nested=True)) for d in range(3)]
    plain = _cases.fuzz_docs(12, base=8300)
    big = [workload.Cfg2Template(3000, 1500, seed=4, commit_every=1, fuse=False).stamp(0)]   # one op row per keystroke: >= 4,096 rows

    def stage_names(docs):
        with Context(_emu.binding()) as c:
            c.stage(docs); c.set_profiling(1); c.run()
            names = {n for n, _ in c.kernel_times()}
            got = c.fetch()
        assert got == _oracle.merge_batch(docs)
        return names
    assert "k_integrate" in stage_names(small)
    assert "k_integrate" not in stage_names(plain)
    assert "k_integrate" not in stage_names(small + big)
    monkeypatch.setenv("LM_SPAN_AUTO", "0")
    assert "k_integrate" not in stage_names(small)


@pytest.mark.parametrize("span", ["1", "0"])
def test_both_integrate_kernels(monkeypatch, span):
    """Both integrate kernels — span-granular (default, lm_k_integrate_span.h) and element-granular (LM_SPAN=0,
    lm_k_integrate.h) — on the edge cases, random concurrent sessions, nested containers, checkouts (incl. cuts through op
    runs) and a trace-shaped document."""
    monkeypatch.setenv("LM_SPAN", span)
    names, docs = _cases.edge_case_docs()
    _check(docs, names)
    _check(_cases.fuzz_docs(16, base=300, steps=80) + _nested_docs(8, first=7300, n_peers=3, n_steps=120) + _cases.trace_docs(3000, n_docs=1))
    cd, cf = _checkout_cases()
    want = _oracle.merge_batch(cd, frontiers=cf)
    got = _emu.merge_batch(cd, cf)
    for g, w in zip(got, want):
        assert (g == w) if w[0] == 0 else (g[0] == w[0])


def test_keystroke_per_change_histories_stay_run_granular():
    """Real-time typing sends one change per keystroke: the span-granular kernel merges the runs back
    (FugueSpan::is_mergeable), so 3,000 keystrokes need a handful of leaves, not one item per character."""
    from loro_amd import wire
    from loro_amd._cabi import Context
    import _fuzz
    docs = [_fuzz.blobs_of(_fuzz.random_session(8800 + s, n_peers=2, n_steps=400, kinds=("text",), sync_prob=0.03, max_ins=1, commit_prob=1.0))
            for s in range(4)]
    r = wire.Replica(5)
    for i in range(3000):
        r.text_insert("text", i, "abcdefghij"[i % 10]); r.commit()
    docs.append([r.export()])
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        leaves_used = c.sizing()[0]
    assert got == want
    assert leaves_used <= 16, leaves_used      # 3,000 sequential keystrokes = one run


def test_json_longer_than_the_optimistic_slab_is_rerendered():
    """Map-typed values reference their keys by index, so `{"someLongerFieldNameHere":null}` costs 4 input bytes and
    renders 36; child maps repeat keys too.  The emitter must never write beyond a document's slab: such a document
    is re-rendered at its exact size and its neighbours in the batch are untouched (lm_pipeline.h slab cap)."""
    from loro_amd import wire
    from loro_amd._cabi import Context
    big = wire.Replica(5)
    big.list_insert("l", 0, [{"someLongerFieldNameHere": None}] * 3000)
    big.commit()
    kids = wire.Replica(6)
    for i in range(40):
        c = kids.map_set_container("root", "child%02d" % i, wire.KIND_MAP)
        for k in ("aRatherLongKeyThatEveryChildRepeats", "anotherLongKeySharedByAllTheChildren"):
            kids.map_set(c, k, i)
    kids.commit()
    small = _cases.fuzz_docs(3, base=8100)
    docs = [small[0], [big.export()], small[1], [kids.export()], small[2]]
    want = _oracle.merge_batch(docs)
    assert len(want[1][1]) > 6 * len(docs[1][0])          # beyond even the old 6x bound
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        sizing = c.sizing()
    assert got == want
    assert sizing[4] >= 1, f"expected a re-render, sizing={sizing}"


def test_forced_rerender_of_every_document(monkeypatch):
    """LM_SLAB_CAP=16: every document overflows its slab in the first emit pass and goes through the exact-size pass."""
    from loro_amd._cabi import Context
    monkeypatch.setenv("LM_SLAB_CAP", "16")
    names, docs = _cases.edge_case_docs()
    docs = docs + _cases.fuzz_docs(6, base=8200) + _nested_docs(4, n_peers=3, n_steps=120)
    want = _oracle.merge_batch(docs)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        assert c.sizing()[4] >= 10
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g == w) if w[0] in (0, 4) and g[0] == w[0] and g[1] else (g[0] == w[0]), i


def _fixture_docs_and_check():
    """(docs, check(got)): the reference fixtures that hold out-of-scope containers next to in-scope ones."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))
    b = {k: bytes.fromhex(v) for k, v in fx["blobs"].items()}
    docs = [[b["updates.blob"]], [b["updates.ts.blob"]], [b["runtime-updates.ts.blob"]],
            [b["concurrent-base.ts.blob"], b["concurrent-left.ts.blob"], b["concurrent-right.ts.blob"]],
            [b["concurrent-base.ts.blob"], b["concurrent-right.ts.blob"], b["concurrent-left.ts.blob"]]]

    def check(got):
        want = _oracle.merge_batch(docs)
        assert got == want
        assert all(g[0] == 4 and g[1] for g in got)
        deep = fx["json"]["snapshot.deep.json"]
        for g in got[:2]:
            v = json.loads(g[1])
            # "list" / "text" end empty in this history: an updates import creates no state for them;
            # snapshot.deep.json is the value of a snapshot import, which carries the exporting document's states
            assert v.get("list", []) == deep["list"] == [] and v.get("text", "") == deep["text"] == ""
            assert v["mlist"] == deep["mlist"] == []      # a MovableList exists once an element was inserted
            for k, x in deep["map"].items():
                if k != "child_tree":
                    assert v["map"][k] == x, k
        rt = fx["json"]["runtime.expected.json"]
        v = json.loads(got[2][1])
        assert v["list"] == rt["list"] and v["text"] == rt["text"] and all(v["map"][k] == rt["map"][k] for k in ("answer", "child", "nested"))
        assert v["movable"] == rt["movable"] == ["z", "x"]
        cc = fx["json"]["concurrent.expected.json"]
        for g in got[3:]:
            v = json.loads(g[1])
            assert all(v[k] == cc[k] for k in ("list", "text", "map"))
    return docs, check


def test_reference_fixtures_with_out_of_scope_containers_render_the_rest():
    """Rust-written `updates.blob` (and the TS-written fixtures) hold Tree / Counter containers: the device
    path renders every in-scope key, shows the out-of-scope containers as null and reports LM_UNSUPPORTED *with* the JSON —
    compared key by key with the oracle and with the reference's expected deep JSON (loro_js_interop.rs:42-126)."""
    docs, check = _fixture_docs_and_check()
    check(_emu.merge_batch(docs))


@pytest.mark.parametrize("variant", ["wave", "wave-unstaged", "wave-slot-64", "wave-slot-160", "wave-slot-320", "wave-two-launches-64-320", "wave-two-launches-160-4096", "wave-columns-launch-320", "lane"])
def test_block_decoders_agree(monkeypatch, variant):
    """The wave-per-block-group decoder (default; LDS-staged, lane = block x column), the same kernel with slots too
    small to stage anything (every parser reads HBM) and the one-lane-per-block decoder (LM_DECODE=0) must all give the
    oracle's results — including WHICH error a damaged block is rejected with."""
    if variant == "wave-unstaged":
        monkeypatch.setenv("LM_DEC_SLOT", "16")
        monkeypatch.setenv("LM_DEC_SLOT_BIG", "0")   # (no second launch with larger slots for the groups whose heads exceed the slot)
    if variant.startswith("wave-slot-"):   # slots between the blocks' column bytes and their heads: some blocks staged whole, some with their op / delete-start columns only, some not at all
        monkeypatch.setenv("LM_DEC_SLOT", variant.rsplit("-", 1)[1])
        monkeypatch.setenv("LM_DEC_SLOT_BIG", "0")
    if variant.startswith("wave-two-launches-"):   # groups with a head beyond the first slot are decoded by a second launch with the larger one
        monkeypatch.setenv("LM_DEC_SLOT", variant.split("-")[3])
        monkeypatch.setenv("LM_DEC_SLOT_BIG", variant.split("-")[4])
        monkeypatch.setenv("LM_DEC_BIG_MODE", "2")   # slots of the second launch sized for the heads (capped by LM_DEC_SLOT_BIG)
    if variant.startswith("wave-columns-launch-"):   # the default mode: the second launch's slots are sized for the op / delete-start columns of those groups
        monkeypatch.setenv("LM_DEC_SLOT", variant.rsplit("-", 1)[1])
    if variant == "lane":
        monkeypatch.setenv("LM_DECODE", "0")
    names, docs = _cases.edge_case_docs()
    _check(docs, names)
    _check(_cases.fuzz_docs(16, base=9300) + _cases.cfg4_docs(3, first=1200, n_steps=300) + _nested_docs(4, n_peers=3, n_steps=150))
    docs, check = _fixture_docs_and_check()
    check(_emu.merge_batch(docs))
    bad = _cases.corrupted_docs(120, seed=11)
    got = _emu.merge_batch(bad)
    monkeypatch.delenv("LM_DEC_SLOT", raising=False)
    monkeypatch.delenv("LM_DEC_SLOT_BIG", raising=False)
    monkeypatch.delenv("LM_DEC_BIG_MODE", raising=False)
    monkeypatch.setenv("LM_DECODE", "0")
    ref = _emu.merge_batch(bad)          # the sequential decoder's verdicts
    assert [g[0] for g in got] == [x[0] for x in ref]
    assert got == ref


def test_two_level_directory_of_deep_histories():
    """the leaf directory's second level (sums of 64 entries, lm_k_integrate_span.h SD_LINEAR) — built with LM_SD_LINEAR=2 so that
    every document uses it, with the structural checker after every op: documents with > 64 leaves (several blocks of sums),
    checkouts, concurrent sessions, MovableLists"""
    import _fuzz
    from loro_amd import workload
    from loro_amd._cabi import Context
    b = _emu.variant(["LM_SD_LINEAR=2", "LM_SD_BSH=2", "LM_EMU_CHECK"])   # blocks of four entries: block boundaries everywhere
    names, edge = _cases.edge_case_docs()
    docs = [edge[names.index("many leaves")], edge[names.index("long pastes")]] + _cases.fuzz_docs(12, base=8100)
    docs += [_fuzz.blobs_of(_fuzz.movable_session(8200 + i, n_steps=80, nested=True)) for i in range(4)]
    import random
    rng = random.Random(9)
    big = wire.Replica(77)
    n = 0
    for i in range(6000):                      # single characters at random positions: every one its own run — > 100 leaves
        big.text_insert("text", rng.randint(0, n), "abcdefghij"[i % 10]); n += 1
        if i % 9 == 0 and n > 4:
            big.text_delete("text", rng.randint(0, n - 2), 1); n -= 1
        if i % 40 == 0:
            big.commit()
    big.commit()
    docs.append([big.export()])
    fr = [None] * len(docs)
    blobs, f = workload.cfg5_doc(3, n_ops=12000, turn=500, n_checkouts=6, commit_every=1)
    docs += [blobs] * 6; fr += f
    want = _oracle.merge_batch(docs, threads=4, frontiers=fr)
    with Context(b) as c:
        got = c.merge_batch(docs, fr)
        assert c.sizing()[0] > 64        # leaves used by the largest document: more than one block of sums
    assert got == want and all(w[0] == 0 for w in want)


@pytest.mark.parametrize("decoder", ["1", "0"])
def test_a_giant_run_in_a_column_costs_nothing(monkeypatch, decoder):
    """a column of a few bytes whose AnyRle run count is near 2^28 is counted per run, not stepped through value by
    value, and a value-type column that announces more rows than the block has op ids sizes no table — DecodeError at once
    (both decoders)."""
    import time
    monkeypatch.setenv("LM_DECODE", decoder)
    names, docs = _cases.huge_run_column_docs()
    t = time.time()
    got = _emu.merge_batch(docs)
    assert time.time() - t < 20.0
    assert [g[0] for g in got] == [1] * len(docs), list(zip(names, [g[0] for g in got]))
    assert [w[0] for w in _oracle.merge_batch(docs)] == [1] * len(docs)


def test_value_level_corruption_in_a_rejected_block_is_named_like_the_reference(monkeypatch):
    """k_block_reclassify: a block a row decoder rejected with DecodeError is read once more, sequentially, with the value reader
    that reports an undefined value tag / a nested key index beyond the key table / an oversized collection as the reference does
    (DecodeDataCorruptionError, value.rs:342-459).  On a 450-document damaged corpus the documents BOTH sides reject agree on the
    code more often with the pass than without it, the pass never changes what is accepted, and a hand-damaged value tag is
    LM_DATA_CORRUPTION on both sides."""
    docs = _cases.corrupted_docs(150, seed=7) + _cases.corrupted_docs(150, seed=8) + _cases.corrupted_docs(150, seed=21)
    want = _oracle.merge_batch(docs, threads=8)

    def differing(env):
        monkeypatch.setenv("LM_RECLASS", env)
        got = _emu.merge_batch(docs)
        assert [g[0] == 0 for g in got] == [g[0] == 0 for g in got0] if env == "0" else True
        return got, sum(1 for g, w in zip(got, want) if g[0] != 0 and w[0] != 0 and g[0] != w[0])
    got0, with_pass = differing("1")
    _, without = differing("0")
    assert with_pass < without and with_pass * 10 <= sum(1 for w in want if w[0] != 0), (with_pass, without)
    # a Map value whose tag byte is undefined: tag 0x3f where the string's tag 5 stood
    r = wire.Replica(3)
    r.map_set("m", "k", "abcdefgh"); r.commit()
    blob = bytearray(r.export())
    at = blob.index(b"\x05\x08abcdefgh")
    blob[at] = 0x3f
    import struct
    body = bytes(blob[20:])
    bad = bytes(blob[:16]) + struct.pack("<I", _oracle.xxh32(body)) + body
    monkeypatch.setenv("LM_RECLASS", "1")
    assert _oracle.merge_batch([[bad]])[0][0] == 3 == _emu.merge_batch([[bad]])[0][0]


def _damaged_checkout_docs(n=220, seed=1):
    """(docs, frontiers): rich sessions with one blob damaged by byte flips (checksum re-fitted), most rendered at a recorded version"""
    import random, struct, _fuzz
    rng = random.Random(seed)
    base = []
    for s in range(10):
        snaps = []
        reps = _fuzz.random_session(9500 + s, n_peers=3, n_steps=80, kinds=("text", "list", "map"), styles="rich", snapshots=snaps)
        base.append((_fuzz.blobs_of(reps), [v for v, _ in snaps]))
    docs, fronts = [], []
    for _ in range(n):
        blobs, vers = rng.choice(base)
        d = list(blobs); j = rng.randrange(len(d)); b = bytearray(d[j])
        for _ in range(rng.choice([0, 1, 1, 2])):
            k = rng.randrange(22, len(b))
            b[k] = rng.choice([b[k] ^ (1 << rng.randrange(8)), rng.randrange(256), 0xFF, 0])
        body = bytes(b[20:])
        d[j] = bytes(b[:16]) + struct.pack("<I", _oracle.xxh32(body)) + body
        docs.append(d)
        fronts.append(wire.encode_frontiers(rng.choice(vers)) if vers and rng.random() < 0.8 else None)
    return docs, fronts


def test_checkouts_through_the_full_import(monkeypatch):
    """LM_CHECKOUT_FULL=1: a checked-out entry imports its whole history and reaches the version by moving the trackers (the
    reference's import + checkout) instead of replaying the version's causal closure.  Healthy documents: the same bytes as ever
    (the checkout known answers and every recorded version of random sessions).  Damaged documents: damage that lies OUTSIDE the
    rendered version is met as the reference meets it — without the knob the closure replay renders a few documents the oracle
    rejects (DESIGN §7 "Checkout"); with it none, and what both accept is rendered alike."""
    from loro_amd._cabi import Context
    docs, fronts = _damaged_checkout_docs()
    want = _oracle.merge_batch(docs, threads=8, frontiers=fronts)

    def dev_only(got):
        return sum(1 for g, w in zip(got, want) if g[0] == 0 and w[0] not in (0, 4))
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")
    assert dev_only(_emu.merge_batch(docs, fronts)) > 0          # the documented deviation of the closure replay
    monkeypatch.delenv("LM_SHARE_REPLAY")
    with Context(_emu.binding()) as c:                           # (the default since round 6; LM_CHECKOUT_FULL=0 switches it off)
        got = c.merge_batch(docs, fronts)
        assert c.b.shared_documents(c.h) == len(docs)
    assert dev_only(got) == 0
    n_both = 0
    for g, w in zip(got, want):
        if g[0] == 0 and w[0] == 0:
            assert g == w
            n_both += 1
    assert n_both > 70
    cd, cf = _checkout_cases()
    cw = _oracle.merge_batch(cd, frontiers=cf)
    for i, (g, w) in enumerate(zip(_emu.merge_batch(cd, cf), cw)):
        assert (g == w) if w[0] == 0 else (g[0] == w[0]), (i, g[:3], w[:3])


def test_a_damaged_option_tag_in_front_of_an_empty_delta_of_delta_column():
    """a block without foreign dependencies (its dependency-counter column is the empty DeltaOfDelta: option tag 00, used bits 00) whose
    tag byte was damaged (3a): DecodeError in the reference (`DeltaOfDeltaDecoder::new`, block_meta_encode.rs:190-214) — dod_finish used
    to return before it looked at the tag of an empty stream.  Found on damaged resident sessions (one blob of 129 bytes)."""
    bad = bytes.fromhex(_cases.DOD_TAG_BLOB_HEX)
    assert _oracle.merge_batch([[bad]])[0][0] == 1
    for dec in ("1", "0"):
        os.environ["LM_DECODE"] = dec
        try:
            assert _emu.merge_batch([[bad]])[0][0] == 1
        finally:
            del os.environ["LM_DECODE"]


def test_damaged_change_meta_columns_are_data_corruption_like_the_reference():
    """every failure of the timestamp / message-length columns is LoroError::DecodeDataCorruptionError in the
    reference (block_encode.rs:563-571 maps both decoders' errors) — the oracle and both device decoders said DecodeError for a
    column that does not decode; the header columns (block_meta_encode.rs) keep DecodeError"""
    names, docs = _cases.damaged_change_meta_docs()
    want = _oracle.merge_batch(docs)
    assert [w[0] for w in want] == [3] * len(docs), list(zip(names, [w[0] for w in want]))
    for dec in ("1", "0"):
        os.environ["LM_DECODE"] = dec
        try:
            got = _emu.merge_batch(docs)
            assert [g[0] for g in got] == [3] * len(docs), (dec, list(zip(names, [g[0] for g in got])))
        finally:
            del os.environ["LM_DECODE"]


# ---- round 6: LWW Map documents without op rows (lm_k_map_fused.h) and the side engine that replays what a configuration has no path for

def _map_env(monkeypatch, on=True):
    monkeypatch.setenv("LM_MF_MIN_ROWS", "1"); monkeypatch.setenv("LM_MF_CHG_RATIO", "0")   # (the product asks for 2,048 rows and 4 per change)
    if not on:
        monkeypatch.setenv("LM_MAP_FUSED", "0")


def test_map_documents_are_resolved_without_op_rows(monkeypatch):
    """configs[2]-shaped documents (both variants) and 160 random Map sessions: op columns folded straight into the LDS LWW table,
    results equal to the oracle's and to the row-table path's; documents the kernel is not built for (nested values, long keys) are
    replayed through the row tables by the side engine"""
    from loro_amd import workload
    _map_env(monkeypatch)
    docs = [workload.cfg3_doc(d, n_peers=4, n_writes=700, n_keys=96, combined=d % 2 == 0, per_change=50) for d in range(4)]
    docs += _cases.map_sessions(160, scalar_only=False)
    want = _oracle.merge_batch(docs, threads=8)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs)
        n_fused, n_redo = c.b.fused_documents(c.h), c.b.redo_documents(c.h)
    assert got == want
    assert n_fused == len(docs) and 20 <= n_redo <= 110, (n_fused, n_redo)
    monkeypatch.setenv("LM_MAP_FUSED", "0")
    with Context(_emu.binding()) as c:
        assert c.merge_batch(docs) == want and c.b.fused_documents(c.h) == 0


def test_map_documents_without_op_rows_at_checked_out_versions(monkeypatch):
    from loro_amd import workload
    _map_env(monkeypatch)
    monkeypatch.setenv("LM_SHARE_REPLAY", "0")   # (a checked-out entry of a folded batch is a resident document: the row tables)
    docs, fr = [], []
    for d in range(6):
        blobs = workload.cfg3_doc(d, n_peers=3, n_writes=200, n_keys=30, combined=(d % 2 == 0), per_change=20)
        for ctr in (0, 57, 199):
            docs.append(blobs); fr.append(wire.encode_frontiers([(d * 1000 + 1, ctr)] + ([(d * 1000 + 2, 100)] if ctr == 57 else [])))
        docs.append(blobs); fr.append(None)
    with Context(_emu.binding()) as c:
        got = c.merge_batch(docs, fr)
        assert c.b.fused_documents(c.h) == len(docs)
    assert got == _oracle.merge_batch(docs, frontiers=fr)


@pytest.mark.parametrize("seed", [1, 2])
def test_damaged_map_documents_get_the_row_decoders_verdicts(monkeypatch, seed):
    """the fused kernel gives no verdict on anything it is not built for — the side engine's row decoders do: status, bytes and
    pending ops of every damaged document are those of the row-table path; nothing the oracle rejects is rendered"""
    docs = _cases.damaged_map_docs(300, seed=seed)
    want = _oracle.merge_batch(docs, threads=8)
    _map_env(monkeypatch)
    with Context(_emu.binding()) as c:
        fused = c.merge_batch(docs)
        assert c.b.fused_documents(c.h) > 100 and c.b.redo_documents(c.h) > 20
    monkeypatch.setenv("LM_MAP_FUSED", "0")
    with Context(_emu.binding()) as c:
        rows = c.merge_batch(docs)
    assert fused == rows
    assert not [i for i in range(len(docs)) if want[i][0] != 0 and fused[i][0] == 0]
    assert not [i for i in range(len(docs)) if want[i][0] == 0 and fused[i][0] == 0 and fused[i] != want[i]]


def test_a_documents_verdict_depends_neither_on_its_neighbours_nor_on_shared_blobs(monkeypatch):
    """ADVICE r5 (medium + low): delete rows that name other elements than the ones at their positions are applied by position by the
    reference (crdt_rope.rs:256-335) and by the span-granular batch kernels.  The element-granular kernel (picked by the batch's
    statistics, LM_SPAN_AUTO) and the resident kernels (every entry of a folded batch — entries that share their blobs, any
    checked-out entry) have no such path: they flag the document DF_REDO and the context replays it through the batch pipeline, so
    every configuration renders what the oracle renders; LM_REDO=0 shows the round-5 behaviour."""
    bad, _good = _cases.misnamed_delete_docs(12)
    empty = b"\x00"
    want_latest = _oracle.merge_batch(bad)
    want_empty = _oracle.merge_batch(bad, frontiers=[empty] * len(bad))
    assert all(w[0] == 0 for w in want_latest)
    docs, fr, want = [], [], []
    for i, d in enumerate(bad):
        docs += [d, d]; fr += [None, empty]; want += [want_latest[i], want_empty[i]]
    for env in ({"LM_SPAN_AUTO": "1", "LM_SHARE_REPLAY": "0"}, {"LM_SPAN_AUTO": "0"}, {"LM_SPAN_AUTO": "1"}, {"LM_SPAN_AUTO": "1", "LM_CHECKOUT_FULL": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with Context(_emu.binding()) as c:
            got = c.merge_batch(docs, fr)
            n_redo = c.b.redo_documents(c.h)
        assert got == want, env
        assert n_redo > 0 or env.get("LM_SPAN_AUTO") == "0" and env.get("LM_SHARE_REPLAY") == "0", env
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("LM_SPAN_AUTO", "0"); monkeypatch.setenv("LM_REDO", "0")
    with Context(_emu.binding()) as c:
        assert [g[0] for g in c.merge_batch(docs, fr)] == [3] * len(docs)
