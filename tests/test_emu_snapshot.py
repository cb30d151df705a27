"""FastSnapshot ingest beyond the reference fixtures: snapshots written by loro_amd.wire (SSTable with many
blocks, large-value blocks, LZ4 frames or stored bodies) through the kernel-logic harness against the oracle, whose snapshot
reader (oracle/lo_snapshot.hpp) is a separate restatement."""
import random, struct
import pytest

import _emu, _fuzz, _oracle, _resident
from loro_amd._cabi import Context
from loro_amd import wire, workload


def _refit(blob: bytes) -> bytes:
    b = bytearray(blob)
    b[16:20] = struct.pack("<I", wire.xxh32(bytes(b[20:])))
    return bytes(b)


def snapshot_docs(n=24, first=0):
    """documents made of written snapshots: alone, next to updates (before / after), two snapshots with different root sets"""
    docs, names = [], []
    for seed in range(first, first + n):
        rng = random.Random(seed)
        mode = seed % 4
        if mode == 3:
            reps = _fuzz.movable_session(seed, n_peers=3, n_steps=80, nested=seed % 8 == 3)
        elif mode == 2:
            reps = _fuzz.nested_session(seed, n_peers=3, n_steps=100)
        else:
            reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 4), n_steps=rng.randint(40, 200), kinds=("text", "list", "map"), styles=True, max_ins=20)
        kw = [dict(), dict(block_size=200), dict(compress=False), dict(block_size=64, max_block=300), dict(block_size=1 << 16, max_block=1 << 15)][seed % 5]
        a, b = reps[0], reps[-1]
        sa, sb = a.export_snapshot(**kw), b.export_snapshot(**kw)
        upd = _fuzz.blobs_of(reps, rng)
        docs += [[sa], [sa] + upd, upd + [sb], [sa, sb], [sb, sa]]
        names += [f"{seed}:{k}" for k in ("snapshot", "snapshot+updates", "updates+snapshot", "two snapshots", "two snapshots reversed")]
        # a root that only the state section knows: nothing of the history touches it
        extra = a.export_snapshot(roots=[(wire.KIND_TEXT, "ghost_text"), (wire.KIND_LIST, "ghost_list"), (wire.KIND_MAP, "ghost_map"), (wire.KIND_MOVABLE, "ghost_ml")], **kw)
        docs.append([extra]); names.append(f"{seed}:state-only roots")
        docs.append([sb, extra]); names.append(f"{seed}:which snapshot initialises the state")
    return names, docs


def damaged_snapshots():
    a = _fuzz.random_session(77, n_peers=2, n_steps=150, kinds=("text", "map"))[0]
    s = a.export_snapshot(block_size=128)
    out = []
    for at in (40, 60, len(s) // 2, len(s) - 30, len(s) - 10, len(s) - 5):
        b = bytearray(s); b[at] ^= 0x21
        out.append([_refit(bytes(b))])       # the envelope checksum fits again: the SSTable's own checksums / structure must catch it
    out.append([s[:len(s) - 7]])
    out.append([_refit(s[:len(s) - 7])])
    return out


def test_written_snapshots_match_the_oracle():
    names, docs = snapshot_docs(24)
    want = _oracle.merge_batch(docs, threads=4)
    got = _emu.merge_batch(docs)
    assert sum(1 for w in want if w[0] == 0) > len(docs) // 2
    for n, g, w in zip(names, got, want):
        assert g == w, (n, g[:2], w[:2])


def test_state_only_roots_and_a_movable_list_root_do_not_flag_the_document():
    a = wire.Replica(5); a.text_insert("text", 0, "x"); a.commit()
    snap = a.export_snapshot(roots=[(wire.KIND_TEXT, "text"), (wire.KIND_MOVABLE, "ml"), (wire.KIND_LIST, "l")])
    got = _emu.merge_batch([[snap], [a.export()]])
    assert got[0][:2] == (0, b'{"l":[],"ml":[],"text":"x"}') and got[1][:2] == (0, b'{"text":"x"}')
    assert got == _oracle.merge_batch([[snap], [a.export()]])
    tree = a.export_snapshot(roots=[(wire.KIND_TEXT, "text"), (3, "tree")])    # a Tree root renders as null and flags the document
    g = _emu.merge_batch([[tree]])[0]
    assert g[:2] == (4, b'{"text":"x","tree":null}') and g == _oracle.merge([tree])


def test_damaged_sstables_fail_like_the_oracle_says():
    docs = damaged_snapshots()
    want = _oracle.merge_batch(docs)
    got = _emu.merge_batch(docs)
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g[0] == 0) == (w[0] == 0), (i, g[0], w[0])
        if w[0] == 0:
            assert g == w, i
    assert sum(1 for w in want if w[0] != 0) >= 5


def test_a_configs1_sized_document_as_one_snapshot():
    """a 30k-op two-peer concurrent text document as ONE snapshot (dozens of SSTable blocks, LZ4 frames) equals its three update blobs"""
    from loro_amd.workload import synthetic_trace, _apply
    acts = synthetic_trace(30000, 7)
    a = wire.Replica(101); _apply(a, acts[:20000], 10)
    b = wire.Replica(102); b.merge_from(a); b.seq = {k: list(v) for k, v in a.seq.items()}
    _apply(a, acts[20000:], 10)
    _apply(b, [(p, dl, ("Z" if ch else "")) for (p, dl, ch) in acts[20000:]], 10)
    a.merge_from(b)
    snap = a.export_snapshot()
    upd = a.export()
    got = _emu.merge_batch([[snap], [upd]])
    assert got[0][0] == 0 and got[0] == got[1] == _oracle.merge([snap])


def test_snapshots_and_resident_documents():
    """a snapshot opens a resident document (its state roots stay), later snapshots arrive as updates"""
    sessions = []
    for seed in range(40, 52):
        rng = random.Random(seed)
        reps = _fuzz.random_session(seed, n_peers=3, n_steps=120, kinds=("text", "list", "map"))
        early = wire.Replica(reps[0].peer)
        k = max(1, len(reps[0].changes.get(reps[0].peer, [])) // 2)
        early.changes = {reps[0].peer: reps[0].changes.get(reps[0].peer, [])[:k]}
        early.vv = {reps[0].peer: early.changes[reps[0].peer][-1].ctr_end} if early.changes[reps[0].peer] else {}
        early.frontiers = [(reps[0].peer, early.vv[reps[0].peer] - 1)] if early.vv else []
        ok_early = all(all(d[0] == reps[0].peer for d in c.deps) for c in early.changes[reps[0].peer])
        first = early.export_snapshot(roots=[(wire.KIND_TEXT, "text"), (wire.KIND_LIST, "never")], block_size=150) if ok_early else reps[0].export_snapshot(block_size=150)
        rest = _resident.chunked_blobs(reps, rng)
        sessions.append([([first], None), (rest[:len(rest) // 2], None), ([reps[-1].export_snapshot()], None), (rest[len(rest) // 2:], None)])
    want = _resident.oracle_sessions(sessions)
    with Context(_emu.binding()) as c:
        got = _resident.run_sessions(c, sessions)
    for k, (g, w) in enumerate(zip(got, want)):
        for i, (x, y) in enumerate(zip(g, w)):
            assert x == y, (k, i, x[:2], y[:2])


# ---- documents staged from their snapshot's STATE section (lm_snapshot.h snapshot_state_to_updates) ----
def real_snapshot(rep, **kw):
    """rep's snapshot with a REAL state section: the entries the checker's state writer (oracle/lo_state_write.hpp) gives for rep's history"""
    st, ents = _oracle.state_entries([rep.export()])
    assert st in (0, 4), st
    return rep.export_snapshot(state=ents, **kw)


def shallow_snapshot(early_blobs, all_blobs, vv, root_frontiers, **kw):
    """a shallow snapshot as shallow_snapshot.rs:32-190 lays it out: third section = the state at the shallow root + `fr`, second section =
    the latest state's entries that differ from the root's, first section = the oplog's `vv` / `fr` (the history above the root is left
    out here: neither state reader needs it, and no reader could replay it without the ops below the root)"""
    st0, root = _oracle.state_entries(early_blobs)
    st1, latest = _oracle.state_entries(all_blobs)
    assert st0 in (0, 4) and st1 in (0, 4)
    same = set(root)
    overlay = [kv for kv in latest if kv not in same]
    root_sst = wire.sstable(root + [(b"fr", wire.encode_frontiers(root_frontiers))], kw.get("block_size", 4096), kw.get("compress", True))
    return wire.encode_snapshot([], [], vv, [], state=overlay, shallow_root_state=root_sst, **kw)


def state_docs(n=40, first=0, styles=False):
    docs = []
    for seed in range(first, first + n):
        rng = random.Random(seed)
        mode = seed % 4
        if mode == 3:
            reps = _fuzz.movable_session(seed, n_peers=3, n_steps=80, nested=seed % 8 == 3)
        elif mode == 2:
            reps = _fuzz.nested_session(seed, n_peers=3, n_steps=100)
        else:
            reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 4), n_steps=rng.randint(40, 200), kinds=("text", "list", "map"), styles=styles, max_ins=20)
        kw = [dict(), dict(block_size=200), dict(compress=False), dict(block_size=64, max_block=300), dict(block_size=1 << 16, max_block=1 << 15)][seed % 5]
        docs += [[real_snapshot(reps[0], **kw)], [real_snapshot(reps[-1], **kw)]]
    return docs


def shallow_docs(n=12, first=300):
    """(docs, frontiers, want): generated shallow snapshots at their latest version and at their shallow root"""
    docs, fronts, want = [], [], []
    for seed in range(first, first + n):
        reps = _fuzz.random_session(seed, n_peers=2, n_steps=120, kinds=("text", "list", "map"), max_ins=12, sync_prob=0.0)
        a = reps[0]
        own = a.changes.get(a.peer, [])
        k = max(1, len(own) // 2)
        if not own or not all(all(d[0] == a.peer for d in c.deps) for c in own[:k]):
            continue
        early = wire.Replica(a.peer); early.changes = {a.peer: own[:k]}
        root_fr = [(a.peer, own[k - 1].ctr_end - 1)]
        full = _fuzz.blobs_of(reps)
        vv = {}
        for r in reps:
            if r.changes.get(r.peer):
                vv[r.peer] = r.changes[r.peer][-1].ctr_end
        snap = shallow_snapshot([early.export()], full, vv, root_fr, **([dict(), dict(block_size=128)][seed % 2]))
        w_latest, w_root = _oracle.merge(full), _oracle.merge([early.export()])
        docs += [[snap], [snap]]; fronts += [None, wire.encode_frontiers(root_fr)]
        want += [w_latest, (w_root[0], w_root[1], w_latest[2], 0)]      # (a checkout does not move the oplog's version vector)
    return docs, fronts, want


def _ref_blobs():
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures.json")))
    return {k: bytes.fromhex(v) for k, v in fx["blobs"].items()}


def check_state_path(ctx_factory, n=40):
    """the state path against the history path and the checker: generated snapshots with real state sections, the reference-held
    snapshots, generated and reference-held shallow snapshots (loro_js_interop.rs:129-147), the fall back when an import follows"""
    import os
    docs = state_docs(n)
    want = _oracle.merge_batch(docs, threads=8)
    assert all(w[0] == 0 for w in want)
    with ctx_factory() as c:
        got = c.merge_batch(docs)
        assert c.b.state_documents(c.h) == len(docs)
        assert got == want
        for d, w in zip(docs, want):                              # the two state readers agree, byte for byte, with the history
            assert _oracle.snapshot_state(d[0]) == (w[0], w[1])
        # lm_richtext needs marks and the real ids of child containers — only the history has them: the batch is staged again through
        # the ChangeStores and run again (same bytes)
        styled = state_docs(8, first=500, styles="rich")
        sw_ = _oracle.merge_batch(styled, threads=8)
        assert c.merge_batch(styled) == sw_ and c.b.state_documents(c.h) == len(styled)
        assert c.richtext() == _oracle.richtext_batch(styled) and c.b.state_documents(c.h) == 0 and c.fetch() == sw_
        os.environ["LM_SNAPSHOT_STATE"] = "0"
        try:
            assert c.merge_batch(docs) == want and c.b.state_documents(c.h) == 0
        finally:
            del os.environ["LM_SNAPSHOT_STATE"]
        # placeholder states (wire's default writer) are declined: the ChangeStore is what such a document is staged from
        plain = [[_fuzz.random_session(7, n_peers=2, n_steps=60, kinds=("text", "map"))[0].export_snapshot()]]
        assert c.merge_batch(plain) == _oracle.merge_batch(plain) and c.b.state_documents(c.h) == 0
        # the reference-held snapshots: three with history (the state gives what the history gives), one shallow
        B = _ref_blobs()
        names = ["snapshot.blob", "snapshot.ts.blob", "runtime-snapshot.ts.blob", "shallow.ts.blob"]
        ref = [[B[k]] for k in names]
        got = c.merge_batch(ref)
        assert c.b.state_documents(c.h) == 4
        hist = _oracle.merge_batch(ref[:3])
        assert got[:3] == hist
        assert got[3] == (0, b'{"text":"0123456789"}', wire.encode_vv({77: 10}), 0)            # loro_js_interop.rs:129-139
        root = c.merge_batch([[B["shallow.ts.blob"]]], [wire.encode_frontiers([(77, 4)])])     # loro_js_interop.rs:141-147 (shallow_since_vv[77] == 4 … frontiers 77@4)
        assert root[0] == (0, b'{"text":"01234"}', wire.encode_vv({77: 10}), 0) and c.b.state_documents(c.h) == 1
        # … any other version of a shallow snapshot needs ops replayed over a state base: reported, not guessed
        other = c.merge_batch([[B["shallow.ts.blob"]]], [wire.encode_frontiers([(77, 6)])])
        assert other[0][0] == 4 and not other[0][1]
        # generated shallow snapshots, latest version and shallow root, next to ordinary documents
        sd, sf, sw = shallow_docs()
        assert len(sd) >= 8
        mix_d, mix_f = sd + docs[:6], sf + [None] * 6
        got = c.merge_batch(mix_d, mix_f)
        assert got == sw + want[:6] and c.b.state_documents(c.h) == len(mix_d)
        # an import into a batch staged from states: the snapshots are staged once more through their ChangeStore
        reps = _fuzz.random_session(90, n_peers=2, n_steps=80, kinds=("text", "map"))
        a, b = reps[0], reps[-1]
        own_b = wire.Replica(b.peer); own_b.changes = {b.peer: b.changes.get(b.peer, [])}
        snap = real_snapshot(a)
        first = c.merge_batch([[snap], docs[0]])
        assert c.b.state_documents(c.h) == 2 and first[0] == _oracle.merge([a.export()])
        c.import_more([[own_b.export()], []])
        c.run()
        after = c.fetch()
        sess = _oracle.Session()
        sess.step([a.export()])
        assert after[0] == sess.step([own_b.export()]) and after[1] == want[0]


def test_documents_staged_from_their_snapshots_state_sections():
    check_state_path(lambda: Context(_emu.binding()), n=24)


# ---- snapshot + updates on a state base (lm_snapshot_base.h): the updates continue the snapshot's history
def state_base_docs(n=24, first=700, kinds=("text", "list", "map")):
    """(docs, want): [snapshot of the history up to a critical version (real state section), the updates beyond it — one blob per peer]"""
    docs, want = [], []
    for seed in range(first, first + n):
        rng = random.Random(seed)
        reps = _fuzz.random_session(seed, n_peers=rng.randint(2, 3), n_steps=rng.randint(30, 120), kinds=kinds, solo_steps=rng.randint(10, 60), max_ins=10)
        p0 = reps[0].peer
        heads = [d[1] for r in reps[1:] for c in r.changes.get(r.peer, [])[:1] for d in c.deps if d[0] == p0]
        if not heads:
            continue
        c_h = min(heads)
        own = reps[0].changes.get(p0, [])
        base = [c for c in own if c.ctr_end <= c_h + 1]
        if not base or base[-1].ctr_end != c_h + 1:
            continue
        early = wire.Replica(p0); early.changes = {p0: base}; early.vv = {p0: c_h + 1}; early.frontiers = [(p0, c_h)]
        snap = real_snapshot(early, **([dict(), dict(block_size=256)][seed % 2]))
        upd = []
        for r in reps:
            o = wire.Replica(r.peer); o.changes = {r.peer: [c for c in r.changes.get(r.peer, []) if r.peer != p0 or c.counter > c_h]}
            if o.changes[r.peer]:
                upd.append(o.export())
        rng.shuffle(upd)
        at = rng.randint(0, len(upd))
        docs.append(upd[:at] + [snap] + upd[at:])                      # (import_batch takes the snapshot first wherever it stands)
        want.append(_oracle.merge(docs[-1]))                           # (the checker replays the snapshot's history)
        full = _oracle.merge(_fuzz.blobs_of(reps))                     # … which is the whole session's, but for roots nothing is visible in
        assert want[-1][2] == full[2] and (want[-1][1] == full[1] or b'""' in want[-1][1] or b"[]" in want[-1][1] or b"{}" in want[-1][1])
    return docs, want


def check_state_base(ctx_factory, n=24):
    import os
    docs, want = state_base_docs(n)
    assert len(docs) >= n // 2 and all(w[0] == 0 for w in want)
    with ctx_factory() as c:
        got = c.merge_batch(docs)
        assert c.b.state_documents(c.h) == len(docs), (c.b.state_documents(c.h), len(docs))
        for i, (g, w) in enumerate(zip(got, want)):
            assert g == w, (i, g[:3], w[:3])
        os.environ["LM_SNAPSHOT_STATE"] = "0"
        try:
            assert c.merge_batch(docs) == want and c.b.state_documents(c.h) == 0
        finally:
            del os.environ["LM_SNAPSHOT_STATE"]
        # child containers of the base, edited by the updates; base text deleted by id
        a = wire.Replica(41)
        a.text_insert("t", 0, "hello world"); cm = a.map_set_container("m", "child", wire.KIND_MAP); a.map_set(cm, "x", 1)
        ct = a.map_set_container("m", "note", wire.KIND_TEXT); a.text_insert(ct, 0, "abc")
        cl = a.list_insert_container("l", 0, wire.KIND_LIST); a.list_insert(cl, 0, [1, 2, 3]); a.commit()
        base_vv = dict(a.vv)
        snap = real_snapshot(a)
        a.map_set(cm, "y", "new"); a.text_insert(ct, 3, "DEF"); a.list_insert(cl, 1, ["mid"]); a.text_delete("t", 0, 6); a.map_set("m", "top", True); a.commit()
        b = wire.Replica(42); b.merge_from(a); b.seq = {k: list(v) for k, v in a.seq.items()}
        b.text_insert(ct, 0, "<"); b.map_set(cm, "x", 2); b.commit()
        a.text_insert(ct, 0, ">"); a.map_set(cm, "x", 3); a.commit()
        ob = wire.Replica(42); ob.changes = {42: b.changes[42]}
        doc = [snap, a.export(from_vv=base_vv), ob.export()]
        full = wire.Replica(41); full.merge_from(a); full.merge_from(b)
        w = _oracle.merge(doc)
        assert w[1] == _oracle.merge([full.export()])[1]
        g = c.merge_batch([doc])
        assert g[0] == w and c.b.state_documents(c.h) == 1, (g[0][:2], w[:2])
        # updates that are concurrent with part of the snapshot's history: declined, replayed from the ChangeStore — same bytes
        reps = _fuzz.random_session(5, n_peers=3, n_steps=120, kinds=("text", "map"))
        snap2 = real_snapshot(reps[0])
        rest = _fuzz.blobs_of(reps[1:])
        conc = [[snap2] + rest]
        assert c.merge_batch(conc) == _oracle.merge_batch(conc) and c.b.state_documents(c.h) == 0
        # an import after such a batch: staged again as history (the resident documents build on it)
        c.merge_batch(docs[:4])
        assert c.b.state_documents(c.h) == 4
        c.import_more([[]] * 4); c.run()
        assert c.fetch() == want[:4] and c.b.state_documents(c.h) == 0


def test_updates_on_top_of_a_snapshots_state():
    check_state_base(lambda: Context(_emu.binding()), n=24)


def check_state_base_large(ctx_factory, n_base=3000, n_branch=1500, n=3):
    """configs[1]-shaped documents whose 50 %-base arrives as a snapshot: thousands of delete rows of the two concurrent branches name base
    content — applied by position, remembered in a per-document list sized by the op rows (Dev::posdel_off / pd_row_idx); a list that
    overflows (LM_PD_STATE_PIECES=0 forces it) sends the document to the side engine, which replays the snapshot's HISTORY"""
    import os
    from loro_amd import workload
    docs = [workload.cfg2_snapshot_doc(s, n_base=n_base, n_branch=n_branch) for s in range(n)]
    want = _oracle.merge_batch(docs)
    assert all(w[0] == 0 for w in want)
    with ctx_factory() as c:
        # the cost rule (lm_snapshot_base.h `pays`): updates that are ONE chain behind the snapshot are always replayed on its state (the
        # linear prefix, by position); concurrent branches only while they are small beside the history
        assert c.merge_batch(docs) == want
        concurrent_taken = 2 * n_branch <= 4096 or 2 * n_branch * 5 <= n_base
        assert c.b.state_documents(c.h) == (n if concurrent_taken else 0) and c.b.redo_documents(c.h) == 0
        chain = [[d[0], d[1]] for d in docs]
        assert c.merge_batch(chain) == _oracle.merge_batch(chain) and c.b.state_documents(c.h) == n
        os.environ["LM_SNAPSHOT_STATE"] = "2"      # (the by-position machinery at size, whatever the rule says)
        try:
            assert c.merge_batch(docs) == want
            assert c.b.state_documents(c.h) == n and c.b.redo_documents(c.h) == 0
            os.environ["LM_PD_STATE_PIECES"] = "0"
            try:
                assert c.merge_batch(docs) == want
                assert c.b.state_documents(c.h) == n and c.b.redo_documents(c.h) == n
            finally:
                del os.environ["LM_PD_STATE_PIECES"]
        finally:
            del os.environ["LM_SNAPSHOT_STATE"]


def check_state_base_many_documents(ctx_factory, n_base=1200, n_branch=600, n_docs=66, dir_opt_max=4):
    """a batch whose documents ALL delete base content by position: k_integrate_span_pos is launched with the optimistic directory first
    (the worst-case one leaves a few waves per CU); LM_DIR_OPT_MAX makes the documents overflow it — replayed by the worst-case pass"""
    import os
    from loro_amd import workload
    base = [workload.cfg2_snapshot_doc(s, n_base=n_base, n_branch=n_branch) for s in range(3)]
    want = _oracle.merge_batch(base)
    docs = [base[i % 3] for i in range(n_docs)]
    os.environ["LM_SNAPSHOT_STATE"] = "2"
    try:
        with ctx_factory() as c:
            got = c.merge_batch(docs)
            assert all(g == want[i % 3] for i, g in enumerate(got)) and c.b.state_documents(c.h) == n_docs and c.b.redo_documents(c.h) == 0
            os.environ["LM_DIR_OPT_MAX"] = str(dir_opt_max)
            try:
                got = c.merge_batch(docs)
                assert all(g == want[i % 3] for i, g in enumerate(got)) and c.b.state_documents(c.h) == n_docs and c.b.redo_documents(c.h) == 0
            finally:
                del os.environ["LM_DIR_OPT_MAX"]
    finally:
        del os.environ["LM_SNAPSHOT_STATE"]


def test_many_base_deletes_on_top_of_a_snapshots_state():
    check_state_base_large(lambda: Context(_emu.binding()))
    check_state_base_many_documents(lambda: Context(_emu.binding()))
