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
