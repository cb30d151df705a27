"""Synthetic workloads of BASELINE.json's configs, minted as real FastUpdates blobs (see wire.py).

The reference's B4 benchmark replays the automerge-paper editing trace
(crates/loro-internal/benches/automerge-paper.json.gz via crates/bench-utils/src/lib.rs:27-56: 259,778
single-character inserts/deletes).  That file is reference data and is not redistributed here; the
generator below reproduces its SHAPE from statistics measured on it in the build container:
  182,315 inserts / 77,463 deletes; 10,712 runs — insert runs mean 29.0 (median 9, p90 81, p99 231, max 1396),
  delete runs mean 17.5 (median 2, p90 17, p99 259, max 5894); cursor jump between runs |Δ| median 1,
  p75 48, p90 239, p99 9.4k; final length 104,852.
"""
from __future__ import annotations

import bisect
import math
import random
from typing import List, Tuple

import numpy as np

from . import wire

# quantile tables (probability, value), log-interpolated
_INS_Q = [(0.0, 1), (0.25, 3), (0.5, 9), (0.75, 33), (0.9, 81), (0.99, 231), (1.0, 1396)]
_DEL_Q = [(0.0, 1), (0.5, 2), (0.75, 6), (0.9, 17), (0.99, 259), (0.999, 1500), (1.0, 5894)]
_JUMP_Q = [(0.0, 0), (0.25, 0), (0.5, 1), (0.75, 48), (0.9, 239), (0.99, 9409), (1.0, 40000)]


def _sample(q, u):
    ps = [p for p, _ in q]
    i = min(max(bisect.bisect_right(ps, u) - 1, 0), len(q) - 2)
    (p0, v0), (p1, v1) = q[i], q[i + 1]
    f = (u - p0) / (p1 - p0) if p1 > p0 else 0.0
    a, b = math.log(v0 + 1), math.log(v1 + 1)
    return int(round(math.exp(a + f * (b - a)) - 1))


def synthetic_trace(n_actions: int, seed: int = 0) -> List[Tuple[int, int, str]]:
    """List of (pos, delete_count∈{0,1}, insert_char or '') single-character actions, automerge-trace shaped."""
    rng = random.Random(seed)
    acts: List[Tuple[int, int, str]] = []
    length, cursor = 0, 0
    letters = "etaoinshrdlucmfwypvbgkqjxz      \n"
    while len(acts) < n_actions:
        want_del = rng.random() < 0.413 and length > 8
        # move the cursor
        j = _sample(_JUMP_Q, rng.random())
        cursor += j if rng.random() < 0.5 else -j
        cursor = max(0, min(length, cursor))
        if want_del:
            n = max(1, min(_sample(_DEL_Q, rng.random()), length))
            if rng.random() < 0.7:  # backspace run
                cursor = max(cursor, 1)
                n = min(n, cursor)
                for _ in range(n):
                    cursor -= 1
                    acts.append((cursor, 1, ""))
                    length -= 1
                    if len(acts) >= n_actions:
                        break
            else:  # forward delete run
                cursor = min(cursor, length - 1)
                n = min(n, length - cursor)
                for _ in range(n):
                    acts.append((cursor, 1, ""))
                    length -= 1
                    if len(acts) >= n_actions:
                        break
        else:
            n = max(1, _sample(_INS_Q, rng.random()))
            for _ in range(n):
                acts.append((cursor, 0, rng.choice(letters)))
                cursor += 1
                length += 1
                if len(acts) >= n_actions:
                    break
    return acts[:n_actions]


def _apply(rep: wire.Replica, acts, commit_every: int, name="text"):
    k = 0
    for (pos, dl, ch) in acts:
        if dl:
            rep.text_delete(name, pos, 1)
        else:
            rep.text_insert(name, pos, ch)
        k += 1
        if commit_every and k % commit_every == 0:
            rep.commit()
    rep.commit()


def fuse_changes(changes: List[wire.Change], max_bytes: int = 4096) -> List[wire.Change]:
    """Default-config Loro fuses a peer's consecutive self-dependent commits into one change until the
    ≈4 KiB block is full (change.rs:268-282, change_store.rs:1007-1017,1666-1683).  Ops keep RLE-merging."""
    out: List[wire.Change] = []
    size = 0
    for c in changes:
        est = sum(len(o.text.encode()) if o.kind == "text_insert" else 8 for o in c.ops)
        self_dep = len(c.deps) == 1 and c.deps[0] == (c.peer, c.counter - 1)
        if out and self_dep and out[-1].peer == c.peer and out[-1].ctr_end == c.counter and size + est <= max_bytes:
            tgt = out[-1]
            for o in c.ops:
                if not (tgt.ops and wire.try_merge(tgt.ops[-1], o)):
                    tgt.ops.append(o)
            tgt._len = None
            size += est
        else:
            out.append(wire.Change(c.peer, c.counter, c.lamport, list(c.deps), [wire.Op(**o.__dict__) for o in c.ops]))
            size = est
    return out


class Cfg2Template:
    """Config 2 of BASELINE.json: base [0,n_base) by peer A; fork; A and B each apply the same
    [n_base, n_base+n_branch) actions concurrently (B's inserted letters are replaced per document).
    Three blobs per document: base, A's branch, B's branch.  One structural template is built once and
    stamped per document (peer ids, B's letters, checksum), which is what makes 10k documents affordable."""

    def __init__(self, n_base=50000, n_branch=25000, seed=0, commit_every=0, fuse=True):
        acts = synthetic_trace(n_base + n_branch, seed)
        self.peer_a, self.peer_b = 0x1111111111111111, 0x2222222222222222
        a = wire.Replica(self.peer_a)
        _apply(a, acts[:n_base], commit_every)
        n_base_changes = len(a.changes[self.peer_a])
        b = wire.Replica(self.peer_b)
        b.merge_from(a)
        b.seq = {k: list(v) for k, v in a.seq.items()}
        branch = acts[n_base:]
        _apply(a, branch, commit_every)
        # B: same positions, inserted letters drawn from a placeholder alphabet so they can be located and re-stamped
        self._b_placeholder = "#"
        _apply(b, [(p, dl, (self._b_placeholder if ch else "")) for (p, dl, ch) in branch], commit_every)
        ch_a = a.changes[self.peer_a]
        base_ch, a_ch, b_ch = ch_a[:n_base_changes], ch_a[n_base_changes:], b.changes[self.peer_b]
        if fuse:
            base_ch, a_ch, b_ch = fuse_changes(base_ch), fuse_changes(a_ch), fuse_changes(b_ch)

        def enc(chs):
            return wire.encode_updates([[c] for c in chs] if fuse else wire.split_blocks(chs))

        self.blobs = [enc(base_ch), enc(a_ch), enc(b_ch)]
        # locate B's inserted letters: re-encode with another placeholder and diff (same lengths everywhere)
        for c in b_ch:
            for o in c.ops:
                if o.kind == "text_insert":
                    o.text = o.text.replace(self._b_placeholder, "$")
        alt = np.frombuffer(enc(b_ch), dtype=np.uint8)
        self.n_ops = n_base + 2 * n_branch
        self.n_runs = sum(len(c.ops) for chs in (base_ch, a_ch, b_ch) for c in chs)
        self.n_changes = sum(len(chs) for chs in (base_ch, a_ch, b_ch))
        # stamp sites
        self._arr = [np.frombuffer(bl, dtype=np.uint8).copy() for bl in self.blobs]
        pa = np.frombuffer(self.peer_a.to_bytes(8, "little"), dtype=np.uint8)
        pb = np.frombuffer(self.peer_b.to_bytes(8, "little"), dtype=np.uint8)
        self._sites_a = [self._find(arr, pa) for arr in self._arr]
        self._sites_b = [self._find(arr, pb) for arr in self._arr]
        assert len(alt) == len(self._arr[2])
        diff = np.nonzero(alt != self._arr[2])[0]
        self._ph = diff[diff >= 22]
        n_b_ins = sum(1 for (_, dl, ch) in branch if ch)
        assert len(self._ph) == n_b_ins, (len(self._ph), n_b_ins)

    @staticmethod
    def _find(arr, pat):
        n = len(pat)
        idx = np.nonzero(arr[: len(arr) - n + 1] == pat[0])[0]
        ok = [i for i in idx if np.array_equal(arr[i:i + n], pat)]
        return np.array(ok, dtype=np.int64)

    def stamp(self, doc_index: int) -> List[bytes]:
        """Blobs of document `doc_index`: peers 2d+1 / 2d+2 (scaled into 53 bits), B's letters seeded by d."""
        pa = (doc_index * 2 + 1) * 0x9E3779B1 % (1 << 53) | 1
        pb = pa + 1
        rng = np.random.default_rng(doc_index)
        letters = rng.integers(ord("A"), ord("Z") + 1, size=len(self._ph), dtype=np.uint8)
        out = []
        for bi, arr in enumerate(self._arr):
            x = arr.copy()
            ba = np.frombuffer(int(pa).to_bytes(8, "little"), dtype=np.uint8)
            bb = np.frombuffer(int(pb).to_bytes(8, "little"), dtype=np.uint8)
            for s in self._sites_a[bi]:
                x[s:s + 8] = ba
            for s in self._sites_b[bi]:
                x[s:s + 8] = bb
            if bi == 2:
                x[self._ph] = letters
            body = x[20:].tobytes()
            x[16:20] = np.frombuffer(wire.xxh32(body).to_bytes(4, "little"), dtype=np.uint8)
            out.append(x.tobytes())
        return out


def cfg1_doc(d: int, n=1000, per_change=10) -> List[bytes]:
    """Config 1: two peers each type n characters at the end of their own replica (text_r.rs:87-101 shape)."""
    blobs = []
    for k in (1, 2):
        r = wire.Replica(2 * d + k)
        for i in range(n):
            r.text_insert("text", i, chr(97 + (i * k + d) % 26))
            if (i + 1) % per_change == 0:
                r.commit()
        r.commit()
        blobs.append(r.export())
    return blobs


def _xorshift64(x):
    x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 7
    x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
    return x & 0xFFFFFFFFFFFFFFFF


def cfg3_doc(d: int, n_peers=16, n_writes=10000, n_keys=1024, combined=True, per_change=100) -> List[bytes]:
    """Config 3: root Map, fully concurrent peers, set(key_k, i64) with k = xorshift64 % n_keys."""
    reps = []
    for p in range(n_peers):
        r = wire.Replica(d * 1000 + p + 1)
        x = (d * 7919 + p * 104729 + 1) & 0xFFFFFFFFFFFFFFFF
        for i in range(n_writes):
            x = _xorshift64(x)
            r.map_set("map", "key_%d" % (x % n_keys), int(x >> 20) - (1 << 42))
            if (i + 1) % per_change == 0:
                r.commit()
        r.commit()
        reps.append(r)
    if combined:
        all_ = wire.Replica(0)
        for r in reps:
            all_.changes[r.peer] = r.changes[r.peer]
        return [all_.export()]
    return [r.export() for r in reps]


def cfg5_doc(d: int, n_ops=20000, turn=1000, mark_prob=0.01, n_checkouts=16, commit_every=10):
    """Config 5: one Text edited by two peers that alternate every `turn` trace actions (a linear
    hand-over history like automerge_x100.rs:16-24), ≈1 % of the actions replaced by a bold mark (StyleStart/StyleEnd
    anchors), rendered at `n_checkouts` random versions.  Returns (blobs, [encoded Frontiers] * n_checkouts)."""
    rng = random.Random(0xC5 + d)
    acts = synthetic_trace(n_ops, seed=d)
    peers = [wire.Replica(2 * d + 1), wire.Replica(2 * d + 2)]
    cid = wire.root_cid("text", wire.KIND_TEXT)
    cur = 0
    for t0 in range(0, n_ops, turn):
        r = peers[cur]
        ids = r.seq.setdefault(cid, [])
        for k, (pos, dl, ch) in enumerate(acts[t0:t0 + turn]):
            if len(ids) >= 2 and rng.random() < mark_prob:
                s = rng.randrange(len(ids) - 1)
                r.text_mark("text", s, rng.randrange(s + 1, min(len(ids), s + 40)), "bold", True)
            elif dl:
                r.text_delete("text", min(pos, len(ids) - 1), 1)
            else:
                r.text_insert("text", min(pos, len(ids)), ch)
            if commit_every and (k + 1) % commit_every == 0:
                r.commit()
        r.commit()
        o = peers[1 - cur]                     # hand the whole history over (nothing concurrent on the other side)
        o.changes = {p: list(v) for p, v in r.changes.items()}
        o.vv = dict(r.vv)
        o.frontiers = list(r.frontiers)
        o.seq = {k: list(v) for k, v in r.seq.items()}
        cur = 1 - cur
    last = peers[1 - cur]
    blob = last.export()
    fronts = []
    for _ in range(n_checkouts):
        p = rng.choice([q for q in last.vv if last.vv[q] > 0])
        fronts.append(wire.encode_frontiers([(p, rng.randrange(last.vv[p]))]))
    return [blob], fronts


def cfg2_snapshot_doc(seed: int, n_base=50000, n_branch=25000, commit_every=10):
    """a configs[1]-shaped history (base by peer A; A and B continue concurrently) delivered as
    [snapshot of the base (real state section: tests/_oracle.state_entries — the checker's state writer), A's branch, B's branch]."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import _oracle
    acts = synthetic_trace(n_base + n_branch, 1000 + seed)
    pa, pb = 0x3111111111111 + 2 * seed, 0x3111111111112 + 2 * seed
    a = wire.Replica(pa)
    _apply(a, acts[:n_base], commit_every)
    base_vv = dict(a.vv)
    st, ents = _oracle.state_entries([a.export()])
    assert st == 0
    snap = a.export_snapshot(state=ents, compress=False)
    b = wire.Replica(pb)
    b.merge_from(a)
    b.seq = {k: list(v) for k, v in a.seq.items()}
    branch = acts[n_base:]
    _apply(a, branch, commit_every)
    _apply(b, [(p, dl, ("Z" if ch else "")) for (p, dl, ch) in branch], commit_every)
    ob = wire.Replica(pb); ob.changes = {pb: b.changes[pb]}
    return [snap, a.export(from_vv=base_vv), ob.export()]
