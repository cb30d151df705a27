"""FastUpdates (EncodeMode 4) writer + a minimal local-editing replica.

Used by the workload generators (bench.py, tests/) to mint real Loro update blobs: the same bytes
`LoroDoc::export(ExportMode::Updates)` produces and `import()` consumes.  The layout follows the
reference writer (paths relative to /root/reference):
  envelope            crates/loro-internal/src/encoding.rs:440-473
  updates framing     crates/loro-internal/src/encoding/fast_snapshot.rs:346-360
  block struct        crates/loro-internal/src/oplog/change_store/block_encode.rs:94-119,137-278
  header/meta         crates/loro-internal/src/oplog/change_store/block_meta_encode.rs:13-88
  op rows / values    crates/loro-internal/src/encoding/outdated_encode_reordered.rs:101-215
  column strategies   docs/encoding.md §8.1 (serde_columnar 0.3.14)
  local text delete   crates/loro-internal/src/handler.rs:2245-2288 (id-contiguous ranges, right to left)
  op RLE merge rules  crates/loro-internal/src/container/list/list_op.rs:189-249,381-424,516-589
Any valid encoding decodes to the same changes (docs/encoding.md §8); byte-exactness with the Rust
writer's segmentation is not claimed here.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

try:  # fast path for the envelope checksum
    import xxhash as _xxhash
except Exception:  # pragma: no cover
    _xxhash = None

XXH_SEED = 0x4F524F4C  # LE("LORO")
KIND_MAP, KIND_LIST, KIND_TEXT, KIND_TREE, KIND_MOVABLE, KIND_COUNTER = 0, 1, 2, 3, 4, 5


# ---------------------------------------------------------------- primitives
def uleb(n: int) -> bytes:
    assert n >= 0
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def zigzag(n: int) -> bytes:
    return uleb((n << 1) if n >= 0 else ((-n << 1) - 1))


def sleb(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if (n == 0 and not (b & 0x40)) or (n == -1 and (b & 0x40)):
            out.append(b)
            return bytes(out)
        out.append(b | 0x80)


def lbytes(b: bytes) -> bytes:
    return uleb(len(b)) + b


def xxh32(data: bytes, seed: int = XXH_SEED) -> int:
    if _xxhash is not None:
        return _xxhash.xxh32(data, seed=seed).intdigest()
    P1, P2, P3, P4, P5 = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D, 0x27D4EB2F, 0x165667B1
    M = 0xFFFFFFFF
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
    n, p = len(data), 0
    if n >= 16:
        v = [(seed + P1 + P2) & M, (seed + P2) & M, seed & M, (seed - P1) & M]
        while p + 16 <= n:
            for i in range(4):
                x = struct.unpack_from("<I", data, p)[0]
                v[i] = (rotl((v[i] + x * P2) & M, 13) * P1) & M
                p += 4
        h = (rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18)) & M
    else:
        h = (seed + P5) & M
    h = (h + n) & M
    while p + 4 <= n:
        x = struct.unpack_from("<I", data, p)[0]
        h = (rotl((h + x * P3) & M, 17) * P4) & M
        p += 4
    while p < n:
        h = (rotl((h + data[p] * P5) & M, 11) * P1) & M
        p += 1
    h ^= h >> 15
    h = (h * P2) & M
    h ^= h >> 13
    h = (h * P3) & M
    h ^= h >> 16
    return h


# ---------------------------------------------------------------- column strategies
def enc_bool_rle(vals) -> bytes:
    out = bytearray()
    cur, run, any_ = False, 0, False
    for v in vals:
        any_ = True
        v = bool(v)
        if v == cur:
            run += 1
        else:
            out += uleb(run)
            cur, run = v, 1
    if any_:
        out += uleb(run)
    return bytes(out)


def _any_rle(vals, wv) -> bytes:
    """runs of >=2 equal values as (k>0, value); everything else as literal groups (k<0)."""
    out = bytearray()
    i, n = 0, len(vals)
    lit: List[Any] = []

    def flush_lit():
        if lit:
            out.extend(zigzag(-len(lit)))
            for x in lit:
                out.extend(wv(x))
            lit.clear()

    while i < n:
        j = i
        while j + 1 < n and vals[j + 1] == vals[i]:
            j += 1
        run = j - i + 1
        if run >= 2:
            flush_lit()
            out.extend(zigzag(run))
            out.extend(wv(vals[i]))
        else:
            lit.append(vals[i])
        i = j + 1
    flush_lit()
    return bytes(out)


def enc_any_rle_uvar(vals) -> bytes:
    return _any_rle(list(vals), uleb)


def enc_rle_u8(vals) -> bytes:
    return _any_rle(list(vals), lambda x: bytes([x]))


def enc_delta_rle(vals) -> bytes:
    deltas, prev = [], 0
    for v in vals:
        deltas.append(v - prev)
        prev = v
    return _any_rle(deltas, zigzag)  # i128 zigzag varint == zigzag for in-range values


class _BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.cur = 0
        self.used = 0

    def write(self, value: int, count: int):
        for shift in range(count - 1, -1, -1):
            self.cur = (self.cur << 1) | ((value >> shift) & 1)
            self.used += 1
            if self.used == 8:
                self.out.append(self.cur)
                self.cur, self.used = 0, 0

    def finish(self) -> Tuple[bytes, int]:
        if self.used == 0:
            return bytes(self.out), 8
        self.out.append((self.cur << (8 - self.used)) & 0xFF)
        return bytes(self.out), self.used


def enc_delta_of_delta(vals) -> bytes:
    vals = list(vals)
    if not vals:
        return b"\x00\x00"
    head = b"\x01" + zigzag(vals[0])
    if len(vals) == 1:
        return head + b"\x00"
    bw = _BitWriter()
    prev_delta = 0
    for i in range(1, len(vals)):
        delta = vals[i] - vals[i - 1]
        dod = delta - prev_delta
        prev_delta = delta
        if dod == 0:
            bw.write(0, 1)
        elif -63 <= dod <= 64:
            bw.write(0b10, 2)
            bw.write(dod + 63, 7)
        elif -255 <= dod <= 256:
            bw.write(0b110, 3)
            bw.write(dod + 255, 9)
        elif -2047 <= dod <= 2048:
            bw.write(0b1110, 4)
            bw.write(dod + 2047, 12)
        elif -1048575 <= dod <= 1048576:
            bw.write(0b11110, 5)
            bw.write(dod + 1048575, 21)
        else:
            bw.write(0b11111, 5)
            bw.write(dod & ((1 << 64) - 1), 64)
    bits, used = bw.finish()
    return head + bytes([used]) + bits


# ---------------------------------------------------------------- model
@dataclass(frozen=True)
class CID:
    root: bool
    kind: int
    name: str = ""
    peer: int = 0
    counter: int = 0


def root_cid(name: str, kind: int) -> CID:
    return CID(True, kind, name)


@dataclass(frozen=True)
class ContainerValue:
    """LoroValue::Container — only the kind goes on the wire (docs/encoding.md §10.1)."""
    kind: int


@dataclass
class Op:
    cid: CID
    counter: int
    kind: str  # text_insert | delete | style_start | style_end | list_insert | map_set | map_delete | list_move | list_set
    pos: int = 0
    text: str = ""  # text_insert
    values: Optional[list] = None  # list_insert
    key: str = ""  # map / style key
    value: Any = None  # map_set / style value
    del_id: Tuple[int, int] = (0, 0)  # leftmost target id
    signed_len: int = 0  # delete
    mark_len: int = 0  # style_start
    mark_info: int = 0x80
    elem: Tuple[int, int] = (0, 0)  # list_move / list_set: the MovableList element (peer, lamport)
    move_from: int = 0  # list_move: source position (pos = destination)

    @property
    def atom_len(self) -> int:
        if self.kind == "text_insert":
            return len(self.text)
        if self.kind == "list_insert":
            return len(self.values)
        if self.kind == "delete":
            return abs(self.signed_len)
        return 1


@dataclass
class Change:
    peer: int
    counter: int
    lamport: int
    deps: List[Tuple[int, int]]
    ops: List[Op] = field(default_factory=list)
    timestamp: int = 0
    msg: Optional[str] = None
    _len: Optional[int] = None
    _len_n: int = -1

    @property
    def atom_len(self) -> int:
        if self._len is None or self._len_n != len(self.ops):
            self._len = sum(o.atom_len for o in self.ops)
            self._len_n = len(self.ops)
        return self._len

    @property
    def ctr_end(self) -> int:
        return self.counter + self.atom_len


class _Registry:
    def __init__(self):
        self.items: list = []
        self.idx: dict = {}

    def get(self, x) -> int:
        i = self.idx.get(x)
        if i is None:
            i = len(self.items)
            self.items.append(x)
            self.idx[x] = i
        return i


def _enc_value(v, keys: _Registry) -> bytes:
    """nested LoroValue: raw tag + payload (docs/encoding.md §10.1)"""
    if v is None:
        return b"\x00"
    if v is True:
        return b"\x01"
    if v is False:
        return b"\x02"
    if isinstance(v, int):
        return b"\x03" + sleb(v)
    if isinstance(v, float):
        return b"\x04" + struct.pack(">d", v)
    if isinstance(v, str):
        return b"\x05" + lbytes(v.encode("utf-8"))
    if isinstance(v, (bytes, bytearray)):
        return b"\x06" + lbytes(bytes(v))
    if isinstance(v, (list, tuple)):
        return b"\x07" + uleb(len(v)) + b"".join(_enc_value(x, keys) for x in v)
    if isinstance(v, dict):
        out = bytearray(b"\x08" + uleb(len(v)))
        for k, x in v.items():
            out += uleb(keys.get(k)) + _enc_value(x, keys)
        return bytes(out)
    if isinstance(v, ContainerValue):
        return b"\x09" + bytes([v.kind])
    raise TypeError(f"unsupported value {type(v)}")


def encode_block(changes: List[Change]) -> bytes:
    """One change block: changes of ONE peer, counter-contiguous (block_encode.rs:137-278)."""
    assert changes
    peer = changes[0].peer
    peers, keys, cids = _Registry(), _Registry(), _Registry()
    peers.get(peer)
    n = len(changes)
    # header columns
    lens = [c.atom_len for c in changes]
    dep_self, dep_counts, dep_peer_idx, dep_counters = [], [], [], []
    for c in changes:
        ds, others = False, []
        for (p, ctr) in c.deps:
            if p == peer and ctr == c.counter - 1:
                ds = True
            else:
                others.append((p, ctr))
        dep_self.append(ds)
        dep_counts.append(len(others))
        for (p, ctr) in others:
            dep_peer_idx.append(peers.get(p))
            dep_counters.append(ctr)
    # ops
    col_c, col_prop, col_vt, col_len = [], [], [], []
    del_peer, del_ctr, del_len = [], [], []
    values = bytearray()

    def cid_idx(cid: CID) -> int:
        if cid in cids.idx:
            return cids.idx[cid]
        if cid.root:
            keys.get(cid.name)
        else:
            peers.get(cid.peer)
        return cids.get(cid)

    for c in changes:
        ctr = c.counter
        for op in c.ops:
            assert op.counter == ctr, (op.counter, ctr)
            col_c.append(cid_idx(op.cid))
            if op.kind == "text_insert":
                col_prop.append(op.pos)
                col_vt.append(5)
                values += lbytes(op.text.encode("utf-8"))
            elif op.kind == "delete":
                col_prop.append(op.pos)
                col_vt.append(9)
                del_peer.append(peers.get(op.del_id[0]))
                del_ctr.append(op.del_id[1])
                del_len.append(op.signed_len)
            elif op.kind == "style_start":
                col_prop.append(op.pos)
                col_vt.append(12)
                values += bytes([op.mark_info]) + uleb(op.mark_len) + uleb(keys.get(op.key)) + _enc_value(op.value, keys)
            elif op.kind == "style_end":
                col_prop.append(0)
                col_vt.append(0)
            elif op.kind == "list_insert":
                col_prop.append(op.pos)
                col_vt.append(11)
                values += _enc_value(list(op.values), keys)
            elif op.kind == "map_set":
                col_prop.append(keys.get(op.key))
                col_vt.append(11)
                values += _enc_value(op.value, keys)
            elif op.kind == "map_delete":
                col_prop.append(keys.get(op.key))
                col_vt.append(8)
            elif op.kind == "list_move":   # docs/encoding.md §10.5
                col_prop.append(op.pos)
                col_vt.append(14)
                values += uleb(op.move_from) + uleb(peers.get(op.elem[0])) + uleb(op.elem[1])
            elif op.kind == "list_set":
                col_prop.append(0)
                col_vt.append(15)
                values += uleb(peers.get(op.elem[0])) + uleb(op.elem[1]) + _enc_value(op.value, keys)
            else:
                raise ValueError(op.kind)
            col_len.append(op.atom_len)
            ctr += op.atom_len
    header = bytearray()
    header += uleb(len(peers.items))
    for p in peers.items:
        header += struct.pack("<Q", p)
    for l in lens[:-1]:
        header += uleb(l)
    header += enc_bool_rle(dep_self)
    header += enc_any_rle_uvar(dep_counts)
    header += enc_any_rle_uvar(dep_peer_idx)
    header += enc_delta_of_delta(dep_counters)
    header += enc_delta_of_delta([c.lamport for c in changes[:-1]])
    meta = bytearray()
    meta += enc_delta_of_delta([c.timestamp for c in changes])
    msgs = [(c.msg or "").encode("utf-8") for c in changes]
    meta += enc_any_rle_uvar([len(m) for m in msgs])
    meta += b"".join(msgs)
    cids_b = bytearray(uleb(len(cids.items)))
    for cid in cids.items:
        cids_b += uleb(4) + (b"\x01" if cid.root else b"\x00") + bytes([cid.kind])
        if cid.root:
            cids_b += uleb(0) + zigzag(keys.idx[cid.name])
        else:
            cids_b += uleb(peers.idx[cid.peer]) + zigzag(cid.counter)
    keys_b = b"".join(lbytes(k.encode("utf-8")) for k in keys.items)
    ops_b = uleb(1) + uleb(4) + lbytes(enc_delta_rle(col_c)) + lbytes(enc_delta_rle(col_prop)) + lbytes(
        enc_rle_u8(col_vt)) + lbytes(enc_any_rle_uvar(col_len))
    if del_peer:
        del_b = uleb(1) + uleb(3) + lbytes(enc_delta_rle(del_peer)) + lbytes(enc_delta_rle(del_ctr)) + lbytes(
            enc_delta_rle(del_len))
    else:
        del_b = b""
    first, last = changes[0], changes[-1]
    counter_len = last.ctr_end - first.counter
    lamport_len = last.lamport + last.atom_len - first.lamport
    assert sum(lens) == counter_len
    out = bytearray()
    out += uleb(first.counter) + uleb(counter_len) + uleb(first.lamport) + uleb(lamport_len) + uleb(n)
    for sec in (bytes(header), bytes(meta), bytes(cids_b), keys_b, b"", ops_b, del_b, bytes(values)):
        out += lbytes(sec)
    return bytes(out)


def envelope(body: bytes, mode: int = 4) -> bytes:
    tail = struct.pack(">H", mode) + body
    return b"loro" + b"\x00" * 12 + struct.pack("<I", xxh32(tail)) + tail


def encode_updates(blocks: List[List[Change]]) -> bytes:
    body = bytearray()
    for blk in blocks:
        b = encode_block(blk)
        body += uleb(len(b)) + b
    return envelope(bytes(body))


def split_blocks(changes: List[Change], max_block: int = 4096) -> List[List[Change]]:
    """Group one peer's contiguous changes into blocks of ≈max_block estimated bytes
    (change_store.rs:42 MAX_BLOCK_SIZE; estimate_storage_size list_op.rs:109-123)."""
    blocks, cur, size = [], [], 0
    for c in changes:
        est = 0
        for o in c.ops:
            if o.kind == "text_insert":
                est += len(o.text.encode("utf-8"))
            elif o.kind == "list_insert":
                est += 4 * len(o.values)
            elif o.kind == "delete":
                est += 8
            else:
                est += 8
        if cur and size + est > max_block:
            blocks.append(cur)
            cur, size = [], 0
        cur.append(c)
        size += est
    if cur:
        blocks.append(cur)
    return blocks


# ---------------------------------------------------------------- FastSnapshot (mode 3) writer — test inputs for the snapshot ingest
# (docs/encoding.md §3-5, §11; crates/kv-store/src/{sstable,block,compress}.rs; docs/encoding-lz4.md)
def lz4_block(data: bytes) -> bytes:
    """one raw LZ4 block, greedy single-probe matcher (any valid LZ4 stream will do: the readers are what is tested)"""
    n, out, i, anchor, table = len(data), bytearray(), 0, 0, {}

    def emit(lit: bytes, mlen: int, off: int):
        ll = len(lit)
        tok_l = 15 if ll >= 15 else ll
        tok_m = 0 if mlen == 0 else (15 if mlen - 4 >= 15 else mlen - 4)
        out.append((tok_l << 4) | tok_m)
        if ll >= 15:
            r = ll - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(lit)
        if mlen:
            out.extend(struct.pack("<H", off))
            if mlen - 4 >= 15:
                r = mlen - 4 - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)

    while i + 4 <= n - 5:   # the last five bytes are always literals (end-of-block rule)
        key = data[i:i + 4]
        j = table.get(key)
        table[key] = i
        if j is not None and i - j <= 0xFFFF:
            m = 4
            while i + m < n - 5 and data[j + m] == data[i + m]:
                m += 1
            emit(data[anchor:i], m, i - j)
            i += m
            anchor = i
        else:
            i += 1
    emit(data[anchor:], 0, 0)
    return bytes(out)


def lz4_frame(data: bytes) -> bytes:
    """LZ4 frame, version 01, independent blocks, no checksums, 4 MiB block size (docs/encoding-lz4.md §3)"""
    flg, bd = 0x60, 0x70
    hc = (xxh32(bytes([flg, bd]), 0) >> 8) & 0xFF
    out = bytearray(struct.pack("<I", 0x184D2204) + bytes([flg, bd, hc]))
    for o in range(0, len(data), 4 << 20):
        chunk = data[o:o + (4 << 20)]
        c = lz4_block(chunk)
        if len(c) < len(chunk):
            out += struct.pack("<I", len(c)) + c
        else:
            out += struct.pack("<I", len(chunk) | 0x80000000) + chunk
    out += struct.pack("<I", 0)
    return bytes(out)


def sstable(kvs, block_size: int = 4096, compress: bool = True) -> bytes:
    """loro-kv-store SSTable of the (key, value) pairs (docs/encoding.md §4): normal blocks with prefix-compressed keys, a
    large-value block for a value beyond the block size that opens a block, per-block and metadata xxh32, LZ4 where it is smaller"""
    kvs = sorted(kvs)
    if not kvs:
        return b""
    out = bytearray(b"LORO\x00")
    metas = []

    def store(body: bytes):
        comp = 0
        if compress:
            f = lz4_frame(body)
            if len(f) <= len(body):
                body, comp = f, 1
        out.extend(body + struct.pack("<I", xxh32(bytes(body))))
        return comp

    i = 0
    while i < len(kvs):
        k0, v0 = kvs[i]
        off = len(out)
        if len(v0) > block_size or len(v0) > 0xFFFF:
            comp = store(v0)
            metas.append((off, k0, 0x80 | comp, None))
            i += 1
            continue
        data, offs, last = bytearray(v0), [0], k0
        i += 1
        while i < len(kvs):
            k, v = kvs[i]
            if len(v) > block_size or len(data) + len(v) + len(k) > block_size:
                break
            pre = 0
            while pre < min(len(k0), len(k), 255) and k0[pre] == k[pre]:
                pre += 1
            offs.append(len(data))
            data += bytes([pre]) + struct.pack("<H", len(k) - pre) + k[pre:] + v
            last = k
            i += 1
        body = bytes(data) + b"".join(struct.pack("<H", o) for o in offs) + struct.pack("<H", len(offs))
        comp = store(body)
        metas.append((off, k0, comp, last))
    m_off = len(out)
    ent = bytearray()
    for off, fk, flags, lk in metas:
        ent += struct.pack("<I", off) + struct.pack("<H", len(fk)) + fk + bytes([flags])
        if lk is not None:
            ent += struct.pack("<H", len(lk)) + lk
    out += struct.pack("<I", len(metas)) + ent + struct.pack("<I", xxh32(bytes(ent)))
    out += struct.pack("<I", m_off)
    return bytes(out)


def encode_snapshot(blocks: List[List["Change"]], roots, vv: Dict[int, int], frontiers, block_size: int = 4096, compress: bool = True,
                    shallow_root_state: bytes = b"", state=None) -> bytes:
    """A FastSnapshot: ChangeStore SSTable (12-byte ID keys -> change blocks, `vv`, `fr`), state SSTable holding the root
    containers `roots` = [(kind, name)] with placeholder values (a reader of state VALUES declines them and takes the history), or —
    `state` = [(key, value)] in key order, e.g. tests/_oracle.state_entries — the entries of a real state section; empty shallow-root section."""
    oplog = [(b"vv", encode_vv(vv)), (b"fr", encode_frontiers(frontiers))]
    for blk in blocks:
        oplog.append((struct.pack(">Qi", blk[0].peer, blk[0].counter), encode_block(blk)))
    if state is None:
        state = [(bytes([0x80 | kind]) + uleb(len(name.encode())) + name.encode(), b"\x00") for kind, name in roots]
    o, st = sstable(oplog, block_size, compress), sstable(state, block_size, compress)
    body = struct.pack("<I", len(o)) + o + struct.pack("<I", len(st)) + st + struct.pack("<I", len(shallow_root_state)) + shallow_root_state
    return envelope(body, mode=3)


def encode_vv(vv: Dict[int, int]) -> bytes:
    """postcard map with entries sorted by peer (the canonical order of the C ABI outputs)."""
    out = bytearray(uleb(len(vv)))
    for p in sorted(vv):
        out += uleb(p) + zigzag(vv[p])
    return bytes(out)


def encode_frontiers(ids) -> bytes:
    """Frontiers::encode (version/frontiers.rs:219-223): postcard Vec<ID> sorted by (peer, counter);
    ID = varint u64 peer, zigzag varint i32 counter.  `ids` = iterable of (peer, counter); [] = the empty version."""
    ids = sorted(ids)
    out = bytearray(uleb(len(ids)))
    for p, c in ids:
        out += uleb(p) + zigzag(c)
    return bytes(out)


# ---------------------------------------------------------------- op RLE merge (RleVec push)
def _del_start(o: Op) -> int:
    return o.pos if o.signed_len > 0 else o.pos + 1 + o.signed_len


def _del_next_pos(o: Op) -> int:  # DeleteSpan::next_pos
    s = _del_start(o)
    return s if o.signed_len > 0 else s - 1


def _del_prev_pos(o: Op) -> int:  # DeleteSpan::prev_pos
    if o.signed_len > 0:
        return o.pos
    return o.pos + 1  # end()


def try_merge(a: Op, b: Op) -> bool:
    """Merge b into a if the reference's RleVec would (same container, counter-contiguous)."""
    if a.cid != b.cid or a.kind != b.kind or a.counter + a.atom_len != b.counter:
        return False
    if a.kind == "text_insert":
        if a.pos + len(a.text) == b.pos:
            a.text += b.text
            return True
        return False
    if a.kind == "list_insert":
        if a.pos + len(a.values) == b.pos:
            a.values = list(a.values) + list(b.values)
            return True
        return False
    if a.kind == "delete":
        abi, bbi = abs(a.signed_len) == 1, abs(b.signed_len) == 1
        a_id_end = (a.del_id[0], a.del_id[1] + abs(a.signed_len))
        b_id_end = (b.del_id[0], b.del_id[1] + abs(b.signed_len))
        inc = lambda i: (i[0], i[1] + 1)
        ok = False
        if abi and bbi:
            ok = (a.pos == b.pos and inc(a.del_id) == b.del_id) or (a.pos == b.pos + 1 and a.del_id == inc(b.del_id))
        elif abi and not bbi:
            if a.pos == _del_prev_pos(b):
                ok = (inc(a.del_id) == b.del_id) if b.signed_len > 0 else (a.del_id == b_id_end)
        elif not abi and bbi:
            if _del_next_pos(a) == b.pos:
                ok = (a_id_end == b.del_id) if a.signed_len > 0 else (a.del_id == inc(b.del_id))
        else:
            da = 1 if a.signed_len > 0 else -1
            db = 1 if b.signed_len > 0 else -1
            if _del_next_pos(a) == b.pos and da == db:
                ok = (a_id_end == b.del_id) if a.signed_len > 0 else (a.del_id == b_id_end)
        if not ok:
            return False
        # DeleteSpan::merge (list_op.rs:396-423)
        if abi and bbi:
            a.signed_len = 2 if a.pos == b.pos else -2
        elif abi and not bbi:
            a.signed_len = b.signed_len + (1 if b.signed_len > 0 else -1)
        elif not abi and bbi:
            a.signed_len += 1 if a.signed_len > 0 else -1
        else:
            a.signed_len += b.signed_len
        a.del_id = (a.del_id[0], min(a.del_id[1], b.del_id[1]))
        return True
    return False


# ---------------------------------------------------------------- replica (local editing, linear)
class Replica:
    """A peer that edits Text/List/Map root containers locally and exchanges changes.

    Local state per sequence container is just the visible element ids; after `merge_from` the
    caller refreshes it with `set_visible` (tests ask the oracle for the merged order)."""

    def __init__(self, peer: int):
        self.peer = peer
        self.changes: Dict[int, List[Change]] = {}
        self.vv: Dict[int, int] = {}
        self.frontiers: List[Tuple[int, int]] = []
        self.lamport_of: Dict[Tuple[int, int], int] = {}  # change start id -> lamport (own + merged)
        self.seq: Dict[CID, List[Tuple[int, int]]] = {}
        self.pending_ops: List[Op] = []
        self.next_counter = 0

    # -- helpers
    @staticmethod
    def _cid(target, kind: int) -> CID:
        """`target` is a root container name, or the CID of any container (e.g. a child made by *_container)."""
        if isinstance(target, CID):
            assert target.kind == kind, "container kind mismatch"
            return target
        return root_cid(target, kind)

    def _push(self, op: Op):
        if self.pending_ops and try_merge(self.pending_ops[-1], op):
            pass
        else:
            self.pending_ops.append(op)
        self.next_counter += 0  # counter advanced by callers

    def _alloc(self, n: int) -> int:
        c = self.next_counter
        self.next_counter += n
        return c

    def _lamport_of_id(self, p: int, ctr: int) -> int:
        chs = self.changes.get(p, [])
        lo, hi = 0, len(chs)
        while lo < hi:  # changes of a peer are appended in counter order
            mid = (lo + hi) // 2
            if chs[mid].ctr_end <= ctr:
                lo = mid + 1
            else:
                hi = mid
        if lo < len(chs) and chs[lo].counter <= ctr:
            return chs[lo].lamport + (ctr - chs[lo].counter)
        raise KeyError((p, ctr))

    # -- text / list
    def text_insert(self, name: str, pos: int, s: str, kind: int = KIND_TEXT):
        cid = self._cid(name, kind)
        ids = self.seq.setdefault(cid, [])
        assert 0 <= pos <= len(ids)
        c0 = self._alloc(len(s))
        self._push(Op(cid, c0, "text_insert", pos=pos, text=s))
        ids[pos:pos] = [(self.peer, c0 + i) for i in range(len(s))]

    def list_insert(self, name: str, pos: int, values: list):
        cid = self._cid(name, KIND_LIST)
        ids = self.seq.setdefault(cid, [])
        assert 0 <= pos <= len(ids)
        c0 = self._alloc(len(values))
        self._push(Op(cid, c0, "list_insert", pos=pos, values=list(values)))
        ids[pos:pos] = [(self.peer, c0 + i) for i in range(len(values))]

    def seq_delete(self, name: str, pos: int, n: int, kind: int = KIND_TEXT):
        """handler.rs:2245-2288: id-contiguous ranges, emitted right to left."""
        cid = self._cid(name, kind)
        ids = self.seq.setdefault(cid, [])
        assert 0 <= pos and pos + n <= len(ids)
        ranges = []  # (start_pos, len, id_start)
        i = pos
        while i < pos + n:
            j = i
            while j + 1 < pos + n and ids[j + 1] == (ids[j][0], ids[j][1] + 1):
                j += 1
            ranges.append((i, j - i + 1, ids[i]))
            i = j + 1
        for (start, ln, id_start) in reversed(ranges):
            c0 = self._alloc(ln)
            self._push(Op(cid, c0, "delete", pos=start, del_id=id_start, signed_len=ln))
        del ids[pos:pos + n]

    def text_delete(self, name: str, pos: int, n: int):
        self.seq_delete(name, pos, n, KIND_TEXT)

    def list_delete(self, name: str, pos: int, n: int):
        self.seq_delete(name, pos, n, KIND_LIST)

    def text_mark(self, name: str, start: int, end: int, key: str, value: Any, info: int = 0x80 | 0x04):
        """StyleStart at entity `start`, StyleEnd after entity `end` (both anchors occupy a position)."""
        cid = self._cid(name, KIND_TEXT)
        ids = self.seq.setdefault(cid, [])
        c0 = self._alloc(2)
        self._push(Op(cid, c0, "style_start", pos=start, key=key, value=value, mark_len=end - start, mark_info=info))
        self._push(Op(cid, c0 + 1, "style_end"))
        ids.insert(start, (self.peer, c0))
        ids.insert(end + 1, (self.peer, c0 + 1))

    # -- map
    def map_set(self, name: str, key: str, value: Any):
        cid = self._cid(name, KIND_MAP)
        self._push(Op(cid, self._alloc(1), "map_set", key=key, value=value))

    def map_set_container(self, name, key: str, kind: int) -> CID:
        """map.insert_container(key, kind): the child's id is the id of this op (docs/encoding.md:967-1003)."""
        cid = self._cid(name, KIND_MAP)
        c0 = self._alloc(1)
        self._push(Op(cid, c0, "map_set", key=key, value=ContainerValue(kind)))
        return CID(False, kind, "", self.peer, c0)

    def list_insert_container(self, name, pos: int, kind: int) -> CID:
        """list.insert_container(pos, kind): the child's id is the id of the new list element."""
        cid = self._cid(name, KIND_LIST)
        ids = self.seq.setdefault(cid, [])
        assert 0 <= pos <= len(ids)
        c0 = self._alloc(1)
        self._push(Op(cid, c0, "list_insert", pos=pos, values=[ContainerValue(kind)]))
        ids[pos:pos] = [(self.peer, c0)]
        return CID(False, kind, "", self.peer, c0)

    def map_delete(self, name: str, key: str):
        cid = self._cid(name, KIND_MAP)
        self._push(Op(cid, self._alloc(1), "map_delete", key=key))

    # -- movable list (handler.rs:3528-4000; state/movable_list_state.rs).  self.seq[cid] holds the ids of the list ITEMS
    # alive at this replica's version in op-index order — the items no element points at any more (losers of concurrent
    # moves) included; the user addresses ELEMENTS, i.e. the items some element points at.
    def _ml_index(self, cid: CID):
        """item id -> element (peer, lamport), element -> id of the item it points at (greatest move by (lamport, peer),
        else its insert) over every op this replica knows."""
        item_elem: Dict[Tuple[int, int], Tuple[int, int]] = {}
        best: Dict[Tuple[int, int], Tuple[int, int, Tuple[int, int]]] = {}

        def visit(peer, lam0, ctr0, ops):
            for o in ops:
                if o.cid != cid:
                    continue
                lam = lam0 + (o.counter - ctr0)
                if o.kind == "list_insert":
                    for i in range(len(o.values)):
                        e = (peer, lam + i)
                        item_elem[(peer, o.counter + i)] = e
                        cand = (lam + i, peer, (peer, o.counter + i))
                        if e not in best or best[e][:2] < cand[:2]:
                            best[e] = cand
                elif o.kind == "list_move":
                    item_elem[(peer, o.counter)] = o.elem
                    cand = (lam, peer, (peer, o.counter))
                    if o.elem not in best or best[o.elem][:2] < cand[:2]:
                        best[o.elem] = cand

        for p, chs in self.changes.items():
            for c in chs:
                visit(p, c.lamport, c.counter, c.ops)
        if self.pending_ops:
            first = self.pending_ops[0].counter
            lam = 0
            for (p, ctr) in self.frontiers:
                lam = max(lam, self._lamport_of_id(p, ctr) + 1)
            visit(self.peer, lam, first, self.pending_ops)
        return item_elem, {e: b[2] for e, b in best.items()}

    def mlist_elements(self, name) -> List[Tuple[int, Tuple[int, int], Tuple[int, int]]]:
        """[(op index, item id, element)] of the items an element points at, in list order (= the user's view)."""
        cid = self._cid(name, KIND_MOVABLE)
        item_elem, pos = self._ml_index(cid)
        out = []
        for i, it in enumerate(self.seq.setdefault(cid, [])):
            e = item_elem.get(it)
            if e is not None and pos.get(e) == it:
                out.append((i, it, e))
        return out

    def mlist_len(self, name) -> int:
        return len(self.mlist_elements(name))

    def _ml_op_index(self, cid: CID, elems, user_pos: int) -> int:  # convert_index(ForUser -> ForOp), movable_list_state.rs:855-876
        return len(self.seq[cid]) if user_pos == len(elems) else elems[user_pos][0]

    def mlist_insert(self, name, pos: int, values: list):
        cid = self._cid(name, KIND_MOVABLE)
        for k, v in enumerate(values):   # one op per value (handler.rs:3552-3585); RleVec merging fuses contiguous ones
            elems = self.mlist_elements(cid)
            assert 0 <= pos + k <= len(elems)
            at = self._ml_op_index(cid, elems, pos + k)
            c0 = self._alloc(1)
            self._push(Op(cid, c0, "list_insert", pos=at, values=[v]))
            self.seq[cid].insert(at, (self.peer, c0))

    def mlist_insert_container(self, name, pos: int, kind: int) -> CID:
        cid = self._cid(name, KIND_MOVABLE)
        c0 = self.next_counter
        self.mlist_insert(cid, pos, [ContainerValue(kind)])
        return CID(False, kind, "", self.peer, c0)

    def mlist_delete(self, name, pos: int, n: int):
        """handler.rs:3918-3959: one single-item delete per element, positions computed up front."""
        cid = self._cid(name, KIND_MOVABLE)
        elems = self.mlist_elements(cid)
        assert 0 <= pos and pos + n <= len(elems)
        plan = [(elems[u][1], elems[u][0] - (u - pos)) for u in range(pos, pos + n)]
        for item, at in plan:
            assert self.seq[cid][at] == item
            c0 = self._alloc(1)
            self._push(Op(cid, c0, "delete", pos=at, del_id=item, signed_len=1))
            del self.seq[cid][at]

    def mlist_move(self, name, frm: int, to: int):
        """handler.rs:3616-3665 + tracker.rs:289-347: the element's item is deleted, a new item (this op's id) inserted."""
        cid = self._cid(name, KIND_MOVABLE)
        if frm == to:
            return
        elems = self.mlist_elements(cid)
        assert 0 <= frm < len(elems) and 0 <= to < len(elems)
        op_from, op_to, elem = elems[frm][0], elems[to][0], elems[frm][2]
        c0 = self._alloc(1)
        self._push(Op(cid, c0, "list_move", pos=op_to, move_from=op_from, elem=elem))
        del self.seq[cid][op_from]
        self.seq[cid].insert(op_to, (self.peer, c0))

    def mlist_set(self, name, index: int, value: Any):
        cid = self._cid(name, KIND_MOVABLE)
        elems = self.mlist_elements(cid)
        assert 0 <= index < len(elems)
        self._push(Op(cid, self._alloc(1), "list_set", elem=elems[index][2], value=value))

    def mlist_set_container(self, name, index: int, kind: int) -> CID:
        c0 = self.next_counter
        self.mlist_set(name, index, ContainerValue(kind))
        return CID(False, kind, "", self.peer, c0)

    # -- commit / exchange
    def commit(self, msg: Optional[str] = None):
        if not self.pending_ops:
            return
        counter = self.pending_ops[0].counter
        lam = 0
        for (p, ctr) in self.frontiers:
            lam = max(lam, self._lamport_of_id(p, ctr) + 1)
        ch = Change(self.peer, counter, lam, list(self.frontiers), self.pending_ops, msg=msg)
        self.pending_ops = []
        self.changes.setdefault(self.peer, []).append(ch)
        self.vv[self.peer] = ch.ctr_end
        self.frontiers = [(self.peer, ch.ctr_end - 1)]

    def merge_from(self, other: "Replica"):
        """Take every change of `other` that this replica lacks (both must be committed)."""
        assert not self.pending_ops and not other.pending_ops
        added = False
        for p, chs in other.changes.items():
            have = self.vv.get(p, 0)
            for c in chs:
                if c.ctr_end <= have:
                    continue
                assert c.counter >= have, "replicas exchange whole changes"
                self.changes.setdefault(p, []).append(c)
                have = c.ctr_end
                added = True
            self.vv[p] = max(self.vv.get(p, 0), have)
        if added:
            heads = set(self.frontiers) | set(other.frontiers)
            # drop heads that are ancestors of other heads: recompute from the DAG
            self.frontiers = self._shrink_heads(heads)
        return added

    def _shrink_heads(self, heads) -> List[Tuple[int, int]]:
        covered = set()
        heads = sorted(heads)

        def closure(p, ctr, acc: Dict[int, int]):
            stack = [(p, ctr)]
            while stack:
                q, c = stack.pop()
                if acc.get(q, 0) > c:
                    continue
                acc[q] = c + 1
                for ch in self.changes.get(q, []):
                    if ch.counter <= c:
                        for d in ch.deps:
                            if acc.get(d[0], 0) <= d[1]:
                                stack.append(d)

        out = []
        for h in heads:
            acc: Dict[int, int] = {}
            for g in heads:
                if g != h:
                    closure(g[0], g[1], acc)
            if acc.get(h[0], 0) <= h[1]:
                out.append(h)
        return out

    def export(self, from_vv: Optional[Dict[int, int]] = None, max_block: int = 4096) -> bytes:
        """ExportMode::Updates{from}: whole changes past from_vv, per-peer blocks."""
        assert not self.pending_ops
        from_vv = from_vv or {}
        blocks = []
        for p in sorted(self.changes):
            sel = [c for c in self.changes[p] if c.ctr_end > from_vv.get(p, 0)]
            if sel:
                assert sel[0].counter >= from_vv.get(p, 0), "export starts on a change boundary"
                blocks += split_blocks(sel, max_block)
        return encode_updates(blocks)

    def export_snapshot(self, roots=None, max_block: int = 4096, block_size: int = 4096, compress: bool = True, state=None) -> bytes:
        """ExportMode::Snapshot of everything this replica knows; `roots` = [(kind, name)] of the state section (default: every
        root container an op of the history addresses)."""
        assert not self.pending_ops
        blocks = []
        for p in sorted(self.changes):
            blocks += split_blocks(self.changes[p], max_block)
        if roots is None:
            seen = []
            for p in sorted(self.changes):
                for c in self.changes[p]:
                    for o in c.ops:
                        if o.cid.root and (o.cid.kind, o.cid.name) not in seen:
                            seen.append((o.cid.kind, o.cid.name))
            roots = seen
        return encode_snapshot(blocks, roots, dict(self.vv), list(self.frontiers), block_size, compress, state=state)

    def set_visible(self, name, kind: int, ids: List[Tuple[int, int]]):
        self.seq[self._cid(name, kind)] = list(ids)
