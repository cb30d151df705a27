// Export side, part two: `lm_export(ctx, doc, from_vv)` — the updates a document holds beyond a version, as
// a FastUpdates blob: what `LoroDoc::export(ExportMode::Updates{from})` produces for a host that keeps its documents in this
// engine (crates/loro/src/lib.rs:1306; crates/loro-internal/src/encoding.rs:399-405 export_fast_updates →
// oplog/change_store.rs:718-752 export_blocks_from: the changes of `latest_vv.sub_iter(start_vv)`, a change that straddles
// the start sliced — change.rs Sliceable, container/list/list_op.rs:251-277,426-433 — and block_encode.rs:137-278 per block).
// Host C++: the document's blobs are read back from the context's arena, every change block is decoded into its tables (the
// inverse of lm_encode.h — block_encode.rs:535-706, block_meta_encode.rs:90-242; serde_columnar strategies docs/encoding.md §8),
// blocks the version already covers are dropped, a straddling block is cut at the version, and the rest is re-encoded by
// lm_encode.h.  Block boundaries stay those of the imported blobs (the reference re-chunks into ~4 KiB blocks when it
// re-inserts the changes, change_store.rs:42; the decoded content is the same), so a blob exported from the empty version
// reproduces, byte for byte, the blocks it was staged with — ordered by (peer, counter) like the reference's BTreeMap.
#pragma once
#include <algorithm>
#include <map>
#include "lm_encode.h"

namespace lmexp {

using lmenc::Bytes;

struct Rd {
  const uint8_t* p; const uint8_t* end;
  Rd(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  size_t left() const { return (size_t)(end - p); }
  uint8_t u8() { if (p >= end) throw std::runtime_error("block: unexpected end"); return *p++; }
  uint64_t uleb() {
    uint64_t v = 0;
    for (int i = 0; i < 10; i++) { uint8_t b = u8(); v |= (uint64_t)(b & 0x7f) << (7 * i); if (!(b & 0x80)) return v; }
    throw std::runtime_error("block: varint too long");
  }
  int64_t zigzag() { uint64_t v = uleb(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  int64_t zigzag_wide() {   // i128 on the wire (DeltaRle); every in-scope delta fits i64
    unsigned __int128 v = 0;
    for (int i = 0; i < 19; i++) { uint8_t b = u8(); v |= (unsigned __int128)(b & 0x7f) << (7 * i); if (!(b & 0x80)) { __int128 s = (__int128)(v >> 1) ^ -(__int128)(v & 1); return (int64_t)s; } }
    throw std::runtime_error("block: varint too long");
  }
  int64_t sleb() {
    int64_t r = 0; int sh = 0; uint8_t b;
    do { b = u8(); if (sh < 64) r |= (int64_t)(b & 0x7f) << sh; sh += 7; } while (b & 0x80);
    if (sh < 64 && (b & 0x40)) r |= -((int64_t)1 << sh);
    return r;
  }
  Rd bytes() { uint64_t n = uleb(); if (n > left()) throw std::runtime_error("block: section beyond its end"); Rd r(p, (size_t)n); p += n; return r; }
  void skip(uint64_t n) { if (n > left()) throw std::runtime_error("block: skip beyond the end"); p += n; }
};

// ---- serde_columnar column readers (docs/encoding.md §8-8.1): `n` values each
inline std::vector<uint8_t> dec_bool_rle(Rd& r, size_t n) {
  std::vector<uint8_t> out;
  bool cur = false;
  while (out.size() < n) { uint64_t run = r.uleb(); if (run > n - out.size()) throw std::runtime_error("BoolRle run too long"); out.insert(out.end(), (size_t)run, cur ? 1 : 0); cur = !cur; }
  return out;
}
template <class T, class F>
inline std::vector<T> dec_any_rle(Rd& r, size_t n, F&& rv) {
  std::vector<T> out;
  while (out.size() < n) {
    int64_t c = r.zigzag();
    if (c > 0) { T v = rv(r); if ((uint64_t)c > n - out.size()) throw std::runtime_error("Rle run too long"); out.insert(out.end(), (size_t)c, v); }
    else if (c < 0) { if ((uint64_t)(-c) > n - out.size()) throw std::runtime_error("Rle literals too long"); for (int64_t i = 0; i < -c; i++) out.push_back(rv(r)); }
    else throw std::runtime_error("Rle zero-length segment");
  }
  return out;
}
// a whole column given as its own byte string: the count is what the bytes hold
template <class T, class F>
inline std::vector<T> dec_any_rle_all(Rd r, F&& rv) {
  std::vector<T> out;
  while (r.left()) {
    int64_t c = r.zigzag();
    if (c > 0) { T v = rv(r); if (c > (1 << 28)) throw std::runtime_error("Rle run too long"); out.insert(out.end(), (size_t)c, v); }
    else if (c < 0) {   // literal run: every literal takes at least one byte (and -INT64_MIN does not exist)
      if (c < -(int64_t)r.left()) throw std::runtime_error("Rle literal run longer than its bytes");
      for (int64_t i = 0; i < -c; i++) out.push_back(rv(r));
    }
    else throw std::runtime_error("Rle zero-length segment");
  }
  return out;
}
inline std::vector<int64_t> dec_delta_rle_all(Rd r) {
  std::vector<int64_t> d = dec_any_rle_all<int64_t>(r, [](Rd& q) { return q.zigzag_wide(); });
  int64_t acc = 0;
  for (auto& x : d) { acc += x; x = acc; }
  return d;
}
inline std::vector<int64_t> dec_delta_of_delta(Rd& r, size_t n) {
  std::vector<int64_t> out;
  uint8_t has = r.u8();
  if (!has) { (void)r.u8(); if (n) throw std::runtime_error("DeltaOfDelta: missing values"); return out; }
  int64_t first = r.zigzag();
  uint8_t last_bits = r.u8();
  out.push_back(first);
  if (n == 0) throw std::runtime_error("DeltaOfDelta: unexpected values");
  // bit reader over the rest; consumes whole bytes
  uint64_t bitpos = 0;
  const uint8_t* base = r.p;
  auto bit = [&]() -> uint32_t { size_t by = (size_t)(bitpos >> 3); if (base + by >= r.end) throw std::runtime_error("DeltaOfDelta: bits beyond the end"); uint32_t b = (base[by] >> (7 - (bitpos & 7))) & 1; bitpos++; return b; };
  auto bits = [&](int c) -> uint64_t { uint64_t v = 0; for (int i = 0; i < c; i++) v = (v << 1) | bit(); return v; };
  int64_t prev = first, prev_delta = 0;
  while (out.size() < n) {
    int64_t dod;
    if (!bit()) dod = 0;
    else if (!bit()) dod = (int64_t)bits(7) - 63;
    else if (!bit()) dod = (int64_t)bits(9) - 255;
    else if (!bit()) dod = (int64_t)bits(12) - 2047;
    else if (!bit()) dod = (int64_t)bits(21) - 1048575;
    else dod = (int64_t)bits(64);
    prev_delta += dod;
    prev += prev_delta;
    out.push_back(prev);
  }
  size_t used = (size_t)((bitpos + 7) >> 3);
  (void)last_bits;
  r.p = base + used;
  return out;
}

// ---- value payloads (docs/encoding.md §10): the extent of one value in the values section
inline void skip_loro_value(Rd& r, int depth = 0) {
  if (depth > 64) throw std::runtime_error("value nesting too deep");
  uint8_t tag = r.u8();
  switch (tag) {
    case 0: case 1: case 2: break;
    case 3: (void)r.sleb(); break;
    case 4: r.skip(8); break;
    case 5: case 6: r.skip(r.uleb()); break;
    case 7: { uint64_t n = r.uleb(); for (uint64_t i = 0; i < n; i++) skip_loro_value(r, depth + 1); break; }
    case 8: { uint64_t n = r.uleb(); for (uint64_t i = 0; i < n; i++) { (void)r.uleb(); skip_loro_value(r, depth + 1); } break; }
    case 9: (void)r.u8(); break;
    default: throw std::runtime_error("unknown LoroValue tag");
  }
}
inline void skip_value(Rd& v, uint32_t vt) {
  switch (vt & 0x7f) {
    case 0: case 1: case 2: case 8: case 9: break;
    case 3: case 10: (void)v.sleb(); break;
    case 4: v.skip(8); break;
    case 5: case 6: v.skip(v.uleb()); break;
    case 7: (void)v.uleb(); break;
    case 11: skip_loro_value(v); break;
    case 12: (void)v.u8(); (void)v.uleb(); (void)v.uleb(); skip_loro_value(v); break;
    case 13: { (void)v.uleb(); uint8_t isn = v.u8(); (void)v.uleb(); if (!isn) (void)v.uleb(); break; }
    case 14: (void)v.uleb(); (void)v.uleb(); (void)v.uleb(); break;
    case 15: (void)v.uleb(); (void)v.uleb(); skip_loro_value(v); break;
    case 16: { (void)v.uleb(); (void)v.uleb(); (void)v.uleb(); uint8_t isn = v.u8(); if (!isn) { (void)v.uleb(); (void)v.uleb(); } break; }
    default: v.skip(v.uleb()); break;
  }
}

// ---- one change block, decoded (owning its arrays)
struct Block {
  uint32_t counter_start = 0, counter_len = 0, lamport_start = 0, lamport_len = 0;
  std::vector<uint64_t> peers;
  std::vector<uint32_t> change_len, dep_count, dep_peer_idx, lamport, msg_len, cid_peer_idx, op_container, op_len, del_peer_idx;
  std::vector<uint8_t> dep_on_self, cid_is_root, cid_kind, op_value_type;
  std::vector<int32_t> dep_counter, cid_key_or_counter, op_prop, del_counter;
  std::vector<int64_t> timestamp, del_len;
  Bytes msgs, positions, values;
  std::vector<Bytes> keys;
  std::vector<const uint8_t*> key_ptr;
  std::vector<size_t> key_len;
  uint64_t peer() const { return peers.at(0); }
  uint32_t end() const { return counter_start + counter_len; }
  lm_block_tables view() {
    key_ptr.clear(); key_len.clear();
    for (auto& k : keys) { key_ptr.push_back(k.data()); key_len.push_back(k.size()); }
    lm_block_tables t;
    memset(&t, 0, sizeof t);
    t.counter_start = counter_start; t.counter_len = counter_len; t.lamport_start = lamport_start; t.lamport_len = lamport_len; t.n_changes = (uint32_t)change_len.size();
    t.peers = peers.data(); t.n_peers = peers.size();
    t.change_len = change_len.data(); t.dep_on_self = dep_on_self.data(); t.dep_count = dep_count.data();
    t.dep_peer_idx = dep_peer_idx.data(); t.dep_counter = dep_counter.data(); t.n_deps = dep_peer_idx.size();
    t.lamport = lamport.data(); t.timestamp = timestamp.data();
    t.msg_len = msg_len.data(); t.msgs = msgs.data(); t.msgs_len = msgs.size();
    t.cid_is_root = cid_is_root.data(); t.cid_kind = cid_kind.data(); t.cid_peer_idx = cid_peer_idx.data(); t.cid_key_or_counter = cid_key_or_counter.data(); t.n_cids = cid_kind.size();
    t.keys = key_ptr.data(); t.key_lens = key_len.data(); t.n_keys = keys.size();
    t.positions = positions.data(); t.positions_len = positions.size();
    t.op_container = op_container.data(); t.op_prop = op_prop.data(); t.op_value_type = op_value_type.data(); t.op_len = op_len.data(); t.n_ops = op_len.size();
    t.del_peer_idx = del_peer_idx.data(); t.del_counter = del_counter.data(); t.del_len = del_len.data(); t.n_dels = del_len.size();
    t.values = values.data(); t.values_len = values.size();
    return t;
  }
};

inline bool op_has_del_row(const Block& b, size_t i) {   // DeleteSeq of a known sequence container carries a delete-start row
  if ((b.op_value_type[i] & 0x7f) != 9) return false;
  uint32_t c = b.op_container[i];
  if (c >= b.cid_kind.size()) return false;
  uint8_t k = b.cid_kind[c];
  return k == 1 || k == 2 || k == 4;
}

inline Block decode_block(const uint8_t* p, size_t n) {
  Rd r(p, n);
  Block b;
  uint64_t cs = r.uleb(), cl = r.uleb(), ls = r.uleb(), ll = r.uleb(), N = r.uleb();
  if (cs > 0x7fffffffu || cl > 0x7fffffffu || ls > 0xffffffffu || ll > 0xffffffffu || N == 0 || N > cl) throw std::runtime_error("block: scalars out of range");
  b.counter_start = (uint32_t)cs; b.counter_len = (uint32_t)cl; b.lamport_start = (uint32_t)ls; b.lamport_len = (uint32_t)ll;
  Rd header = r.bytes(), meta = r.bytes(), cids = r.bytes(), keys = r.bytes(), positions = r.bytes(), ops = r.bytes(), dels = r.bytes(), values = r.bytes();
  if (r.left()) throw std::runtime_error("block: trailing bytes");
  {   // header (block_meta_encode.rs:90-242)
    uint64_t np = header.uleb();
    if (np == 0 || np > 1u << 20) throw std::runtime_error("block: peer table");
    for (uint64_t i = 0; i < np; i++) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)header.u8() << (8 * k); b.peers.push_back(v); }
    uint64_t sum = 0;
    for (uint64_t i = 0; i + 1 < N; i++) { uint64_t l = header.uleb(); if (l == 0 || l > cl) throw std::runtime_error("block: change length"); b.change_len.push_back((uint32_t)l); sum += l; }
    if (sum >= cl) throw std::runtime_error("block: change lengths");
    b.change_len.push_back((uint32_t)(cl - sum));
    b.dep_on_self = dec_bool_rle(header, (size_t)N);
    std::vector<uint64_t> dc = dec_any_rle<uint64_t>(header, (size_t)N, [](Rd& q) { return q.uleb(); });
    uint64_t nd = 0;
    for (uint64_t x : dc) { if (x > np + 1) throw std::runtime_error("block: dependency count"); b.dep_count.push_back((uint32_t)x); nd += x; }
    std::vector<uint64_t> dp = dec_any_rle<uint64_t>(header, (size_t)nd, [](Rd& q) { return q.uleb(); });
    for (uint64_t x : dp) b.dep_peer_idx.push_back((uint32_t)x);
    std::vector<int64_t> dctr = dec_delta_of_delta(header, (size_t)nd);
    for (int64_t x : dctr) b.dep_counter.push_back((int32_t)x);
    std::vector<int64_t> lam = dec_delta_of_delta(header, (size_t)(N - 1));
    for (int64_t x : lam) b.lamport.push_back((uint32_t)x);
    b.lamport.push_back((uint32_t)(ls + ll - b.change_len.back()));
  }
  {   // change meta (block_encode.rs:176-196): timestamps, message lengths, message bytes
    b.timestamp = dec_delta_of_delta(meta, (size_t)N);
    std::vector<uint64_t> ml = dec_any_rle<uint64_t>(meta, (size_t)N, [](Rd& q) { return q.uleb(); });
    for (uint64_t x : ml) b.msg_len.push_back((uint32_t)x);
    b.msgs.assign(meta.p, meta.end);
  }
  {   // container arena rows (encoding/arena.rs:39-105)
    uint64_t nc = cids.uleb();
    for (uint64_t i = 0; i < nc; i++) {
      if (cids.uleb() != 4) throw std::runtime_error("block: container row");
      b.cid_is_root.push_back(cids.u8() ? 1 : 0);
      b.cid_kind.push_back(cids.u8());
      b.cid_peer_idx.push_back((uint32_t)cids.uleb());
      b.cid_key_or_counter.push_back((int32_t)cids.zigzag());
    }
  }
  while (keys.left()) { Rd k = keys.bytes(); b.keys.emplace_back(k.p, k.end); }
  b.positions.assign(positions.p, positions.end);
  {   // EncodedOp columns (block_encode.rs:417-445)
    if (ops.uleb() != 1 || ops.uleb() != 4) throw std::runtime_error("block: op columns");
    std::vector<int64_t> c = dec_delta_rle_all(ops.bytes()), pr = dec_delta_rle_all(ops.bytes());
    std::vector<uint8_t> vt = dec_any_rle_all<uint8_t>(ops.bytes(), [](Rd& q) { return q.u8(); });
    std::vector<uint64_t> ln = dec_any_rle_all<uint64_t>(ops.bytes(), [](Rd& q) { return q.uleb(); });
    if (c.size() != pr.size() || c.size() != vt.size() || c.size() != ln.size() || c.empty()) throw std::runtime_error("block: op column lengths");
    uint64_t atoms = 0;
    for (size_t i = 0; i < c.size(); i++) {
      if (c[i] < 0 || (uint64_t)c[i] >= b.cid_kind.size() || pr[i] < INT32_MIN || pr[i] > INT32_MAX || ln[i] == 0 || ln[i] > cl) throw std::runtime_error("block: op row");
      b.op_container.push_back((uint32_t)c[i]); b.op_prop.push_back((int32_t)pr[i]); b.op_value_type.push_back(vt[i]); b.op_len.push_back((uint32_t)ln[i]);
      atoms += ln[i];
    }
    if (atoms != cl) throw std::runtime_error("block: op lengths do not add up to the block's counters");
  }
  if (dels.left()) {   // delete-start ids (outdated_encode_reordered.rs:480-489)
    if (dels.uleb() != 1 || dels.uleb() != 3) throw std::runtime_error("block: delete columns");
    std::vector<int64_t> a = dec_delta_rle_all(dels.bytes()), c = dec_delta_rle_all(dels.bytes()), l = dec_delta_rle_all(dels.bytes());
    if (a.size() != c.size() || a.size() != l.size()) throw std::runtime_error("block: delete column lengths");
    for (size_t i = 0; i < a.size(); i++) { b.del_peer_idx.push_back((uint32_t)a[i]); b.del_counter.push_back((int32_t)c[i]); b.del_len.push_back(l[i]); }
  }
  b.values.assign(values.p, values.end);
  size_t need = 0;
  for (size_t i = 0; i < b.op_len.size(); i++) need += op_has_del_row(b, i) ? 1 : 0;
  if (need != b.del_len.size()) throw std::runtime_error("block: delete rows do not match the delete ops");
  return b;
}

// keep the ops with counters in [from, to) — `from` may fall inside a change and inside an op (the op is sliced like the
// reference slices it, list_op.rs:251-277,426-433; the change then depends on its peer's previous op, change.rs Sliceable);
// `to` lies on a change boundary (what is dropped there are changes still waiting for their dependencies)
inline void cut_block(Block& b, uint32_t from, uint32_t to) {
  if (from <= b.counter_start && to >= b.end()) return;
  if (from < b.counter_start) from = b.counter_start;
  if (to > b.end()) to = b.end();
  if (from >= to) throw std::runtime_error("cut_block: empty range");
  size_t N = b.change_len.size();
  // ---- changes
  std::vector<uint32_t> ch_start(N);
  { uint32_t c = b.counter_start; for (size_t i = 0; i < N; i++) { ch_start[i] = c; c += b.change_len[i]; } }
  size_t k0 = 0, k1 = N;
  while (k0 + 1 < N && ch_start[k0 + 1] <= from) k0++;
  while (k1 > k0 + 1 && ch_start[k1 - 1] >= to) k1--;
  if (ch_start[k1 - 1] + b.change_len[k1 - 1] > to) throw std::runtime_error("cut_block: the end is not a change boundary");
  uint32_t co = from - ch_start[k0];
  std::vector<uint32_t> n_len, n_dc, n_dp, n_lam, n_ml;
  std::vector<uint8_t> n_self;
  std::vector<int32_t> n_dctr;
  std::vector<int64_t> n_ts;
  Bytes n_msgs;
  size_t dep_at = 0, msg_at = 0;
  for (size_t i = 0; i < N; i++) {
    size_t nd = b.dep_count[i], ml = b.msg_len[i];
    if (i >= k0 && i < k1) {
      bool sliced = i == k0 && co > 0;
      n_len.push_back(b.change_len[i] - (sliced ? co : 0));
      n_lam.push_back(b.lamport[i] + (sliced ? co : 0));
      n_self.push_back(sliced ? 1 : b.dep_on_self[i]);
      n_dc.push_back(sliced ? 0 : (uint32_t)nd);
      if (dep_at + nd > b.dep_peer_idx.size() || dep_at + nd > b.dep_counter.size()) throw std::runtime_error("block: dependency rows do not match the dependency counts");
      if (!sliced) for (size_t d = 0; d < nd; d++) { n_dp.push_back(b.dep_peer_idx[dep_at + d]); n_dctr.push_back(b.dep_counter[dep_at + d]); }
      n_ts.push_back(b.timestamp[i]);
      n_ml.push_back((uint32_t)ml);
      if (msg_at + ml > b.msgs.size()) throw std::runtime_error("cut_block: commit messages");
      n_msgs.insert(n_msgs.end(), b.msgs.begin() + msg_at, b.msgs.begin() + msg_at + ml);
    }
    dep_at += nd; msg_at += ml;
  }
  // ---- ops, delete rows, values
  std::vector<uint32_t> o_c, o_l, d_p;
  std::vector<int32_t> o_p, d_c;
  std::vector<uint8_t> o_vt;
  std::vector<int64_t> d_l;
  Bytes n_values;
  Rd v(b.values.data(), b.values.size());
  uint32_t c = b.counter_start;
  size_t del_at = 0;
  for (size_t i = 0; i < b.op_len.size(); i++) {
    uint32_t len = b.op_len[i], vt = b.op_value_type[i];
    const uint8_t* v0 = v.p;
    skip_value(v, vt);
    const uint8_t* v1 = v.p;
    bool has_row = op_has_del_row(b, i);
    size_t row = del_at;
    if (has_row) del_at++;
    uint32_t s = c, e = c + len;
    c = e;
    if (e <= from || s >= to) continue;
    if (e > to) throw std::runtime_error("cut_block: an op crosses the end");
    uint32_t off = s < from ? from - s : 0;
    int32_t prop = b.op_prop[i];
    if (off == 0) {
      n_values.insert(n_values.end(), v0, v1);
      if (has_row) { d_p.push_back(b.del_peer_idx[row]); d_c.push_back(b.del_counter[row]); d_l.push_back(b.del_len[row]); }
    } else if ((vt & 0x7f) == 5) {   // Text insert: the string from its off-th unicode scalar on
      Rd q(v0, (size_t)(v1 - v0));
      uint64_t bl = q.uleb();
      const uint8_t* sp = q.p;
      size_t at = 0;
      for (uint32_t k = 0; k < off; k++) { if (at >= bl) throw std::runtime_error("cut_block: text shorter than its op"); at++; while (at < bl && (sp[at] & 0xC0) == 0x80) at++; }
      lmenc::put_uleb(n_values, bl - at);
      n_values.insert(n_values.end(), sp + at, sp + bl);
      prop += (int32_t)off;
    } else if ((vt & 0x7f) == 11 && v0 < v1 && *v0 == 7) {   // List insert: the items from the off-th on
      Rd q(v0 + 1, (size_t)(v1 - v0 - 1));
      uint64_t cnt = q.uleb();
      if (cnt != len) throw std::runtime_error("cut_block: list insert length");
      for (uint32_t k = 0; k < off; k++) skip_loro_value(q);
      n_values.push_back(7);
      lmenc::put_uleb(n_values, cnt - off);
      n_values.insert(n_values.end(), q.p, v1);
      prop += (int32_t)off;
    } else if (has_row) {   // DeleteSpanWithId::slice(off, len)
      int64_t L = b.del_len[row];
      d_p.push_back(b.del_peer_idx[row]);
      if (L > 0) { d_c.push_back(b.del_counter[row] + (int32_t)off); d_l.push_back(L - off); }
      else { d_c.push_back(b.del_counter[row]); d_l.push_back(L + off); prop -= (int32_t)off; }
    } else throw std::runtime_error("cut_block: an op of this kind cannot be sliced");
    o_c.push_back(b.op_container[i]); o_p.push_back(prop); o_vt.push_back((uint8_t)vt); o_l.push_back(len - off);
  }
  uint32_t lam_end = b.lamport_start + b.lamport_len;
  if (k1 < N) lam_end = n_lam.back() + n_len.back();
  b.counter_start = from; b.counter_len = to - from;
  b.lamport_start = n_lam.front(); b.lamport_len = lam_end - n_lam.front();
  b.change_len.swap(n_len); b.lamport.swap(n_lam); b.dep_on_self.swap(n_self); b.dep_count.swap(n_dc); b.dep_peer_idx.swap(n_dp); b.dep_counter.swap(n_dctr);
  b.timestamp.swap(n_ts); b.msg_len.swap(n_ml); b.msgs.swap(n_msgs);
  b.op_container.swap(o_c); b.op_prop.swap(o_p); b.op_value_type.swap(o_vt); b.op_len.swap(o_l);
  b.del_peer_idx.swap(d_p); b.del_counter.swap(d_c); b.del_len.swap(d_l);
  b.values.swap(n_values);
}

// the change blocks of one FastUpdates blob (envelope already verified by the device when it was imported)
inline void blocks_of_blob(const uint8_t* blob, size_t len, std::vector<Block>& out) {
  if (len < 22 || memcmp(blob, "loro", 4) != 0 || blob[20] != 0 || blob[21] != 4) throw std::runtime_error("export: not a FastUpdates blob");
  Rd r(blob + 22, len - 22);
  while (r.left()) { Rd b = r.bytes(); out.push_back(decode_block(b.p, b.left())); }
}

// postcard VersionVector (version.rs:962-968): map peer → exclusive counter
inline std::map<uint64_t, uint32_t> decode_vv(const uint8_t* p, size_t n) {
  std::map<uint64_t, uint32_t> vv;
  if (!p || !n) return vv;
  Rd r(p, n);
  uint64_t cnt = r.uleb();
  for (uint64_t i = 0; i < cnt; i++) { uint64_t peer = r.uleb(); int64_t c = r.zigzag(); if (c > 0) vv[peer] = (uint32_t)c; }
  if (r.left()) throw std::runtime_error("export: trailing bytes after the version vector");
  return vv;
}

// blocks (any order, overlaps and duplicates allowed) → the updates beyond `from`, up to `applied` (per peer: the end of what
// the document's oplog holds; changes behind a gap or waiting for a dependency are not part of it)
inline Bytes export_updates(std::vector<Block>& blocks, const std::map<uint64_t, uint32_t>& from, const std::map<uint64_t, uint32_t>& applied) {
  std::map<uint64_t, std::vector<Block*>> by_peer;
  for (Block& b : blocks) by_peer[b.peer()].push_back(&b);
  std::vector<Bytes> enc;
  for (auto& kv : by_peer) {
    auto ai = applied.find(kv.first);
    uint32_t end = ai == applied.end() ? 0 : ai->second;
    auto fi = from.find(kv.first);
    uint32_t covered = fi == from.end() ? 0 : fi->second;
    if (covered >= end) continue;
    std::vector<Block*>& v = kv.second;
    std::stable_sort(v.begin(), v.end(), [](const Block* a, const Block* b) { return a->counter_start < b->counter_start; });
    for (Block* b : v) {
      if (b->end() <= covered) continue;
      if (b->counter_start > covered && b->counter_start >= end) break;
      if (b->counter_start > covered) {
        // a hole between what was exported so far and this block: only allowed below `from` (the version itself covers it)
        throw std::runtime_error("export: the document's blocks do not cover the applied history");
      }
      uint32_t lo = covered, hi = b->end() < end ? b->end() : end;
      if (lo >= hi) continue;
      cut_block(*b, lo, hi);
      lm_block_tables t = b->view();
      enc.push_back(lmenc::encode_block(t));
      covered = hi;
      if (covered >= end) break;
    }
  }
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> lens;
  for (auto& e : enc) { ptrs.push_back(e.data()); lens.push_back(e.size()); }
  return lmenc::encode_updates(ptrs.data(), lens.data(), enc.size());
}

}  // namespace lmexp
