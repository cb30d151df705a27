// Export / encode side of the change-block path — host C++, no device work: the inverse of the decode
// stage.  Given the tables of one change block (exactly what k_block_decode produces from a block: header columns, change
// meta, container arena, keys, op columns, delete-start ids; the value and position sections travel as bytes) it writes
// the block the reference's writer writes, byte for byte, and frames blocks into a FastUpdates blob.
// Reference (paths relative to /root/reference/crates/loro-internal/src):
//   EncodedBlock (postcard struct of 5 scalars + 8 byte sections)       oplog/change_store/block_encode.rs:94-119,137-278
//   header: peers, N-1 lengths, dep_on_self BoolRle, dep counts / peer idx AnyRle, dep counters / lamports DeltaOfDelta
//                                                                        oplog/change_store/block_meta_encode.rs:13-88
//   change meta: timestamps DeltaOfDelta, message lengths AnyRle, message bytes                  block_encode.rs:176-196
//   EncodedOp columns: container_index DeltaRle, prop DeltaRle, value_type Rle, len Rle          block_encode.rs:417-445
//   EncodedDeleteStartId columns: peer_idx, counter, len — all DeltaRle          encoding/outdated_encode_reordered.rs:480-489
//   container arena rows                                                                         encoding/arena.rs:39-105
//   blob = "loro" + 12 zero bytes + xxh32(mode..end, seed "LORO") LE + mode u16 BE + (uleb len + block)*
//                                                                        encoding.rs:440-473, encoding/fast_snapshot.rs:346-360
// Column strategies are serde_columnar 0.3.14's (not under /root/reference; docs/encoding.md §8): AnyRle runs of >= 2 equal
// values as (zigzag +n, value), everything else as literal groups (zigzag -n, values) — the segmentation the Rust-written
// fixture updates.blob shows; DeltaRle = AnyRle over i128 zigzag deltas; BoolRle alternating run lengths starting with
// `false`; DeltaOfDelta = first value + bit-packed second differences in five buckets.
// tests/test_encode_roundtrip.py: every block of updates.blob re-encodes to its own bytes, and the reframed blob is identical.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/loro_block_tables.h"

namespace lmenc {

typedef std::vector<uint8_t> Bytes;

inline void put_uleb(Bytes& o, uint64_t v) {
  for (;;) {
    uint8_t b = (uint8_t)(v & 0x7f);
    v >>= 7;
    if (v) o.push_back(b | 0x80); else { o.push_back(b); return; }
  }
}
inline void put_zigzag(Bytes& o, int64_t v) { put_uleb(o, v >= 0 ? ((uint64_t)v << 1) : ((((uint64_t)(-(v + 1))) << 1) | 1)); }
inline void put_bytes(Bytes& o, const uint8_t* p, size_t n) { put_uleb(o, n); o.insert(o.end(), p, p + n); }
inline void put_bytes(Bytes& o, const Bytes& b) { put_bytes(o, b.data(), b.size()); }

// ---- serde_columnar strategies
inline Bytes enc_bool_rle(const uint8_t* v, size_t n) {
  Bytes o;
  bool cur = false;
  uint64_t run = 0;
  for (size_t i = 0; i < n; i++) {
    bool x = v[i] != 0;
    if (x == cur) run++;
    else { put_uleb(o, run); cur = x; run = 1; }
  }
  if (n) put_uleb(o, run);
  return o;
}
// AnyRle over values written by `wv`: maximal runs of >= 2 equal values, literal groups in between
template <class T, class W>
inline Bytes enc_any_rle(const T* v, size_t n, W&& wv) {
  Bytes o;
  size_t i = 0;
  std::vector<T> lit;
  auto flush = [&]() {
    if (lit.empty()) return;
    put_zigzag(o, -(int64_t)lit.size());
    for (const T& x : lit) wv(o, x);
    lit.clear();
  };
  while (i < n) {
    size_t j = i;
    while (j + 1 < n && v[j + 1] == v[i]) j++;
    size_t run = j - i + 1;
    if (run >= 2) { flush(); put_zigzag(o, (int64_t)run); wv(o, v[i]); }
    else lit.push_back(v[i]);
    i = j + 1;
  }
  flush();
  return o;
}
inline Bytes enc_any_rle_uvar(const std::vector<uint64_t>& v) { return enc_any_rle<uint64_t>(v.data(), v.size(), [](Bytes& o, uint64_t x) { put_uleb(o, x); }); }
inline Bytes enc_rle_u8(const uint8_t* v, size_t n) { return enc_any_rle<uint8_t>(v, n, [](Bytes& o, uint8_t x) { o.push_back(x); }); }
inline Bytes enc_delta_rle(const std::vector<int64_t>& v) {
  std::vector<int64_t> d(v.size());
  int64_t prev = 0;
  for (size_t i = 0; i < v.size(); i++) { d[i] = v[i] - prev; prev = v[i]; }
  return enc_any_rle<int64_t>(d.data(), d.size(), [](Bytes& o, int64_t x) { put_zigzag(o, x); });
}
struct BitWriter {
  Bytes out;
  uint32_t cur = 0, used = 0;
  void write(uint64_t value, int count) {
    for (int s = count - 1; s >= 0; s--) {
      cur = (cur << 1) | (uint32_t)((value >> s) & 1);
      if (++used == 8) { out.push_back((uint8_t)cur); cur = 0; used = 0; }
    }
  }
};
inline Bytes enc_delta_of_delta(const std::vector<int64_t>& v) {
  Bytes o;
  if (v.empty()) { o.push_back(0); o.push_back(0); return o; }
  o.push_back(1);
  put_zigzag(o, v[0]);
  if (v.size() == 1) { o.push_back(0); return o; }
  BitWriter bw;
  int64_t prev_delta = 0;
  for (size_t i = 1; i < v.size(); i++) {
    int64_t delta = v[i] - v[i - 1], dod = delta - prev_delta;
    prev_delta = delta;
    if (dod == 0) bw.write(0, 1);
    else if (dod >= -63 && dod <= 64) { bw.write(0b10, 2); bw.write((uint64_t)(dod + 63), 7); }
    else if (dod >= -255 && dod <= 256) { bw.write(0b110, 3); bw.write((uint64_t)(dod + 255), 9); }
    else if (dod >= -2047 && dod <= 2048) { bw.write(0b1110, 4); bw.write((uint64_t)(dod + 2047), 12); }
    else if (dod >= -1048575 && dod <= 1048576) { bw.write(0b11110, 5); bw.write((uint64_t)(dod + 1048575), 21); }
    else { bw.write(0b11111, 5); bw.write((uint64_t)dod, 64); }
  }
  uint32_t used = bw.used;
  if (used) bw.out.push_back((uint8_t)((bw.cur << (8 - used)) & 0xff));
  o.push_back((uint8_t)(used ? used : 8));
  o.insert(o.end(), bw.out.begin(), bw.out.end());
  return o;
}

// ---- xxh32 (envelope checksum, docs/encoding-xxhash32.md)
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t xxh32(const uint8_t* p, size_t len, uint32_t seed) {
  const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
  auto rd = [](const uint8_t* q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); };
  size_t i = 0;
  uint32_t h;
  if (len >= 16) {
    uint32_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
    for (; i + 16 <= len; i += 16)
      for (int k = 0; k < 4; k++) v[k] = rotl32(v[k] + rd(p + i + 4 * k) * P2, 13) * P1;
    h = rotl32(v[0], 1) + rotl32(v[1], 7) + rotl32(v[2], 12) + rotl32(v[3], 18);
  } else h = seed + P5;
  h += (uint32_t)len;
  for (; i + 4 <= len; i += 4) h = rotl32(h + rd(p + i) * P3, 17) * P4;
  for (; i < len; i++) h = rotl32(h + p[i] * P5, 11) * P1;
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}

// ---- one change block (block_encode.rs:137-278)
inline Bytes encode_block(const lm_block_tables& t) {
  size_t N = t.n_changes;
  if (N == 0 || t.n_peers == 0) throw std::runtime_error("a change block holds at least one change and its peer");
  Bytes header;
  put_uleb(header, t.n_peers);
  for (size_t i = 0; i < t.n_peers; i++) for (int k = 0; k < 8; k++) header.push_back((uint8_t)(t.peers[i] >> (8 * k)));
  for (size_t i = 0; i + 1 < N; i++) put_uleb(header, t.change_len[i]);
  {
    Bytes b = enc_bool_rle(t.dep_on_self, N);
    header.insert(header.end(), b.begin(), b.end());
    std::vector<uint64_t> dc(t.dep_count, t.dep_count + N), dp(t.dep_peer_idx, t.dep_peer_idx + t.n_deps);
    b = enc_any_rle_uvar(dc); header.insert(header.end(), b.begin(), b.end());
    b = enc_any_rle_uvar(dp); header.insert(header.end(), b.begin(), b.end());
    std::vector<int64_t> dctr(t.dep_counter, t.dep_counter + t.n_deps), lam(t.lamport, t.lamport + (N - 1));
    b = enc_delta_of_delta(dctr); header.insert(header.end(), b.begin(), b.end());
    b = enc_delta_of_delta(lam); header.insert(header.end(), b.begin(), b.end());
  }
  Bytes meta;
  {
    std::vector<int64_t> ts(t.timestamp, t.timestamp + N);
    Bytes b = enc_delta_of_delta(ts);
    meta.insert(meta.end(), b.begin(), b.end());
    std::vector<uint64_t> ml(t.msg_len, t.msg_len + N);
    b = enc_any_rle_uvar(ml);
    meta.insert(meta.end(), b.begin(), b.end());
    meta.insert(meta.end(), t.msgs, t.msgs + t.msgs_len);
  }
  Bytes cids;
  {
    put_uleb(cids, t.n_cids);
    for (size_t i = 0; i < t.n_cids; i++) {
      put_uleb(cids, 4);
      cids.push_back(t.cid_is_root[i] ? 1 : 0);
      cids.push_back(t.cid_kind[i]);
      put_uleb(cids, t.cid_peer_idx[i]);
      put_zigzag(cids, t.cid_key_or_counter[i]);
    }
  }
  Bytes keys;
  for (size_t i = 0; i < t.n_keys; i++) put_bytes(keys, t.keys[i], t.key_lens[i]);
  Bytes ops;
  {
    std::vector<int64_t> c(t.n_ops), p(t.n_ops);
    std::vector<uint64_t> l(t.n_ops);
    for (size_t i = 0; i < t.n_ops; i++) { c[i] = t.op_container[i]; p[i] = t.op_prop[i]; l[i] = t.op_len[i]; }
    put_uleb(ops, 1);
    put_uleb(ops, 4);
    put_bytes(ops, enc_delta_rle(c));
    put_bytes(ops, enc_delta_rle(p));
    put_bytes(ops, enc_rle_u8(t.op_value_type, t.n_ops));
    put_bytes(ops, enc_any_rle_uvar(l));
  }
  Bytes dels;
  if (t.n_dels) {   // no delete in the block: the section stays empty (block_encode.rs:237-244)
    std::vector<int64_t> a(t.n_dels), b(t.n_dels), c(t.n_dels);
    for (size_t i = 0; i < t.n_dels; i++) { a[i] = t.del_peer_idx[i]; b[i] = t.del_counter[i]; c[i] = t.del_len[i]; }
    put_uleb(dels, 1);
    put_uleb(dels, 3);
    put_bytes(dels, enc_delta_rle(a));
    put_bytes(dels, enc_delta_rle(b));
    put_bytes(dels, enc_delta_rle(c));
  }
  Bytes out;
  put_uleb(out, t.counter_start); put_uleb(out, t.counter_len); put_uleb(out, t.lamport_start); put_uleb(out, t.lamport_len); put_uleb(out, N);
  put_bytes(out, header); put_bytes(out, meta); put_bytes(out, cids); put_bytes(out, keys);
  put_bytes(out, t.positions, t.positions_len);
  put_bytes(out, ops); put_bytes(out, dels);
  put_bytes(out, t.values, t.values_len);
  return out;
}

// ---- a FastUpdates blob from encoded blocks (encoding.rs:440-473, fast_snapshot.rs:346-360)
inline Bytes encode_updates(const uint8_t* const* blocks, const size_t* lens, size_t n) {
  Bytes o(22, 0);
  memcpy(o.data(), "loro", 4);
  o[20] = 0; o[21] = 4;   // EncodeMode::FastUpdates, big endian
  for (size_t i = 0; i < n; i++) put_bytes(o, blocks[i], lens[i]);
  uint32_t h = xxh32(o.data() + 20, o.size() - 20, 0x4F524F4Cu);
  for (int k = 0; k < 4; k++) o[16 + k] = (uint8_t)(h >> (8 * k));
  return o;
}

}  // namespace lmenc
