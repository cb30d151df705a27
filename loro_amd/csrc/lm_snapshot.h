// FastSnapshot (EncodeMode 3) ingest — host C++, run by lm_stage before the blobs go to the device.
// A snapshot's first section is the ChangeStore: an SSTable (loro-kv-store) whose 12-byte keys hold the very same postcard
// change blocks a FastUpdates blob frames (docs/encoding.md §3-5).  This file reads that table — LZ4-framed blocks included —
// and reframes the change blocks as a FastUpdates blob, which is the path the reference itself takes when a snapshot is
// imported into a document that is not empty (`decode_oplog`, encoding/fast_snapshot.rs:326-344, loro.rs:582-638): only the
// ChangeStore is used as incoming changes.  The state sections are not read: the device replays the history.
//   envelope + three u32le-prefixed sections           docs/encoding.md §2-3, encoding/fast_snapshot.rs:47-95
//   SSTable: "LORO" 00 | blocks | metadata | u32le M   docs/encoding.md §4, crates/kv-store/src/sstable.rs:164-307,369-429
//   normal / large blocks, prefix-compressed keys      crates/kv-store/src/block.rs:18-228
//   LZ4 frame (lz4_flex 0.11.5 profile)                docs/encoding-lz4.md, crates/kv-store/src/compress.rs:7-69
//   ChangeStore keys: vv / fr / sv / sf / ID::to_bytes docs/encoding.md §5, oplog/change_store.rs:134-137,633-725
// Shallow snapshots (a non-empty third section: history trimmed below a shallow root) stay LM_UNSUPPORTED.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <map>
#include <string>
#include <algorithm>
#include "lm_encode.h"

namespace lmsnap {

enum { SN_OK = 0, SN_DECODE = 1, SN_CHECKSUM = 2, SN_UNSUPPORTED = 4 };

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// one raw LZ4 block (docs/encoding-lz4.md §4)
inline bool lz4_block(const uint8_t* p, size_t n, std::vector<uint8_t>& out, size_t base) {
  size_t i = 0;
  while (i < n) {
    uint8_t tok = p[i++];
    size_t lit = tok >> 4;
    if (lit == 15) { uint8_t e; do { if (i >= n) return false; e = p[i++]; lit += e; } while (e == 255); }
    if (lit > n - i || out.size() - base + lit > (4u << 20)) return false;
    out.insert(out.end(), p + i, p + i + lit);
    i += lit;
    if (i >= n) return true;   // the last sequence has literals only
    if (n - i < 2) return false;
    size_t off = rd16(p + i);
    i += 2;
    size_t ml = 4 + (tok & 15);
    if ((tok & 15) == 15) { uint8_t e; do { if (i >= n) return false; e = p[i++]; ml += e; } while (e == 255); }
    if (off == 0 || off > out.size() - base) return false;
    if (ml > (4u << 20) || out.size() - base + ml > (4u << 20)) return false;   // an LZ4 data block decodes to at most 4 MiB (BD 0x70)
    size_t s = out.size() - off;
    for (size_t k = 0; k < ml; k++) out.push_back(out[s + k]);   // may overlap its own output
  }
  return true;
}
// one LZ4 frame (§3): magic, FLG, BD, HC, data blocks, end mark; optional fields are parsed where the flags announce them
inline bool lz4_frame(const uint8_t* p, size_t n, std::vector<uint8_t>& out) {
  if (n < 7 || rd32(p) != 0x184D2204u) return false;
  uint8_t flg = p[4];
  if ((flg >> 6) != 1) return false;
  bool has_bsum = flg & 0x10, has_csize = flg & 0x08, has_csum = flg & 0x04, has_dict = flg & 0x01;
  size_t i = 6;
  if (has_csize) i += 8;
  if (has_dict) i += 4;
  i += 1;   // HC
  if (i > n) return false;
  for (;;) {
    if (n - i < 4) return false;
    uint32_t info = rd32(p + i);
    i += 4;
    if (info == 0) break;
    if (out.size() > (256u << 20)) return false;   // an SSTable block body is a few KiB; a frame that expands beyond this is not one of ours
    size_t len = info & 0x7fffffffu;
    if (len > n - i) return false;
    if (info & 0x80000000u) out.insert(out.end(), p + i, p + i + len);
    else if (!lz4_block(p + i, len, out, out.size())) return false;
    i += len;
    if (has_bsum) { if (n - i < 4) return false; i += 4; }
  }
  if (has_csum) { if (n - i < 4) return false; i += 4; }
  return true;
}

// every (key, value) of an SSTable, in key order
template <class F>
inline bool sstable_for_each(const uint8_t* p, size_t n, F&& f) {
  if (n == 0) return true;   // an empty KV store
  if (n < 13 || memcmp(p, "LORO", 4) != 0 || p[4] != 0) return false;
  size_t M = rd32(p + n - 4);
  if (M < 5 || M + 8 > n - 4) return false;   // metadata = count + entries + checksum, in front of the footer
  // the metadata (count + entries) is followed by its xxh32, every stored block by its own (crates/kv-store/src/sstable.rs:164-307,
  // block.rs:18-228; seed = "LORO" little endian, the envelope's)
  if (lmenc::xxh32(p + M + 4, n - 4 - 4 - (M + 4), 0x4F524F4Cu) != rd32(p + n - 8)) return false;   // (the entries, without the count in front: sstable.rs:89,110)
  size_t i = M;
  uint32_t nb = rd32(p + i);
  i += 4;
  if (nb == 0 || nb > 10000000u) return false;
  struct Meta { size_t off; const uint8_t* fk; size_t fkl; bool large; int comp; };
  std::vector<Meta> metas;
  for (uint32_t b = 0; b < nb; b++) {
    if (n - 4 - i < 7) return false;
    Meta m;
    m.off = rd32(p + i); i += 4;
    m.fkl = rd16(p + i); i += 2;
    if (m.fkl > n - 4 - i) return false;
    m.fk = p + i; i += m.fkl;
    if (i >= n - 4) return false;
    uint8_t flags = p[i++];
    m.large = flags & 0x80; m.comp = flags & 0x7f;
    if (m.comp > 1) return false;
    if (!m.large) { if (n - 4 - i < 2) return false; size_t lk = rd16(p + i); i += 2; if (lk > n - 4 - i) return false; i += lk; }
    if (m.off < 5 || m.off >= M || (!metas.empty() && m.off <= metas.back().off)) return false;
    metas.push_back(m);
  }
  std::vector<uint8_t> body, key;
  for (uint32_t b = 0; b < nb; b++) {
    size_t end = b + 1 < nb ? metas[b + 1].off : M;
    if (end - metas[b].off < 4) return false;
    const uint8_t* sp = p + metas[b].off;
    size_t sl = end - metas[b].off - 4;   // stored payload; the last four bytes are its xxh32
    if (lmenc::xxh32(sp, sl, 0x4F524F4Cu) != rd32(sp + sl)) return false;
    const uint8_t* bp = sp;
    size_t bl = sl;
    if (metas[b].comp == 1) { body.clear(); if (!lz4_frame(sp, sl, body)) return false; bp = body.data(); bl = body.size(); }
    if (metas[b].large) { f(metas[b].fk, metas[b].fkl, bp, bl); continue; }
    if (bl < 4) return false;
    size_t cnt = rd16(bp + bl - 2);
    if (cnt == 0 || 2 * cnt + 2 > bl) return false;
    size_t data_len = bl - 2 - 2 * cnt;
    const uint8_t* offs = bp + data_len;
    for (size_t e = 0; e < cnt; e++) {
      size_t o0 = rd16(offs + 2 * e), o1 = e + 1 < cnt ? rd16(offs + 2 * (e + 1)) : data_len;
      if (o0 > o1 || o1 > data_len) return false;
      const uint8_t* ep = bp + o0;
      size_t el = o1 - o0;
      if (e == 0) { f(metas[b].fk, metas[b].fkl, ep, el); continue; }
      if (el < 3) return false;
      size_t pre = ep[0], suf = rd16(ep + 1);
      if (pre > metas[b].fkl || 3 + suf > el) return false;
      key.assign(metas[b].fk, metas[b].fk + pre);
      key.insert(key.end(), ep + 3, ep + 3 + suf);
      f(key.data(), key.size(), ep + 3 + suf, el - 3 - suf);
    }
  }
  return true;
}

// a mode-3 blob → a FastUpdates blob holding its ChangeStore's change blocks (in key order: peer, counter)
// `roots` (optional) receives the root containers of the state section as [kind u8, uleb name_len, name]* — the keys of
// the state SSTable (docs/encoding-container-states.md §1.1: root key = kind | 0x80, uleb len, name).  An empty document
// importing the snapshot initialises its state store from that section (fast_snapshot.rs:168-258), so these roots are
// part of the value even when nothing is visible in them.
inline int snapshot_to_updates(const uint8_t* blob, size_t len, std::vector<uint8_t>& out, std::vector<uint8_t>* roots = nullptr, size_t* n_changes = nullptr) {
  if (len < 22 || memcmp(blob, "loro", 4) != 0) return SN_DECODE;
  if (blob[20] != 0 || blob[21] != 3) return SN_DECODE;
  if (lmenc::xxh32(blob + 20, len - 20, 0x4F524F4Cu) != rd32(blob + 16)) return SN_CHECKSUM;
  const uint8_t* p = blob + 22;
  size_t n = len - 22;
  const uint8_t* sec[3];
  size_t sl[3];
  for (int s = 0; s < 3; s++) {
    if (n < 4) return SN_DECODE;
    size_t l = rd32(p);
    p += 4; n -= 4;
    if (l > n) return SN_DECODE;
    sec[s] = p; sl[s] = l;
    p += l; n -= l;
  }
  if (n != 0) return SN_DECODE;
  if (sl[2] != 0) return SN_UNSUPPORTED;                 // shallow snapshot: history below the shallow root is gone
  std::vector<std::vector<uint8_t>> blocks;
  bool shallow_keys = false;
  bool ok = sstable_for_each(sec[0], sl[0], [&](const uint8_t* k, size_t kl, const uint8_t* v, size_t vl) {
    if (kl == 12) blocks.emplace_back(v, v + vl);
    else if (kl == 2 && (memcmp(k, "sv", 2) == 0 || memcmp(k, "sf", 2) == 0)) { if (!(vl == 1 && v[0] == 0)) shallow_keys = true; }
  });
  if (!ok) return SN_DECODE;
  if (shallow_keys) return SN_UNSUPPORTED;
  if (roots) {
    roots->clear();
    bool ok2 = sstable_for_each(sec[1], sl[1], [&](const uint8_t* k, size_t kl, const uint8_t*, size_t) {
      if (kl < 2 || !(k[0] & 0x80)) return;
      size_t i = 1, nl = 0, sh = 0;
      for (;;) { if (i >= kl || sh > 28) return; uint8_t c = k[i++]; nl |= (size_t)(c & 0x7f) << sh; sh += 7; if (!(c & 0x80)) break; }
      if (nl != kl - i) return;
      roots->push_back(k[0] & 0x7f);
      lmenc::put_uleb(*roots, nl);
      roots->insert(roots->end(), k + i, k + kl);
    });
    if (!ok2) return SN_DECODE;
  }
  if (n_changes) {   // Σ n_changes over the change blocks (the fifth postcard varint of a block, block_encode.rs:94-119)
    *n_changes = 0;
    for (auto& b : blocks) {
      size_t i = 0;
      uint64_t v = 0;
      for (int f = 0; f < 5; f++) { v = 0; int sh = 0; while (i < b.size()) { uint8_t c = b[i++]; v |= (uint64_t)(c & 0x7f) << sh; sh += 7; if (!(c & 0x80) || sh > 63) break; } }
      *n_changes += (size_t)v;
    }
  }
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> lens;
  for (auto& b : blocks) { ptrs.push_back(b.data()); lens.push_back(b.size()); }
  out = lmenc::encode_updates(ptrs.data(), lens.data(), blocks.size());
  return SN_OK;
}


// ---- N3, round 6: the STATE section as the materialised base (encoding/fast_snapshot.rs:168-258 — an empty document that imports
// a snapshot initialises its state store from that section and never replays the history; shallow snapshots, whose history below
// the shallow root is gone, have nothing else to be rendered from: shallow_snapshot.rs, docs/encoding.md "shallow root / overlay").
// For a document given as ONE snapshot rendered at its latest version, lm_stage hands the device — instead of the ChangeStore's whole
// history — a blob that holds the state itself: one synthetic change of one synthetic peer that writes every visible Map entry, every
// List / MovableList item and every Text as ONE op each (parents in front of their children, a child container addressed by the id
// of the synthetic op that creates it), which the pipeline replays as a linear history in time proportional to the STATE, not to the
// history.  The version vector of such a document is the snapshot's own (`vv` of its ChangeStore section), carried beside the blob
// and written out by the renderer in place of the synthetic peer's.  Formats: docs/encoding-container-states.md — ContainerID keys
// §1.1, ContainerWrapper §2, postcard LoroValue §3, Map §4, List §5, Text §6, MovableList §8 (ids, marks and tombstones are not read:
// nothing can refer to them when no update follows).  Anything this reader does not take — a state value it cannot parse (the
// placeholder states of loro_amd/wire.py's snapshot writer included), a container value that names a root — makes it decline, and
// the document goes through its ChangeStore as before (a shallow snapshot then stays LM_UNSUPPORTED).
struct PRd {
  const uint8_t* p; const uint8_t* end; bool bad = false;
  PRd(const uint8_t* a, size_t n) : p(a), end(a + n) {}
  uint8_t u8() { if (p >= end) { bad = true; return 0; } return *p++; }
  uint64_t uleb() { uint64_t v = 0; for (int sh = 0; sh < 70; sh += 7) { uint8_t c = u8(); if (bad) return 0; v |= (uint64_t)(c & 0x7f) << (sh < 64 ? sh : 63); if (!(c & 0x80)) return v; } bad = true; return 0; }
  int64_t zz() { uint64_t v = uleb(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
  const uint8_t* take(size_t n) { if ((size_t)(end - p) < n) { bad = true; return p; } const uint8_t* q = p; p += n; return q; }
};
struct StateWriter {
  std::vector<lmenc::Bytes> keys;                      // the synthetic block's key table
  std::map<std::string, uint32_t> key_idx;
  std::vector<uint8_t> cid_root, cid_kind; std::vector<uint32_t> cid_peer; std::vector<int32_t> cid_koc;
  std::vector<uint32_t> op_c, op_len; std::vector<int32_t> op_prop; std::vector<uint8_t> op_vt;
  lmenc::Bytes values;
  uint32_t ctr = 0;
  uint32_t key(const uint8_t* p, size_t n) {
    std::string k((const char*)p, n);
    auto it = key_idx.find(k);
    if (it != key_idx.end()) return it->second;
    uint32_t i = (uint32_t)keys.size();
    keys.emplace_back(p, p + n); key_idx.emplace(std::move(k), i);
    return i;
  }
  uint32_t cid(bool root, uint8_t kind, uint32_t key_or_counter) {
    for (size_t i = 0; i < cid_kind.size(); i++) if (cid_root[i] == (root ? 1 : 0) && cid_kind[i] == kind && cid_koc[i] == (int32_t)key_or_counter) return (uint32_t)i;
    cid_root.push_back(root ? 1 : 0); cid_kind.push_back(kind); cid_peer.push_back(0); cid_koc.push_back((int32_t)key_or_counter);
    return (uint32_t)cid_kind.size() - 1;
  }
};
static inline void put_sleb(lmenc::Bytes& o, int64_t v) {
  for (;;) { uint8_t b = v & 0x7f; v >>= 7; if ((v == 0 && !(b & 0x40)) || (v == -1 && (b & 0x40))) { o.push_back(b); return; } o.push_back(b | 0x80); }
}
// the historical kind byte of a postcard ContainerID (§2.1) -> the raw kind of the wire (§1.1)
static inline int hist_kind_to_raw(uint8_t h) { static const int m[6] = {2, 0, 1, 4, 3, 5}; return h < 6 ? m[h] : (int)h; }
// one postcard LoroValue -> the op-value codec (docs/encoding.md §10.1: tag + payload; map keys as indices of the block's key table).
// A Container value is written as `9, raw kind` and reported through `child` (its real id bytes in the state-key form of §1.1;
// `top` = it is the value itself, not nested inside a list / map value): -1 = none, else the offset of the real id in `child_id`
static inline bool pv_to_op(PRd& r, lmenc::Bytes& out, StateWriter& w, int depth, std::vector<lmenc::Bytes>* child_ids) {
  if (depth > 64) return false;
  uint64_t tag = r.uleb();
  if (r.bad) return false;
  switch (tag) {
    case 0: out.push_back(0); return true;
    case 1: { uint8_t b = r.u8(); if (r.bad || b > 1) return false; out.push_back(b ? 1 : 2); return true; }
    case 2: { const uint8_t* q = r.take(8); if (r.bad) return false; out.push_back(4); for (int k = 7; k >= 0; k--) out.push_back(q[k]); return true; }   // f64: postcard little endian, the op codec big endian
    case 3: { int64_t v = r.zz(); if (r.bad) return false; out.push_back(3); put_sleb(out, v); return true; }
    case 4: case 8: { uint64_t n = r.uleb(); const uint8_t* q = r.take((size_t)n); if (r.bad) return false; out.push_back(tag == 4 ? 5 : 6); lmenc::put_bytes(out, q, (size_t)n); return true; }
    case 5: {
      uint64_t n = r.uleb(); if (r.bad || n > (1u << 28)) return false;
      out.push_back(7); lmenc::put_uleb(out, n);
      for (uint64_t i = 0; i < n; i++) if (!pv_to_op(r, out, w, depth + 1, nullptr)) return false;
      return true;
    }
    case 6: {
      uint64_t n = r.uleb(); if (r.bad || n > (1u << 28)) return false;
      out.push_back(8); lmenc::put_uleb(out, n);
      for (uint64_t i = 0; i < n; i++) {
        uint64_t kl = r.uleb(); const uint8_t* kp = r.take((size_t)kl); if (r.bad) return false;
        lmenc::put_uleb(out, w.key(kp, (size_t)kl));
        if (!pv_to_op(r, out, w, depth + 1, nullptr)) return false;
      }
      return true;
    }
    case 7: {
      uint64_t variant = r.uleb();
      if (r.bad || variant != 1) return false;                 // (a value that names a ROOT container: mergeable containers — not this reader's)
      uint64_t peer = r.uleb(); int64_t c = r.zz(); uint8_t hk = r.u8();
      if (r.bad || c < 0 || c > 0x7fffffff) return false;
      int raw = hist_kind_to_raw(hk);
      out.push_back(9); out.push_back((uint8_t)raw);
      if (child_ids) {   // the child's state key (§1.1 normal form): kind, u64le peer, i32le counter
        lmenc::Bytes id(13);
        id[0] = (uint8_t)raw;
        for (int k = 0; k < 8; k++) id[1 + k] = (uint8_t)(peer >> (8 * k));
        for (int k = 0; k < 4; k++) id[9 + k] = (uint8_t)((uint32_t)c >> (8 * k));
        child_ids->push_back(std::move(id));
      } else return false;                                     // (a container nested inside a plain list / map VALUE: the device reports those LM_UNSUPPORTED anyway)
      return true;
    }
    default: return false;
  }
}
// a mode-3 blob -> (a FastUpdates blob that replays to its state, its version vector as the C ABI writes it); false: declined
// `root_only`: the state AT the shallow root (the third section alone) — what LoroDoc::checkout(shallow_since_frontiers) shows
// (loro_js_interop.rs:141-147); the oplog's version vector stays the whole snapshot's
// what a caller that stages UPDATES on top of the state needs to know about the base (lm_snapshot_base.h)
struct StateBase {
  uint64_t synth_peer = 0;                         // the peer of the synthetic change
  uint32_t synth_len = 0;                          // its ops: counters [0, synth_len)
  std::map<uint64_t, uint32_t> vv;                 // the snapshot's version vector (exclusive ends)
  std::vector<std::pair<uint64_t, uint32_t>> frontiers;   // … and frontiers (ChangeStore `fr`)
  std::map<lmenc::Bytes, uint32_t> child_ctr;      // real id of a child container (13-byte state key) -> counter of the synthetic op that creates it
  bool has_movable = false;                        // a MovableList holds items (their element ids are synthetic: move / set rows cannot name them)
};
inline bool snapshot_state_to_updates(const uint8_t* blob, size_t len, std::vector<uint8_t>& out, std::vector<uint8_t>& vv_out, std::vector<uint8_t>* roots, bool root_only = false, StateBase* sb = nullptr) {
  if (len < 22 || memcmp(blob, "loro", 4) != 0 || blob[20] != 0 || blob[21] != 3) return false;
  if (lmenc::xxh32(blob + 20, len - 20, 0x4F524F4Cu) != rd32(blob + 16)) return false;
  const uint8_t* p = blob + 22;
  size_t n = len - 22;
  const uint8_t* sec[3]; size_t sl[3];
  for (int s = 0; s < 3; s++) { if (n < 4) return false; size_t l = rd32(p); p += 4; n -= 4; if (l > n) return false; sec[s] = p; sl[s] = l; p += l; n -= l; }
  if (n != 0) return false;
  // the version vector of the oplog (ChangeStore key `vv`, VersionVector::encode: a postcard map) in the C ABI's order: ascending peer, no zero entry
  std::map<uint64_t, uint32_t> vv;
  std::vector<std::pair<uint64_t, uint32_t>> fr_ids;
  bool have_vv = false, have_fr = false, ok = sstable_for_each(sec[0], sl[0], [&](const uint8_t* k, size_t kl, const uint8_t* v, size_t vl) {
    if (kl == 2 && memcmp(k, "vv", 2) == 0) {
      PRd r(v, vl);
      uint64_t cnt = r.uleb();
      for (uint64_t i = 0; i < cnt && !r.bad; i++) { uint64_t peer = r.uleb(); int64_t c = r.zz(); if (c > 0 && c <= 0x7fffffff) vv[peer] = (uint32_t)c; else if (c != 0) r.bad = true; }
      have_vv = !r.bad && r.p == r.end;
    } else if (kl == 2 && memcmp(k, "fr", 2) == 0) {   // Frontiers::encode: postcard Vec<ID>
      PRd r(v, vl);
      uint64_t cnt = r.uleb();
      for (uint64_t i = 0; i < cnt && !r.bad; i++) { uint64_t peer = r.uleb(); int64_t c = r.zz(); if (c >= 0 && c <= 0x7fffffff) fr_ids.emplace_back(peer, (uint32_t)c); else r.bad = true; }
      have_fr = !r.bad && r.p == r.end;
    }
  });
  if (!ok || !have_vv) return false;
  if (sb && !have_fr) return false;
  // the state: the shallow root's entries first, the (end-)state's over them (docs/encoding-container-states.md §1); `fr` is not a container
  std::map<lmenc::Bytes, lmenc::Bytes> st;
  auto load = [&](const uint8_t* sp, size_t sn) {
    return sstable_for_each(sp, sn, [&](const uint8_t* k, size_t kl, const uint8_t* v, size_t vl) {
      if (kl == 2 && memcmp(k, "fr", 2) == 0) return;
      st[lmenc::Bytes(k, k + kl)] = lmenc::Bytes(v, v + vl);
    });
  };
  if (sl[2] && !load(sec[2], sl[2])) return false;
  if (root_only ? sl[2] == 0 : !load(sec[1], sl[1])) return false;
  // containers by depth: a parent's ops come in front of its children's
  struct Cont { const lmenc::Bytes* key; const lmenc::Bytes* val; uint8_t kind; uint64_t depth; size_t body; };
  std::vector<Cont> conts;
  if (roots) roots->clear();
  for (auto& kv : st) {
    const lmenc::Bytes& k = kv.first;
    if (k.empty()) return false;
    PRd r(kv.second.data(), kv.second.size());
    uint8_t kind = r.u8();
    uint64_t depth = r.uleb();
    uint64_t opt = r.uleb();
    if (r.bad || opt > 1) return false;
    if (opt == 1) {   // the parent: a postcard ContainerID — stepped over (children are found through their parents' values)
      uint64_t variant = r.uleb();
      if (variant == 0) { uint64_t nl = r.uleb(); r.take((size_t)nl); r.u8(); }
      else if (variant == 1) { r.uleb(); r.zz(); r.u8(); }
      else return false;
    }
    if (r.bad || kind != (k[0] & 0x7f)) return false;
    conts.push_back(Cont{&k, &kv.second, kind, depth, (size_t)(r.p - kv.second.data())});
    if ((k[0] & 0x80) && roots) {
      PRd kr(k.data() + 1, k.size() - 1);
      uint64_t nl = kr.uleb();
      if (kr.bad || nl != (uint64_t)(kr.end - kr.p)) return false;
      roots->push_back(kind);
      lmenc::put_uleb(*roots, nl);
      roots->insert(roots->end(), kr.p, kr.end);
    }
  }
  std::stable_sort(conts.begin(), conts.end(), [](const Cont& a, const Cont& b) { return a.depth < b.depth; });
  StateWriter w;
  std::map<lmenc::Bytes, uint32_t> child_ctr;            // real id of a child container -> the counter of the synthetic op that creates it
  for (const Cont& c : conts) {
    const lmenc::Bytes& k = *c.key;
    uint32_t ci;
    if (k[0] & 0x80) {
      PRd kr(k.data() + 1, k.size() - 1);
      uint64_t nl = kr.uleb();
      if (kr.bad) return false;
      ci = w.cid(true, c.kind, w.key(kr.p, (size_t)nl));
    } else {
      if (k.size() != 13) return false;
      auto it = child_ctr.find(k);
      if (it == child_ctr.end()) continue;               // nothing visible refers to it: unreachable from the value
      ci = w.cid(false, c.kind, it->second);
    }
    PRd r(c.val->data() + c.body, c.val->size() - c.body);
    std::vector<lmenc::Bytes> kids;
    if (c.kind == 0) {            // Map: postcard(FxHashMap<String, LoroValue>) visible_values, then metadata this reader does not need
      uint64_t cnt = r.uleb();
      if (r.bad || cnt > (1u << 28)) return false;
      for (uint64_t i = 0; i < cnt; i++) {
        uint64_t kl = r.uleb(); const uint8_t* kp = r.take((size_t)kl);
        if (r.bad) return false;
        kids.clear();
        const size_t v0 = w.values.size();
        if (!pv_to_op(r, w.values, w, 0, &kids)) return false;
        (void)v0;
        w.op_c.push_back(ci); w.op_prop.push_back((int32_t)w.key(kp, (size_t)kl)); w.op_vt.push_back(11); w.op_len.push_back(1);
        for (auto& id : kids) child_ctr[id] = w.ctr;       // (at most one: the value itself)
        w.ctr += 1;
      }
    } else if (c.kind == 1 || c.kind == 4) {   // List / MovableList: postcard(Vec<LoroValue>) visible_values — ONE insert of all items at position 0
      uint64_t cnt = r.uleb();
      if (r.bad || cnt > (1u << 24)) return false;
      if (cnt) {
        w.values.push_back(7); lmenc::put_uleb(w.values, cnt);
        for (uint64_t i = 0; i < cnt; i++) {
          kids.clear();
          if (!pv_to_op(r, w.values, w, 1, &kids)) return false;
          for (auto& id : kids) child_ctr[id] = w.ctr + (uint32_t)i;   // an item that is a container: created by the item's own op id
        }
        w.op_c.push_back(ci); w.op_prop.push_back(0); w.op_vt.push_back(11); w.op_len.push_back((uint32_t)cnt);
        w.ctr += (uint32_t)cnt;
        if (c.kind == 4 && sb) sb->has_movable = true;
      }
    } else if (c.kind == 2) {     // Text: postcard(String) full_text (spans, ids and marks behind it are not needed for the value)
      uint64_t bl = r.uleb(); const uint8_t* tp = r.take((size_t)bl);
      if (r.bad) return false;
      uint32_t scalars = 0;
      for (uint64_t i = 0; i < bl; i++) scalars += (tp[i] & 0xC0) != 0x80 ? 1u : 0u;
      if (scalars) {
        lmenc::put_bytes(w.values, tp, (size_t)bl);
        w.op_c.push_back(ci); w.op_prop.push_back(0); w.op_vt.push_back(5); w.op_len.push_back(scalars);
        w.ctr += scalars;
      }
    }
    // (Tree / Counter / unknown kinds: no op — a root of such a kind is known through `roots`, a child through its parent's value: both
    // render as null and flag the document LM_UNSUPPORTED like everywhere else)
    if (w.ctr > (1u << 24) - 16) return false;             // (the device's per-peer counter limit)
  }
  vv_out.clear();
  lmenc::put_uleb(vv_out, vv.size());
  for (auto& e : vv) { lmenc::put_uleb(vv_out, e.first); lmenc::put_uleb(vv_out, (uint64_t)e.second << 1); }
  if (w.ctr == 0) {   // nothing visible anywhere: an empty update blob (the roots still come through `roots`)
    if (sb) { sb->synth_len = 0; sb->vv = vv; sb->frontiers = fr_ids; }
    out = lmenc::encode_updates(nullptr, nullptr, 0);
    return true;
  }
  lm_block_tables t;
  memset(&t, 0, sizeof t);
  uint64_t peer = 1;
  if (sb) {   // (updates follow: the synthetic peer must be nobody's)
    peer = 0xFFFFFFFFFFFFFF00ull;
    while (vv.count(peer)) peer--;
    sb->synth_peer = peer; sb->synth_len = w.ctr; sb->vv = vv; sb->frontiers = fr_ids; sb->child_ctr = child_ctr;
  }
  const uint32_t one_len = w.ctr, zero32 = 0; const uint8_t zero8 = 0; const int64_t zero64 = 0;
  std::vector<const uint8_t*> kp; std::vector<size_t> kl;
  for (auto& k2 : w.keys) { kp.push_back(k2.data()); kl.push_back(k2.size()); }
  t.counter_start = 0; t.counter_len = w.ctr; t.lamport_start = 0; t.lamport_len = w.ctr; t.n_changes = 1;
  t.peers = &peer; t.n_peers = 1;
  t.change_len = &one_len; t.dep_on_self = &zero8; t.dep_count = &zero32; t.dep_peer_idx = nullptr; t.dep_counter = nullptr; t.n_deps = 0;
  t.lamport = &zero32; t.timestamp = &zero64; t.msg_len = &zero32; t.msgs = nullptr; t.msgs_len = 0;
  t.cid_is_root = w.cid_root.data(); t.cid_kind = w.cid_kind.data(); t.cid_peer_idx = w.cid_peer.data(); t.cid_key_or_counter = w.cid_koc.data(); t.n_cids = w.cid_kind.size();
  t.keys = kp.data(); t.key_lens = kl.data(); t.n_keys = kp.size();
  t.positions = nullptr; t.positions_len = 0;
  t.op_container = w.op_c.data(); t.op_prop = w.op_prop.data(); t.op_value_type = w.op_vt.data(); t.op_len = w.op_len.data(); t.n_ops = w.op_len.size();
  t.values = w.values.data(); t.values_len = w.values.size();
  lmenc::Bytes blk = lmenc::encode_block(t);
  const uint8_t* bp = blk.data(); size_t bl = blk.size();
  out = lmenc::encode_updates(&bp, &bl, 1);
  return true;
}

// the frontiers of a shallow snapshot's root (the `fr` entry of its third section: Frontiers::encode, shallow_snapshot.rs:174) — the
// bytes a caller passes as checkout_frontiers to see the document at its shallow root; false: not a shallow snapshot / unreadable
inline bool snapshot_shallow_root_frontiers(const uint8_t* blob, size_t len, std::vector<uint8_t>& fr) {
  if (len < 22 || memcmp(blob, "loro", 4) != 0 || blob[20] != 0 || blob[21] != 3) return false;
  const uint8_t* p = blob + 22;
  size_t n = len - 22;
  const uint8_t* sec[3]; size_t sl[3];
  for (int s = 0; s < 3; s++) { if (n < 4) return false; size_t l = rd32(p); p += 4; n -= 4; if (l > n) return false; sec[s] = p; sl[s] = l; p += l; n -= l; }
  if (n != 0 || sl[2] == 0) return false;
  bool have = false;
  bool ok = sstable_for_each(sec[2], sl[2], [&](const uint8_t* k, size_t kl, const uint8_t* v, size_t vl) {
    if (kl == 2 && memcmp(k, "fr", 2) == 0) { fr.assign(v, v + vl); have = true; }
  });
  return ok && have;
}

}  // namespace lmsnap
