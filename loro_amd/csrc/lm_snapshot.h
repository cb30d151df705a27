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

}  // namespace lmsnap
