// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// FastSnapshot (EncodeMode 3) reader of the checker: the ChangeStore section's change blocks and the root containers of the
// state section.  Restates (reference file:line, relative to /root/reference):
//   three u32le-prefixed sections, exact EOF               crates/loro-internal/src/encoding/fast_snapshot.rs:47-95, docs/encoding.md §3
//   SSTable: magic, schema, blocks, metadata, footer        crates/kv-store/src/sstable.rs:41-147,369-451, docs/encoding.md §4.1-4.2
//   normal block (prefix-compressed keys) / large block     crates/kv-store/src/block.rs:18-228, docs/encoding.md §4.3-4.4
//   block + metadata checksums (xxh32, seed "LORO")         crates/kv-store/src/sstable.rs:89,110,516-540; MemKvStore::import_all verifies them (mem_store.rs:282-290)
//   LZ4 frame / block                                       crates/kv-store/src/compress.rs:7-69, docs/encoding-lz4.md
//   ChangeStore keys (vv, fr, sv, sf, 12-byte ID)           crates/loro-internal/src/oplog/change_store.rs:134-137,633-725, docs/encoding.md §5
//   state section: root key = kind|0x80, uleb len, name     docs/encoding-container-states.md §1.1; an EMPTY document initialises its
//                                                           state store from it (fast_snapshot.rs:168-258)
// The state VALUES are not read: the checker replays the history (decode_oplog, fast_snapshot.rs:326-344).
#pragma once
#include <map>
#include "lo_codec.hpp"

namespace lo {

struct SnapshotParts {
  std::vector<Change> changes;                               // every change of the ChangeStore, block by block in key order
  std::vector<std::pair<uint8_t, std::string>> roots;        // (kind, name) of the root containers the state section holds
  size_t n_changes = 0;
};

inline void lz4_block_into(Reader r, std::string& out, size_t base) {
  while (!r.eof()) {
    uint8_t tok = r.u8();
    size_t lit = tok >> 4;
    if (lit == 15) for (;;) { uint8_t e = r.u8(); lit += e; if (e != 255) break; }
    if (lit > r.remaining()) fail(ST_DECODE_ERROR, "lz4 literals beyond the block");
    out.append((const char*)r.p, lit);
    r.p += lit;
    if (r.eof()) return;                                       // the last sequence carries literals only
    if (r.remaining() < 2) fail(ST_DECODE_ERROR, "lz4 offset missing");
    size_t off = (size_t)r.p[0] | ((size_t)r.p[1] << 8);
    r.p += 2;
    size_t ml = (tok & 15);
    if (ml == 15) for (;;) { uint8_t e = r.u8(); ml += e; if (e != 255) break; }
    ml += 4;
    if (off == 0 || off > out.size() - base) fail(ST_DECODE_ERROR, "lz4 offset out of range");
    if (out.size() - base + ml > (4u << 20)) fail(ST_DECODE_ERROR, "lz4 block too large");
    size_t from = out.size() - off;
    for (size_t k = 0; k < ml; k++) out.push_back(out[from + k]);
  }
}
inline std::string lz4_frame(Reader r) {
  if (r.remaining() < 7 || rd32le(r.p) != 0x184D2204u) fail(ST_DECODE_ERROR, "lz4 frame magic");
  r.p += 4;
  uint8_t flg = r.u8();
  (void)r.u8();   // BD
  if ((flg >> 6) != 1) fail(ST_DECODE_ERROR, "lz4 frame version");
  if (flg & 0x08) { if (r.remaining() < 8) fail(ST_DECODE_ERROR, "lz4 frame"); r.p += 8; }
  if (flg & 0x01) { if (r.remaining() < 4) fail(ST_DECODE_ERROR, "lz4 frame"); r.p += 4; }
  (void)r.u8();   // header checksum
  std::string out;
  for (;;) {
    if (r.remaining() < 4) fail(ST_DECODE_ERROR, "lz4 frame truncated");
    uint32_t info = rd32le(r.p);
    r.p += 4;
    if (info == 0) break;
    size_t len = info & 0x7fffffffu;
    if (len > r.remaining()) fail(ST_DECODE_ERROR, "lz4 data block beyond the frame");
    if (info & 0x80000000u) out.append((const char*)r.p, len);
    else lz4_block_into(Reader(r.p, len), out, out.size());
    r.p += len;
    if (flg & 0x10) { if (r.remaining() < 4) fail(ST_DECODE_ERROR, "lz4 frame"); r.p += 4; }
    if (out.size() > (256u << 20)) fail(ST_DECODE_ERROR, "lz4 frame too large");
  }
  return out;
}

// every (key, value) of one SSTable section, in key order
inline std::vector<std::pair<std::string, std::string>> sstable_entries(const uint8_t* p, size_t n) {
  std::vector<std::pair<std::string, std::string>> out;
  if (n == 0) return out;                                       // an empty KV store
  if (n < 13 || memcmp(p, "LORO", 4) != 0 || p[4] != 0) fail(ST_DECODE_ERROR, "sstable magic / schema");
  size_t M = rd32le(p + n - 4);
  if (M < 5 || M + 8 > n - 4) fail(ST_DECODE_ERROR, "sstable metadata offset");
  const uint8_t* meta = p + M;
  size_t meta_len = n - 4 - M;                                  // count | entries | checksum
  if (xxh32(meta + 4, meta_len - 8, LORO_XXH_SEED) != rd32le(meta + meta_len - 4)) fail(ST_DECODE_ERROR, "sstable metadata checksum");
  Reader r(meta + 4, meta_len - 8);
  uint32_t nb = rd32le(meta);
  if (nb == 0 || nb > 10000000u) fail(ST_DECODE_ERROR, "sstable block count");
  struct BM { size_t off; std::string first; bool large; int comp; };
  std::vector<BM> bms;
  for (uint32_t b = 0; b < nb; b++) {
    if (r.remaining() < 7) fail(ST_DECODE_ERROR, "sstable metadata entry");
    BM m;
    m.off = rd32le(r.p); r.p += 4;
    size_t fkl = (size_t)r.p[0] | ((size_t)r.p[1] << 8); r.p += 2;
    if (fkl > r.remaining()) fail(ST_DECODE_ERROR, "sstable first key");
    m.first.assign((const char*)r.p, fkl); r.p += fkl;
    uint8_t flags = r.u8();
    m.large = flags & 0x80; m.comp = flags & 0x7f;
    if (m.comp > 1) fail(ST_DECODE_ERROR, "sstable compression type");
    if (!m.large) {
      if (r.remaining() < 2) fail(ST_DECODE_ERROR, "sstable last key");
      size_t lkl = (size_t)r.p[0] | ((size_t)r.p[1] << 8); r.p += 2;
      if (lkl > r.remaining()) fail(ST_DECODE_ERROR, "sstable last key");
      r.p += lkl;
    }
    if (m.off < 5 || m.off >= M || (!bms.empty() && m.off <= bms.back().off)) fail(ST_DECODE_ERROR, "sstable block offsets");
    bms.push_back(m);
  }
  for (uint32_t b = 0; b < nb; b++) {
    size_t end = b + 1 < nb ? bms[b + 1].off : M;
    if (end - bms[b].off < 4) fail(ST_DECODE_ERROR, "sstable block too short");
    const uint8_t* sp = p + bms[b].off;
    size_t sl = end - bms[b].off - 4;
    if (xxh32(sp, sl, LORO_XXH_SEED) != rd32le(sp + sl)) fail(ST_DECODE_ERROR, "sstable block checksum");
    std::string body = bms[b].comp == 1 ? lz4_frame(Reader(sp, sl)) : std::string((const char*)sp, sl);
    if (bms[b].large) { out.push_back({bms[b].first, body}); continue; }
    if (body.size() < 4) fail(ST_DECODE_ERROR, "sstable normal block");
    const uint8_t* bp = (const uint8_t*)body.data();
    size_t bl = body.size();
    size_t cnt = (size_t)bp[bl - 2] | ((size_t)bp[bl - 1] << 8);
    if (cnt == 0 || 2 * cnt + 2 > bl) fail(ST_DECODE_ERROR, "sstable entry count");
    size_t data_len = bl - 2 - 2 * cnt;
    auto off_at = [&](size_t e) { return (size_t)bp[data_len + 2 * e] | ((size_t)bp[data_len + 2 * e + 1] << 8); };
    for (size_t e = 0; e < cnt; e++) {
      size_t o0 = off_at(e), o1 = e + 1 < cnt ? off_at(e + 1) : data_len;
      if (o0 > o1 || o1 > data_len) fail(ST_DECODE_ERROR, "sstable entry offsets");
      if (e == 0) { out.push_back({bms[b].first, body.substr(o0, o1 - o0)}); continue; }
      if (o1 - o0 < 3) fail(ST_DECODE_ERROR, "sstable entry");
      size_t pre = bp[o0], suf = (size_t)bp[o0 + 1] | ((size_t)bp[o0 + 2] << 8);
      if (pre > bms[b].first.size() || 3 + suf > o1 - o0) fail(ST_DECODE_ERROR, "sstable key prefix");
      out.push_back({bms[b].first.substr(0, pre) + body.substr(o0 + 3, suf), body.substr(o0 + 3 + suf, o1 - o0 - 3 - suf)});
    }
  }
  return out;
}

inline uint16_t blob_mode(const uint8_t* blob, size_t len) { return len >= 22 ? (uint16_t)((blob[20] << 8) | blob[21]) : 0; }

inline void decode_snapshot_blob(const uint8_t* blob, size_t len, SnapshotParts& out) {
  if (len < 22) fail(ST_DECODE_ERROR, "Invalid import data");
  if (memcmp(blob, "loro", 4) != 0) fail(ST_DECODE_ERROR, "Invalid magic");
  if (blob_mode(blob, len) != 3) fail(ST_DECODE_ERROR, "not a FastSnapshot");
  if (xxh32(blob + 20, len - 20, LORO_XXH_SEED) != rd32le(blob + 16)) fail(ST_CHECKSUM_MISMATCH, "checksum mismatch");
  const uint8_t* p = blob + 22;
  size_t n = len - 22;
  const uint8_t* sec[3];
  size_t sl[3];
  for (int s = 0; s < 3; s++) {
    if (n < 4) fail(ST_DECODE_ERROR, "snapshot section length");
    size_t l = rd32le(p);
    p += 4; n -= 4;
    if (l > n) fail(ST_DECODE_ERROR, "snapshot section beyond the blob");
    sec[s] = p; sl[s] = l;
    p += l; n -= l;
  }
  if (n != 0) fail(ST_DECODE_ERROR, "bytes after the third section");
  if (sl[2] != 0) fail(ST_UNSUPPORTED, "shallow snapshot: history below the shallow root is gone");
  for (auto& kv : sstable_entries(sec[0], sl[0])) {
    if (kv.first.size() == 12) {
      size_t before = out.changes.size();
      decode_block(Reader((const uint8_t*)kv.second.data(), kv.second.size()), out.changes);
      out.n_changes += out.changes.size() - before;
    } else if (kv.first == "sv" || kv.first == "sf") {
      if (!(kv.second.size() == 1 && kv.second[0] == 0)) fail(ST_UNSUPPORTED, "shallow snapshot");
    }
  }
  for (auto& kv : sstable_entries(sec[1], sl[1])) {
    const std::string& k = kv.first;
    if (k.size() < 2 || !((uint8_t)k[0] & 0x80)) continue;     // normal (child) containers, `fr`
    Reader r((const uint8_t*)k.data() + 1, k.size() - 1);
    uint64_t nl = 0;
    try { nl = r.uleb(); } catch (const DecodeErr&) { continue; }
    if (nl != r.remaining()) continue;
    out.roots.push_back({(uint8_t)((uint8_t)k[0] & 0x7f), std::string((const char*)r.p, (size_t)nl)});
  }
}

}  // namespace lo
