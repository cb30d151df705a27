// ORACLE — TEST INFRASTRUCTURE ONLY.
// CPU restatement of Loro's FastUpdates (mode 4) wire decoder.  Nothing under oracle/ is part of
// the shipped MI355X path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load it.
//
// Restates (reference file:line, relative to /root/reference):
//   envelope + checksum      crates/loro-internal/src/encoding.rs:302-373
//   updates framing          crates/loro-internal/src/encoding/fast_snapshot.rs:372-400
//   block struct             crates/loro-internal/src/oplog/change_store/block_encode.rs:94-119
//   block header             crates/loro-internal/src/oplog/change_store/block_meta_encode.rs:90-242
//   op columns / row walk    crates/loro-internal/src/oplog/change_store/block_encode.rs:417-445,535-706
//   op/value mapping         crates/loro-internal/src/encoding/outdated_encode_reordered.rs:215-476
//   container arena          crates/loro-internal/src/encoding/arena.rs:39-105
//   value tags               crates/loro-internal/src/encoding/value.rs:39-161,342-459,608-859
// Third-party arithmetic not in tree (restated from docs/encoding.md §1,§8 and
// docs/encoding-xxhash32.md; pinned by loro-js/tests/serde-columnar.test.ts:27-109 known answers and
// by the Rust-written fixture loro-js/tests/fixtures/rust/updates.blob):
//   serde_columnar 0.3.14 (BoolRle/AnyRle/DeltaRle/DeltaOfDelta), postcard 1.1.3, xxhash-rust 0.8.15.
#pragma once
#include "../include/loro_block_tables.h"
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <memory>
#include <algorithm>

namespace lo {

typedef uint64_t PeerID;
typedef int32_t Counter;
typedef uint32_t Lamport;

// status codes shared with include/loro_merge.h (LM_*)
enum Status : int32_t {
  ST_OK = 0,
  ST_DECODE_ERROR = 1,            // LoroError::DecodeError
  ST_CHECKSUM_MISMATCH = 2,       // LoroError::DecodeChecksumMismatchError
  ST_DATA_CORRUPTION = 3,         // LoroError::DecodeDataCorruptionError
  ST_UNSUPPORTED = 4,             // container kind / feature outside the hot-path scope
  ST_INTERNAL = 5,
  ST_FRONTIERS_NOT_FOUND = 6,     // LoroError::FrontiersNotFound (checkout target outside the OpLog)
};

struct DecodeErr {
  int32_t st;
  const char* what;
};
[[noreturn]] inline void fail(int32_t st, const char* what) { throw DecodeErr{st, what}; }

struct ID {
  PeerID peer;
  Counter counter;
  bool operator==(const ID& o) const { return peer == o.peer && counter == o.counter; }
  bool operator!=(const ID& o) const { return !(*this == o); }
};

// ---------------------------------------------------------------- xxHash32 (docs/encoding-xxhash32.md)
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t rd32le(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint32_t xxh32(const uint8_t* p, size_t len, uint32_t seed) {
  const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
  const uint8_t* end = p + len;
  uint32_t h;
  if (len >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* limit = end - 16;
    do {
      v1 = rotl32(v1 + rd32le(p) * P2, 13) * P1; p += 4;
      v2 = rotl32(v2 + rd32le(p) * P2, 13) * P1; p += 4;
      v3 = rotl32(v3 + rd32le(p) * P2, 13) * P1; p += 4;
      v4 = rotl32(v4 + rd32le(p) * P2, 13) * P1; p += 4;
    } while (p <= limit);
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) { h = rotl32(h + rd32le(p) * P3, 17) * P4; p += 4; }
  while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}
static const uint32_t LORO_XXH_SEED = 0x4f524f4cu;  // LE("LORO"), encoding.rs:302

// ---------------------------------------------------------------- byte reader
struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  size_t remaining() const { return (size_t)(end - p); }
  bool eof() const { return p >= end; }
  uint8_t u8() {
    if (p >= end) fail(ST_DECODE_ERROR, "unexpected eof");
    return *p++;
  }
  // unsigned LEB128 / postcard varint (same grammar); max 10 bytes for u64
  uint64_t uleb() {
    uint64_t v = 0;
    for (int i = 0; i < 10; i++) {
      uint8_t b = u8();
      v |= (uint64_t)(b & 0x7f) << (7 * i);
      if (!(b & 0x80)) return v;
    }
    fail(ST_DECODE_ERROR, "varint overflow");
  }
  // u128 varint truncated to the low 64 bits + a flag whether the value fits in i64 after zigzag
  // (DeltaRle deltas are i128 on the wire; every in-scope column fits i64).
  int64_t zigzag128_as_i64() {
    unsigned __int128 v = 0;
    for (int i = 0; i < 19; i++) {
      uint8_t b = u8();
      v |= (unsigned __int128)(b & 0x7f) << (7 * i);
      if (!(b & 0x80)) {
        __int128 s = (__int128)(v >> 1) ^ -(__int128)(v & 1);
        if (s > INT64_MAX || s < INT64_MIN) fail(ST_DECODE_ERROR, "delta out of range");
        return (int64_t)s;
      }
    }
    fail(ST_DECODE_ERROR, "varint128 overflow");
  }
  int64_t zigzag() {
    uint64_t v = uleb();
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  // signed LEB128 (leb128 crate), used by value payloads I64/DeltaInt
  int64_t sleb() {
    int64_t result = 0;
    int shift = 0;
    uint8_t b;
    do {
      b = u8();
      if (shift < 64) result |= (int64_t)(uint64_t)(b & 0x7f) << shift;
      shift += 7;
      if (shift > 70) fail(ST_DECODE_ERROR, "sleb overflow");
    } while (b & 0x80);
    if (shift < 64 && (b & 0x40)) result |= -((int64_t)1 << shift);
    return result;
  }
  Reader bytes() {  // uleb length + bytes
    uint64_t n = uleb();
    if (n > remaining()) fail(ST_DECODE_ERROR, "bytes field beyond input");
    Reader r(p, (size_t)n);
    p += n;
    return r;
  }
  const uint8_t* take(size_t n) {
    if (n > remaining()) fail(ST_DECODE_ERROR, "take beyond input");
    const uint8_t* q = p;
    p += n;
    return q;
  }
};

// ---------------------------------------------------------------- serde_columnar strategies
// BoolRle: alternating run lengths starting with a FALSE run (docs/encoding.md:700-701)
inline std::vector<uint8_t> take_bool_rle(Reader& r, size_t n) {
  std::vector<uint8_t> out;
  out.reserve(n);
  bool v = false;
  while (out.size() < n) {
    uint64_t run = r.uleb();
    if (out.size() + run > n) fail(ST_DECODE_ERROR, "BoolRle too many");
    out.insert(out.end(), (size_t)run, v ? 1 : 0);
    v = !v;
  }
  return out;
}

// AnyRle<T>: zigzag k; k>0 → one value repeated k; k<0 → |k| literals (docs/encoding.md:703-708)
template <class F>
inline void any_rle_segment(Reader& r, size_t limit, size_t& count, F&& emit_value_reader) {
  int64_t k = r.zigzag();
  if (k == 0) fail(ST_DECODE_ERROR, "AnyRle zero segment");
  uint64_t len = k < 0 ? (uint64_t)(-k) : (uint64_t)k;
  if (count + len > limit) fail(ST_DECODE_ERROR, "AnyRle too many");
  emit_value_reader(k > 0, (size_t)len);
  count += len;
}
inline std::vector<uint64_t> take_any_rle_uvar(Reader& r, size_t n) {
  std::vector<uint64_t> out;
  size_t cnt = 0;
  while (cnt < n) {
    if (r.eof()) fail(ST_DECODE_ERROR, "AnyRle too few");
    any_rle_segment(r, n, cnt, [&](bool run, size_t len) {
      if (run) {
        uint64_t v = r.uleb();
        out.insert(out.end(), len, v);
      } else
        for (size_t i = 0; i < len; i++) out.push_back(r.uleb());
    });
  }
  return out;
}
// whole-payload decoders (column payload ends at reader end)
inline std::vector<uint64_t> decode_any_rle_uvar(Reader r) {
  std::vector<uint64_t> out;
  size_t cnt = 0;
  while (!r.eof())
    any_rle_segment(r, (size_t)1 << 31, cnt, [&](bool run, size_t len) {
      if (run) {
        uint64_t v = r.uleb();
        out.insert(out.end(), len, v);
      } else
        for (size_t i = 0; i < len; i++) out.push_back(r.uleb());
    });
  return out;
}
inline std::vector<uint8_t> decode_any_rle_u8(Reader r) {  // Rle<u8>: values are raw bytes
  std::vector<uint8_t> out;
  size_t cnt = 0;
  while (!r.eof())
    any_rle_segment(r, (size_t)1 << 31, cnt, [&](bool run, size_t len) {
      if (run) {
        uint8_t v = r.u8();
        out.insert(out.end(), len, v);
      } else
        for (size_t i = 0; i < len; i++) out.push_back(r.u8());
    });
  return out;
}
// DeltaRle: AnyRle<i128> over first differences, first delta from 0 (docs/encoding.md:710-711)
inline std::vector<int64_t> decode_delta_rle(Reader r) {
  std::vector<int64_t> out;
  size_t cnt = 0;
  int64_t cur = 0;
  while (!r.eof())
    any_rle_segment(r, (size_t)1 << 31, cnt, [&](bool run, size_t len) {
      // (the reference sums in i128 and fails when narrowing to the column type; no in-scope column exceeds i64)
      auto add = [&](int64_t d) { if (__builtin_add_overflow(cur, d, &cur)) fail(ST_DECODE_ERROR, "DeltaRle sum beyond i64"); out.push_back(cur); };
      if (run) {
        int64_t d = r.zigzag128_as_i64();
        for (size_t i = 0; i < len; i++) add(d);
      } else
        for (size_t i = 0; i < len; i++) add(r.zigzag128_as_i64());
    });
  return out;
}

// DeltaOfDelta (docs/encoding.md:713-726): Option<i64> first | u8 used-bits-of-last-byte | MSB-first bits
struct BitReader {
  const uint8_t* p;
  size_t nbits, pos;
  uint64_t bits(int n) {
    if (pos + n > nbits) fail(ST_DECODE_ERROR, "DeltaOfDelta bitstream eof");
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
      v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1);
      pos++;
    }
    return v;
  }
};
inline int64_t dod_value(BitReader& b) {
  if (b.bits(1) == 0) return 0;
  if (b.bits(1) == 0) return (int64_t)b.bits(7) - 63;
  if (b.bits(1) == 0) return (int64_t)b.bits(9) - 255;
  if (b.bits(1) == 0) return (int64_t)b.bits(12) - 2047;
  if (b.bits(1) == 0) return (int64_t)b.bits(21) - 1048575;
  return (int64_t)b.bits(64);
}
inline std::vector<int64_t> take_delta_of_delta(Reader& r, size_t n) {
  std::vector<int64_t> out;
  uint8_t tag = r.u8();
  bool has_first = false;
  int64_t first = 0;
  if (tag == 1) { has_first = true; first = r.zigzag(); }
  else if (tag != 0) fail(ST_DECODE_ERROR, "DeltaOfDelta option tag");
  uint8_t last_used = r.u8();
  if (!has_first) {
    if (n != 0) fail(ST_DECODE_ERROR, "DeltaOfDelta too few");
    if (last_used != 0) fail(ST_DECODE_ERROR, "DeltaOfDelta empty with bits");
    return out;
  }
  if (n == 0) fail(ST_DECODE_ERROR, "DeltaOfDelta too many");
  BitReader b{r.p, r.remaining() * 8, 0};
  out.push_back(first);
  int64_t prev = first, delta = 0;
  while (out.size() < n) {
    delta = (int64_t)((uint64_t)delta + (uint64_t)dod_value(b));   // wrapping, like the release-mode Rust reader (damaged input only)
    prev = (int64_t)((uint64_t)prev + (uint64_t)delta);
    out.push_back(prev);
  }
  if (n == 1) {
    if (last_used != 0) fail(ST_DECODE_ERROR, "DeltaOfDelta single-value bits");
  } else {
    unsigned expect = (unsigned)(b.pos % 8 ? b.pos % 8 : 8);
    if (last_used != expect) fail(ST_DECODE_ERROR, "DeltaOfDelta last-used-bits mismatch");
  }
  r.p += (b.pos + 7) / 8;
  return out;
}

// ---------------------------------------------------------------- values
enum ContainerKind : uint8_t { CK_MAP = 0, CK_LIST = 1, CK_TEXT = 2, CK_TREE = 3, CK_MOVABLE = 4, CK_COUNTER = 5 };

struct ContainerID {
  bool root = true;
  uint8_t kind = 0;
  std::string name;  // root
  PeerID peer = 0;   // normal
  Counter counter = 0;
  bool operator==(const ContainerID& o) const {
    return root == o.root && kind == o.kind && (root ? name == o.name : (peer == o.peer && counter == o.counter));
  }
  bool operator<(const ContainerID& o) const {
    if (root != o.root) return root > o.root;
    if (kind != o.kind) return kind < o.kind;
    if (root) return name < o.name;
    if (peer != o.peer) return peer < o.peer;
    return counter < o.counter;
  }
};

enum ValueKindTag : uint8_t { V_NULL = 0, V_BOOL, V_I64, V_F64, V_STR, V_BIN, V_LIST, V_MAP, V_CONTAINER };
struct Value {
  uint8_t kind = V_NULL;
  bool b = false;
  int64_t i = 0;
  double f = 0;
  std::string s;  // string / binary
  std::vector<Value> list;
  std::vector<std::pair<std::string, Value>> map;
  ContainerID cid;
};

struct DecodeArena {
  const std::vector<PeerID>* peers;
  const std::vector<std::string>* keys;
};

// nested LoroValue (docs/encoding.md §10.1; value.rs:608-859).  `id` is the contextual op id;
// direct elements of a top-level list take id.inc(i).
inline Value read_loro_value(Reader& r, const DecodeArena& a, ID id, int depth, bool top) {
  if (depth > 256) fail(ST_DATA_CORRUPTION, "value nesting too deep");
  Value v;
  uint8_t tag = r.u8();
  switch (tag) {
    case 0: v.kind = V_NULL; break;
    case 1: v.kind = V_BOOL; v.b = true; break;
    case 2: v.kind = V_BOOL; v.b = false; break;
    case 3: v.kind = V_I64; v.i = r.sleb(); break;
    case 4: {
      v.kind = V_F64;
      const uint8_t* q = r.take(8);
      uint64_t bits = 0;
      for (int i = 0; i < 8; i++) bits = (bits << 8) | q[i];  // big-endian
      memcpy(&v.f, &bits, 8);
      break;
    }
    case 5: { v.kind = V_STR; Reader s = r.bytes(); v.s.assign((const char*)s.p, s.remaining()); break; }
    case 6: { v.kind = V_BIN; Reader s = r.bytes(); v.s.assign((const char*)s.p, s.remaining()); break; }
    case 7: {
      v.kind = V_LIST;
      uint64_t n = r.uleb();
      if (n > (1u << 28)) fail(ST_DATA_CORRUPTION, "collection too large");
      for (uint64_t i = 0; i < n; i++) {
        ID eid = top ? ID{id.peer, (Counter)(id.counter + (Counter)i)} : id;
        v.list.push_back(read_loro_value(r, a, eid, depth + 1, false));
      }
      break;
    }
    case 8: {
      v.kind = V_MAP;
      uint64_t n = r.uleb();
      if (n > (1u << 28)) fail(ST_DATA_CORRUPTION, "collection too large");
      for (uint64_t i = 0; i < n; i++) {
        uint64_t k = r.uleb();
        if (k >= a.keys->size()) fail(ST_DATA_CORRUPTION, "key index");
        Value e = read_loro_value(r, a, id, depth + 1, false);
        v.map.emplace_back((*a.keys)[(size_t)k], std::move(e));
      }
      break;
    }
    case 9: {
      v.kind = V_CONTAINER;
      v.cid.root = false;
      v.cid.kind = r.u8();
      v.cid.peer = id.peer;
      v.cid.counter = id.counter;
      break;
    }
    default: fail(ST_DATA_CORRUPTION, "unknown LoroValue tag");
  }
  return v;
}

// ---------------------------------------------------------------- ops / changes
enum OpKind : uint8_t {
  OP_TEXT_INSERT, OP_SEQ_DELETE, OP_STYLE_START, OP_STYLE_END, OP_LIST_INSERT, OP_MAP_SET, OP_MAP_DELETE, OP_OTHER,
  OP_LIST_MOVE, OP_LIST_SET   // MovableList (docs/encoding.md §10.5; outdated_encode_reordered.rs:388-459)
};
struct Op {
  uint32_t container;  // index into Doc-level container table after registration (filled by caller)
  Counter counter;
  int32_t len;         // atom length from the wire `len` column
  uint8_t kind = OP_OTHER;
  int32_t prop = 0;
  // text insert
  std::vector<uint32_t> cps;  // unicode scalars
  // delete
  ID del_id_start{0, 0};
  int64_t del_signed_len = 0;
  // style start: end position
  uint32_t style_end = 0;
  // list insert
  std::vector<Value> values;
  // map (and the value of a MovableList set)
  std::string key;
  Value value;
  // MovableList move / set: the element addressed (IdLp = peer + lamport), move: the source position (prop = destination)
  PeerID elem_peer = 0;
  Lamport elem_lamport = 0;
  uint32_t move_from = 0;
};
struct Change {
  ID id;
  Lamport lamport = 0;  // wire lamport; recomputed on import (outdated_encode_reordered.rs:61-62)
  std::vector<ID> deps;
  std::vector<Op> ops;
  int32_t len = 0;  // atom length
  std::vector<ContainerID> cids;  // container of ops[i] = cids[ops[i].container] until registered
  Counter ctr_end() const { return id.counter + len; }
};

inline void utf8_to_cps(const uint8_t* p, size_t n, std::vector<uint32_t>& out) {
  size_t i = 0;
  while (i < n) {
    uint8_t c = p[i];
    uint32_t cp;
    int extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c & 0xE0) == 0xC0) { cp = c & 0x1F; extra = 1; }
    else if ((c & 0xF0) == 0xE0) { cp = c & 0x0F; extra = 2; }
    else if ((c & 0xF8) == 0xF0) { cp = c & 0x07; extra = 3; }
    else fail(ST_DATA_CORRUPTION, "invalid utf8");
    for (int k = 1; k <= extra; k++) {
      if (i + k >= n || (p[i + k] & 0xC0) != 0x80) fail(ST_DATA_CORRUPTION, "invalid utf8");
      cp = (cp << 6) | (p[i + k] & 0x3F);
    }
    out.push_back(cp);
    i += extra + 1;
  }
}
inline void cp_to_utf8(uint32_t cp, std::string& out) {
  if (cp < 0x80) out.push_back((char)cp);
  else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    out.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// Skip/parse one value payload of the given outer tag (docs/encoding.md §10; value.rs:342-459).
// Fills `op` for the in-scope container kinds, otherwise just advances the cursor.
struct BlockCtx {
  std::vector<PeerID> peers;
  std::vector<std::string> keys;
  std::vector<ContainerID> cids;
};

inline void skip_future_value(Reader& r) { (void)r.bytes(); }

// The tables of one block exactly as the wire holds them (test infrastructure for the product's encode side:
// tests/test_encode_roundtrip.py hands them to lm_encode_block through the C ABI and expects the block's own bytes back).
struct RawBlock {
  std::vector<uint64_t> peers;
  std::vector<uint32_t> change_len, dep_count, dep_peer_idx, lamport, msg_len, cid_peer_idx, op_container, op_len, del_peer_idx;
  std::vector<uint8_t> dep_on_self, msgs, cid_is_root, cid_kind, op_value_type, positions, values;
  std::vector<int32_t> dep_counter, cid_key_or_counter, op_prop, del_counter;
  std::vector<int64_t> timestamp, del_len;
  std::vector<std::string> keys;
  std::vector<const uint8_t*> key_ptr;
  std::vector<size_t> key_len;
  lm_block_tables t;
  void finish() {
    key_ptr.clear(); key_len.clear();
    for (auto& k : keys) { key_ptr.push_back((const uint8_t*)k.data()); key_len.push_back(k.size()); }
    t.peers = peers.data(); t.n_peers = peers.size();
    t.change_len = change_len.data(); t.dep_on_self = dep_on_self.data(); t.dep_count = dep_count.data();
    t.dep_peer_idx = dep_peer_idx.data(); t.dep_counter = dep_counter.data(); t.n_deps = dep_peer_idx.size();
    t.lamport = lamport.data(); t.timestamp = timestamp.data();
    t.msg_len = msg_len.data(); t.msgs = msgs.data(); t.msgs_len = msgs.size();
    t.cid_is_root = cid_is_root.data(); t.cid_kind = cid_kind.data(); t.cid_peer_idx = cid_peer_idx.data();
    t.cid_key_or_counter = cid_key_or_counter.data(); t.n_cids = cid_kind.size();
    t.keys = key_ptr.data(); t.key_lens = key_len.data(); t.n_keys = keys.size();
    t.positions = positions.data(); t.positions_len = positions.size();
    t.op_container = op_container.data(); t.op_prop = op_prop.data(); t.op_value_type = op_value_type.data(); t.op_len = op_len.data();
    t.n_ops = op_len.size();
    t.del_peer_idx = del_peer_idx.data(); t.del_counter = del_counter.data(); t.del_len = del_len.data(); t.n_dels = del_len.size();
    t.values = values.data(); t.values_len = values.size();
  }
};

inline void decode_block(Reader blk, std::vector<Change>& out, RawBlock* raw = nullptr) {
  // postcard EncodedBlock (block_encode.rs:94-119)
  uint64_t counter_start = blk.uleb(), counter_len = blk.uleb(), lamport_start = blk.uleb(),
           lamport_len = blk.uleb(), n_changes = blk.uleb();
  if (counter_start > INT32_MAX || counter_len > INT32_MAX || lamport_start > UINT32_MAX || lamport_len > UINT32_MAX ||
      n_changes > UINT32_MAX)
    fail(ST_DECODE_ERROR, "block scalar out of range");
  Reader header = blk.bytes(), change_meta = blk.bytes(), cids_b = blk.bytes(), keys_b = blk.bytes(),
         positions = blk.bytes(), ops_b = blk.bytes(), del_b = blk.bytes(), values_b = blk.bytes();
  if (raw) {
    raw->t.counter_start = (uint32_t)counter_start; raw->t.counter_len = (uint32_t)counter_len; raw->t.lamport_start = (uint32_t)lamport_start;
    raw->t.lamport_len = (uint32_t)lamport_len; raw->t.n_changes = (uint32_t)n_changes;
    raw->positions.assign(positions.p, positions.p + positions.remaining());
    raw->values.assign(values_b.p, values_b.p + values_b.remaining());
  }
  if (n_changes == 0) fail(ST_DECODE_ERROR, "empty change block");
  size_t N = (size_t)n_changes;
  BlockCtx ctx;
  // ---- header (block_meta_encode.rs:90-242)
  uint64_t peer_num = header.uleb();
  if (peer_num == 0 || peer_num > header.remaining() / 8) fail(ST_DECODE_ERROR, "invalid peer table");
  for (uint64_t i = 0; i < peer_num; i++) {
    const uint8_t* q = header.take(8);
    uint64_t v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | q[k];
    ctx.peers.push_back(v);
  }
  std::vector<int64_t> lengths;
  int64_t known = 0;
  for (size_t i = 0; i + 1 < N; i++) {
    uint64_t l = header.uleb();
    if (l > INT32_MAX) fail(ST_DECODE_ERROR, "change len");
    lengths.push_back((int64_t)l);
    known += (int64_t)l;
    if (known > INT32_MAX) fail(ST_DECODE_ERROR, "counter length overflow");
  }
  if ((int64_t)counter_len < known) fail(ST_DECODE_ERROR, "invalid counter length");
  lengths.push_back((int64_t)counter_len - known);
  std::vector<uint8_t> dep_self = take_bool_rle(header, N);
  std::vector<uint64_t> deps_len = take_any_rle_uvar(header, N);
  uint64_t other = 0;
  for (auto d : deps_len) { other += d; if (other > (1u << 28)) fail(ST_DECODE_ERROR, "too many deps"); }
  std::vector<uint64_t> dep_peers = take_any_rle_uvar(header, (size_t)other);
  std::vector<int64_t> dep_counters = take_delta_of_delta(header, (size_t)other);
  std::vector<int64_t> lamports = take_delta_of_delta(header, N - 1);
  {
    int64_t last_len = lengths.back();
    // block_meta_encode.rs:215-221: lamport_start.checked_add(lamport_len) in u32, then checked_sub(last_len)
    int64_t lend = (int64_t)lamport_start + (int64_t)lamport_len;
    if (lend > (int64_t)UINT32_MAX || lend < last_len) fail(ST_DECODE_ERROR, "invalid lamport");
    int64_t ll = lend - last_len;
    lamports.push_back(ll);
  }
  std::vector<int64_t> counters;
  {
    int64_t last = (int64_t)counter_start;
    for (auto l : lengths) { counters.push_back(last); last += l; if (last > INT32_MAX) fail(ST_DECODE_ERROR, "counter overflow"); }
    counters.push_back((int64_t)counter_start + (int64_t)counter_len);
  }
  if (raw) {
    raw->peers = ctx.peers;
    for (auto l : lengths) raw->change_len.push_back((uint32_t)l);
    raw->dep_on_self = dep_self;
    for (auto x : deps_len) raw->dep_count.push_back((uint32_t)x);
    for (auto x : dep_peers) raw->dep_peer_idx.push_back((uint32_t)x);
    for (auto x : dep_counters) raw->dep_counter.push_back((int32_t)x);
    for (auto x : lamports) raw->lamport.push_back((uint32_t)x);
  }
  size_t base = out.size();
  {
    size_t di = 0;
    for (size_t i = 0; i < N; i++) {
      Change c;
      c.id = ID{ctx.peers[0], (Counter)counters[i]};
      c.lamport = (Lamport)lamports[i];
      c.len = (int32_t)lengths[i];
      if (dep_self[i]) {
        if (counters[i] < 1) fail(ST_DECODE_ERROR, "invalid self dependency");
        c.deps.push_back(ID{ctx.peers[0], (Counter)(counters[i] - 1)});
      }
      for (uint64_t k = 0; k < deps_len[i]; k++, di++) {
        if (dep_peers[di] >= ctx.peers.size()) fail(ST_DECODE_ERROR, "invalid peer index");
        c.deps.push_back(ID{ctx.peers[(size_t)dep_peers[di]], (Counter)dep_counters[di]});
      }
      out.push_back(std::move(c));
    }
  }
  // ---- change_meta (block_encode.rs:563-571): validated for shape, content not result-bearing
  {
    // (both decoders' failures are mapped to DecodeDataCorruptionError here — `.map_err(|_| LoroError::DecodeDataCorruptionError)`,
    // block_encode.rs:563-571 — unlike the header columns of block_meta_encode.rs, which stay DecodeError)
    std::vector<int64_t> ts;
    std::vector<uint64_t> msg_lens;
    try {
      ts = take_delta_of_delta(change_meta, N);
      msg_lens = take_any_rle_uvar(change_meta, N);
    } catch (const DecodeErr&) { fail(ST_DATA_CORRUPTION, "change_meta column"); }
    uint64_t tot = 0;
    for (auto l : msg_lens) tot += l;
    if (tot > change_meta.remaining()) fail(ST_DATA_CORRUPTION, "commit message bytes");
    if (raw) {
      raw->timestamp = ts;
      for (auto l : msg_lens) raw->msg_len.push_back((uint32_t)l);
      raw->msgs.assign(change_meta.p, change_meta.p + change_meta.remaining());
    }
  }
  // ---- keys (block_encode.rs:280-305)
  while (!keys_b.eof()) {
    Reader k = keys_b.bytes();
    ctx.keys.emplace_back((const char*)k.p, k.remaining());
  }
  if (raw) raw->keys = ctx.keys;
  // ---- cids (arena.rs:39-105)
  {
    uint64_t n = cids_b.eof() ? 0 : cids_b.uleb();
    for (uint64_t i = 0; i < n; i++) {
      uint64_t fields = cids_b.uleb();
      if (fields != 4) fail(ST_DECODE_ERROR, "EncodedContainer field count");
      uint8_t is_root = cids_b.u8();
      uint8_t kind = cids_b.u8();
      uint64_t peer_idx = cids_b.uleb();
      int64_t koc = cids_b.zigzag();
      if (raw) { raw->cid_is_root.push_back(is_root); raw->cid_kind.push_back(kind); raw->cid_peer_idx.push_back((uint32_t)peer_idx); raw->cid_key_or_counter.push_back((int32_t)koc); }
      ContainerID c;
      c.kind = kind;
      if (is_root) {
        c.root = true;
        if (koc < 0 || (uint64_t)koc >= ctx.keys.size()) fail(ST_DATA_CORRUPTION, "root key idx");
        c.name = ctx.keys[(size_t)koc];
      } else {
        c.root = false;
        if (peer_idx >= ctx.peers.size()) fail(ST_DATA_CORRUPTION, "cid peer idx");
        c.peer = ctx.peers[(size_t)peer_idx];
        c.counter = (Counter)koc;
      }
      ctx.cids.push_back(c);
    }
  }
  // ---- ops columns (block_encode.rs:417-434)
  std::vector<int64_t> col_container, col_prop;
  std::vector<uint8_t> col_vt;
  std::vector<uint64_t> col_len;
  {
    uint64_t outer = ops_b.uleb();
    if (outer != 1) fail(ST_DECODE_ERROR, "EncodedOps outer field count");
    uint64_t ncols = ops_b.uleb();
    if (ncols != 4) fail(ST_DECODE_ERROR, "EncodedOp column count");
    col_container = decode_delta_rle(ops_b.bytes());
    col_prop = decode_delta_rle(ops_b.bytes());
    col_vt = decode_any_rle_u8(ops_b.bytes());
    col_len = decode_any_rle_uvar(ops_b.bytes());
    size_t n = col_container.size();
    if (col_prop.size() != n || col_vt.size() != n || col_len.size() != n) fail(ST_DECODE_ERROR, "op column length mismatch");
  }
  // ---- delete start ids (outdated_encode_reordered.rs:480-489)
  std::vector<int64_t> del_peer, del_counter, del_len;
  if (!del_b.eof()) {
    uint64_t outer = del_b.uleb();
    if (outer != 1) fail(ST_DECODE_ERROR, "EncodedDeleteStartIds outer");
    uint64_t ncols = del_b.uleb();
    if (ncols != 3) fail(ST_DECODE_ERROR, "EncodedDeleteStartId columns");
    del_peer = decode_delta_rle(del_b.bytes());
    del_counter = decode_delta_rle(del_b.bytes());
    del_len = decode_delta_rle(del_b.bytes());
    if (del_counter.size() != del_peer.size() || del_len.size() != del_peer.size()) fail(ST_DECODE_ERROR, "delete column mismatch");
  }
  if (raw) {
    for (size_t i = 0; i < col_container.size(); i++) {
      raw->op_container.push_back((uint32_t)col_container[i]); raw->op_prop.push_back((int32_t)col_prop[i]);
      raw->op_value_type.push_back(col_vt[i]); raw->op_len.push_back((uint32_t)col_len[i]);
    }
    for (size_t i = 0; i < del_peer.size(); i++) { raw->del_peer_idx.push_back((uint32_t)del_peer[i]); raw->del_counter.push_back((int32_t)del_counter[i]); raw->del_len.push_back(del_len[i]); }
    raw->finish();
  }
  size_t del_i = 0;
  // ---- row walk (block_encode.rs:651-704)
  DecodeArena arena{&ctx.peers, &ctx.keys};
  int64_t counter = (int64_t)counter_start;
  size_t change_index = 0;
  PeerID peer = ctx.peers[0];
  for (size_t row = 0; row < col_container.size(); row++) {
    uint8_t vt = col_vt[row] & 0x7f;  // value.rs: mask bit 7
    int64_t ci = col_container[row];
    if (ci < 0 || (uint64_t)ci >= ctx.cids.size()) fail(ST_DATA_CORRUPTION, "container index");
    const ContainerID& cid = ctx.cids[(size_t)ci];
    int32_t prop = (int32_t)col_prop[row];
    ID op_id{peer, (Counter)counter};
    Op op;
    op.counter = (Counter)counter;
    op.len = (int32_t)col_len[row];
    op.prop = prop;
    // value payload
    Value lv;
    bool have_lv = false;
    uint32_t mark_len = 0;
    std::string mark_key;   // MarkStart: the style's key and value (container/richtext.rs:31-57 StyleOp)
    Value mark_val;
    uint64_t mv_from = 0, mv_peer = 0, mv_lamport = 0;
    std::string str_payload;
    switch (vt) {
      case 0: case 1: case 2: case 8: case 9: break;
      case 3: (void)values_b.sleb(); break;
      case 4: (void)values_b.take(8); break;
      case 5: { Reader s = values_b.bytes(); str_payload.assign((const char*)s.p, s.remaining()); break; }
      case 6: (void)values_b.bytes(); break;
      case 7: (void)values_b.uleb(); break;
      case 10: (void)values_b.sleb(); break;
      case 11: lv = read_loro_value(values_b, arena, op_id, 0, true); have_lv = true; break;
      case 12: {  // MarkStart (value.rs:936-955)
        (void)values_b.u8();
        uint64_t ml = values_b.uleb();
        uint64_t key_idx = values_b.uleb();
        if (key_idx >= ctx.keys.size()) fail(ST_DATA_CORRUPTION, "mark key idx");
        mark_val = read_loro_value(values_b, arena, op_id, 0, true);
        if (key_idx < ctx.keys.size()) mark_key = ctx.keys[(size_t)key_idx];
        mark_len = (uint32_t)ml;
        break;
      }
      case 13: {  // older TreeMove
        (void)values_b.uleb();
        uint8_t is_null = values_b.u8();
        (void)values_b.uleb();
        if (!is_null) (void)values_b.uleb();
        break;
      }
      case 14: mv_from = values_b.uleb(); mv_peer = values_b.uleb(); mv_lamport = values_b.uleb(); break;   // ListMove (value.rs:342-459)
      case 15: mv_peer = values_b.uleb(); mv_lamport = values_b.uleb(); lv = read_loro_value(values_b, arena, op_id, 0, true); have_lv = true; break;
      case 16: {
        (void)values_b.uleb(); (void)values_b.uleb(); (void)values_b.uleb();
        uint8_t is_null = values_b.u8();
        if (!is_null) { (void)values_b.uleb(); (void)values_b.uleb(); }
        break;
      }
      default: skip_future_value(values_b); break;
    }
    // decode_op (outdated_encode_reordered.rs:215-476)
    auto take_del = [&]() {
      if (del_i >= del_peer.size()) fail(ST_DATA_CORRUPTION, "missing delete start id");
      int64_t pi = del_peer[del_i];
      if (pi < 0 || (uint64_t)pi >= ctx.peers.size()) fail(ST_DATA_CORRUPTION, "delete peer idx");
      op.kind = OP_SEQ_DELETE;
      op.del_id_start = ID{ctx.peers[(size_t)pi], (Counter)del_counter[del_i]};
      op.del_signed_len = del_len[del_i];
      if (op.del_signed_len == 0) fail(ST_DATA_CORRUPTION, "zero delete len");
      del_i++;
    };
    switch (cid.kind) {
      case CK_TEXT:
        if (vt == 5) {
          op.kind = OP_TEXT_INSERT;
          utf8_to_cps((const uint8_t*)str_payload.data(), str_payload.size(), op.cps);
        } else if (vt == 9) take_del();
        else if (vt == 12) { op.kind = OP_STYLE_START; op.style_end = (uint32_t)prop + mark_len; op.key = mark_key; op.value = std::move(mark_val); }
        else if (vt == 0) op.kind = OP_STYLE_END;
        else fail(ST_DATA_CORRUPTION, "bad text op value");
        break;
      case CK_MAP:
        if (prop < 0 || (size_t)prop >= ctx.keys.size()) fail(ST_DATA_CORRUPTION, "map key idx");
        op.key = ctx.keys[(size_t)prop];
        if (vt == 8) op.kind = OP_MAP_DELETE;
        else if (vt == 11 && have_lv) { op.kind = OP_MAP_SET; op.value = std::move(lv); }
        else fail(ST_DATA_CORRUPTION, "bad map op value");
        break;
      case CK_LIST:
        if (vt == 11 && have_lv) {
          if (lv.kind != V_LIST) fail(ST_DATA_CORRUPTION, "list insert value not a list");
          op.kind = OP_LIST_INSERT;
          op.values = std::move(lv.list);
        } else if (vt == 9) take_del();
        else fail(ST_DATA_CORRUPTION, "bad list op value");
        break;
      case CK_MOVABLE:   // outdated_encode_reordered.rs:388-459
        if (vt == 11 && have_lv) {
          if (lv.kind != V_LIST) fail(ST_DATA_CORRUPTION, "movable list insert value not a list");
          op.kind = OP_LIST_INSERT;
          op.values = std::move(lv.list);
        } else if (vt == 9) take_del();
        else if (vt == 14 || (vt == 15 && have_lv)) {
          if (mv_peer >= ctx.peers.size()) fail(ST_DATA_CORRUPTION, "movable list element peer idx");
          if (mv_lamport > 0xFFFFFFFFull || mv_from > 0x7FFFFFFFull || prop < 0) fail(ST_DATA_CORRUPTION, "movable list element id");
          // canonical invariant (docs/encoding.md §9.6): move / set rows are one atom long
          if (col_len[row] != 1) fail(ST_DATA_CORRUPTION, "movable list move/set len");
          op.elem_peer = ctx.peers[(size_t)mv_peer];
          op.elem_lamport = (Lamport)mv_lamport;
          if (vt == 14) { op.kind = OP_LIST_MOVE; op.move_from = (uint32_t)mv_from; }
          else { op.kind = OP_LIST_SET; op.value = std::move(lv); }
        } else fail(ST_DATA_CORRUPTION, "bad movable list op value");
        break;
      default: op.kind = OP_OTHER; break;
    }
    // canonical invariant (docs/encoding.md §10.6): an insert's content spans exactly the atoms its `len` column entry
    // claims.  The Rust reader does not check this; slicing such an op (Sliceable, below in lo_doc.hpp) would index past
    // its content, so the restatement rejects it here — the device does the same where it fills elements (k_elem_fill).
    if (op.kind == OP_TEXT_INSERT && op.cps.size() != (size_t)col_len[row]) fail(ST_DATA_CORRUPTION, "text insert len");
    if (op.kind == OP_LIST_INSERT && op.values.size() != (size_t)col_len[row]) fail(ST_DATA_CORRUPTION, "list insert len");
    if (change_index >= N) fail(ST_DATA_CORRUPTION, "op beyond last change");
    Change& ch = out[base + change_index];
    // ops carry a per-change cid table index until the Doc registers containers
    {
      size_t k = 0;
      for (; k < ch.cids.size(); k++) if (ch.cids[k] == cid) break;
      if (k == ch.cids.size()) ch.cids.push_back(cid);
      op.container = (uint32_t)k;
    }
    ch.ops.push_back(std::move(op));
    counter += (int64_t)col_len[row];
    if (counter > INT32_MAX) fail(ST_DATA_CORRUPTION, "counter overflow");
    if (change_index + 1 >= counters.size()) fail(ST_DATA_CORRUPTION, "change index");
    // canonical invariant (docs/encoding.md §10.6: "operation atom lengths partition the reconstructed change counter ranges
    // exactly, without crossing a boundary … independent validators should reject these inputs"): the Rust reader advances at
    // most one change on `counter >= next_counter` (block_encode.rs:697-703) and lets an op that crosses a boundary — or a change of
    // length zero — through; what the oplog then makes of overlapping changes is not a value anybody specified.  Rejected here, and by
    // the device decoders (lm_k_decode_wave.h / lm_k_decode.h), like the other canonical invariants above.
    if (counter > counters[change_index + 1]) fail(ST_DATA_CORRUPTION, "op crosses a change boundary");
    if (counter >= counters[change_index + 1]) change_index++;
  }
}

// envelope (encoding.rs:334-373) + updates framing (fast_snapshot.rs:372-400).
inline void decode_updates_blob(const uint8_t* blob, size_t len, std::vector<Change>& out) {
  if (len < 22) fail(ST_DECODE_ERROR, "Invalid import data");
  if (memcmp(blob, "loro", 4) != 0) fail(ST_DECODE_ERROR, "Invalid magic");
  uint32_t expect = rd32le(blob + 16);
  uint16_t mode = (uint16_t)((blob[20] << 8) | blob[21]);
  if (mode != 4) {
    if (mode == 3) fail(ST_UNSUPPORTED, "FastSnapshot (mode 3) is outside the hot-path scope");
    fail(ST_DECODE_ERROR, "Unknown encode mode");
  }
  if (xxh32(blob + 20, len - 20, LORO_XXH_SEED) != expect) fail(ST_CHECKSUM_MISMATCH, "checksum mismatch");
  Reader r(blob + 22, len - 22);
  while (!r.eof()) {
    uint64_t bl = r.uleb();
    if (bl == 0 || bl > r.remaining()) fail(ST_DECODE_ERROR, "Invalid bytes");
    Reader blk(r.p, (size_t)bl);
    r.p += bl;
    decode_block(blk, out);
  }
}

}  // namespace lo
