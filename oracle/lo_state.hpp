// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// The STATE section of a FastSnapshot (EncodeMode 3) read for its values: groundwork for SURVEY §8f N3 "the state section lets merges
// start from a materialised base" (NEXT.md §0.2) — the checker renders the deep value a snapshot's state section holds WITHOUT
// replaying the history, so that a device path that seeds itself from the section has something to be compared with, and so that
// this repository's reading of docs/encoding-container-states.md is pinned on the Rust-written fixtures (tests/test_oracle_golden.py:
// snapshot.blob / snapshot.ts.blob / runtime-snapshot.ts.blob must give the value their history gives and the value the reference
// expects, shallow.ts.blob the value at its latest version).  Restates (paths relative to /root/reference):
//   state SSTable keys = ContainerID::to_bytes, `fr` in a shallow root      docs/encoding-container-states.md §1; loro-common/src/lib.rs:604-686
//   shallow import: root entries first, overlay entries replace equal keys   state/container_store.rs:192-223, fast_snapshot.rs:168-258
//   ContainerWrapper: kind, uleb depth, postcard Option<ContainerID> parent  state/container_store/container_wrapper.rs:428-489
//   postcard LoroValue (tags 0-8), postcard ContainerID + historical kinds   loro-common/src/value.rs:719-795, lib.rs:589-602,805-903
//   Map: postcard map of visible values first                               state/map_state.rs:492-607
//   List / MovableList: postcard Vec<LoroValue> of visible values first      state/list_state.rs:759-858, movable_list_state.rs:1389-1430
//   Text: postcard String full_text first                                   state/richtext_state.rs:1219-1468
// Only the visible values are read (what get_deep_value shows); ids, peers, marks and tombstone metadata behind them are skipped.
// Tree / Counter states render as null and flag the result unsupported, like the history path.
#pragma once
#include "lo_doc.hpp"

namespace lo {

struct StateStore {
  std::map<ContainerID, std::string> wrappers;   // container → ContainerWrapper bytes
  bool unsupported = false;
};

inline uint8_t hist_kind_to_raw(uint8_t h) {   // historical postcard tag → raw kind (lib.rs:805-903 vs 765-803)
  switch (h) { case 0: return CK_TEXT; case 1: return CK_MAP; case 2: return CK_LIST; case 3: return CK_MOVABLE; case 4: return CK_TREE; case 5: return CK_COUNTER; default: return h; }
}
inline ContainerID read_postcard_cid(Reader& r) {
  ContainerID c;
  uint64_t variant = r.uleb();
  if (variant == 0) {
    Reader nm = r.bytes();
    c.root = true;
    c.name.assign((const char*)nm.p, nm.remaining());
  } else if (variant == 1) {
    c.root = false;
    c.peer = r.uleb();
    int64_t ctr = r.zigzag();
    if (ctr < INT32_MIN || ctr > INT32_MAX) fail(ST_DECODE_ERROR, "container counter");
    c.counter = (Counter)ctr;
  } else fail(ST_DECODE_ERROR, "ContainerID variant");
  c.kind = hist_kind_to_raw(r.u8());
  return c;
}
inline Value read_postcard_value(Reader& r, int depth) {
  if (depth > 256) fail(ST_DATA_CORRUPTION, "value nesting too deep");
  Value v;
  uint64_t tag = r.uleb();
  switch (tag) {
    case 0: v.kind = V_NULL; break;
    case 1: v.kind = V_BOOL; v.b = r.u8() != 0; break;
    case 2: { const uint8_t* q = r.take(8); uint64_t bits = 0; for (int k = 7; k >= 0; k--) bits = (bits << 8) | q[k]; memcpy(&v.f, &bits, 8); v.kind = V_F64; break; }
    case 3: v.kind = V_I64; v.i = r.zigzag(); break;
    case 4: { Reader s = r.bytes(); v.kind = V_STR; v.s.assign((const char*)s.p, s.remaining()); break; }
    case 5: { uint64_t n = r.uleb(); if (n > r.remaining()) fail(ST_DECODE_ERROR, "list length"); v.kind = V_LIST; for (uint64_t i = 0; i < n; i++) v.list.push_back(read_postcard_value(r, depth + 1)); break; }
    case 6: {
      uint64_t n = r.uleb();
      if (n > r.remaining()) fail(ST_DECODE_ERROR, "map length");
      v.kind = V_MAP;
      for (uint64_t i = 0; i < n; i++) { Reader k = r.bytes(); std::string key((const char*)k.p, k.remaining()); v.map.emplace_back(key, read_postcard_value(r, depth + 1)); }
      break;
    }
    case 7: v.kind = V_CONTAINER; v.cid = read_postcard_cid(r); break;
    case 8: { Reader s = r.bytes(); v.kind = V_BIN; v.s.assign((const char*)s.p, s.remaining()); break; }
    default: fail(ST_DATA_CORRUPTION, "postcard LoroValue tag");
  }
  return v;
}
inline bool cid_from_state_key(const std::string& k, ContainerID& c) {   // ContainerID::to_bytes (lib.rs:604-686)
  if (k.empty()) return false;
  uint8_t b0 = (uint8_t)k[0];
  if (b0 & 0x80) {
    Reader r((const uint8_t*)k.data() + 1, k.size() - 1);
    uint64_t nl = r.uleb();
    if (nl != r.remaining()) return false;
    c.root = true; c.kind = b0 & 0x7f; c.name.assign((const char*)r.p, (size_t)nl);
    return true;
  }
  if (k.size() != 13) return false;
  c.root = false; c.kind = b0;
  uint64_t peer = 0;
  for (int i = 7; i >= 0; i--) peer = (peer << 8) | (uint8_t)k[1 + i];
  c.peer = peer;
  c.counter = (Counter)(int32_t)rd32le((const uint8_t*)k.data() + 9);
  return true;
}

// the three sections of a mode-3 blob → the state store it initialises (shallow: the root's entries, then the overlay's)
inline StateStore snapshot_state_store(const uint8_t* blob, size_t len, bool root_only = false) {
  if (len < 22 || memcmp(blob, "loro", 4) != 0) fail(ST_DECODE_ERROR, "Invalid import data");
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
  StateStore st;
  for (int s : {2, 1}) {
    if (sl[s] == 0 || (root_only && s == 1)) continue;   // root_only: the state AT the shallow root (what a checkout to shallow_since_frontiers shows)
    for (auto& kv : sstable_entries(sec[s], sl[s])) {
      ContainerID c;
      if (kv.first == "fr" || !cid_from_state_key(kv.first, c)) continue;
      st.wrappers[c] = kv.second;
    }
  }
  return st;
}

inline void state_value_json(StateStore& st, const Value& v, std::string& out, int depth);
inline void state_container_json(StateStore& st, const ContainerID& cid, std::string& out, int depth) {
  if (depth > 200) { out += "null"; return; }
  auto it = st.wrappers.find(cid);
  if (it == st.wrappers.end()) {   // a child nobody wrote to: the empty value of its kind (state.rs:1550-1616)
    if (cid.kind == CK_TEXT) out += "\"\"";
    else if (cid.kind == CK_MAP) out += "{}";
    else if (cid.kind == CK_LIST || cid.kind == CK_MOVABLE) out += "[]";
    else { out += "null"; st.unsupported = true; }
    return;
  }
  Reader r((const uint8_t*)it->second.data(), it->second.size());
  uint8_t kind = r.u8();
  (void)r.uleb();                                   // hierarchy depth
  if (r.uleb() == 1) (void)read_postcard_cid(r);    // Option<ContainerID> parent
  if (kind == CK_MAP) {
    uint64_t n = r.uleb();
    std::map<std::string, Value> m;
    for (uint64_t i = 0; i < n; i++) { Reader k = r.bytes(); std::string key((const char*)k.p, k.remaining()); m[key] = read_postcard_value(r, 0); }
    out.push_back('{');
    bool first = true;
    for (auto& kv : m) {
      if (!first) out.push_back(',');
      first = false;
      json_escape(kv.first, out);
      out.push_back(':');
      state_value_json(st, kv.second, out, depth + 1);
    }
    out.push_back('}');
  } else if (kind == CK_LIST || kind == CK_MOVABLE) {
    uint64_t n = r.uleb();
    out.push_back('[');
    for (uint64_t i = 0; i < n; i++) { if (i) out.push_back(','); state_value_json(st, read_postcard_value(r, 0), out, depth + 1); }
    out.push_back(']');
  } else if (kind == CK_TEXT) {
    Reader s = r.bytes();
    json_escape(std::string((const char*)s.p, s.remaining()), out);
  } else { out += "null"; st.unsupported = true; }
}
inline void state_value_json(StateStore& st, const Value& v, std::string& out, int depth) {
  if (v.kind == V_CONTAINER) { state_container_json(st, v.cid, out, depth); return; }
  if (v.kind == V_LIST) { out.push_back('['); for (size_t i = 0; i < v.list.size(); i++) { if (i) out.push_back(','); state_value_json(st, v.list[i], out, depth + 1); } out.push_back(']'); return; }
  if (v.kind == V_MAP) {
    std::map<std::string, const Value*> m;
    for (auto& e : v.map) m[e.first] = &e.second;
    out.push_back('{');
    bool first = true;
    for (auto& kv : m) { if (!first) out.push_back(','); first = false; json_escape(kv.first, out); out.push_back(':'); state_value_json(st, *kv.second, out, depth + 1); }
    out.push_back('}');
    return;
  }
  Doc none;
  json_value(none, v, out, depth);   // scalars, strings, binary
}
// the deep value of the state section: every root container it holds, names bytewise sorted
inline std::string snapshot_state_json(const uint8_t* blob, size_t len, bool& unsupported, bool root_only = false) {
  StateStore st = snapshot_state_store(blob, len, root_only);
  std::map<std::string, ContainerID> roots;
  for (auto& kv : st.wrappers) if (kv.first.root) roots[kv.first.name] = kv.first;
  std::string out = "{";
  bool first = true;
  for (auto& kv : roots) {
    if (!first) out.push_back(',');
    first = false;
    json_escape(kv.first, out);
    out.push_back(':');
    state_container_json(st, kv.second, out, 0);
  }
  out.push_back('}');
  unsupported = st.unsupported;
  return out;
}

}  // namespace lo
