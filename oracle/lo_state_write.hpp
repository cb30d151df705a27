// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// A STATE-section WRITER for the test generators: the key / value pairs of the state SSTable of ExportMode::Snapshot for the document
// the checker holds after an import (docs/encoding-container-states.md: keys = ContainerID::to_bytes §1.1, values = ContainerWrapper
// §2 — kind, uleb depth, postcard Option<ContainerID> parent — followed by the container's VISIBLE values: Map §4 postcard map,
// List §5 / MovableList §8 postcard Vec<LoroValue>, Text §6 postcard String).  The metadata behind the visible values (ids, peers,
// lamports, marks) is NOT written: neither lo_state.hpp nor the product's reader (loro_amd/csrc/lm_snapshot.h snapshot_state_to_updates)
// looks at it.  tests/ use it to give loro_amd.wire's snapshots a real state section, so that the product's state path
//  is compared with the history path on generated documents, not only on the four reference-held snapshots.
// This is the inverse of lo_state.hpp's reader — which IS pinned on the Rust- and TS-written fixtures — and is checked against it
// (tests/test_oracle_golden.py: write, read back, same JSON as the history gives).
#pragma once
#include "lo_doc.hpp"

namespace lo {

inline void pc_uleb(std::string& o, uint64_t v) { do { uint8_t b = v & 0x7f; v >>= 7; if (v) b |= 0x80; o.push_back((char)b); } while (v); }
inline void pc_zigzag(std::string& o, int64_t v) { pc_uleb(o, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
inline uint8_t raw_kind_to_hist(uint8_t k) {   // the inverse of hist_kind_to_raw (lib.rs:805-903)
  switch (k) { case CK_TEXT: return 0; case CK_MAP: return 1; case CK_LIST: return 2; case CK_MOVABLE: return 3; case CK_TREE: return 4; case CK_COUNTER: return 5; default: return k; }
}
inline void pc_cid(std::string& o, const ContainerID& c) {
  if (c.root) { pc_uleb(o, 0); pc_uleb(o, c.name.size()); o += c.name; }
  else { pc_uleb(o, 1); pc_uleb(o, c.peer); pc_zigzag(o, c.counter); }
  o.push_back((char)raw_kind_to_hist(c.kind));
}
inline std::string state_key(const ContainerID& c) {
  std::string k;
  if (c.root) { k.push_back((char)(0x80 | c.kind)); pc_uleb(k, c.name.size()); k += c.name; return k; }
  k.push_back((char)c.kind);
  for (int i = 0; i < 8; i++) k.push_back((char)(uint8_t)(c.peer >> (8 * i)));
  for (int i = 0; i < 4; i++) k.push_back((char)(uint8_t)((uint32_t)c.counter >> (8 * i)));
  return k;
}
inline void pc_value(std::string& o, const Value& v, std::vector<ContainerID>& kids) {
  switch (v.kind) {
    case V_NULL: pc_uleb(o, 0); break;
    case V_BOOL: pc_uleb(o, 1); o.push_back(v.b ? 1 : 0); break;
    case V_F64: { pc_uleb(o, 2); uint64_t bits; memcpy(&bits, &v.f, 8); for (int k = 0; k < 8; k++) o.push_back((char)(uint8_t)(bits >> (8 * k))); break; }
    case V_I64: pc_uleb(o, 3); pc_zigzag(o, v.i); break;
    case V_STR: pc_uleb(o, 4); pc_uleb(o, v.s.size()); o += v.s; break;
    case V_BIN: pc_uleb(o, 8); pc_uleb(o, v.s.size()); o += v.s; break;
    case V_LIST: pc_uleb(o, 5); pc_uleb(o, v.list.size()); for (auto& x : v.list) pc_value(o, x, kids); break;
    case V_MAP: {
      // (a map VALUE with a repeated key: the last one wins, as json_value renders it)
      std::map<std::string, const Value*> m;
      for (auto& e : v.map) m[e.first] = &e.second;
      pc_uleb(o, 6); pc_uleb(o, m.size());
      for (auto& kv : m) { pc_uleb(o, kv.first.size()); o += kv.first; pc_value(o, *kv.second, kids); }
      break;
    }
    case V_CONTAINER: pc_uleb(o, 7); pc_cid(o, v.cid); kids.push_back(v.cid); break;
  }
}

// the state store's entries for the document at its rendered version: every root to_json shows, every child a visible value names
inline std::vector<std::pair<std::string, std::string>> state_entries(Doc& d) {
  d.materialize();
  std::vector<std::pair<std::string, std::string>> out;
  struct Todo { ContainerID cid; uint32_t depth; bool has_parent; ContainerID parent; };
  std::vector<Todo> todo;
  for (uint32_t i = 0; i < d.containers.size(); i++) {
    const ContainerID& c = d.containers[i];
    if (!c.root || !(d.touched.count(i) || d.state_roots.count(i))) continue;
    if ((c.kind == CK_TEXT || c.kind == CK_LIST || c.kind == CK_MOVABLE) && !d.seq_exists.count(i) && !d.state_roots.count(i)) continue;
    todo.push_back(Todo{c, 1, false, ContainerID()});
  }
  std::set<std::string> seen;
  for (size_t t = 0; t < todo.size(); t++) {
    const Todo td = todo[t];
    const std::string key = state_key(td.cid);
    if (!seen.insert(key).second) continue;
    std::string w;
    w.push_back((char)td.cid.kind);
    pc_uleb(w, td.depth);
    if (td.has_parent) { pc_uleb(w, 1); pc_cid(w, td.parent); } else pc_uleb(w, 0);
    std::vector<ContainerID> kids;
    auto ci = d.container_idx.find(td.cid);
    const bool known = ci != d.container_idx.end();
    const uint32_t idx = known ? ci->second : 0;
    if (td.cid.kind == CK_MAP) {
      std::string body; uint64_t n = 0;
      if (known) { auto it = d.maps.find(idx); if (it != d.maps.end()) for (auto& kv : it->second) { if (!kv.second.has) continue; n++; pc_uleb(body, kv.first.size()); body += kv.first; pc_value(body, kv.second.v, kids); } }
      pc_uleb(w, n); w += body;
    } else if (td.cid.kind == CK_LIST || td.cid.kind == CK_MOVABLE) {
      std::string body; uint64_t n = 0;
      if (known) {
        auto it = d.seqs.find(idx);
        if (it != d.seqs.end()) {
          const SeqState& st = *it->second;
          for (Span* sp = st.tr.head; sp; sp = sp->next)
            if (sp->active())
              for (int32_t k = 0; k < sp->len; k++) {
                size_t cix = (size_t)sp->content + (uint32_t)k;
                if (td.cid.kind == CK_LIST) {
                  if (cix >= st.values.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
                  n++; pc_value(body, st.values[cix], kids);
                } else {
                  if (cix >= st.items.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
                  const SeqState::ItemRec& rec = st.items[cix];
                  auto pw = st.pos_win.find(rec.elem);
                  if (!((pw == st.pos_win.end() ? rec.elem : pw->second) == rec.item)) continue;
                  auto vw = st.val_win.find(rec.elem);
                  n++; pc_value(body, vw != st.val_win.end() ? vw->second.v : st.values[st.elem_init.at(rec.elem)], kids);
                }
              }
        }
      }
      pc_uleb(w, n); w += body;
    } else if (td.cid.kind == CK_TEXT) {
      std::string s;
      bool anchors = false;
      if (known) {
        auto it = d.seqs.find(idx);
        if (it != d.seqs.end())
          for (Span* sp = it->second->tr.head; sp; sp = sp->next)
            if (sp->active())
              for (int32_t k = 0; k < sp->len; k++) {
                if ((uint64_t)sp->content + (uint32_t)k >= it->second->cps.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
                uint32_t cp = it->second->cps[sp->content + (uint32_t)k];
                if (cp != 0xFFFFFFFFu) cp_to_utf8(cp, s); else anchors = true;
              }
      }
      pc_uleb(w, s.size()); w += s;
      // the metadata's SHAPE (§6.1): an empty peer table, EncodedText = 3 fields — 4 (empty) span columns, the style keys, no mark rows.
      // Only what a reader of VALUES looks at is true to the document: there is a style key exactly when a style anchor is visible
      // (the product declines such a Text: lm_richtext needs its marks, which only the history holds)
      pc_uleb(w, 0); pc_uleb(w, 3); pc_uleb(w, 4);
      for (int c4 = 0; c4 < 4; c4++) pc_uleb(w, 0);
      if (anchors) { pc_uleb(w, 1); pc_uleb(w, 4); w += "bold"; } else pc_uleb(w, 0);
      pc_uleb(w, 0);
    }
    // (Tree / Counter: the wrapper alone — both readers render null and flag the document)
    out.emplace_back(key, w);
    for (auto& k : kids) todo.push_back(Todo{k, td.depth + 1, true, td.cid});
  }
  std::sort(out.begin(), out.end());
  return out;
}

}  // namespace lo
