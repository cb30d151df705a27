// LWW map resolve and state emission: canonical JSON of get_deep_value() + postcard VersionVector.
// Reference semantics (paths relative to /root/reference/crates):
//   Map LWW winner = max (lamport, peer)      loro-internal/src/diff_calc.rs:515-538, delta/map_delta.rs:20-46
//   deletes hide the key                      loro-internal/src/state/map_state.rs:438-449
//   deep value of roots                       loro-internal/src/state.rs:1294-1329
//   JSON rendering                            loro-common/src/value.rs:719-738 (serde_json, keys sorted)
//   VersionVector::encode                     loro-internal/src/version.rs:962-964 (entries sorted by peer here)
#pragma once
#include "lm_f64.h"
#include "lm_k_integrate_span.h"

namespace lm {

static constexpr unsigned long long HT_EMPTY = ~0ull;

LM_DEV uint64_t fnv1a(const uint8_t* p, uint32_t n, uint64_t h) {
  for (uint32_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
LM_DEV bool bytes_eq(const uint8_t* a, const uint8_t* b, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) if (a[i] != b[i]) return false;
  return true;
}
// bytewise string order (what serde_json::Value / BTreeMap<String,_> uses)
LM_DEV int bytes_cmp(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {
  uint32_t n = na < nb ? na : nb;
  for (uint32_t i = 0; i < n; i++) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  return na < nb ? -1 : (na > nb ? 1 : 0);
}

// K10: one lane per op row — running LWW maximum per (container, key) in the doc's hash table.
// The table of a document is sized OPTIMISTICALLY (lm_pipeline.h: a few thousand slots, however many Map rows it has — an LWW
// history writes few keys many times, and a table sized for the rows is megabytes per document that every probe misses the caches
// in); a document that claims more than half of its slots is flagged DF_LWW_RETRY and counted in retry_count[2]: the host gives it
// a table sized for its rows and launches the kernel again for those documents only (`only_retry`).
LM_KERNEL void k_map_lww(Dev d, uint32_t n_ops, uint32_t* retry_count, uint32_t only_retry, uint32_t skip_lds) {
  uint32_t t = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  if (t >= n_ops) return;
  OpRow r = d.op[t];
  uint32_t kind = (r.cidx_kind >> 16) & 0xff;
  if (kind != OK_MAP_SET && kind != OK_MAP_DEL && kind != OK_OTHER) return;
  if (!d.chg_flag[r.chg]) return;
  uint32_t blk = d.op_blk[t];
  uint32_t doc = d.blk[blk].doc;
  const DocMeta& m = d.doc[doc];
  if (status_fatal(m.status)) return;
  // (skip_lds: the documents whose table fits LDS were resolved by k_map_lww_doc, lm_k_lww_doc.h — 2,048 = its LWW_LDS_CAP)
  if (skip_lds && !only_retry && d.ht_cap[doc] <= 2048u && !(m.flags & DF_MOVABLE)) return;
  const ChangeRow& ch = d.chg[r.chg];
  if (kind == OK_OTHER) {
    // an applied op of a container outside the device scope (Tree / Counter): the container is known to
    // the state store (it renders as null) and the document is reported LM_UNSUPPORTED together with its JSON
    if (r.ctr + r.len <= ch.ctr + d.chg_skip[r.chg]) return;
    uint32_t ci = r.cidx_kind & 0xffff;
    uint32_t ock = d.cont[m.cid0 + ci].kind_root & 0xff;
    if (ock > CK_TEXT && ock != CK_MOVABLE) {
      d.cont[m.cid0 + ci].touched = 1;
      lmw::atomic_or(&d.doc[doc].flags, DF_SOFT_UNSUPPORTED);
    }
    return;
  }
  if (r.ctr < ch.ctr + d.chg_skip[r.chg]) return;  // already-known prefix of a sliced change
  if (((m.flags & DF_LWW_RETRY) != 0) != (only_retry != 0)) return;   // (first pass: a document already found too big is not worth more work)
  auto overflow = [&]() {
    if (only_retry) { LM_SETERR(d.doc[doc].status, ST_INTERNAL); return; }   // (the second table holds every row's key)
    if (!(lmw::atomic_or(&d.doc[doc].flags, DF_LWW_RETRY) & DF_LWW_RETRY)) lmw::atomic_add(retry_count + 2, 1u);
  };
  uint32_t cap = d.ht_cap[doc];
  if (cap == 0) { LM_SETERR(d.doc[doc].status, ST_INTERNAL); return; }
  uint32_t cidx = r.cidx_kind & 0xffff;
  // checked-out documents: writes after the version do not compete (MapHistoryCache::get_container_latest_op_at_vv,
  // history_cache.rs:630-703); the container itself stays known to the state store
  if (r.ctr >= d.peer_end[m.praw0 + ch.peer]) { d.cont[m.cid0 + cidx].touched = 1; return; }
  uint32_t krow = r.a0;   // (k_remap: the block's first key row + prop)
  const uint8_t* ks = d.data + d.key_off[krow];
  uint32_t kl = d.key_len[krow];
  uint64_t h = fnv1a(ks, kl, 0xcbf29ce484222325ull ^ cidx);
  unsigned long long mine = ((unsigned long long)cidx << 32) | krow;
  unsigned long long* keys = d.ht_key + d.ht0[doc];
  unsigned long long* best = d.ht_best + d.ht0[doc];
  uint32_t slot = (uint32_t)h & (cap - 1);
  for (uint32_t probe = 0; probe < cap; probe++, slot = (slot + 1) & (cap - 1)) {
    unsigned long long cur = keys[slot];
    if (cur == HT_EMPTY) {
      cur = lmw::atomic_cas64(&keys[slot], HT_EMPTY, mine);
      if (cur == HT_EMPTY) {   // this thread claimed the slot: one list entry per distinct (container, key)
        cur = mine;
        uint32_t at = lmw::atomic_add(&d.ht_cnt[doc], 1u);
        if (at >= cap / 2) { overflow(); return; }
        d.ht_list[2 * d.ht0[doc] + at] = slot;
      }
    }
    bool same = cur == mine;
    if (!same && (uint32_t)(cur >> 32) == cidx) {
      uint32_t orow = (uint32_t)cur;
      same = d.key_len[orow] == kl && bytes_eq(d.data + d.key_off[orow], ks, kl);
    }
    if (same) {
      uint32_t lam = d.chg_lamport[r.chg] + (r.ctr - ch.ctr);
      uint32_t rel = t - m.op0;
      if (rel >= (1u << 24)) { LM_SETERR(d.doc[doc].status, ST_UNSUPPORTED); return; }
      unsigned long long v = ((unsigned long long)lam << 32) | ((unsigned long long)ch.peer << 24) | rel;
      // (+1 so that 0 stays "no write".  The slot only ever grows: a row that a plain load already shows beaten needs no atomic — with
      // 16 peers x 10,000 writes on 1,024 keys all but a few dozen rows per key lose, and 328 M atomics on 2 M addresses were the kernel)
      if (best[slot] < v + 1) lmw::atomic_max64(&best[slot], v + 1);
      d.cont[m.cid0 + cidx].touched = 1;
      return;
    }
  }
  overflow();
}

// K9b: documents rendered at a checked-out version only, one wave per document, after the integrate stage.
// The reference's state store holds a root Text / List once a diff for it was not empty (diff_calc.rs:299, state.rs:1365):
// importing the batch is ONE diff from the empty version to the latest one (LoroDoc::import_batch, loro.rs:1432-1523) —
// not empty iff something is visible at the LATEST version — and the checkout a second one, latest → version.  The
// integrate stage replayed only the version's causal closure (it set `touched` when something is visible there), so
// "visible at the latest version" is decided here from the op rows alone: an applied element is visible at the latest
// version iff no applied delete row targets it.  loc[] is free at this point and serves as the mark array.
LM_KERNEL void k_seq_alive_latest(Dev d) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  if (d.front_off[doc + 1] == d.front_off[doc]) return;
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  bool need = false;
  for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) {
    const ContRow o = d.cont[m.cid0 + c];
    uint32_t ck = o.kind_root & 0xff;
    if ((ck == CK_TEXT || ck == CK_LIST) && (o.kind_root & 0x100) && !o.touched) need = true;
  }
  if (!lmw::any(need)) return;
  static constexpr uint32_t MARK = 0xFFFFFFFEu;
  uint32_t* loc = d.loc + (((uint64_t)m.elem0_hi << 32) | m.elem0_lo);
  for (uint32_t i = (uint32_t)lane; i < m.atoms; i += 64) loc[i] = NONE;
  lmw::mem_fence();
  lmw::block_sync();
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t ci = 0; ci < m.n_valid_chg; ci++) {
      uint32_t crow = d.chg_sorted[m.chg0 + ci];
      const ChangeRow ch = d.chg[crow];
      uint32_t lo = ch.ctr + d.chg_skip[crow], hi = d.peer_end_all[m.praw0 + ch.peer];
      for (uint32_t r0 = 0; r0 < ch.n_op; r0 += 64) {
        uint32_t ri = r0 + (uint32_t)lane;
        if (ri >= ch.n_op) continue;
        const OpRow r = d.op[ch.op0 + ri];
        uint32_t cidx = r.cidx_kind & 0xffff, kind = (r.cidx_kind >> 16) & 0xff;
        if (cidx >= m.n_cont) continue;
        uint32_t ck = d.cont[m.cid0 + cidx].kind_root & 0xff;
        if (ck != CK_TEXT && ck != CK_LIST) continue;
        uint32_t a = lo > r.ctr ? lo - r.ctr : 0, b = hi < r.ctr + r.len ? (hi > r.ctr ? hi - r.ctr : 0) : r.len;
        if (a >= b) continue;
        if (pass == 0 && kind == OK_DEL) {
          uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2), t0, t1;
          if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; } else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }
          if (r.a0 >= m.n_peers) continue;
          uint32_t ext = d.peer_ext[m.praw0 + r.a0], eb = d.elem_base[m.praw0 + r.a0];
          if (t1 > ext) t1 = ext;
          for (uint32_t c = t0; c < t1; c++) loc[eb + c] = MARK;
        } else if (pass == 1 && (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END)) {
          if (d.cont[m.cid0 + cidx].touched) continue;
          uint32_t eb = d.elem_base[m.praw0 + ch.peer];
          for (uint32_t c = r.ctr + a; c < r.ctr + b; c++)
            if (loc[eb + c] != MARK) { d.cont[m.cid0 + cidx].touched = 1; break; }
        }
      }
    }
    lmw::mem_fence();
    lmw::block_sync();
  }
}

// K9c: documents whose first blob was a snapshot.  An empty reference document initialises its state store from the
// snapshot's state section (fast_snapshot.rs:168-258), so every root container that section holds is part of the value —
// also one in which nothing is visible.  lm_stage passed the section's root keys; one wave per document marks them.
LM_KERNEL void k_state_roots(Dev d) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  uint64_t f0 = d.froot_off[doc], f1 = d.froot_off[doc + 1];
  if (f1 == f0) return;
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  Rd r = rd_make(d.froot + f0, f1 - f0);
  while (r.p < r.end && !r.bad) {
    uint32_t kind = rd_u8(r);
    uint64_t nl = rd_uleb(r);
    if (r.bad || nl > rd_left(r)) break;
    const uint8_t* name = r.p;
    r.p += nl;
    for (uint32_t c0 = 0; c0 < m.n_cont; c0 += 64) {
      uint32_t c = c0 + (uint32_t)lane;
      if (c >= m.n_cont) continue;
      const ContRow o = d.cont[m.cid0 + c];
      if ((o.kind_root & 0x100) && (o.kind_root & 0xff) == kind && o.name_len == nl && bytes_eq(d.data + o.name_off, name, (uint32_t)nl)) {
        d.cont[m.cid0 + c].touched = 1;
        if (kind > CK_TEXT && kind != CK_MOVABLE) lmw::atomic_or(&d.doc[doc].flags, DF_SOFT_UNSUPPORTED);   // Tree / Counter root: renders as null, the document is flagged
      }
    }
  }
}

// ---- MovableList (diff_calc.rs:1669-1993, history_cache.rs:754-1003, state/movable_list_state.rs:953-964)
// A list ITEM is placed by an insert (one per inserted value) or by a move; an ELEMENT is what an insert created, named by
// the IdLp (peer, lamport) of its insert, and it points at ONE item: that of its greatest move by (lamport, peer) inside the
// rendered version, else its insert's (last_pos).  Its value is that of its greatest set, else the inserted one (last_value).
// The value of the list = the visible items some element points at, in list order.  Both maxima live in the document's LWW
// table next to the Map keys (key = bit 63 | set-bit 62 | packed id of the element's insert item).
LM_DEV unsigned long long ml_key(uint32_t pid_e, bool is_set) { return (1ull << 63) | ((unsigned long long)(is_set ? 1u : 0u) << 62) | pid_e; }
LM_DEV uint32_t ml_hash(unsigned long long k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 29; return (uint32_t)k; }
LM_DEV uint32_t ml_ht_claim(unsigned long long* keys, uint32_t cap, unsigned long long mine) {
  uint32_t slot = ml_hash(mine) & (cap - 1);
  for (uint32_t probe = 0; probe < cap; probe++, slot = (slot + 1) & (cap - 1)) {
    unsigned long long cur = keys[slot];
    if (cur == HT_EMPTY) { cur = lmw::atomic_cas64(&keys[slot], HT_EMPTY, mine); if (cur == HT_EMPTY) return slot; }
    if (cur == mine) return slot;
  }
  return NONE;
}
LM_DEV uint32_t ml_ht_find(const unsigned long long* keys, uint32_t cap, unsigned long long mine) {
  if (cap == 0) return NONE;
  uint32_t slot = ml_hash(mine) & (cap - 1);
  for (uint32_t probe = 0; probe < cap; probe++, slot = (slot + 1) & (cap - 1)) {
    unsigned long long cur = keys[slot];
    if (cur == HT_EMPTY) return NONE;
    if (cur == mine) return slot;
  }
  return NONE;
}
// element (peer idx, lamport) → packed id of its insert item, NONE unless an applied insert row of container `cidx` made it
LM_DEV uint32_t ml_resolve(const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t peer, uint32_t lam) {
  if (peer >= m.n_peers) return NONE;
  uint32_t lo = d.peer_chg0[m.praw0 + peer], hi = d.peer_chg1[m.praw0 + peer], first = lo;
  while (lo < hi) {   // first change whose lamport is beyond `lam` (a peer's lamports grow with its counters)
    uint32_t mid = (lo + hi) >> 1;
    if (d.chg_lamport[d.chg_sorted[m.chg0 + mid]] <= lam) lo = mid + 1; else hi = mid;
  }
  if (lo == first) return NONE;
  uint32_t crow = d.chg_sorted[m.chg0 + lo - 1];
  const ChangeRow ch = d.chg[crow];
  uint32_t off = lam - d.chg_lamport[crow];
  if (off >= ch.len) return NONE;
  uint32_t c = ch.ctr + off;
  uint32_t rl = ch.op0, rh = ch.op0 + ch.n_op;
  while (rl < rh) { uint32_t mid = (rl + rh) >> 1; if (d.op[mid].ctr + d.op[mid].len <= c) rl = mid + 1; else rh = mid; }
  if (rl >= ch.op0 + ch.n_op) return NONE;
  const OpRow r = d.op[rl];
  if (r.ctr > c || (r.cidx_kind & 0xffff) != cidx || ((r.cidx_kind >> 16) & 0xff) != OK_LIST_INS) return NONE;
  return pid_make(peer, c);
}

// K9c: documents holding a MovableList only (DF_MOVABLE), one wave per document, after the integrate stage (loc[] is free
// then): loc[item] := the element the item positions, the per-element maxima of the move / set rows inside the rendered
// version, and the state-store rule — a MovableList exists once an element was inserted, even if nothing is visible any
// more (its diff lists every element the version knows: delta/movable_list.rs:32-34, diff_calc.rs:1880-1924).
LM_KERNEL void k_mlist_post(Dev d, DevDag g) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status) || !(m.flags & DF_MOVABLE)) return;
  uint32_t* loc = d.loc + (((uint64_t)m.elem0_hi << 32) | m.elem0_lo);
  uint32_t cap = d.ht_cap[doc];
  unsigned long long* keys = d.ht_key + d.ht0[doc];
  unsigned long long* best = d.ht_best + d.ht0[doc];
  int32_t err = 0;
  for (uint32_t ci = 0; ci < m.n_valid_chg; ci++) {
    uint32_t crow = d.chg_sorted[m.chg0 + ci];
    const ChangeRow ch = d.chg[crow];
    uint32_t lo = ch.ctr + d.chg_skip[crow], pe = d.peer_end[m.praw0 + ch.peer];
    uint32_t eb = d.elem_base[m.praw0 + ch.peer], ext = d.peer_ext[m.praw0 + ch.peer];
    for (uint32_t r0 = 0; r0 < ch.n_op; r0 += 64) {
      uint32_t ri = r0 + (uint32_t)lane;
      if (ri >= ch.n_op) continue;
      const OpRow r = d.op[ch.op0 + ri];
      uint32_t cidx = r.cidx_kind & 0xffff, kind = (r.cidx_kind >> 16) & 0xff;
      if (cidx >= m.n_cont || (d.cont[m.cid0 + cidx].kind_root & 0xff) != CK_MOVABLE) continue;
      if (r.ctr + r.len <= lo || r.ctr + r.len > ext) continue;   // already-known prefix of a sliced change; beyond the peer's element slots
      if (kind == OK_LIST_INS) {
        d.cont[m.cid0 + cidx].touched = 1;
        uint32_t a = lo > r.ctr ? lo - r.ctr : 0u, b = r.ctr + r.len <= pe ? r.len : (pe > r.ctr ? pe - r.ctr : 0u);
        for (uint32_t i = a; i < b; i++) loc[eb + r.ctr + i] = pid_make(ch.peer, r.ctr + i);   // an inserted item positions its own element
      } else if (kind == OK_LIST_MOVE || kind == OK_LIST_SET) {
        // the element must exist ("moved element should have a visible source position", diff_calc.rs:1927-1931)
        uint32_t pid_e = ml_resolve(d, m, cidx, r.a0, r.a1);
        if (pid_e == NONE) { err = ST_DATA_CORRUPTION; continue; }
        {
          // … and its insert must lie in the causal PAST of the row (the version at the head of the row's node, or the row's own peer's
          // earlier ops).  A writer can only name an element it has seen; a damaged dependency can leave the insert concurrent with the
          // row — the reference then fails or not depending on which of the two its iteration happens to replay first (the element is
          // looked up when the row is replayed, diff_calc.rs:1927-1931): no value that is independent of the replay order, LM_DATA_CORRUPTION
          // (found by tests/_richtext.py damaged_mixed_docs(600, seed=901), document 427: rendered where the oracle fails)
          const uint32_t q = pid_peer(pid_e), c = pid_ctr(pid_e);
          const uint64_t vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
          const uint32_t node = g.chg_node[m.chg0 + ci];
          const bool past = c < d.vvh[vvh0 + (uint64_t)node * m.n_peers + q] || (q == ch.peer && c < r.ctr);
          if (!past) { err = ST_DATA_CORRUPTION; continue; }
        }
        if (r.ctr >= pe) continue;   // past the version being rendered: does not compete (last_pos / last_value take the version)
        if (kind == OK_LIST_MOVE) loc[eb + r.ctr] = pid_e;
        uint32_t rel = ch.op0 + ri - m.op0;
        if (cap == 0 || rel >= (1u << 24)) { err = ST_UNSUPPORTED; continue; }
        uint32_t slot = ml_ht_claim(keys, cap, ml_key(pid_e, kind == OK_LIST_SET));
        if (slot == NONE) { err = ST_INTERNAL; continue; }
        uint32_t lam = d.chg_lamport[crow] + (r.ctr - ch.ctr);
        lmw::atomic_max64(&best[slot], (((unsigned long long)lam << 32) | ((unsigned long long)ch.peer << 24) | rel) + 1);
      }
    }
  }
  if (err) LM_SETERR(d.doc[doc].status, err);
}

// ------------------------------------------------------------------------------------------------ sink
struct Sink {
  uint8_t* out;       // nullptr in the sizing pass
  uint64_t pos;       // wave-uniform; keeps counting past `cap`, so the exact size is known even when the slab was too small
  uint64_t cap;       // bytes available at `out`: nothing is ever written at or beyond it
};
LM_DEV void sink_byte(Sink& s, uint8_t b) {
  if (s.out && s.pos < s.cap && lmw::lane() == 0) s.out[s.pos] = b;
  s.pos++;
}
LM_DEV void sink_lit(Sink& s, const char* lit, uint32_t n) {   // n <= 64: lane i writes byte i
  if (s.out && s.pos + n <= s.cap && (uint32_t)lmw::lane() < n) s.out[s.pos + (uint32_t)lmw::lane()] = (uint8_t)lit[lmw::lane()];
  s.pos += n;
}
// each lane contributes `n` (<= 8) bytes packed little-endian in `bytes`; lanes are concatenated in lane order
LM_DEV void sink_lanes(Sink& s, uint64_t bytes, uint32_t n) {
  uint32_t inc = lmw::scan_incl_add(n);
  uint32_t tot = lmw::bcast(inc, 63);
  if (s.out && s.pos + tot <= s.cap) {
    uint64_t at = s.pos + inc - n;
    for (uint32_t i = 0; i < n; i++) s.out[at + i] = (uint8_t)(bytes >> (8 * i));
  }
  s.pos += tot;
}
LM_DEV char hexd(uint32_t v) { return (char)(v < 10 ? '0' + v : 'a' + (v - 10)); }
// JSON escape of one byte of a UTF-8 string (bytes >= 0x80 pass through)
LM_DEV void esc_byte(uint32_t c, uint64_t& bytes, uint32_t& n) {
  if (c == '"') { bytes = (uint64_t)'\\' | ((uint64_t)'"' << 8); n = 2; }
  else if (c == '\\') { bytes = (uint64_t)'\\' | ((uint64_t)'\\' << 8); n = 2; }
  else if (c >= 0x20) { bytes = c; n = 1; }
  else {
    char e = 0;
    if (c == 8) e = 'b'; else if (c == 12) e = 'f'; else if (c == 10) e = 'n'; else if (c == 13) e = 'r'; else if (c == 9) e = 't';
    if (e) { bytes = (uint64_t)'\\' | ((uint64_t)e << 8); n = 2; }
    else {
      bytes = (uint64_t)'\\' | ((uint64_t)'u' << 8) | ((uint64_t)'0' << 16) | ((uint64_t)'0' << 24) |
              ((uint64_t)hexd(c >> 4) << 32) | ((uint64_t)hexd(c & 15) << 40);
      n = 6;
    }
  }
}
LM_DEV void sink_escaped(Sink& s, const uint8_t* p, uint32_t len) {  // without the quotes
  int lane = lmw::lane();
  for (uint32_t c0 = 0; c0 < len; c0 += 64) {
    uint32_t i = c0 + (uint32_t)lane;
    uint64_t bytes = 0;
    uint32_t n = 0;
    if (i < len) esc_byte(p[i], bytes, n);
    sink_lanes(s, bytes, n);
  }
}
LM_DEV void sink_string(Sink& s, const uint8_t* p, uint32_t len) {
  sink_byte(s, '"');
  sink_escaped(s, p, len);
  sink_byte(s, '"');
}
// (digits from three 9-digit chunks in 32-bit arithmetic: a 64-bit `% 10` / `/ 10` per digit is a ≈150-instruction software
// division each on this target — ≈4,000 instructions for a 13-digit Map value, 90 % of the renderer's time on configs[2] and
// most of the List renderer's on configs[3], tests/tools/gpu_prof_emit.py)
LM_DEV void sink_i64(Sink& s, int64_t v) {
  // (every lane works the digits out — wave-uniform arithmetic, least significant first, one digit per byte of three registers — and
  // lane i writes byte i.  Rounds 2-4a kept them in an indexed private array, which is scratch memory on this target: a memory round
  // trip per digit for lane 0, then one single-byte store after the other.)
  uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
  // (a value below 10^9 — list indices, small counters, the bytes of a binary value — needs neither of the two 64-bit divisions)
  uint64_t q1 = u < 1000000000ull ? 0ull : u / 1000000000ull, q2 = q1 < 1000000000ull ? 0ull : q1 / 1000000000ull;
  uint32_t c0 = (uint32_t)(u - q1 * 1000000000ull), c1 = (uint32_t)(q1 - q2 * 1000000000ull), c2 = (uint32_t)q2;
  const int top = c2 ? 2 : (c1 ? 1 : 0);
  uint64_t d0 = 0, d1 = 0, d2 = 0;
  uint32_t nd = 0;
  for (int ci = 0; ci <= top; ci++) {
    uint32_t c = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
    for (int i = 0; i < 9 && (ci < top || c || i == 0); i++) {
      uint64_t dgt = '0' + c % 10u;
      c /= 10u;
      if (nd < 8) d0 |= dgt << (8 * nd); else if (nd < 16) d1 |= dgt << (8 * (nd - 8)); else d2 |= dgt << (8 * (nd - 16));
      nd++;
    }
  }
  const uint32_t n = nd + (v < 0 ? 1u : 0u), lane = (uint32_t)lmw::lane();
  if (s.out && s.pos + n <= s.cap && lane < n) {
    const uint32_t j = n - 1 - lane;      // digit index from the least significant one; j == nd: the sign
    s.out[s.pos + lane] = j == nd ? (uint8_t)'-' : (uint8_t)(j < 8 ? d0 >> (8 * j) : (j < 16 ? d1 >> (8 * (j - 8)) : d2 >> (8 * (j - 16))));
  }
  s.pos += n;
}
// one unicode scalar of a Text container → escaped UTF-8 (anchors contribute nothing)
LM_DEV void cp_bytes(uint32_t cp, uint64_t& bytes, uint32_t& n) {
  if (cp >= CP_ANCHOR) { n = 0; bytes = 0; return; }
  if (cp < 0x80) { esc_byte(cp, bytes, n); return; }
  if (cp < 0x800) { bytes = (0xC0 | (cp >> 6)) | ((uint64_t)(0x80 | (cp & 0x3F)) << 8); n = 2; }
  else if (cp < 0x10000) {
    bytes = (0xE0 | (cp >> 12)) | ((uint64_t)(0x80 | ((cp >> 6) & 0x3F)) << 8) | ((uint64_t)(0x80 | (cp & 0x3F)) << 16);
    n = 3;
  } else {
    bytes = (0xF0 | (cp >> 18)) | ((uint64_t)(0x80 | ((cp >> 12) & 0x3F)) << 8) | ((uint64_t)(0x80 | ((cp >> 6) & 0x3F)) << 16) |
            ((uint64_t)(0x80 | (cp & 0x3F)) << 24);
    n = 4;
  }
}

// render one plain LoroValue at `r` (wave-uniform parse): scalars, f64, binary, nested lists and map values.
// Map values are written with their keys in bytewise order (canonical JSON): the entries of a map frame are re-scanned
// for the next larger key each time — O(K²) parses, but map VALUES are small (a child container is the tool for big
// maps).  Key indices refer to the key table of the value's own block (`key0`, `n_keys`).  A container below the top
// level of a value was rejected at decode; a top-level child container is resolved by the caller.
// `vb` = the value's block, or NONE: then it is looked up (last block of [blk0, blk0+n_blk) starting at or before the
// value) only if a map-typed value actually turns up.
LM_DEV void sink_value(Sink& s, Rd& r, int32_t& err, const Dev& d, uint32_t vb, uint32_t blk0, uint32_t n_blk) {
  const uint8_t* base = r.p;
  uint32_t key0 = 0, n_keys = 0;
  bool have_keys = false;
  // frame stack in LDS (the parse is wave-uniform; per-lane arrays would be scratch memory for every wave of the kernel):
  // 5 words per frame = remaining items | items written | first entry offset | end offset | last key row; lane 0 writes
  LM_SHARED(uint32_t, s_vf, 5 * 16);
  int lane = lmw::lane();
  auto fset = [&](int f, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4) {
    lmw::wave_sync();
    if (lane == 0) { s_vf[5 * f] = c0; s_vf[5 * f + 1] = c1; s_vf[5 * f + 2] = c2; s_vf[5 * f + 3] = c3; s_vf[5 * f + 4] = c4; }
    lmw::wave_sync();
  };
  uint32_t f_map = 0;   // bit i: frame i is a map
  // Map frames of up to 64 entries are ORDERED ONCE when they open (f_sorted): the walk that finds the map's end also notes every
  // entry's key row and value position, each lane then takes one entry — its key's first eight bytes as a big-endian word, the whole
  // strings only on a tie — ranks it against the others (of equal keys the last occurrence wins) and the entries are laid out in
  // key order in an LDS pool the frame reads one by one.  (Re-scanning the frame for the next larger key each time — the fallback
  // for larger maps and an exhausted pool — is K passes of K entries with four dependent loads per comparison: a two-entry map
  // nested in a list item cost as much as twenty scalar items.)
  static constexpr uint32_t MPOOL = 256;
  LM_SHARED(uint32_t, s_mrow, MPOOL);
  LM_SHARED(uint32_t, s_mpos, MPOOL);
  uint32_t f_sorted = 0, pool_top = 0;
#ifdef LM_PROF_EMIT
  uint32_t pf_tag = 0; uint64_t pf_t0 = 0;
#endif
  int sp = 0;
  bool pending = true;  // a value waits at r.p
  for (uint32_t guard = 0; guard < (1u << 28); guard++) {
    if (r.bad) { err = ST_DATA_CORRUPTION; return; }
    if (!pending) {
#ifdef LM_PROF_EMIT
      if (sp == 0 && lane == 0) atomicAdd(&d.prof[pf_tag], (unsigned long long)(lmw::clock() - pf_t0));
#endif
      if (sp == 0) return;
      int top = sp - 1;
      lmw::wave_sync();
      uint32_t cnt = s_vf[5 * top], done = s_vf[5 * top + 1], start = s_vf[5 * top + 2], endo = s_vf[5 * top + 3], last = s_vf[5 * top + 4];
      if (!((f_map >> top) & 1)) {
        if (cnt == 0) { sink_byte(s, ']'); sp--; continue; }
        if (done) sink_byte(s, ',');
        fset(top, cnt - 1, done + 1, start, endo, last);
      } else if ((f_sorted >> top) & 1) {
        // (cnt = entries that show, done = written so far, start = the frame's first pool slot)
        if (done >= cnt) { sink_byte(s, '}'); r.p = base + endo; pool_top = start; sp--; continue; }
        const uint32_t krow = s_mrow[start + done], vpos = s_mpos[start + done];
        if (done) sink_byte(s, ',');
        sink_string(s, d.data + d.key_off[krow], d.key_len[krow]);
        sink_byte(s, ':');
        fset(top, cnt, done + 1, start, endo, last);
        r.p = base + vpos;
      } else {
        bool fin = done >= cnt;
        uint32_t best_row = NONE, best_pos = 0;
        if (!fin) {
          // next key in bytewise order: the smallest key greater than the last one written (of equal keys the last
          // occurrence wins, like a map built by successive inserts)
          const uint8_t* sp0 = base + start;
          Rd q = rd_make(sp0, (uint64_t)(r.end - sp0));
          for (uint32_t i = 0; i < cnt && !q.bad; i++) {
            uint64_t kidx = rd_uleb(q);
            if (kidx >= n_keys) { err = ST_DATA_CORRUPTION; return; }
            uint32_t row = key0 + (uint32_t)kidx;
            uint32_t vpos = (uint32_t)(q.p - base);
            bool u = false;
            skip_loro_value(q, u);
            bool gt = last == NONE || bytes_cmp(d.data + d.key_off[row], d.key_len[row], d.data + d.key_off[last], d.key_len[last]) > 0;
            if (gt && (best_row == NONE ||
                       bytes_cmp(d.data + d.key_off[row], d.key_len[row], d.data + d.key_off[best_row], d.key_len[best_row]) <= 0)) {
              best_row = row; best_pos = vpos;
            }
          }
          if (q.bad) { err = ST_DATA_CORRUPTION; return; }
          if (best_row == NONE) fin = true;   // the remaining entries repeat keys already written
        }
        if (fin) { sink_byte(s, '}'); r.p = base + endo; sp--; continue; }
        if (done) sink_byte(s, ',');
        sink_string(s, d.data + d.key_off[best_row], d.key_len[best_row]);
        sink_byte(s, ':');
        fset(top, cnt, done + 1, start, endo, best_row);
        r.p = base + best_pos;
      }
      pending = true;
      continue;
    }
    pending = false;
    uint32_t tag = rd_u8(r);
#ifdef LM_PROF_EMIT   // (experiment build: ticks per kind of top-level value, d.prof[0..7] — closed when the next top-level value starts)
    if (sp == 0) { pf_tag = tag < 3 ? 0u : (tag - 2 < 7 ? tag - 2 : 7u); pf_t0 = lmw::clock(); }
#endif
    switch (tag) {
      case 0: sink_lit(s, "null", 4); break;
      case 1: sink_lit(s, "true", 4); break;
      case 2: sink_lit(s, "false", 5); break;
      case 3: sink_i64(s, rd_sleb(r)); break;
      case 4: {  // f64, big endian (docs/encoding.md §10.1)
        uint64_t bits = 0;
        for (int k = 0; k < 8; k++) bits = (bits << 8) | rd_u8(r);
        // one lane formats (bignum workspace and text in LDS), every lane learns the length
        LM_SHARED(Big, s_big, 6);
        LM_SHARED(uint32_t, s_f64, 9);
        lmw::block_sync();
        if (lane == 0) s_f64[8] = (uint32_t)f64_json(bits, (char*)s_f64, s_big);
        lmw::block_sync();
        uint32_t n = s_f64[8];
        if (s.out && s.pos + n <= s.cap && (uint32_t)lane < n) s.out[s.pos + (uint32_t)lane] = ((const uint8_t*)s_f64)[lane];   // (n <= 32)
        s.pos += n;
        break;
      }
      case 5: { uint64_t l = rd_uleb(r); if (l > rd_left(r)) { err = ST_DATA_CORRUPTION; return; } sink_string(s, r.p, (uint32_t)l); rd_skip(r, l); break; }
      case 6: {  // binary → array of ints
        uint64_t l = rd_uleb(r);
        if (l > rd_left(r)) { err = ST_DATA_CORRUPTION; return; }
        sink_byte(s, '[');
        for (uint64_t i = 0; i < l; i++) { if (i) sink_byte(s, ','); sink_i64(s, r.p[i]); }
        sink_byte(s, ']');
        rd_skip(r, l);
        break;
      }
      case 7: case 8: {
        uint64_t n = rd_uleb(r);
        if (sp >= 16 || n > (1u << 28)) { err = ST_UNSUPPORTED; return; }
        uint32_t start = (uint32_t)(r.p - base), endo = 0;
        if (tag == 8 && !have_keys) {
          if (vb == NONE) {
            uint32_t lo = 0, hi = n_blk;
            uint64_t at = (uint64_t)(base - d.data);
            while (lo + 1 < hi) { uint32_t mid = (lo + hi) >> 1; if (d.blk[blk0 + mid].base <= at) lo = mid; else hi = mid; }
            vb = blk0 + lo;
          }
          key0 = d.boff[(uint64_t)vb * BCN + BC_KEY]; n_keys = d.bcnt[(uint64_t)vb * BCN + BC_KEY];
          have_keys = true;
        }
        if (tag == 8) {
          const bool sorted = n <= 64 && pool_top + (uint32_t)n <= MPOOL;
          Rd q = r;   // the map frame jumps around inside its entries: find where the map ends first
          lmw::wave_sync();
          for (uint64_t i = 0; i < n && !q.bad; i++) {
            uint64_t kidx = rd_uleb(q);
            if (sorted) {
              if (kidx >= n_keys) { err = ST_DATA_CORRUPTION; return; }
              if (lane == 0) { s_mrow[pool_top + i] = key0 + (uint32_t)kidx; s_mpos[pool_top + i] = (uint32_t)(q.p - base); }
            }
            bool u = false;
            skip_loro_value(q, u);
          }
          if (q.bad) { err = ST_DATA_CORRUPTION; return; }
          endo = (uint32_t)(q.p - base);
          f_map |= 1u << sp;
          f_sorted &= ~(1u << sp);
          sink_byte(s, '{');
          if (sorted) {
            lmw::wave_sync();
            const uint32_t cnt = (uint32_t)n;
            const bool in = (uint32_t)lane < cnt;
            const uint32_t my_row = in ? s_mrow[pool_top + lane] : 0u, my_pos = in ? s_mpos[pool_top + lane] : 0u;
            const uint8_t* kp = d.data + (in ? d.key_off[my_row] : 0ull);
            const uint32_t kl = in ? d.key_len[my_row] : 0u;
            uint32_t p_hi = 0, p_lo = 0;   // the key's first eight bytes, big endian, zero padded
            for (uint32_t b8 = 0; b8 < 8; b8++) { uint32_t c = (in && b8 < kl) ? kp[b8] : 0u; if (b8 < 4) p_hi = (p_hi << 8) | c; else p_lo = (p_lo << 8) | c; }
            uint64_t less_m = 0;           // bit j: key j sorts in front of this lane's key
            bool shadowed = false;         // a later entry carries the same key
            for (uint32_t j = 0; j < cnt; j++) {
              const uint32_t jh = lmw::shfl(p_hi, (int)j), jl = lmw::shfl(p_lo, (int)j), jrow = lmw::shfl(my_row, (int)j), jlen = lmw::shfl(kl, (int)j);
              int c = jh != p_hi ? (jh < p_hi ? -1 : 1) : (jl != p_lo ? (jl < p_lo ? -1 : 1) : 0);
              if (in && c == 0 && (kl > 8 || jlen > 8 || kl != jlen)) c = bytes_cmp(d.data + d.key_off[jrow], jlen, kp, kl);   // (a tie of the padded words: the strings decide)
              if (c < 0) less_m |= 1ull << j;
              if (c == 0 && j > (uint32_t)lane) shadowed = true;
            }
            const uint64_t sh_m = lmw::ballot(in && shadowed);
            const uint32_t rank = (uint32_t)lmw::popc64(less_m & ~sh_m);
            lmw::wave_sync();
            if (in && !shadowed) { s_mrow[pool_top + rank] = my_row; s_mpos[pool_top + rank] = my_pos; }
            lmw::wave_sync();
            f_sorted |= 1u << sp;
            fset(sp, cnt - (uint32_t)lmw::popc64(sh_m), 0, pool_top, endo, NONE);
            pool_top += cnt;
            sp++;
            break;
          }
        } else {
          f_map &= ~(1u << sp);
          sink_byte(s, '[');
        }
        fset(sp, (uint32_t)n, 0, start, endo, NONE);
        sp++;
        break;
      }
      default: err = ST_UNSUPPORTED; return;
    }
  }
}

static constexpr uint32_t EMIT_MAX_DEPTH = 16;   // nesting depth of child containers the emitter follows

// K11: one wave per doc — JSON of the deep value (mode 0: size only, mode 1: write) and the VV bytes.
// Two instantiations share this body: documents made of Text containers only (the streaming pipeline, few registers,
// five waves per SIMD) and everything else (List / Map rendering, child containers, map-typed values, f64).
// `pass` 0 renders every document into its optimistic slab [out_off[doc], out_off[doc+1]); a document whose JSON does not fit
// is left flagged DF_REEMIT with its exact size in out_len.  `pass` 1 re-renders only those documents (the host has moved
// their slabs to exactly-sized ones).
#ifdef LM_PROF_EMIT   // experiment build: where the renderer's time goes (tests/tools/gpu_prof_emit.py) — ticks per kind of container
#define EMIT_PH(i) do { uint64_t n_ = lmw::clock(); epacc[i] += n_ - eptp; eptp = n_; } while (0)
#else
#define EMIT_PH(i) do {} while (0)
#endif
template <bool TEXT_ONLY>
LM_DEV void emit_doc(Dev d, int mode, int pass) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  LM_SHARED(uint32_t, s_order, MAX_ROOTS);
  DocMeta m = d.doc[doc];
  uint32_t C = m.n_cont;
  bool failed = status_fatal(m.status);
  {
    bool other = false;
    if (!failed) for (uint32_t c0 = (uint32_t)lane; c0 < C; c0 += 64) other |= (d.cont[m.cid0 + c0].kind_root & 0xff) != CK_TEXT;
    if (TEXT_ONLY == lmw::any(other)) return;   // failed documents are closed by the text-only instantiation
  }
  if (failed) { if (lane == 0 && pass == 0) { d.doc[doc].out_len = 0; d.doc[doc].vv_len = 0; } return; }
  if (pass != 0 && !(m.flags & DF_REEMIT)) return;
  int32_t err = 0;
  bool soft = false;   // met a child container of a kind outside the device scope that never received an op
  // ---- root containers that received an applied op, ordered bytewise by name
  LM_SHARED(uint32_t, s_root, MAX_ROOTS);
  uint32_t n_roots = 0;
  for (uint32_t c0 = 0; c0 < C; c0 += 64) {
    bool isr = false;
    if (c0 + (uint32_t)lane < C) { const ContRow o = d.cont[m.cid0 + c0 + lane]; isr = (o.kind_root & 0x100) && o.touched; }
    uint64_t rm = lmw::ballot(isr);
    uint32_t at = n_roots + (uint32_t)lmw::popc64(rm & ((1ull << lane) - 1));
    if (isr && at < MAX_ROOTS) s_root[at] = c0 + (uint32_t)lane;
    n_roots += (uint32_t)lmw::popc64(rm);
  }
  // roots that only the state section of the initialising snapshot knows (k_state_roots marks those the history addresses too):
  // no op ever touched them, the state store holds them all the same (fast_snapshot.rs:168-258) — they render as the empty value
  // of their kind.  Entry = GHOST | byte offset of its [kind, uleb len, name] record in froot.
  static constexpr uint32_t GHOST = 0x80000000u;
  {
    uint64_t f0 = d.froot_off[doc], f1 = d.froot_off[doc + 1];
    Rd fr = rd_make(d.froot + f0, f1 - f0);
    while (fr.p < fr.end && !fr.bad && n_roots <= MAX_ROOTS) {
      uint32_t at = (uint32_t)(fr.p - (d.froot + f0));
      uint32_t gk = rd_u8(fr);
      uint64_t nl = rd_uleb(fr);
      if (fr.bad || nl > rd_left(fr)) break;
      const uint8_t* nm = fr.p;
      fr.p += nl;
      bool known = false;
      for (uint32_t c0 = (uint32_t)lane; c0 < C; c0 += 64) {
        const ContRow o = d.cont[m.cid0 + c0];
        known |= (o.kind_root & 0x100) && (o.kind_root & 0xff) == gk && o.name_len == nl && bytes_eq(d.data + o.name_off, nm, (uint32_t)nl);
      }
      if (lmw::any(known)) continue;
      if (n_roots < MAX_ROOTS && lane == 0) s_root[n_roots] = GHOST | at;
      n_roots++;
    }
  }
  if (n_roots > MAX_ROOTS) { err = ST_UNSUPPORTED; n_roots = 0; }
  lmw::block_sync();
  // name (and, for a state-only root, kind) of an entry of s_root / s_order
  auto root_name = [&](uint32_t e, const uint8_t*& np, uint32_t& nl, uint32_t& gkind) {
    if (e & GHOST) {
      Rd fr = rd_make(d.froot + d.froot_off[doc] + (e & ~GHOST), d.froot_off[doc + 1] - d.froot_off[doc] - (e & ~GHOST));
      gkind = rd_u8(fr);
      nl = (uint32_t)rd_uleb(fr);
      np = fr.p;
    } else {
      const ContRow o = d.cont[m.cid0 + e];
      np = d.data + o.name_off; nl = o.name_len; gkind = o.kind_root & 0xff;
    }
  };
  {
    bool mine = (uint32_t)lane < n_roots;
    const uint8_t* my_p = d.data;
    uint32_t my_l = 0, my_k = 0;
    if (mine) root_name(s_root[lane], my_p, my_l, my_k);
    uint32_t rank = 0;
    bool dup = false;
    for (uint32_t j = 0; j < n_roots; j++) {
      const uint8_t* op = d.data;
      uint32_t ol = 0, ok = 0;
      root_name(s_root[j], op, ol, ok);
      if (mine && j != (uint32_t)lane) {
        int c = bytes_cmp(op, ol, my_p, my_l);
        if (c < 0) rank++;
        if (c == 0) dup = true;
      }
    }
    if (lmw::any(mine && dup)) err = ST_UNSUPPORTED;  // two roots share a name (state.rs:1352-1392 picks by registration order)
    if (mine && !dup) s_order[rank] = s_root[lane];
  }
  lmw::block_sync();
  uint64_t elem0 = ((uint64_t)m.elem0_hi << 32) | m.elem0_lo;
  uint64_t doc_data0 = d.blob_off[d.doc_blob[doc]];
  // which items show: a replay from the empty version of exactly the rendered version's ops leaves "never deleted" as the
  // answer; a resident tracker (lm_k_integrate_span.h, DevRes) holds every applied op and stands AT the rendered version:
  // an item shows iff it is active there (not future, delete count 0)
  // (only when it was moved to a checked-out version: rendered at the latest version, every applied op is in the tracker and
  // "never deleted" is the answer as well — the integrate stage then leaves the tracker where its last change put it)
  const uint32_t vis_mask = (d.res_vis && d.front_off[doc + 1] > d.front_off[doc] && !(m.flags & DF_FRONT_ERR)) ? (ST_FUT | ST_DELMASK) : ST_EVER;
  LM_SHARED(uint32_t, s_eb, MAX_PEERS);   // element base per peer (the text gather reads it once per lane and leaf)
  for (uint32_t p = (uint32_t)lane; p < m.n_peers && p < MAX_PEERS; p += 64) s_eb[p] = d.elem_base[m.praw0 + p];
  lmw::block_sync();
  Sink s;
  s.out = mode ? d.out + d.out_off[doc] : nullptr;
  s.pos = 0;
  s.cap = d.out_off[doc + 1] - d.out_off[doc];
  // Containers nest: a Map value or a List item may be a child container (LoroValue::Container — on the wire only
  // the kind; its id is the id of the op / list element that created it, docs/encoding.md:967-1003, state.rs:1550-1616).
  // Rendering is therefore a small stack machine; a frame = (container, resume position).
  //   List:  a = next leaf of the container's directory, b = next slot in that leaf, c = 1 until an item was written
  //   Map :  a = next entry of the sorted key list, b = start of that list in the document's sort scratch, c = #entries
  LM_SHARED(uint32_t, s_frame, 4 * EMIT_MAX_DEPTH);
  auto frame_set = [&](int i, uint32_t cidx, uint32_t fa, uint32_t fb, uint32_t fc) {
    lmw::block_sync();
    if (lane == 0) { s_frame[4 * i] = cidx; s_frame[4 * i + 1] = fa; s_frame[4 * i + 2] = fb; s_frame[4 * i + 3] = fc; }
    lmw::block_sync();
  };
  // child container (peer idx, counter, kind) → container index of the document, NONE when it never received an op
  auto find_child = [&](uint32_t peer, uint32_t ctr, uint32_t ckind) -> uint32_t {
    for (uint32_t c0 = 0; c0 < C; c0 += 64) {
      bool hit = false;
      if (c0 + (uint32_t)lane < C) { const ContRow o = d.cont[m.cid0 + c0 + lane]; hit = !(o.kind_root & 0x100) && (o.kind_root & 0xff) == ckind && o.peer == peer && o.counter == ctr; }
      uint64_t hm = lmw::ballot(hit);
      if (hm) return c0 + (uint32_t)lmw::ffs64(hm);
    }
    return NONE;
  };
  auto empty_child = [&](uint32_t ckind) {
    if (ckind == CK_TEXT) sink_lit(s, "\"\"", 2);
    else if (ckind == CK_LIST || ckind == CK_MOVABLE) sink_lit(s, "[]", 2);
    else if (ckind == CK_MAP) sink_lit(s, "{}", 2);
    else { sink_lit(s, "null", 4); soft = true; }   // Tree / Counter child: outside the device scope
  };
  const uint32_t ht_capd = d.ht_cap[doc];
  const unsigned long long* keys = d.ht_key + d.ht0[doc];
  const unsigned long long* best = d.ht_best + d.ht0[doc];
  unsigned long long* pfx = d.ht_pfx + d.ht0[doc];                  // per slot: key prefix for the sort
  const uint32_t* claimed = d.ht_list + 2 * d.ht0[doc];            // [0, cap/2): one slot per distinct (container, key)
  uint32_t* scratch = d.ht_list + 2 * d.ht0[doc] + ht_capd / 2;     // [cap/2, 2·cap): sorted key lists of the open maps
  uint32_t scratch_top = 0;
  uint64_t doc_end = 0;   // list item values are bounded by the end of the document's last blob
  if (d.doc_blob[doc + 1] > d.doc_blob[doc]) doc_end = d.blob_off[d.doc_blob[doc + 1] - 1] + d.blob_len[d.doc_blob[doc + 1] - 1];
#ifdef LM_PROF_EMIT
  uint64_t epacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, eptp = lmw::clock();   // 0 set-up (roots, order) | 1 text | 2 list | 3 map sort | 4 map entries | 5 version vector | 6 root names
#endif
  sink_byte(s, '{');
  EMIT_PH(0);
  for (uint32_t oi = 0; oi < n_roots && !err; oi++) {
    {
      const uint8_t* np = d.data;
      uint32_t nl = 0, gk = 0;
      root_name(s_order[oi], np, nl, gk);
      if (oi) sink_byte(s, ',');
      sink_string(s, np, nl);
      sink_byte(s, ':');
      if (s_order[oi] & GHOST) { empty_child(gk); continue; }
    }
    int sp = 1;
    EMIT_PH(6);
    frame_set(0, s_order[oi], 0, 0, 0x3);   // c bit 1 = frame not entered yet
    while (sp > 0 && !err) {
      lmw::block_sync();
      uint32_t cidx = s_frame[4 * (sp - 1)], fa = s_frame[4 * (sp - 1) + 1], fb = s_frame[4 * (sp - 1) + 2], fc = s_frame[4 * (sp - 1) + 3];
      uint32_t kind = d.cont[m.cid0 + cidx].kind_root & 0xff;
      if (kind == CK_TEXT && d.span) {
        // span-granular leaves: per leaf, the visible runs are flattened 64 elements per step — lane → (run, offset)
        // by a search over the running lengths kept in LDS; a run's elements are consecutive in tb[] — one BYTE per element:
        // the scalar itself when it is ASCII, TB_WIDE when cp[] holds it (a multi-byte scalar), TB_ANCHOR for a style anchor
        sink_byte(s, '"');
        uint32_t r0 = d.cont_root0[m.cid0 + cidx], nr = d.cont_nroot[m.cid0 + cidx];
        const uint32_t* dirp = d.dir_out + m.leaf0 + r0;
        LM_SHARED(uint32_t, s_inc, 64);
        LM_SHARED(uint32_t, s_g0, 64);
        // (one wave renders a document, so every HBM round trip is on its critical path: the leaf records are requested one
        // leaf ahead, the directory entries two, and the byte gathers of four steps go out together — a step-at-a-time loop
        // took ≈1,200 dependent round trips per configs[1] document)
        #ifndef LM_EMIT_EU
#define LM_EMIT_EU 4
#endif
        static constexpr int EU = LM_EMIT_EU;
        uint32_t de1 = nr > 0 ? dirp[0] : 0u, de2 = nr > 1 ? dirp[1] : 0u;
        uint32_t p_id = NONE, p_ln = 0, p_st = ST_EVER;
        if (nr > 0 && (uint32_t)lane < de_n(de1)) {
          const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + de_leaf(de1)) * SP_REC;
          p_id = rec[lane]; p_ln = rec[64 + lane]; p_st = rec[256 + lane];
        }
        for (uint32_t ri = 0; ri < nr && !err; ri++) {
          uint32_t id0 = p_id, ln = p_ln, st = p_st;
          de1 = de2;
          de2 = ri + 2 < nr ? dirp[ri + 2] : 0u;
          p_id = NONE; p_ln = 0; p_st = ST_EVER;
          if (ri + 1 < nr && (uint32_t)lane < de_n(de1)) {
            const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + de_leaf(de1)) * SP_REC;
            p_id = rec[lane]; p_ln = rec[64 + lane]; p_st = rec[256 + lane];
          }
          uint32_t vl = (id0 != NONE && !(st & vis_mask)) ? ln : 0u;
          uint32_t inc = lmw::scan_incl_add(vl);
          uint32_t total = lmw::bcast(inc, 63);
          lmw::block_sync();
          s_inc[lane] = inc;
          s_g0[lane] = vl ? s_eb[pid_peer(id0)] + pid_ctr(id0) - (inc - vl) : 0u;   // element index minus position: tb index = g0 + position
          lmw::block_sync();
          for (uint32_t e0 = 0; e0 < total; e0 += 64 * EU) {
            uint32_t cv[EU], gv[EU];
#pragma unroll
            for (int u = 0; u < EU; u++) {
              uint32_t e = e0 + 64u * (uint32_t)u + (uint32_t)lane;
              cv[u] = 0x20; gv[u] = 0;
              if (e < total) {
                uint32_t lo = 0, hi = 63;                  // first run whose running length exceeds e
                while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (s_inc[mid] > e) hi = mid; else lo = mid + 1; }
                gv[u] = s_g0[lo] + e;                      // 32-bit wrap-around arithmetic: g0 may be "negative"
                cv[u] = d.tb[elem0 + gv[u]];
              }
            }
#pragma unroll
            for (int u = 0; u < EU; u++) {
              uint32_t eu = e0 + 64u * (uint32_t)u;
              if (eu >= total) break;
              uint32_t e = eu + (uint32_t)lane, cpv = cv[u];
              // a step of plain ASCII (nothing to escape): scalar = byte, lane = output position — no scan, no byte loop
              if (!lmw::ballot(cpv < 0x20 || cpv >= 0x80 || cpv == '"' || cpv == '\\')) {
                uint32_t cnt = total - eu < 64 ? total - eu : 64;
                if (s.out && s.pos + cnt <= s.cap && e < total) s.out[s.pos + (uint32_t)lane] = (uint8_t)cpv;
                s.pos += cnt;
              } else {
                uint64_t bytes = 0;
                uint32_t nb = 0;
                if (e < total) {
                  if (cpv == TB_WIDE) cpv = d.cp[elem0 + gv[u]]; else if (cpv == TB_ANCHOR) cpv = 0xFFFFFFFFu;
                  cp_bytes(cpv, bytes, nb);
                }
                sink_lanes(s, bytes, nb);
              }
            }
          }
        }
        sink_byte(s, '"');
        sp--;
        EMIT_PH(1);
      } else if (kind == CK_TEXT) {
        sink_byte(s, '"');
        uint32_t r0 = d.cont_root0[m.cid0 + cidx], nr = d.cont_nroot[m.cid0 + cidx];
        // software pipeline over the leaves, six iterations deep: the directory entry of leaf i+6, the (id, status) of
        // leaf i+4 and the payload gather of leaf i+2 are in flight while leaf i is rendered — every dependent HBM
        // round trip has two iterations to complete (element bases come from LDS, so the gather is a single load)
        const uint32_t* dirp = d.dir_out + m.leaf0 + r0;
        uint32_t de_a = 0, de_b = 0, id_a = NONE, id_b = NONE, st_a = ST_EVER, st_b = ST_EVER, pay_a = 0, pay_b = 0;
        bool dev_a = false, dev_b = false, vis_a = false, vis_b = false;
        for (uint32_t it = 0; it < nr + 6 && !err; it++) {
          if (it >= 6) {                                   // render leaf it-6
            uint64_t bytes = 0;
            uint32_t nb = 0;
            if (vis_b) cp_bytes(pay_b, bytes, nb);
            sink_lanes(s, bytes, nb);
          }
          bool n_vis = id_b != NONE && !(st_b & ST_EVER);   // gather for leaf it-4
          uint32_t n_pay = n_vis ? d.cp[elem0 + s_eb[pid_peer(id_b)] + pid_ctr(id_b)] : 0u;
          uint32_t n_id = NONE, n_st = ST_EVER;             // (id, status) of leaf it-2
          if (dev_b) {
            uint32_t L = de_leaf(de_b), n = de_n(de_b);
            if ((uint32_t)lane < n) { const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + L) * 256; n_id = rec[lane]; n_st = rec[192 + lane]; }
          }
          bool n_dev = it < nr;                             // directory entry of leaf it
          uint32_t n_de = n_dev ? dirp[it] : 0u;
          pay_b = pay_a; vis_b = vis_a; pay_a = n_pay; vis_a = n_vis;
          id_b = id_a; st_b = st_a; id_a = n_id; st_a = n_st;
          de_b = de_a; dev_b = dev_a; de_a = n_de; dev_a = n_dev;
        }
        sink_byte(s, '"');
        sp--;
      } else if (!TEXT_ONLY && (kind == CK_LIST || kind == CK_MOVABLE)) {
        if (fc & 2) { sink_byte(s, '['); fc &= ~2u; }
        uint32_t r0 = d.cont_root0[m.cid0 + cidx], nr = d.cont_nroot[m.cid0 + cidx];
        const uint32_t* dirp = d.dir_out + m.leaf0 + r0;
        bool pushed = false;
        uint32_t ri = fa, slot0 = fb, k0 = fc >> 2;   // resume: leaf, slot, element of the slot's run (span-granular leaves)
        fc &= 3u;
        for (; ri < nr && !err && !pushed; ri++, slot0 = 0, k0 = 0) {
          uint32_t de = dirp[ri];
          uint32_t L = de_leaf(de), n = de_n(de);
          uint32_t id = NONE, st = ST_EVER, ln = 1;
          if ((uint32_t)lane < n) {
            const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + L) * (d.span ? SP_REC : 256u);
            id = rec[lane];
            if (d.span) { ln = rec[64 + lane]; st = rec[256 + lane]; } else st = rec[192 + lane];
          }
          bool vis = id != NONE && !(st & vis_mask) && (uint32_t)lane >= slot0;
          uint64_t vm = lmw::ballot(vis);
          while (vm && !err && !pushed) {
            int l0 = lmw::ffs64(vm);
            vm &= vm - 1;
            uint32_t eid0 = lmw::bcast(id, l0), elen = lmw::bcast(ln, l0);
            uint64_t g = elem0 + s_eb[pid_peer(eid0)] + pid_ctr(eid0);
            for (uint32_t k = ((uint32_t)l0 == slot0 ? k0 : 0u); k < elen && !err; k++) {
              uint64_t vabs, vlim = doc_end;
              uint32_t vblk = NONE, eid = eid0 + k;   // eid: the id a child container created by this value carries
              if (kind == CK_MOVABLE) {
                // the item shows iff its element points at it; the value is the element's (last set, else the inserted one)
                uint32_t pid_e = d.loc[g + k];
                bool show = pid_e == eid;
                uint32_t sl = ml_ht_find(keys, ht_capd, ml_key(pid_e, false));
                if (sl != NONE && best[sl] != 0) {
                  unsigned long long w = best[sl] - 1;
                  show = pid_make((uint32_t)(w >> 24) & 0xffu, d.op[m.op0 + (uint32_t)(w & 0xffffffu)].ctr) == eid;
                }
                if (!show) continue;
                sl = ml_ht_find(keys, ht_capd, ml_key(pid_e, true));
                if (sl != NONE && best[sl] != 0) {
                  uint32_t row = m.op0 + (uint32_t)((best[sl] - 1) & 0xffffffu);
                  const OpRow wr = d.op[row];
                  vblk = d.op_blk[row];
                  const BlockDesc& gb = d.blk[vblk];
                  vabs = d.op_val[row];
                  vlim = gb.base + gb.sec_rel[SEC_VALUES] + gb.sec_len[SEC_VALUES];
                  eid = pid_make(d.chg[wr.chg].peer, wr.ctr);
                } else {
                  if (pid_peer(pid_e) >= m.n_peers || (uint64_t)s_eb[pid_peer(pid_e)] + pid_ctr(pid_e) >= m.atoms) { err = ST_INTERNAL; break; }
                  vabs = doc_data0 + d.cp[elem0 + s_eb[pid_peer(pid_e)] + pid_ctr(pid_e)];
                  eid = pid_e;
                }
              } else vabs = doc_data0 + d.cp[g + k];
              if (!(fc & 1)) sink_byte(s, ',');
              fc &= ~1u;
              if (vabs > vlim) { err = ST_INTERNAL; break; }
              Rd r = rd_make(d.data + vabs, vlim - vabs);
              if (r.p < r.end && *r.p == 9) {   // child container created by this list element (or by the set that wrote it)
                (void)rd_u8(r);
                uint32_t ckind = rd_u8(r);
                uint32_t child = find_child(pid_peer(eid), pid_ctr(eid), ckind);
                if (child == NONE) { empty_child(ckind); continue; }
                if (sp >= (int)EMIT_MAX_DEPTH) { err = ST_UNSUPPORTED; break; }
                frame_set(sp - 1, cidx, ri, (uint32_t)l0, fc | ((k + 1) << 2));
                frame_set(sp, child, 0, 0, 0x3);
                sp++;
                pushed = true;
                break;
              }
              EMIT_PH(2);
              sink_value(s, r, err, d, vblk, m.blk0, m.n_blk);   // NONE: the item's block is found only if its key table is needed
              EMIT_PH(7);
            }
          }
        }
        if (!pushed && !err) { sink_byte(s, ']'); sp--; }
        EMIT_PH(2);
      } else if (!TEXT_ONLY && kind == CK_MAP) {
        if (fc & 2) {
          // first visit: this container's winning SET entries out of the document's claimed slots, bitonic-sorted by key
          uint32_t n_claimed = ht_capd ? d.ht_cnt[doc] : 0;
          uint32_t* sorted = scratch + scratch_top;
          uint32_t K = 0;
          for (uint32_t c0 = 0; c0 < n_claimed; c0 += 64) {
            uint32_t i = c0 + (uint32_t)lane;
            bool live = false;
            uint32_t sl = 0;
            if (i < n_claimed) {
              sl = claimed[i];
              unsigned long long k = keys[sl], bv = best[sl];
              if ((uint32_t)(k >> 32) == cidx && bv != 0) {
                uint32_t row = m.op0 + (uint32_t)((bv - 1) & 0xffffffu);
                live = ((d.op[row].cidx_kind >> 16) & 0xff) == OK_MAP_SET;
              }
            }
            uint64_t lm_ = lmw::ballot(live);
            if (live) {
              sorted[K + (uint32_t)lmw::popc64(lm_ & ((1ull << lane) - 1))] = sl;
              // the key's first eight bytes, big endian and zero padded, next to its slot: the sort compares these words and
              // only reads the strings on a tie (a 1,024-key map took 55 passes of four dependent loads per comparison)
              uint32_t kr = (uint32_t)keys[sl];
              const uint8_t* kp = d.data + d.key_off[kr];
              uint32_t kl = d.key_len[kr];
              uint64_t pf = 0;
              for (uint32_t q = 0; q < 8; q++) pf = (pf << 8) | (q < kl ? kp[q] : 0u);
              pfx[sl] = pf;
            }
            K += (uint32_t)lmw::popc64(lm_);
          }
          uint32_t Kp = 1;
          while (Kp < K) Kp <<= 1;
          if (scratch_top + Kp > ht_capd + ht_capd / 2) { err = ST_INTERNAL; break; }
          for (uint32_t i = K + (uint32_t)lane; i < Kp; i += 64) sorted[i] = NONE;   // padding sorts last
          lmw::block_sync();
          auto key_less = [&](uint32_t sa, uint32_t sb) -> bool {   // NONE is the largest
            if (sa == NONE) return false;
            if (sb == NONE) return true;
            unsigned long long pa = pfx[sa], pb = pfx[sb];
            if (pa != pb) return pa < pb;
            uint32_t ra = (uint32_t)keys[sa], rb = (uint32_t)keys[sb];
            return bytes_cmp(d.data + d.key_off[ra], d.key_len[ra], d.data + d.key_off[rb], d.key_len[rb]) < 0;
          };
          for (uint32_t k2 = 2; k2 <= Kp; k2 <<= 1) {
            for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
              for (uint32_t t0 = 0; t0 < Kp / 2; t0 += 64) {
                uint32_t t = t0 + (uint32_t)lane;
                if (t < Kp / 2) {
                  uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                  uint32_t q = i | j;
                  uint32_t x = sorted[i], y = sorted[q];
                  bool up = (i & k2) == 0;
                  bool swap = up ? key_less(y, x) : key_less(x, y);
                  if (swap) { sorted[i] = y; sorted[q] = x; }
                }
              }
              lmw::block_sync();
            }
          }
          sink_byte(s, '{');
          fa = 0; fb = scratch_top; fc = K << 2;
          scratch_top += Kp;
          EMIT_PH(3);
        }
        uint32_t K = fc >> 2;
        const uint32_t* sorted = scratch + fb;
        bool pushed = false;
        uint32_t e = fa;
        // The chain sorted list → claimed slot → winning op row → key / value location is eight dependent loads per entry;
        // it is walked for 64 entries at once, one entry per lane, and the entries are then rendered from registers (a
        // 1,024-key map cost 12 µs per entry when every entry walked the chain on its own)
        uint32_t g_base = NONE, g_row = 0, g_blk = 0, g_klen = 0, g_koff_lo = 0, g_koff_hi = 0, g_voff_lo = 0, g_voff_hi = 0, g_lim_lo = 0, g_lim_hi = 0;
        for (; e < K && !err; e++) {
          if (g_base == NONE || e >= g_base + 64) {
            g_base = e;
            uint32_t ge = e + (uint32_t)lane;
            if (ge < K) {
              uint32_t sl = sorted[ge];
              uint32_t krow = (uint32_t)keys[sl];
              g_row = m.op0 + (uint32_t)((best[sl] - 1) & 0xffffffu);
              uint64_t ko = d.key_off[krow];
              g_koff_lo = (uint32_t)ko; g_koff_hi = (uint32_t)(ko >> 32); g_klen = d.key_len[krow];
              g_blk = d.op_blk[g_row];
              const BlockDesc& gb = d.blk[g_blk];
              uint64_t lim_o = gb.base + gb.sec_rel[SEC_VALUES] + gb.sec_len[SEC_VALUES], vo = d.op_val[g_row];
              g_lim_lo = (uint32_t)lim_o; g_lim_hi = (uint32_t)(lim_o >> 32); g_voff_lo = (uint32_t)vo; g_voff_hi = (uint32_t)(vo >> 32);
            }
          }
#ifndef LM_NO_EMIT_MAP_FAST
          if (e == g_base) {
            // 64 entries at once, one per lane, when every one of them is PLAIN: a key of at most 24 bytes that needs no escape and
            // an integer / bool / null value — what an LWW map of counters, flags and ids holds.  Each lane formats its own entry
            // and the wave writes them side by side (one scan of the lengths).  Entry by entry the wave spent ≈30 single-byte
            // stores of lane 0 and a handful of dependent loads on each of a configs[2] document's 1,024 entries.  Anything else in
            // the group — a long or escaped key, a string / float / nested value, a child container — and the group is rendered
            // entry by entry below.
            const uint32_t ge = e + (uint32_t)lane;
            const bool act = ge < K;
            bool plain = true;
            uint32_t vlen = 0, lit = 0, nd = 0;
            uint64_t dg0 = 0, dg1 = 0, dg2 = 0;   // decimal digits, least significant first, one per byte
            bool neg = false;
            const uint8_t* kp = d.data + (((uint64_t)g_koff_hi << 32) | g_koff_lo);
            if (act) {
              if (g_klen > 24) plain = false;
              else for (uint32_t i = 0; i < g_klen; i++) { uint32_t c = kp[i]; if (c < 0x20 || c == '"' || c == '\\') plain = false; }
              const uint8_t* vp = d.data + (((uint64_t)g_voff_hi << 32) | g_voff_lo);
              const uint8_t* vl = d.data + (((uint64_t)g_lim_hi << 32) | g_lim_lo);
              if (vp >= vl) plain = false;
              else {
                uint32_t tag = *vp;
                if (tag == 0) { lit = 1; vlen = 4; }
                else if (tag == 1) { lit = 2; vlen = 4; }
                else if (tag == 2) { lit = 3; vlen = 5; }
                else if (tag == 3) {
                  Rd vr = rd_make(vp + 1, (uint64_t)(vl - (vp + 1)));
                  int64_t v = rd_sleb(vr);
                  if (vr.bad) plain = false;
                  neg = v < 0;
                  uint64_t u = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
                  uint64_t q1 = u / 1000000000ull, q2 = q1 / 1000000000ull;
                  uint32_t c0 = (uint32_t)(u - q1 * 1000000000ull), c1 = (uint32_t)(q1 - q2 * 1000000000ull), c2 = (uint32_t)q2;
                  const int top = c2 ? 2 : (c1 ? 1 : 0);
                  for (int ci = 0; ci <= top; ci++) {
                    uint32_t c = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
                    for (int i = 0; i < 9 && (ci < top || c || i == 0); i++) {
                      uint64_t dgt = '0' + c % 10u;
                      c /= 10u;
                      if (nd < 8) dg0 |= dgt << (8 * nd); else if (nd < 16) dg1 |= dgt << (8 * (nd - 8)); else dg2 |= dgt << (8 * (nd - 16));
                      nd++;
                    }
                  }
                  vlen = nd + (neg ? 1u : 0u);
                } else plain = false;
              }
            }
            if (!lmw::any(act && !plain)) {
              uint32_t len = act ? (ge ? 1u : 0u) + 2u + g_klen + 1u + vlen : 0u;
              uint32_t inc = lmw::scan_incl_add(len);
              uint32_t tot = lmw::bcast(inc, 63);
              if (s.out && s.pos + tot <= s.cap && act) {
                uint8_t* o = s.out + s.pos + (inc - len);
                if (ge) *o++ = ',';
                *o++ = '"';
                for (uint32_t i = 0; i < g_klen; i++) *o++ = kp[i];
                *o++ = '"'; *o++ = ':';
                if (lit == 1) { o[0] = 'n'; o[1] = 'u'; o[2] = 'l'; o[3] = 'l'; }
                else if (lit == 2) { o[0] = 't'; o[1] = 'r'; o[2] = 'u'; o[3] = 'e'; }
                else if (lit == 3) { o[0] = 'f'; o[1] = 'a'; o[2] = 'l'; o[3] = 's'; o[4] = 'e'; }
                else {
                  if (neg) *o++ = '-';
                  for (uint32_t j = nd; j-- > 0;) *o++ = (uint8_t)(j < 8 ? dg0 >> (8 * j) : (j < 16 ? dg1 >> (8 * (j - 8)) : dg2 >> (8 * (j - 16))));
                }
              }
              s.pos += tot;
              uint32_t n_grp = K - e < 64 ? K - e : 64u;
              e += n_grp - 1;       // (the loop's own increment takes the last step)
              g_base = NONE;
              continue;
            }
          }
#endif
          int gj = (int)(e - g_base);
          uint32_t row = lmw::bcast(g_row, gj), vblk = lmw::bcast(g_blk, gj);
          uint64_t koff = ((uint64_t)lmw::bcast(g_koff_hi, gj) << 32) | lmw::bcast(g_koff_lo, gj);
          uint64_t voff = ((uint64_t)lmw::bcast(g_voff_hi, gj) << 32) | lmw::bcast(g_voff_lo, gj);
          uint64_t limo = ((uint64_t)lmw::bcast(g_lim_hi, gj) << 32) | lmw::bcast(g_lim_lo, gj);
          if (e) sink_byte(s, ',');
          sink_string(s, d.data + koff, lmw::bcast(g_klen, gj));
          sink_byte(s, ':');
          const uint8_t* lim = d.data + limo;
          const uint8_t* p = d.data + voff;
          Rd r = rd_make(p, (uint64_t)(lim - p));
          if (r.p < r.end && *r.p == 9) {   // child container created by the winning set op
            (void)rd_u8(r);
            uint32_t ckind = rd_u8(r);
            const OpRow wr = d.op[row];
            uint32_t child = find_child(d.chg[wr.chg].peer, wr.ctr, ckind);
            if (child == NONE) { empty_child(ckind); continue; }
            if (sp >= (int)EMIT_MAX_DEPTH) { err = ST_UNSUPPORTED; break; }
            frame_set(sp - 1, cidx, e + 1, fb, K << 2);
            frame_set(sp, child, 0, 0, 0x3);
            sp++;
            pushed = true;
            break;
          }
          sink_value(s, r, err, d, vblk, m.blk0, m.n_blk);
        }
        if (!pushed && !err) {
          sink_byte(s, '}');
          scratch_top = fb;   // lists of deeper maps were released when those frames closed
          sp--;
        }
        EMIT_PH(4);
      } else {
        sink_lit(s, "null", 4);
        sp--;
      }
    }
  }
  sink_byte(s, '}');
  // ---- VersionVector: postcard map, entries sorted by peer, only peers with applied ops
  uint32_t vvn = 0;
  {
    uint32_t P = m.n_peers;
    uint8_t* vo = mode ? d.vv_out + d.vv_off[doc] : nullptr;
    const uint64_t vcap = d.vv_off[doc + 1] - d.vv_off[doc];   // 16 bytes per peer + 16: a true bound (10 + 5 bytes per entry)
    auto put_uleb = [&](uint64_t v) {
      do { uint8_t b = v & 0x7f; v >>= 7; if (v) b |= 0x80; if (vo && lane == 0 && vvn < vcap) vo[vvn] = b; vvn++; } while (v);
    };
    if (vvo_has(d, doc)) {
      // a document staged on a snapshot's STATE (lm_snapshot.h, lm_snapshot_base.h): its oplog version is the snapshot's version merged
      // with what the staged updates added (k_dag_a started every peer at its base end) — without the synthetic peer that wrote the state
      const uint64_t synth = vvo_synth_peer(d, doc);
      const uint8_t* bp = d.vvo + d.vvo_off[doc] + 8;
      const uint64_t bl = d.vvo_off[doc + 1] - d.vvo_off[doc] - 8;
      for (int pass2 = 0; pass2 < 2; pass2++) {
        Rd r = rd_make(bp, bl);
        uint64_t nb = rd_uleb(r), j = 0, bq = 0, be = 0;
        bool hb = false;
        auto next_b = [&]() { hb = false; while (j < nb && !r.bad) { j++; bq = rd_uleb(r); be = rd_uleb(r) >> 1; if (be) { hb = true; break; } } };
        next_b();
        uint32_t i = 0, cnt = 0;
        for (;;) {
          while (i < P && (d.peer_end[m.praw0 + i] == 0 || d.peer_uniq[m.praw0 + i] == synth)) i++;
          if (i >= P && !hb) break;
          uint64_t q, e;
          const uint64_t dq = i < P ? d.peer_uniq[m.praw0 + i] : 0;
          if (i < P && (!hb || dq <= bq)) { q = dq; e = d.peer_end[m.praw0 + i]; if (hb && bq == dq) { if (be > e) e = be; next_b(); } i++; }
          else { q = bq; e = be; next_b(); }
          cnt++;
          if (pass2) { put_uleb(q); put_uleb(e << 1); }
        }
        if (!pass2) put_uleb(cnt);
      }
    } else {
    uint32_t cntp = 0;
    for (uint32_t p = 0; p < P; p++) if (d.peer_end[m.praw0 + p] > 0) cntp++;
    put_uleb(cntp);
    for (uint32_t p = 0; p < P; p++) {
      uint32_t e = d.peer_end[m.praw0 + p];
      if (e == 0) continue;
      put_uleb(d.peer_uniq[m.praw0 + p]);
      put_uleb((uint64_t)e << 1);  // zigzag of a non-negative i32
    }
    }
  }
#ifdef LM_PROF_EMIT
  EMIT_PH(5);
  if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&d.prof[8 + i], (unsigned long long)epacc[i]);
#endif
  if (s.pos > 0xfffffff0ull) err = ST_UNSUPPORTED;   // a document's JSON is addressed with 32 bits
  if (lane == 0) {
    if (err) { d.doc[doc].status = err; d.doc[doc].out_len = 0; d.doc[doc].vv_len = 0; d.doc[doc].flags = m.flags & ~DF_REEMIT; }
    else {
      d.doc[doc].out_len = (uint32_t)s.pos; d.doc[doc].vv_len = vvn;
      uint32_t f = (m.flags & ~DF_REEMIT) | (soft ? DF_SOFT_UNSUPPORTED : 0u) | ((mode && s.pos > s.cap) ? DF_REEMIT : 0u);
      if (f != m.flags) d.doc[doc].flags = f;
    }
  }
}

LM_KERNEL void k_emit_text(Dev d, int mode, int pass) { emit_doc<true>(d, mode, pass); }
LM_KERNEL void k_emit_any(Dev d, int mode, int pass) { emit_doc<false>(d, mode, pass); }

// K12: one wave per doc — copy the rendered JSON / VV from the worst-case slabs into the compact result buffers.
// All four offset tables are 16-byte aligned, so the copy runs on 16-byte vectors.
LM_KERNEL void k_compact(Dev d, const uint64_t* src_addr, const uint64_t* vv_slab_off, const uint8_t* vv_slab) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  const DocMeta& m = d.doc[doc];
  if (status_fatal(m.status)) return;
  struct V16 { uint32_t x, y, z, w; };
  {
    const V16* src = (const V16*)(uintptr_t)src_addr[doc];
    V16* dst = (V16*)(d.out + d.out_off[doc]);
    uint32_t n = (m.out_len + 15) / 16;
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
  }
  {
    const V16* src = (const V16*)(vv_slab + vv_slab_off[doc]);
    V16* dst = (V16*)(d.vv_out + d.vv_off[doc]);
    uint32_t n = (m.vv_len + 15) / 16;
    for (uint32_t i = (uint32_t)lane; i < n; i += 64) dst[i] = src[i];
  }
}

// K12: xxh64 (seed 0) of every document's rendered JSON — the content word of the merged-state summary a sharded deployment
// all-gathers, so ranks can compare states without moving the JSON.
// SIXTEEN documents per wave: the 32-byte stripe loop is a serial chain of 64-bit multiplies per accumulator, and a multiply
// occupies the SIMD for the whole wave whatever its exec mask — with one document per wave (rounds 1-5: accumulators in lanes
// 0..3, sixty lanes idle) the kernel was VALU-issue bound at 1/16 of the lanes (0.36 ms per 5,000 configs[1] documents).
// Lanes 4g..4g+3 hold the four accumulators of document 16 * block + g (each reads its own 8 bytes of a stripe); the tail and
// the avalanche are computed by all four lanes of the group, lane 4g stores.  Launch: ceil(n_docs / HASH_DOCS) waves.
static constexpr uint32_t HASH_DOCS = 16;
LM_DEV uint64_t xx_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
LM_KERNEL void k_hash_json(Dev d, uint64_t* out_hash) {
  static constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                            P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
  int lane = lmw::lane();
  const int g4 = lane & ~3, j = lane & 3;
  uint32_t doc = (uint32_t)lmw::bid() * HASH_DOCS + (uint32_t)(lane >> 2);
  const bool in = doc < d.n_docs;
  const bool live = in && !status_fatal(d.doc[in ? doc : 0].status);
  const uint8_t* p = live ? d.out + d.out_off[doc] : d.out;   // 16-byte aligned
  const uint64_t len = live ? d.doc[doc].out_len : 0;
  const uint64_t n_str = len / 32;
  uint64_t v = j == 0 ? P1 + P2 : j == 1 ? P2 : j == 2 ? 0ull : 0ull - P1;
  {
    const uint64_t* q = (const uint64_t*)p + j;
#pragma unroll 8
    for (uint64_t s = 0; s < n_str; s++) v = xx_rotl(v + q[s * 4] * P2, 31) * P1;
  }
  uint64_t v1 = lmw::shfl64(v, g4), v2 = lmw::shfl64(v, g4 + 1), v3 = lmw::shfl64(v, g4 + 2), v4 = lmw::shfl64(v, g4 + 3);
  uint64_t h;
  if (n_str) {
    h = xx_rotl(v1, 1) + xx_rotl(v2, 7) + xx_rotl(v3, 12) + xx_rotl(v4, 18);
    h = (h ^ (xx_rotl(v1 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xx_rotl(v2 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xx_rotl(v3 * P2, 31) * P1)) * P1 + P4;
    h = (h ^ (xx_rotl(v4 * P2, 31) * P1)) * P1 + P4;
  } else h = P5;
  h += len;
  uint64_t i = n_str * 32;
  for (; i + 8 <= len; i += 8) { uint64_t k = *(const uint64_t*)(p + i); h ^= xx_rotl(k * P2, 31) * P1; h = xx_rotl(h, 27) * P1 + P4; }
  if (i + 4 <= len) { h ^= (uint64_t)(*(const uint32_t*)(p + i)) * P1; h = xx_rotl(h, 23) * P2 + P3; i += 4; }
  for (; i < len; i++) { h ^= (uint64_t)p[i] * P5; h = xx_rotl(h, 11) * P1; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  if (in && j == 0) out_hash[doc] = live ? h : 0;
}

// The per-document row of the merged-state summary a sharded deployment exchanges (layout = loro_amd/dist.py
// SUMMARY_WORDS): document id, status, pending ops, JSON length, VV length, xxh64 of the JSON — written on the device straight
// into the buffer the all-gather sends (lm_summary_layout): no host copy of the table, no D2H → pack → H2D in front of the
// collective.  Statuses as lm_result_meta reports them.  One lane per document.
LM_KERNEL void k_summary_rows(Dev d, const uint64_t* hashes, long long* rows, long long id0, long long stride) {
  uint32_t i = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  if (i >= d.n_docs) return;
  const DocMeta m = d.doc[i];
  bool ok = m.status == ST_OK;
  int32_t st = m.status;
  if (ok && (m.flags & DF_SOFT_UNSUPPORTED)) st = ST_UNSUPPORTED;               // rendered, with the out-of-scope containers as null
  if (ok && (m.flags & DF_FRONT_ERR)) { st = m.front_err; ok = false; }          // resident: the import went through, the checkout was refused
  long long* w = rows + (uint64_t)i * 6;
  w[0] = id0 + (long long)i * stride;
  w[1] = st;
  w[2] = ok ? (long long)(((uint64_t)m.pending_hi << 32) | m.pending_lo) : 0;
  w[3] = ok ? (long long)m.out_len : 0;
  w[4] = ok ? (long long)m.vv_len : 0;
  w[5] = ok ? (long long)hashes[i] : 0;
}

}  // namespace lm
