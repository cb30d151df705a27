// Richtext values (second half): TextHandler::get_richtext_value (handler.rs:1502 → container/richtext/
// richtext_state.rs:2500-2584) of every Text container of a document, rendered on the device from the trackers the integrate
// stage left behind — a kernel of its own behind lm_run (lm_richtext), because the JSON renderer is at its register limit and
// get_deep_value never shows styles.
//
// What the reference keeps as a StyleRangeMap next to the rope (container/richtext/style_range_map.rs) is a function of the
// anchors in the sequence: the End anchor's insertion annotates start..=end with the StyleOp (state/richtext_state.rs:730-812),
// an element inserted inside a range inherits it and at a boundary the intersection of both sides (style_range_map.rs insert()),
// a deleted anchor takes its range along (richtext_state.rs:2275-2300).  So a scalar carries the StyleOps whose Start anchor
// stands in front of it and whose End anchor stands behind it, both visible at the rendered version; per key the op with the
// greatest (lamport, peer) decides (StyleValue::get = BTreeSet::last under StyleOp::cmp, container/richtext.rs:120-126), a null
// value removes the key (StyleMeta::to_value, delta/text.rs:125-140) and neighbouring spans with equal attributes are one span
// (richtext_state.rs:2546-2584).  A span is the canonical JSON of the LoroValue map it is: {"attributes":{…},"insert":"…"},
// keys bytewise sorted, no attributes entry when the map is empty.
//
// One wave per document, two walks over each Text container's leaves in sequence order:
//   A  every visible anchor gets CP_ALIVE in cp[] (k_elem_fill left CP_ANCHOR | op row there) — a Start anchor counts iff the slot
//      behind it (same peer, counter + 1: its End, every writer emits the pair) carries that mark;
//   B  64 visible elements per step: the scalars between two anchors are escaped and stored by all lanes at once, an anchor
//      opens / closes its StyleOp in the active set (LDS) and takes its mark off again; before scalars are written after a change
//      of the set, the winners per key are worked out and compared BY VALUE (key bytes, encoded value bytes) with the open span's.
// Output per document: {"<container id>":[span,…],…} for the Text containers in which something — a scalar or an anchor — is visible
// at the rendered version; the kernel writes the members in the order of the document's container table, the host (Engine::richtext)
// puts the members of a document that lists several into the bytewise order of their JSON-encoded keys (ContainerID Display: cid:root-<name>:Text / cid:<counter>@<peer>:Text).
// Limits: at most RT_MAX StyleOps open at one scalar and RT_MAX distinct style keys per Text (LM_UNSUPPORTED beyond); two values are "equal" when their encodings are
// (map-typed style values with the same entries in another order split a span the reference would merge).
#pragma once
#include "lm_k_emit.h"

namespace lm {

static constexpr uint32_t RT_MAX = 64;

struct RtStyle { const uint8_t* kp; const uint8_t* vp; uint32_t kl, vl, lam, peer, blk; bool null; };

// the StyleOp of a StyleStart row (row inside the document): key, encoded value, (lamport, peer) — value.rs:936-955 MarkStart =
// info u8, len, key idx, value
LM_DEV RtStyle rt_style(const Dev& d, const DocMeta& m, uint32_t srow, int32_t& err) {
  RtStyle s;
  const uint32_t row = m.op0 + srow;
  const OpRow r = d.op[row];
  const ChangeRow c = d.chg[r.chg];
  s.lam = d.chg_lamport[r.chg] + (r.ctr - c.ctr);
  s.peer = c.peer;
  s.blk = d.op_blk[row];
  const BlockDesc& bd = d.blk[s.blk];
  const uint8_t* lim = d.data + bd.base + bd.sec_rel[SEC_VALUES] + bd.sec_len[SEC_VALUES];
  const uint8_t* p = d.data + d.op_val[row];
  Rd v = rd_make(p, p < lim ? (uint64_t)(lim - p) : 0ull);
  (void)rd_u8(v);
  (void)rd_uleb(v);
  uint64_t kidx = rd_uleb(v);
  const uint32_t nk = d.bcnt[(uint64_t)s.blk * BCN + BC_KEY];
  s.kp = d.data; s.kl = 0;
  if (v.bad || kidx >= nk) err = err ? err : ST_DATA_CORRUPTION;
  else { uint32_t krow = d.boff[(uint64_t)s.blk * BCN + BC_KEY] + (uint32_t)kidx; s.kp = d.data + d.key_off[krow]; s.kl = d.key_len[krow]; }
  s.vp = v.p;
  s.null = v.p < v.end && *v.p == 0;
  uint32_t vf = 0;
  skip_loro_value(v, vf, -1);
  if (v.bad) err = err ? err : ST_DATA_CORRUPTION;
  s.vl = (uint32_t)(v.p - s.vp);
  return s;
}

LM_DEV void sink_u64(Sink& s, uint64_t u) {
  uint32_t n = 1;
  for (uint64_t t = u; t >= 10; t /= 10) n++;
  const uint32_t lane = (uint32_t)lmw::lane();
  if (lane < n) {
    uint64_t p = 1;
    for (uint32_t i = 0; i + 1 + lane < n; i++) p *= 10;
    if (s.out && s.pos + n <= s.cap) s.out[s.pos + lane] = (uint8_t)('0' + (uint32_t)((u / p) % 10));
  }
  s.pos += n;
}

// visible elements of a sequence container in order, 64 per step: f(has, g) — g = element slot inside the document — is called by
// all lanes together; lane order is sequence order
template <class F>
LM_DEV void rt_walk(const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t vis_mask, const uint32_t* s_eb, uint32_t* s_inc, uint32_t* s_g0, F&& f) {
  const int lane = lmw::lane();
  const uint32_t r0 = d.cont_root0[m.cid0 + cidx], nr = d.cont_nroot[m.cid0 + cidx];
  const uint32_t* dirp = d.dir_out + m.leaf0 + r0;
  const bool span = d.span != 0;
  const uint32_t rec_words = span ? SP_REC : 256u, st_at = span ? 256u : 192u;
  // (one wave walks the document: the record of the next leaf and the directory entry behind it are requested before this leaf's
  // elements are handled — f() loads and stores global memory, the compiler cannot move these loads across it)
  uint32_t de1 = nr > 0 ? dirp[0] : 0u, de2 = nr > 1 ? dirp[1] : 0u;
  uint32_t p_id = NONE, p_ln = 1, p_st = ST_EVER;
  if (nr > 0 && (uint32_t)lane < de_n(de1)) {
    const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + de_leaf(de1)) * rec_words;
    p_id = rec[lane]; p_st = rec[st_at + lane]; if (span) p_ln = rec[64 + lane];
  }
  for (uint32_t ri = 0; ri < nr; ri++) {
    const uint32_t id0 = p_id, ln = p_ln, st = p_st;
    de1 = de2;
    de2 = ri + 2 < nr ? dirp[ri + 2] : 0u;
    p_id = NONE; p_ln = 1; p_st = ST_EVER;
    if (ri + 1 < nr && (uint32_t)lane < de_n(de1)) {
      const uint32_t* rec = d.it + (uint64_t)(m.leaf0 + de_leaf(de1)) * rec_words;
      p_id = rec[lane]; p_st = rec[st_at + lane]; if (span) p_ln = rec[64 + lane];
    }
    if (span) {
      const uint32_t vl = (id0 != NONE && !(st & vis_mask)) ? ln : 0u;
      const uint32_t inc = lmw::scan_incl_add(vl);
      const uint32_t total = lmw::bcast(inc, 63);
      lmw::block_sync();
      s_inc[lane] = inc;
      s_g0[lane] = vl ? s_eb[pid_peer(id0)] + pid_ctr(id0) - (inc - vl) : 0u;
      lmw::block_sync();
      for (uint32_t e0 = 0; e0 < total; e0 += 64) {
        const uint32_t e = e0 + (uint32_t)lane;
        uint32_t g = 0;
        if (e < total) {
          uint32_t lo = 0, hi = 63;
          while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (s_inc[mid] > e) hi = mid; else lo = mid + 1; }
          g = s_g0[lo] + e;
        }
        f(e < total, g);
      }
    } else {
      const bool vis = id0 != NONE && !(st & vis_mask);
      if (lmw::any(vis)) f(vis, vis ? s_eb[pid_peer(id0)] + pid_ctr(id0) : 0u);
    }
  }
}

// mode 1 (the only one the host uses): write into out + out_off[doc] (capacity out_off[doc + 1] - out_off[doc]; nothing is written beyond it) and report the exact size — Engine::richtext launches once into optimistic slabs and a second time, at exact sizes, only when a document overflowed; mode 0 (sizes only) is kept for experiments.
// rt_len[doc] = the exact size either way
LM_KERNEL void k_richtext(Dev d, uint8_t* out, const uint64_t* out_off, uint32_t* rt_len, int32_t* rt_status, uint32_t* rt_cnt, int mode) {
  const uint32_t doc = (uint32_t)lmw::bid();
  const int lane = lmw::lane();
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) { if (lane == 0) { rt_len[doc] = 0; rt_status[doc] = m.status; rt_cnt[doc] = 0; } return; }
  const uint32_t C = m.n_cont;
  const uint64_t elem0 = ((uint64_t)m.elem0_hi << 32) | m.elem0_lo;
  const uint32_t vis_mask = (d.res_vis && d.front_off[doc + 1] > d.front_off[doc] && !(m.flags & DF_FRONT_ERR)) ? (ST_FUT | ST_DELMASK) : ST_EVER;
  LM_SHARED(uint32_t, s_eb, MAX_PEERS);
  LM_SHARED(uint32_t, s_inc, 64);
  LM_SHARED(uint32_t, s_g0, 64);
  LM_SHARED(uint32_t, s_act, RT_MAX);    // StyleStart rows (inside the document) of the open StyleOps
  LM_SHARED(uint32_t, s_alam, RT_MAX);   // … their lamports, peer | null value << 31, and key numbers (keys are numbered per container as they
  LM_SHARED(uint32_t, s_apk, RT_MAX);    //   turn up — s_kp / s_kl — so that the winners per key are worked out in LDS alone: resolving a StyleOp
  LM_SHARED(uint32_t, s_akid, RT_MAX);   //   from its row is five dependent loads, measured ≈10 µs per anchor when it was redone for every comparison)
  LM_SHARED(uint32_t, s_wsl, RT_MAX);    // winners(): the deciding SLOT of s_act per key
  LM_SHARED(unsigned long long, s_kp, RT_MAX);
  LM_SHARED(uint32_t, s_kl, RT_MAX);
  LM_SHARED(uint32_t, s_win, RT_MAX);    // … the deciding op of every key with a value, now
  LM_SHARED(uint32_t, s_open, RT_MAX);   // … and when the span being written was opened
  for (uint32_t p = (uint32_t)lane; p < m.n_peers && p < MAX_PEERS; p += 64) s_eb[p] = d.elem_base[m.praw0 + p];
  lmw::block_sync();
  Sink s;
  s.out = mode ? out + out_off[doc] : nullptr;
  s.pos = 0;
  s.cap = mode ? out_off[doc + 1] - out_off[doc] : 0;
  int32_t err = 0;
  sink_byte(s, '{');
  bool first_cont = true;
  uint32_t n_listed = 0;   // members written (the host orders the members of a document that lists several, lm_pipeline.h)
  for (uint32_t cidx = 0; cidx < C && !err; cidx++) {
    const ContRow o = d.cont[m.cid0 + cidx];
    if ((o.kind_root & 0xff) != CK_TEXT) continue;
    // ---- walk A: which anchors are visible (a container in which nothing is visible at the rendered version is not listed)
    bool any_vis = false;
    rt_walk(d, m, cidx, vis_mask, s_eb, s_inc, s_g0, [&](bool has, uint32_t g) {
      any_vis = true;
      if (!has) return;
      if (d.span ? d.tb[elem0 + g] == TB_ANCHOR : d.cp[elem0 + g] >= CP_ANCHOR) d.cp[elem0 + g] |= CP_ALIVE;
    });
    if (!any_vis) continue;
    if (!first_cont) sink_byte(s, ',');
    first_cont = false;
    n_listed++;
    if (o.kind_root & 0x100) {
      sink_lit(s, "\"cid:root-", 10);
      sink_escaped(s, d.data + o.name_off, o.name_len);
    } else {
      sink_lit(s, "\"cid:", 5);
      sink_i64(s, (int64_t)(int32_t)o.counter);
      sink_byte(s, '@');
      sink_u64(s, d.peer_uniq[m.praw0 + o.peer]);
    }
    sink_lit(s, ":Text\":[", 8);
    lmw::mem_fence();
    lmw::block_sync();
    // ---- walk B
    uint32_t n_act = 0, n_win = 0, n_open = 0;
    bool dirty = false, span_open = false, first_span = true;
    auto lds_set = [&](uint32_t* a, uint32_t i, uint32_t v) { lmw::block_sync(); if (lane == 0) a[i] = v; lmw::block_sync(); };
    uint32_t n_keys = 0;
    auto winners = [&]() {   // the op that decides each key, keys without a value dropped
      n_win = 0;
      for (uint32_t i = 0; i < n_act; i++) {
        const uint32_t kid = s_akid[i], lam = s_alam[i], pk = s_apk[i] & 0x7fffffffu;
        uint32_t found = NONE;
        for (uint32_t j = 0; j < n_win && found == NONE; j++) {
          const uint32_t w = s_wsl[j];
          if (s_akid[w] == kid) { found = j; const uint32_t wl = s_alam[w], wp = s_apk[w] & 0x7fffffffu; if (lam > wl || (lam == wl && pk > wp)) lds_set(s_wsl, j, i); }
        }
        if (found == NONE) { lds_set(s_wsl, n_win, i); n_win++; }
      }
      uint32_t k = 0;
      for (uint32_t j = 0; j < n_win; j++) {
        const uint32_t w = s_wsl[j];
        if (!(s_apk[w] >> 31)) { lds_set(s_win, k, s_act[w]); k++; }
      }
      n_win = k;
    };
    auto same_as_open = [&]() -> bool {
      if (n_win != n_open) return false;
      for (uint32_t i = 0; i < n_win; i++) {
        const uint32_t w = s_win[i];
        bool hit = false;
        for (uint32_t j = 0; j < n_open && !hit; j++) hit = s_open[j] == w;
        if (hit) continue;
        const RtStyle A = rt_style(d, m, w, err);
        for (uint32_t j = 0; j < n_open && !hit; j++) {
          const RtStyle B = rt_style(d, m, s_open[j], err);
          hit = A.kl == B.kl && A.vl == B.vl && bytes_eq(A.kp, B.kp, A.kl) && bytes_eq(A.vp, B.vp, A.vl);
        }
        if (!hit) return false;
      }
      return true;
    };
    auto open_span = [&]() {
      if (!first_span) sink_byte(s, ',');
      first_span = false;
      sink_byte(s, '{');
      if (n_open) {
        sink_lit(s, "\"attributes\":{", 14);
        uint32_t last = NONE;   // keys in bytewise order: the smallest one greater than the one written last, n_open times
        for (uint32_t k = 0; k < n_open && !err; k++) {
          uint32_t best = NONE;
          RtStyle Lst = last != NONE ? rt_style(d, m, last, err) : RtStyle{d.data, d.data, 0, 0, 0, 0, 0, false};
          RtStyle Bst = Lst;
          for (uint32_t j = 0; j < n_open; j++) {
            const uint32_t w = s_open[j];
            const RtStyle W = rt_style(d, m, w, err);
            if (last != NONE && bytes_cmp(W.kp, W.kl, Lst.kp, Lst.kl) <= 0) continue;
            if (best == NONE || bytes_cmp(W.kp, W.kl, Bst.kp, Bst.kl) < 0) { best = w; Bst = W; }
          }
          if (best == NONE) break;
          if (k) sink_byte(s, ',');
          sink_string(s, Bst.kp, Bst.kl);
          sink_byte(s, ':');
          Rd vr = rd_make(Bst.vp, Bst.vl);
          sink_value(s, vr, err, d, Bst.blk, m.blk0, m.n_blk);
          last = best;
        }
        sink_lit(s, "},", 2);
      }
      sink_lit(s, "\"insert\":\"", 10);
    };
    rt_walk(d, m, cidx, vis_mask, s_eb, s_inc, s_g0, [&](bool has, uint32_t g) {
      uint32_t cpv = 0;
      bool anc = false;
      if (has) {
        if (d.span) {
          const uint32_t t = d.tb[elem0 + g];
          if (t == TB_ANCHOR) { anc = true; cpv = d.cp[elem0 + g]; }
          else cpv = t == TB_WIDE ? d.cp[elem0 + g] : t;
        } else { cpv = d.cp[elem0 + g]; anc = cpv >= CP_ANCHOR; }
      }
      uint64_t am = lmw::ballot(anc);
      const uint64_t hm = lmw::ballot(has && !anc);
      uint32_t lo = 0;
      for (;;) {
        const uint32_t a = am ? (uint32_t)lmw::ffs64(am) : 64u;
        const uint64_t below_a = a >= 64 ? ~0ull : (1ull << a) - 1, below_lo = lo >= 64 ? ~0ull : (1ull << lo) - 1;
        const uint64_t seg = hm & below_a & ~below_lo;
        if (seg && !err) {
          if (dirty) {
            winners();
            dirty = false;
            if (span_open && !same_as_open()) { sink_lit(s, "\"}", 2); span_open = false; }
          }
          if (!span_open) {
            lmw::block_sync();
            if ((uint32_t)lane < n_win) s_open[lane] = s_win[lane];
            lmw::block_sync();
            n_open = n_win;
            open_span();
            span_open = true;
          }
          uint64_t bytes = 0;
          uint32_t nb = 0;
          if ((seg >> lane) & 1) cp_bytes(cpv, bytes, nb);
          sink_lanes(s, bytes, nb);
        }
        if (a >= 64 || err) break;
        // the anchor in lane a
        const uint32_t av = lmw::bcast(cpv, (int)a), ag = lmw::bcast(g, (int)a);
        if ((uint32_t)lane == a) d.cp[elem0 + ag] = av & ~CP_ALIVE;
        const uint32_t rrel = av & ~(CP_ANCHOR | CP_ALIVE);
        const OpRow r = d.op[m.op0 + rrel];
        const uint32_t kind = (r.cidx_kind >> 16) & 0xff;
        if (kind == OK_STYLE_START) {
          const uint32_t peer = d.chg[r.chg].peer;
          const bool pair = r.ctr + 2 <= d.peer_ext[m.praw0 + peer] && d.cp[elem0 + ag + 1] == (CP_ANCHOR | CP_ALIVE | (rrel + 1));
          if (pair) {
            const RtStyle A = rt_style(d, m, rrel, err);
            uint32_t kid = NONE;
            for (uint32_t q = 0; q < n_keys && kid == NONE; q++)
              if (s_kl[q] == A.kl && (s_kp[q] == (unsigned long long)(uintptr_t)A.kp || bytes_eq((const uint8_t*)(uintptr_t)s_kp[q], A.kp, A.kl))) kid = q;
            if (kid == NONE && n_keys < RT_MAX) {
              lmw::block_sync();
              if (lane == 0) { s_kp[n_keys] = (unsigned long long)(uintptr_t)A.kp; s_kl[n_keys] = A.kl; }
              lmw::block_sync();
              kid = n_keys++;
            }
            if (n_act >= RT_MAX || kid == NONE) err = err ? err : ST_UNSUPPORTED;
            else {
              lmw::block_sync();
              if (lane == 0) { s_act[n_act] = rrel; s_alam[n_act] = A.lam; s_apk[n_act] = A.peer | (A.null ? 0x80000000u : 0u); s_akid[n_act] = kid; }
              lmw::block_sync();
              n_act++; dirty = true;
            }
          }
        } else if (kind == OK_STYLE_END && rrel > 0) {
          for (uint32_t i = 0; i < n_act; i++)
            if (s_act[i] == rrel - 1) {
              const uint32_t l = n_act - 1, v0 = s_act[l], v1 = s_alam[l], v2 = s_apk[l], v3 = s_akid[l];
              lmw::block_sync();
              if (lane == 0) { s_act[i] = v0; s_alam[i] = v1; s_apk[i] = v2; s_akid[i] = v3; }
              lmw::block_sync();
              n_act--; dirty = true;
              break;
            }
        }
        am &= am - 1;
        lo = a + 1;
      }
    });
    if (span_open) sink_lit(s, "\"}", 2);
    sink_byte(s, ']');
    if (err)   // walk B stopped early: take the remaining marks off
      rt_walk(d, m, cidx, vis_mask, s_eb, s_inc, s_g0, [&](bool has, uint32_t g) {
        if (has && (d.span ? d.tb[elem0 + g] == TB_ANCHOR : d.cp[elem0 + g] >= CP_ANCHOR)) d.cp[elem0 + g] &= ~CP_ALIVE;
      });
    lmw::mem_fence();
  }
  sink_byte(s, '}');
  if (lane == 0) {
    rt_len[doc] = err ? 0u : (uint32_t)s.pos;
    rt_cnt[doc] = n_listed;
    rt_status[doc] = err ? err : (int32_t)ST_OK;   // (s.pos > s.cap: the host sees the size and launches again with room for it)
  }
}

}  // namespace lm
