// Integrate stage, the LINEAR PREFIX of a batch replay (included by lm_k_integrate_span.h; PLAIN batch instantiations only).
//
// The reference does not run its CRDT at all while a history is one chain: DiffMode::Linear (diff_calc.rs:1391-1465) composes the
// ops' positions directly, and a tracker that is needed later starts at the common ancestors with everything before them collapsed
// into ONE span without origins or tombstones (Tracker::new_with_unknown, tracker.rs:40-62).  k_dag_b finds the same boundary for a
// batch: the first node of the replay order with the "a concurrent section begins here" flag (node_done bit 1) — the version in
// front of it is critical (every later node depends on all of it) and so is every version before it.  The nodes in front of that
// flag are replayed here as what they are, a positional rope:
//   * an insert lands at its position; no origin_right is looked for, no sibling is scanned (nothing can be future), nothing is
//     remembered by id (loc[] is not kept; ts_build_loc writes it when the chain ends);
//   * a delete REMOVES its targets from the leaves (the reference's unknown span has no tombstones either): every later op has
//     this version in its causal past, so nothing can name a deleted element again — by id or as an origin.  The ids at the
//     row's position are still compared with the row's target span (what ts_del_pos_ok does for the tracker);
//   * items keep what the tracker reads from NON-future items later on: id, length, origin_left of the head (element k > 0 has
//     origin_left id0 + k - 1).  origin_right is written as NONE — it is only ever read from future items (sibling scan) and
//     copied along when an item is cut.
// The leaves and the LDS directory are the span kernel's own (SP_REC records, word A / word B), so the replay simply goes on in
// integrate_span_body's row loop behind the prefix: its first ts_goto builds loc[] and makes this version the tracker's base.
// Cost per op row: ≈35 instructions for typing that continues an item, ≈80 for a new item / a delete inside the cached leaf —
// no LDS traffic (count and length of the cached leaf live in registers until it leaves the cache) — against ≈190 on the
// tracker's fast paths.
#pragma once

namespace lm {

#ifndef LM_LINEAR
#define LM_LINEAR 1     // 0: no linear prefix — every node goes through the tracker (rounds 1-4; A/B builds)
#endif

struct Tl {               // the cached leaf of the linear replay: lane i = item i; lanes >= n hold id NONE, len 0, ol NONE
  uint32_t n, tot;        // items / elements (n = 255: no leaf cached — then tot = 0, pre = 0, p = NONE)
  uint32_t id, len, ol;
  uint32_t leaf, p, pre;  // leaf record, directory position, elements in front of it
  bool dirty;
};
LM_DEV void tl_none(Tl& c) { c.n = 255; c.tot = 0; c.id = NONE; c.len = 0; c.ol = NONE; c.leaf = NONE; c.p = NONE; c.pre = 0; c.dirty = false; }
// write-back: the leaf record's id / length / origin_left words and both directory words (status 0 and origin_right NONE are the same
// for every item of the prefix: tl_finish writes them once per leaf — 40 % of the prefix's stores otherwise)
LM_DEV void tl_store(Ts& t, Tl& c) {
  if (c.n == 255 || !c.dirty) return;
  int lane = lmw::lane();
  uint32_t* rec = t.it + (uint64_t)c.leaf * SP_REC;
  if ((uint32_t)lane < c.n) { rec[lane] = c.id; rec[64 + lane] = c.len; rec[128 + lane] = c.ol; }
  lmw::wave_sync();
  if (lane == 0) { t.da[c.p] = sa_make(c.leaf, c.n, c.n != 0); t.db[c.p] = c.tot; }
  lmw::wave_sync();
  c.dirty = false;
}
LM_DEV void tl_load(Ts& t, Tl& c, uint32_t p, uint32_t pre) {
  int lane = lmw::lane();
  lmw::wave_sync();
  uint32_t a = lmw::first(t.da[p]);
  c.leaf = sa_leaf(a); c.n = sa_n(a); c.tot = lmw::first(t.db[p]); c.p = p; c.pre = pre; c.dirty = false;
  const uint32_t* rec = t.it + (uint64_t)c.leaf * SP_REC;
  bool in = (uint32_t)lane < c.n;
  c.id = in ? rec[lane] : NONE; c.len = in ? rec[64 + lane] : 0u; c.ol = in ? rec[128 + lane] : NONE;
}
// the leaf that holds the pos-th element (1 <= pos <= tot_active) becomes the cached leaf
LM_DEV void tl_seek(Ts& t, Tl& c, uint32_t pos) {
  int lane = lmw::lane();
  tl_store(t, c);
  lmw::wave_sync();
  uint32_t acc = 0;
  for (uint32_t i0 = 0; i0 < t.n_dir; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    uint32_t b = i < t.n_dir ? t.db[i] : 0u;
    uint32_t inc = lmw::scan_incl_add(b);
    uint64_t m = lmw::ballot(acc + inc >= pos);
    if (m) { int s = lmw::ffs64(m); tl_load(t, c, i0 + (uint32_t)s, acc + lmw::bcast(inc, s) - lmw::bcast(b, s)); return; }
    acc += lmw::bcast(inc, 63);
  }
  LM_SETERR(t.err, ST_INTERNAL);
  tl_none(c);
}
LM_DEV void tl_dir_insert_after(Ts& t, uint32_t p, uint32_t a, uint32_t b) {
  int lane = lmw::lane();
  if (t.n_dir >= t.dir_cap) { t.err = ST_RETRY; return; }
  lmw::wave_sync();
  uint32_t q = p + 1;
  for (uint32_t hi = t.n_dir; hi > q;) {
    uint32_t c0 = hi > q + 64 ? hi - 64 : q;
    uint32_t i = c0 + (uint32_t)lane;
    bool in = i < hi;
    uint32_t va = in ? t.da[i] : 0u, vb = in ? t.db[i] : 0u;
    lmw::wave_sync();
    if (in) { t.da[i + 1] = va; t.db[i + 1] = vb; }
    lmw::wave_sync();
    hi = c0;
  }
  if (lane == 0) { t.da[q] = a; t.db[q] = b; }
  t.n_dir++;
  lmw::wave_sync();
}
// the cached leaf is full: its upper items move to a new leaf; the half that holds position `pos` stays cached
LM_DEV void tl_split(Ts& t, Tl& c, uint32_t pos) {
  int lane = lmw::lane();
  if (t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
  uint32_t NL = t.n_leaf++;
  uint32_t nu = c.n - 32;
  uint32_t inc = lmw::scan_incl_add(c.len);
  uint32_t tot_lo = lmw::bcast(inc, 31);
  uint32_t uid = lmw::shfl(c.id, (lane + 32) & 63), uln = lmw::shfl(c.len, (lane + 32) & 63), uol = lmw::shfl(c.ol, (lane + 32) & 63);
  if ((uint32_t)lane >= nu) { uid = NONE; uln = 0; uol = NONE; }
  bool keep_upper = pos > c.pre + tot_lo;
  if (keep_upper) {
    uint32_t* rec = t.it + (uint64_t)c.leaf * SP_REC;
    if (lane < 32) { rec[lane] = c.id; rec[64 + lane] = c.len; rec[128 + lane] = c.ol; }
    lmw::wave_sync();
    if (lane == 0) { t.da[c.p] = sa_make(c.leaf, 32, true); t.db[c.p] = tot_lo; }
    tl_dir_insert_after(t, c.p, sa_make(NL, nu, true), c.tot - tot_lo);
    c.leaf = NL; c.p = c.p + 1; c.pre += tot_lo; c.n = nu; c.tot -= tot_lo;
    c.id = uid; c.len = uln; c.ol = uol;
  } else {
    uint32_t* rec = t.it + (uint64_t)NL * SP_REC;
    if ((uint32_t)lane < nu) { rec[lane] = uid; rec[64 + lane] = uln; rec[128 + lane] = uol; }
    tl_dir_insert_after(t, c.p, sa_make(NL, nu, true), c.tot - tot_lo);
    c.n = 32; c.tot = tot_lo;
    if (lane >= 32) { c.id = NONE; c.len = 0; c.ol = NONE; }
  }
  c.dirty = true;
  if (t.err) tl_none(c);
}
// one item (A) drops in at lane idx; the items from idx on move up by one (the leaf has room)
LM_DEV void tl_shift_in(Tl& c, uint32_t idx, uint32_t a_id, uint32_t a_len, uint32_t a_ol) {
  uint32_t lane = (uint32_t)lmw::lane();
  uint32_t pid = lmw::shift_up0(c.id, 1), pln = lmw::shift_up0(c.len, 1), pol = lmw::shift_up0(c.ol, 1);
  bool sh = lane > idx, isA = lane == idx;
  c.id = isA ? a_id : (sh ? pid : c.id);
  c.len = isA ? a_len : (sh ? pln : c.len);
  c.ol = isA ? a_ol : (sh ? pol : c.ol);
  c.n++;
}
// the leaf that holds the cursor behind the pos-th element (pos = 0: the first leaf) becomes the cached leaf, with room for `room`
// more items — the rare part of an edit, kept out of its straight-line part (a loop around the edit that retried after a seek or a
// split carried the cached leaf through its phi nodes: 26-72 register moves per op row)
LM_DEV void tl_prepare(Ts& t, Tl& c, uint32_t pos, uint32_t room) {
  if (pos == 0) { if (c.p != 0) { tl_store(t, c); tl_load(t, c, 0, 0); } }
  else if (pos <= c.pre || pos - c.pre > c.tot) tl_seek(t, c, pos);   // (no leaf cached: pre = 0, tot = 0)
  if (!t.err && c.n + room > 64) tl_split(t, c, pos);
}
// insert of run [pid0, pid0 + len) behind the pos-th element
LM_DEV void tl_insert(Ts& t, Tl& c, uint32_t pos, uint32_t pid0, uint32_t len) {
  uint32_t lane = (uint32_t)lmw::lane();
  t.beyond |= pos > t.tot_active ? 1u : 0u;   // (see ts_insert: the document is closed with LM_DATA_CORRUPTION)
  pos = pos > t.tot_active ? t.tot_active : pos;
  t.tot_active += len;
  {
    // inside the cached leaf (pre < pos <= pre + tot, or the very start while the first leaf is cached) with room for two items?
    // (as integer arithmetic: a boolean expression of wave-uniform compares becomes a chain of lane-mask selects)
    uint32_t k0 = pos - c.pre - 1;                                   // pos <= pre wraps to a huge value
    uint32_t miss = (k0 >= c.tot ? 1u : 0u) & ((pos | c.p) != 0 ? 1u : 0u);
    if (miss | (c.n > 62 ? 1u : 0u)) tl_prepare(t, c, pos, 2);   // (no leaf cached: n = 255)
    if (t.err) { tl_none(c); return; }
  }
  const uint32_t k = pos - c.pre;
  uint32_t inc = lmw::scan_incl_add(c.len);
  uint64_t hit = lmw::ballot(inc >= k);                               // (k = 0, or an empty first leaf: lane 0)
  uint32_t slot = (uint32_t)lmw::ffs64(hit);
  uint32_t sln = lmw::bcast(c.len, (int)slot), sid = lmw::bcast(c.id, (int)slot);
  uint32_t off = k - (lmw::bcast(inc, (int)slot) - sln);             // 1..sln: the cursor sits right behind element off-1 of the item (k = 0: 0)
  c.tot += len; c.dirty = true;
  if ((off ^ sln) | ((sid + sln) ^ pid0) | (k == 0 ? 1u : 0u)) {
    // a new item.  Inside an item (0 < off < sln) that one is cut and the new run goes between the halves
    uint32_t idx = k ? slot + 1 : 0u;
    if (k != 0 && off < sln) {
      c.len = lane == slot ? off : c.len;
      tl_shift_in(c, idx, sid + off, sln - off, sid + off - 1);
    }
    tl_shift_in(c, idx, pid0, len, k ? sid + off - 1 : NONE);
  } else {
    // typing goes on where it stopped: the item grows (FugueSpan::is_mergeable, fugue_span.rs:281-300 — next id of the same
    // peer, origin_left = the item's last element; sid + sln == pid0 cannot hold across peers: a run never crosses 2^24)
    c.len = lane == slot ? sln + len : c.len;
  }
}
// delete of the Ln elements from position pos0 (0-based) on — the part of it that lies in ONE leaf; returns the number of elements left
// (the range runs on into the next leaf: the row loop queues the rest as a row of its own — a loop in here put 49 register moves in
// front of every delete row).  The row names its targets as ids, ascending with the position (list_op.rs:288-379); the reference
// deletes by position and never looks at the ids while it does (crdt_rope.rs:256-335) — so does the prefix: nothing here is ever
// retreated, what a damaged row names instead of the elements at its position is of no consequence (the tracker behind the prefix:
// ts_del_positional).
LM_DEV uint32_t tl_delete(Ts& t, Tl& c, uint32_t pos0, uint32_t Ln, bool& emptied) {
  uint32_t lane = (uint32_t)lmw::lane();
  if (pos0 > t.tot_active || Ln > t.tot_active - pos0) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return 0; }
  const uint32_t pos = pos0 + 1;
  {
    uint32_t k0 = pos - c.pre - 1;
    if ((k0 >= c.tot ? 1u : 0u) | (c.n > 63 ? 1u : 0u)) tl_prepare(t, c, pos, 1);
    if (t.err) { tl_none(c); return 0; }
  }
  uint32_t s = pos - c.pre - 1, e = s + Ln < c.tot ? s + Ln : c.tot;     // elements [s, e) of the leaf
  uint32_t take = e - s;
  uint32_t inc = lmw::scan_incl_add(c.len), start = inc - c.len;
  uint32_t lo = start > s ? start : s, hi = inc < e ? inc : e;
  bool has = hi > lo;
  uint64_t mid = lmw::ballot(has & (lo > start) & (hi < inc));
  uint32_t cut = has ? hi - lo : 0u;
  if (mid) {
    // strictly inside one item: left part | (deleted) | right part — one more item
    uint32_t slot = (uint32_t)lmw::ffs64(mid);
    uint32_t sid = lmw::bcast(c.id, (int)slot), sln = lmw::bcast(c.len, (int)slot), a = s - lmw::bcast(start, (int)slot);
    c.len = lane == slot ? a : c.len;
    tl_shift_in(c, slot + 1, sid + a + take, sln - a - take, sid + a + take - 1);
  } else {
    bool keep_tail = has & (lo == start) & (hi < inc);         // the item loses its head (an item that loses its tail only shrinks)
    c.len -= cut;
    c.id = keep_tail ? c.id + cut : c.id;
    c.ol = keep_tail ? c.id - 1 : c.ol;
    uint64_t mf = lmw::ballot(has & (c.len == 0));
    if (mf) {   // whole items go: the items behind them move down (they are one contiguous range of lanes)
      uint32_t r0 = (uint32_t)lmw::ffs64(mf), dn = (uint32_t)lmw::popc64(mf);
      uint32_t src = (lane + dn) & 63;
      uint32_t gid = lmw::shfl(c.id, (int)src), gln = lmw::shfl(c.len, (int)src), gol = lmw::shfl(c.ol, (int)src);
      bool mv = lane >= r0, gone = lane + dn >= c.n;
      c.id = mv ? (gone ? NONE : gid) : c.id;
      c.len = mv ? (gone ? 0u : gln) : c.len;
      c.ol = mv ? (gone ? NONE : gol) : c.ol;
      c.n -= dn;
      emptied |= c.n == 0;
    }
  }
  c.tot -= take; c.dirty = true;
  t.tot_active -= take;
  return Ln - take;
}
// the prefix is done: everything goes back to HBM / LDS in the tracker's form; leaves that lost every item leave the directory
LM_DEV void tl_finish(Ts& t, Tl& c, bool emptied) {
  int lane = lmw::lane();
  tl_store(t, c);
  lmw::wave_sync();
  if (emptied) {
    uint32_t w = 0;
    for (uint32_t i0 = 0; i0 < t.n_dir; i0 += 64) {
      uint32_t i = i0 + (uint32_t)lane;
      uint32_t a = i < t.n_dir ? t.da[i] : 0u, b = i < t.n_dir ? t.db[i] : 0u;
      bool keep = i < t.n_dir && sa_n(a) != 0;
      uint64_t m = lmw::ballot(keep);
      uint32_t dst = w + (uint32_t)lmw::popc64(m & ((1ull << lane) - 1));
      lmw::wave_sync();
      if (keep) { t.da[dst] = a; t.db[dst] = b; }
      lmw::wave_sync();
      w += (uint32_t)lmw::popc64(m);
    }
    t.n_dir = w ? w : 1u;   // (nothing left: entry 0 still is the container's first leaf, empty — the state a replay starts from)
    if (!w && lane == 0) { t.da[0] = sa_make(sa_leaf(t.da[0]), 0, false); t.db[0] = 0; }
    lmw::wave_sync();
  }
  // origin_right / status of every item the prefix leaves behind (all 64 slots of a leaf: what lies beyond its items is never read)
  for (uint32_t q = 0; q < t.n_dir; q++) {
    uint32_t* rec = t.it + (uint64_t)sa_leaf(lmw::first(t.da[q])) * SP_REC;
    rec[192 + lane] = NONE; rec[256 + lane] = 0u;
  }
  t.n_alive = t.tot_active;
  t.cache_leaf = NONE; t.cr.n = 255; t.cache_pre = NONE; t.dirty = false; t.loc_pend = 0;
  t.ds_on = t.n_dir > SD_LINEAR;
  if (t.ds_on) sd_sums_from(t, 0);
}

}  // namespace lm
