// Integrate stage, span-granular kernel — the default (LM_SPAN=0 selects the element-granular lm_k_integrate.h).
// Same replay as lm_k_integrate.h — one wavefront per document, Event-Graph-Walker over every Text / List container —
// but a leaf slot holds a RUN of elements (the reference's FugueSpan, container/richtext/fugue_span.rs:191-279):
//   id0 (peer:8 | counter:24), len, origin_left (of the first element), origin_right (shared), status (shared);
//   element k > 0 of a run has origin_left = id0 + k - 1.
// A document typed in runs of ≈20 characters needs ≈10x fewer leaves than with one slot per element: the LDS directory
// shrinks to a few hundred entries (searched linearly, 64 entries per step), a concurrent run of the other peer is ONE
// item for the sibling scan instead of a chain of leaves, and the emit stage streams runs.
// Reference algorithm: container/richtext/tracker.rs:88-160,193-252,354-546, tracker/crdt_rope.rs:63-247,256-381.
#pragma once
#include "lm_k_integrate.h"

namespace lm {

static constexpr uint32_t SP_REC = 320;   // dwords per leaf record: id0[64] | len[64] | origin_left[64] | origin_right[64] | status[64]
// directory word A: leaf (17 bits) | holds a non-future item (bit 17) | item count (7 bits at 18) ; word B: active length
LM_DEV uint32_t sa_make(uint32_t leaf, uint32_t n, bool nf) { return leaf | (nf ? DIR_NF : 0u) | (n << 18); }
LM_DEV uint32_t sa_leaf(uint32_t a) { return a & DIR_LEAF_MASK; }
LM_DEV uint32_t sa_n(uint32_t a) { return (a >> 18) & 0x7f; }
LM_DEV bool sa_nf(uint32_t a) { return (a & DIR_NF) != 0; }

struct SpanRegs { uint32_t n, id, len, ol, orr, st; };   // one leaf in registers: lane i holds item i
struct SpanItem { uint32_t id, len, ol, orr, st; };

struct Ts {   // wave-uniform context of one (document, sequence container) replay
  uint32_t* it;                   // HBM leaf records of the document: [leaf * SP_REC + field * 64 + slot]
  uint32_t* loc;                  // doc element → leaf
  const uint32_t* ebase;          // LDS: element base per peer
  const uint32_t* cur;            // LDS: tracker version per peer at the head of the node being replayed
  const uint32_t* end;            // LDS: version being rendered per peer (no element exists at or beyond it)
  uint32_t* da;                   // LDS directory, word A per leaf in document order
  uint32_t* db;                   // LDS directory, word B (active length)
  uint32_t* ds;                   // LDS: per 64 directory entries the sum of their active lengths — kept once the directory is longer than
                                  // SD_LINEAR entries (ds_on); exact for every block except the cached leaf's, whose entry the fast paths
                                  // change in place: that block is re-summed when the leaf leaves the cache and before a search
  bool ds_on;
  uint32_t n_dir, dir_cap, n_leaf, leaf_cap, tot_active;
  uint32_t n_alive;               // elements inserted and never deleted by a replayed op = visible at the rendered version
  uint32_t cache_leaf;            // leaf mirrored in `cr` (NONE = none); WRITE-BACK: `cr` is the truth, HBM is updated by sp_flush
  uint32_t cache_p;               // its directory position (kept in step with directory inserts)
  uint32_t cache_pre;             // active elements in front of the cached leaf, NONE when not known: an edit at a position
                                  // inside the cached leaf then needs no directory search at all
  SpanRegs cr;
  bool dirty;                     // `cr` differs from the leaf record in HBM
  uint32_t loc_pend;              // per lane: the item of `cr` in this lane entered the leaf since the last flush, its loc[] entries are pending
                                  // (a lane flag, not a 64-bit mask: it shifts with the items on the vector unit — the kernel is scalar-issue bound)
  int32_t err;
  uint32_t beyond;                // an insert row named a position beyond the end (sticky; LM_DATA_CORRUPTION when the document is closed)
#ifdef LM_PROF
  mutable uint64_t prof[PF_N];
#endif
};
// internal verdict of a delete row whose target ids are not the elements at its position: the row loop applies the rest of the row
// BY POSITION, as the reference applies every delete (crdt_rope.rs:256-335), and remembers what it deleted for later retreats
static constexpr int32_t ST_POSDEL = 101;
static constexpr uint32_t PD_CAP = 63;   // pieces per document (3 words each; one lane per piece when the list is searched)
// The bookkeeping of positional deletes lives in eight LDS words in front of the peer tables (t.ebase - PD_LDS .. t.ebase) and in the
// document's list in HBM — NOT in the wave-uniform context: the kernel sits at its SGPR limit, seven more live scalars cost the
// replay of every healthy document 3 % (profiles/r05_posdel_ab.log).  [0] position where the row stopped matching  [1] ids left
// [2] first id of the row  [3] ids that matched  [4] pieces in the list  [5,6] the list (pointer; 0 = positional deletes are off)
// [7] the list's capacity in pieces  [8,9] row index (pointer; 0 = none): per op row of the batch the first piece of that row in its
// document's list — a document staged on a snapshot's state (lm_snapshot_base.h) deletes base content by position as a matter of
// course, thousands of rows, where a damaged document has a handful (PD_CAP pieces, found by one lane per piece)
static constexpr uint32_t PD_LDS = 12;
LM_DEV uint32_t* pd_w(const Ts& t) { return const_cast<uint32_t*>(t.ebase) - PD_LDS; }
template <bool POS>
LM_DEV void pd_stop(Ts& t, uint32_t k, uint32_t left, uint32_t first, uint32_t matched) {   // (cold: a delete row that does not match its position)
  // (every kernel but k_integrate_span_pos: the document leaves with this verdict and is replayed by that kernel; POS with the optimistic
  // directory, retry_pass 3: a directory overflow met on the way here stays the verdict — the document is replayed with the worst-case one)
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_BASE") && lmw::lane() == 0 && t.err) fprintf(stderr, "pd_stop<%d> with err %d\n", (int)POS, (int)t.err);
#endif
  if (!POS || t.err != ST_RETRY) t.err = ST_POSDEL;
  if (POS && lmw::lane() == 0) { uint32_t* w = pd_w(t); w[0] = k; w[1] = left; w[2] = first; w[3] = matched; }
}

LM_DEV uint32_t ts_g(const Ts& t, uint32_t pid) { return t.ebase[pid_peer(pid)] + pid_ctr(pid); }
LM_DEV bool sp_has(uint32_t id0, uint32_t len, uint32_t x) { return x >= id0 && x - id0 < len; }   // element x inside the run (same peer implied)

// ---- leaf access
LM_DEV SpanRegs sp_load(const Ts& t, uint32_t L, uint32_t n) {
  int lane = lmw::lane();
  lmw::wave_sync();
  if (L == t.cache_leaf) return t.cr;
  PROF_CNT(t, PF_NEXTRA, 1);   // leaf fetched from HBM
  SpanRegs r;
  r.n = n;
  bool in = (uint32_t)lane < n;
  const uint32_t* rec = t.it + (uint64_t)L * SP_REC;
  r.id = in ? rec[lane] : NONE;
  r.len = in ? rec[64 + lane] : 0u;
  r.ol = in ? rec[128 + lane] : NONE;
  r.orr = in ? rec[192 + lane] : NONE;
  r.st = in ? rec[256 + lane] : ST_FUT;
  return r;
}
// leaf record L in HBM := R
LM_DEV void sp_write(Ts& t, uint32_t L, const SpanRegs& R) {
  int lane = lmw::lane();
  if ((uint32_t)lane < R.n) {
    uint32_t* rec = t.it + (uint64_t)L * SP_REC;
    rec[lane] = R.id; rec[64 + lane] = R.len; rec[128 + lane] = R.ol; rec[192 + lane] = R.orr; rec[256 + lane] = R.st;
  }
}
// (register invariant of SpanRegs: lanes >= n hold id NONE, len 0, status ST_FUT — sp_load, sp_shift_in and the split keep it —
// so the per-lane predicates below need no `lane < n` term)
LM_DEV uint32_t sp_alen(const SpanRegs& R) { return st_active(R.st) ? R.len : 0u; }
LM_DEV bool sp_nf(const SpanRegs& R) { return lmw::ballot(!(R.st & ST_FUT)) != 0; }
// lanes whose run holds element x: ids of different peers are >= 2^24 apart and a run never crosses 2^24, so one unsigned
// compare covers "same peer, inside the run" (and NONE / len 0 of the unused lanes never match)
LM_DEV uint64_t sp_hit(const SpanRegs& R, uint32_t x) { return lmw::ballot(x - R.id < R.len); }
// loc[] := leaf L for every element of the items in the lanes with `pend` set.  The first LOC_SHORT elements of each item
// are written by the item's own lane (one divergent loop for all items at once: an item-by-item walk cost ≈40 scalar
// instructions per item and this kernel is scalar-issue bound); what lies beyond — long runs — is written 64 elements
// per store by the whole wave, item by item.
#ifndef LM_LOC_SHORT
#define LM_LOC_SHORT 16
#endif
static constexpr uint32_t LOC_SHORT = LM_LOC_SHORT;
// loc[] layout.  Default (LM_LOC16): loc[] is kept only for the HEAD of every item and for the elements whose counter is a
// multiple of LOC_W (64; 16 in round 2); -DLM_LOC_FULL builds the per-element layout of rounds 1-2 instead (kept as the A/B and parity counterpart).
// Measured on configs[1], 5,000 documents per launch (profiles/r02_ab_prepared.log): 14.95 -> 14.50 ms, 12.98 ms with the
// PLAIN + SWEEP instantiation on top.
#if !defined(LM_LOC_FULL) && !defined(LM_LOC16)
#define LM_LOC16 1
#endif
#ifdef LM_LOC16
// (LOC_W: the window — 16 in round 2.  Typing comes in long runs — 70 % of a configs[1] document's text sits in items longer than
// 64 elements — and a flush wrote 1 + len / 16 entries per pending item; a lookup reads one window with one wave load whatever
// its width up to 64, and lookups are rare: ≈170 per document against ≈2,900 flushes)
#ifndef LM_LOC_W
#define LM_LOC_W 64
#endif
static constexpr uint32_t LOC_W = LM_LOC_W;   // a power of two <= 64
// loc[] is kept only for the HEAD of every item and for the elements whose
// counter is a multiple of LOC_W — everything else stays NONE.  Items are only ever cut or appended to, never joined, so a head
// stays a head; the nearest kept entry at or below an element of an item, inside its LOC_W-aligned counter window, is therefore an
// element of the same item (ts_loc_find).  A flush writes 1 + len/16 entries per pending item instead of len.
LM_DEV void sp_set_loc_lanes(Ts& t, const SpanRegs& R, bool pend, uint32_t L) {
  if (!t.loc) return;   // loc[] is not kept yet (ts_build_loc): nothing can ask for an element by id while the replay is one chain
  uint32_t len = pend ? R.len : 0u;
  uint32_t c0 = pid_ctr(R.id);
  uint32_t g = pend ? t.ebase[pid_peer(R.id)] + c0 : 0u;
  if (pend) t.loc[g] = L;
  uint32_t k = LOC_W - (c0 & (LOC_W - 1));             // offset of the first multiple of LOC_W beyond the head
  // (most pending items are short: the trips are taken only while some item still has such an element — four masked trips
  // for every call were 4.5 % of the kernel)
  for (uint32_t i = 0; i < 4 && lmw::any(k < len); i++, k += LOC_W) if (k < len) t.loc[g + k] = L;
  uint64_t m = lmw::ballot(k < len);                   // items with more than four such elements
  while (m) {
    int j = lmw::ffs64(m);
    m &= m - 1;
    uint32_t gj = lmw::bcast(g, j), lj = lmw::bcast(len, j), kj = lmw::bcast(k, j);
    for (uint32_t kk = kj + LOC_W * (uint32_t)lmw::lane(); kk < lj; kk += 64 * LOC_W) t.loc[gj + kk] = L;
  }
}
// leaf of element `pid` (wave-uniform), NONE when no kept entry lies at or below it in its window
LM_DEV uint32_t ts_loc_find(const Ts& t, uint32_t pid) {
  if (!t.loc) return NONE;
  uint32_t ctr = pid_ctr(pid), lo = ctr & ~(LOC_W - 1), eb = t.ebase[pid_peer(pid)];
  uint32_t lane = (uint32_t)lmw::lane();
  uint32_t v = (lane < LOC_W && lo + lane <= ctr) ? t.loc[eb + lo + lane] : NONE;
  uint64_t m = lmw::ballot(v != NONE);
  if (!m) return NONE;
  int top = 63 - __builtin_clzll((unsigned long long)m);
  return lmw::bcast(v, top);
}
#else
LM_DEV void sp_set_loc_lanes(Ts& t, const SpanRegs& R, bool pend, uint32_t L) {
  uint32_t len = pend ? R.len : 0u;
  uint32_t g = pend ? t.ebase[pid_peer(R.id)] + pid_ctr(R.id) : 0u;
  uint32_t lim = len < LOC_SHORT ? len : LOC_SHORT;
  for (uint32_t k = 0; k < lim; k++) t.loc[g + k] = L;
  uint64_t m = lmw::ballot(len > LOC_SHORT);
  while (m) {
    int j = lmw::ffs64(m);
    m &= m - 1;
    uint32_t gj = lmw::bcast(g, j), lj = lmw::bcast(len, j);
    for (uint32_t k = LOC_SHORT + (uint32_t)lmw::lane(); k < lj; k += 64) t.loc[gj + k] = L;
  }
}
#endif
// The cached leaf is write-back: an edit that stays inside it issues NO global store (on gfx9-class hardware stores share
// the load counter, so a store per edit makes the next op-row fetch wait a full write round trip).  HBM and loc[] catch up
// when another leaf takes the cache and at the end of the replay; a sibling scan that must read loc[] writes the pending loc[]
// entries first (sp_flush_loc).
LM_DEV void sp_flush(Ts& t) {
  if (t.cache_leaf == NONE) return;
  if (t.dirty) { sp_write(t, t.cache_leaf, t.cr); t.dirty = false; }
  if (lmw::any(t.loc_pend != 0)) { sp_set_loc_lanes(t, t.cr, t.loc_pend != 0, t.cache_leaf); t.loc_pend = 0; }
}
// only the pending loc[] entries (before a loc[] read that may concern an item of the cached leaf); the leaf stays cached
LM_DEV void sp_flush_loc(Ts& t) {
  if (t.cache_leaf != NONE && lmw::any(t.loc_pend != 0)) { sp_set_loc_lanes(t, t.cr, t.loc_pend != 0, t.cache_leaf); t.loc_pend = 0; }
}
// leaf L becomes the cached leaf (its registers are set by the caller); another cached leaf is written back first
LM_DEV void sd_sync_cached(Ts& t);
LM_DEV void sp_take(Ts& t, uint32_t L) {
  if (t.cache_leaf != L) { sp_flush(t); sd_sync_cached(t); t.cache_leaf = L; t.dirty = false; t.loc_pend = 0; }
}

// ---- directory (LDS).  Up to SD_LINEAR entries it is searched linearly, 64 entries per step (configs[1]: ≈120 leaves, two
// steps); a longer one — deep histories: ≈1,200 leaves for a 1M-op document, 19 steps per search — gets a second level, the sums
// of 64 entries each, so that a search is one step over the sums and one inside a block.
#ifndef LM_SD_LINEAR
#define LM_SD_LINEAR 128
#endif
static constexpr uint32_t SD_LINEAR = LM_SD_LINEAR;
#ifndef LM_SD_BSH
#define LM_SD_BSH 6
#endif
static constexpr uint32_t SD_BSH = LM_SD_BSH, SD_BS = 1u << LM_SD_BSH;   // entries per block of sums (tests: LM_SD_BSH=2 — block boundaries everywhere)   // (tests build the harness with LM_SD_LINEAR=2: every document gets the second level)
LM_DEV void sd_block_sum(Ts& t, uint32_t j) {   // ds[j] := Σ db over block j
  uint32_t i = (j << SD_BSH) + (uint32_t)lmw::lane();
  lmw::wave_sync();
  uint32_t s = lmw::reduce_add(((uint32_t)lmw::lane() < SD_BS && i < t.n_dir) ? t.db[i] : 0u);
  if (lmw::lane() == 0) t.ds[j] = s;
  lmw::wave_sync();
}
LM_DEV void sd_sums_from(Ts& t, uint32_t j0) { for (uint32_t j = j0; (j << SD_BSH) < t.n_dir; j++) sd_block_sum(t, j); }
LM_DEV void sd_sync_cached(Ts& t) { if (t.ds_on && t.cache_leaf != NONE && t.cache_p < t.n_dir) sd_block_sum(t, t.cache_p >> SD_BSH); }
LM_DEV void sd_set(Ts& t, uint32_t p, uint32_t a, uint32_t b) {
  lmw::wave_sync();
  uint32_t old = t.db[p];
  lmw::wave_sync();
  if (lmw::lane() == 0) { t.da[p] = a; t.db[p] = b; if (t.ds_on) t.ds[p >> SD_BSH] += b - old; }
  t.tot_active += b - old;
  lmw::wave_sync();
}
LM_DEV void sd_insert_after(Ts& t, uint32_t p, uint32_t a, uint32_t b) {
  int lane = lmw::lane();
  if (t.n_dir >= t.dir_cap) { LM_SETERR(t.err, ST_RETRY); return; }
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
  t.tot_active += b;
  if (t.cache_leaf != NONE && t.cache_p >= q) t.cache_p++;
  lmw::wave_sync();
  if (!t.ds_on && t.n_dir > SD_LINEAR) { t.ds_on = true; sd_sums_from(t, 0); }
  else if (t.ds_on) sd_sums_from(t, q >> SD_BSH);   // every entry behind q moved up by one
}
// k-th active element (1 <= k <= tot_active) → directory position; k becomes the rank inside that leaf
LM_DEV uint32_t sd_find_kth(Ts& t, uint32_t& k) {
  int lane = lmw::lane();
  lmw::wave_sync();
  uint32_t first = 0;
  if (t.ds_on) {
    sd_sync_cached(t);
    uint32_t nb = (t.n_dir + SD_BS - 1) >> SD_BSH, blk = NONE;
    for (uint32_t j0 = 0; j0 < nb && blk == NONE; j0 += 64) {
      uint32_t j = j0 + (uint32_t)lane;
      uint32_t a = j < nb ? t.ds[j] : 0u;
      uint32_t inc = lmw::scan_incl_add(a);
      uint64_t m = lmw::ballot(inc >= k);
      if (m) { int s2 = lmw::ffs64(m); k -= lmw::bcast(inc, s2) - lmw::bcast(a, s2); blk = j0 + (uint32_t)s2; }
      else k -= lmw::bcast(inc, 63);
    }
    if (blk == NONE) return NONE;
    first = blk << SD_BSH;
  }
  for (uint32_t i0 = first; i0 < t.n_dir; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    uint32_t a = i < t.n_dir ? t.db[i] : 0u;
    uint32_t inc = lmw::scan_incl_add(a);
    uint64_t m = lmw::ballot(inc >= k);
    if (m) {
      int s = lmw::ffs64(m);
      k -= lmw::bcast(inc, s) - lmw::bcast(a, s);
      return i0 + (uint32_t)s;
    }
    k -= lmw::bcast(inc, 63);
  }
  return NONE;
}
LM_DEV uint32_t sd_find_leaf(const Ts& t, uint32_t L) {
  int lane = lmw::lane();
  lmw::wave_sync();
  for (uint32_t i0 = 0; i0 < t.n_dir; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    uint64_t m = lmw::ballot(i < t.n_dir && sa_leaf(t.da[i]) == L);
    if (m) return i0 + (uint32_t)lmw::ffs64(m);
  }
  return NONE;
}
// active elements in front of directory position p
LM_DEV uint32_t sd_prefix(Ts& t, uint32_t p) {
  int lane = lmw::lane();
  lmw::wave_sync();
  uint32_t acc = 0, first = 0;
  if (t.ds_on) {
    sd_sync_cached(t);
    uint32_t nbp = p >> SD_BSH;
    for (uint32_t j = (uint32_t)lane; j < nbp; j += 64) acc += t.ds[j];
    first = nbp << SD_BSH;
  }
  for (uint32_t i0 = first; i0 < p; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    acc += i < p ? t.db[i] : 0u;
  }
  return lmw::reduce_add(acc);
}
// A delete row of the replay carries the position of its leftmost target (DeleteSpan, list_op.rs:288-379) next to the target
// ids.  The reference deletes BY POSITION (crdt_rope.rs:256-335) and only records which ids it met; this kernel deletes by id.
// For every blob a writer produces the two agree; for a damaged one they may not — so every piece of such a row is checked:
// it must be active and sit exactly at the row's position (everything deleted before it in this row is inactive by then).
// A disagreement is LM_DATA_CORRUPTION, never a value the reference would not have computed.
LM_DEV bool ts_del_pos_ok(Ts& t, const SpanRegs& R, uint32_t slot, uint32_t s_off, uint32_t st0, uint32_t hint_k) {
  if (t.cache_pre == NONE) t.cache_pre = sd_prefix(t, t.cache_p);
  uint32_t al = sp_alen(R);
  uint32_t inc = lmw::scan_incl_add(al);
  uint32_t before = lmw::bcast(inc, (int)slot) - lmw::bcast(al, (int)slot);
  return st_active(st0) && t.cache_pre + before + s_off + 1 == hint_k;
}
LM_DEV void sd_refresh(Ts& t, uint32_t p, uint32_t L, const SpanRegs& R) {   // directory entry of leaf L from its registers
  sd_set(t, p, sa_make(L, R.n, sp_nf(R)), lmw::reduce_add(sp_alen(R)));
}

// ---- leaf edits
// lanes >= idx move up by cnt (1 or 2) and items A (, B) drop in at idx (, idx+1); the leaf has room
LM_DEV SpanRegs sp_shift_in(const SpanRegs& R, uint32_t& lp, uint32_t idx, const SpanItem& A, const SpanItem& B, uint32_t cnt, bool setA, bool setB) {
  // items idx.. move up by cnt, A (and B) take lanes idx (idx + 1).  `lp` (the pending-loc[] lane flag) follows the same shift;
  // setA / setB: the item's loc[] must be (re)written.  A part cut off an item comes from the item at lane idx-1: if that
  // item's loc[] is still pending, so is the part's.  Selects only — no exec-masked blocks — and the unused lanes need no
  // clearing: they are shifted in from unused lanes (register invariant above).
  uint32_t lane = (uint32_t)lmw::lane();
  SpanRegs N;
  N.n = R.n + cnt;
  uint32_t pid, pln, pol, por, pst, plp;
  // (the second shift and the second item under a branch on cnt: the kernel runs at the machine's issue rate, where a wave-wide
  // VALU operation costs what a scalar one costs — twelve shifts and selects saved for cnt == 1 beat the three scalar instructions of the branch)
  pid = lmw::shift_up0(R.id, 1); pln = lmw::shift_up0(R.len, 1); pol = lmw::shift_up0(R.ol, 1); por = lmw::shift_up0(R.orr, 1); pst = lmw::shift_up0(R.st, 1); plp = lmw::shift_up0(lp, 1);
  if (cnt == 2) {
    pid = lmw::shift_up0(pid, 1); pln = lmw::shift_up0(pln, 1); pol = lmw::shift_up0(pol, 1);
    por = lmw::shift_up0(por, 1); pst = lmw::shift_up0(pst, 1); plp = lmw::shift_up0(plp, 1);
  }
  uint32_t inh = idx ? lmw::bcast(lp, (int)idx - 1) : 0u;
  bool sh = lane >= idx + cnt, isA = lane == idx;
  N.id = sh ? pid : R.id; N.len = sh ? pln : R.len; N.ol = sh ? pol : R.ol; N.orr = sh ? por : R.orr; N.st = sh ? pst : R.st;
  uint32_t nlp = sh ? plp : lp;
  N.id = isA ? A.id : N.id; N.len = isA ? A.len : N.len; N.ol = isA ? A.ol : N.ol; N.orr = isA ? A.orr : N.orr; N.st = isA ? A.st : N.st;
#ifdef LM_LOC16
  nlp = isA ? 1u : nlp;   // every new item has a new head
#else
  nlp = isA ? ((setA ? 1u : 0u) | inh) : nlp;
#endif
  if (cnt == 2) {
    bool isB = lane == idx + 1;
    N.id = isB ? B.id : N.id; N.len = isB ? B.len : N.len; N.ol = isB ? B.ol : N.ol; N.orr = isB ? B.orr : N.orr; N.st = isB ? B.st : N.st;
#ifdef LM_LOC16
    nlp = isB ? 1u : nlp;
#else
    nlp = isB ? ((setB ? 1u : 0u) | inh) : nlp;
#endif
  }
  lp = nlp;
  return N;
}
// Insert `cnt` (1 or 2) items — A, then B — as items idx, idx+1 of the leaf at directory position p (registers R, which
// may carry edits the caller made in registers): the leaf becomes the cached leaf and nothing is stored.  A leaf without
// room is split first (lower half keeps 32 items); the half that does not hold the edit is written to HBM.  newA / newB:
// the item's elements are new to the container (loc[] must be written); a part of a split run only needs loc[] when it
// lands in another leaf.  On return (p, idx) address item A and R holds its leaf.
LM_DEV void sp_insert_items(Ts& t, uint32_t& p, SpanRegs& R, uint32_t& idx, const SpanItem& A, const SpanItem& B, uint32_t cnt,
                            bool newA, bool newB, uint32_t pre) {   // pre: active elements before leaf p (NONE = unknown)
  int lane = lmw::lane();
  lmw::wave_sync();
  sd_sync_cached(t);   // (a split below may hand the cache to the new leaf without going through sp_take)
  uint32_t L = sa_leaf(lmw::first(t.da[p]));
  sp_take(t, L);
  uint32_t lp = t.loc_pend;
  bool moved = false;   // the items end up in a leaf other than L
  if (R.n + cnt > 64) {
    if (t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
    uint32_t NL = t.n_leaf++;
    uint32_t nu = R.n - 32;
    SpanRegs U;   // items [32, n) → new leaf
    U.n = nu;
    U.id = lmw::shfl(R.id, (lane + 32) & 63); U.len = lmw::shfl(R.len, (lane + 32) & 63); U.ol = lmw::shfl(R.ol, (lane + 32) & 63);
    U.orr = lmw::shfl(R.orr, (lane + 32) & 63); U.st = lmw::shfl(R.st, (lane + 32) & 63);
    if ((uint32_t)lane >= nu) { U.id = NONE; U.len = 0; U.ol = NONE; U.orr = NONE; U.st = ST_FUT; }
    SpanRegs Lo = R;
    Lo.n = 32;
    if (lane >= 32) { Lo.id = NONE; Lo.len = 0; Lo.ol = NONE; Lo.orr = NONE; Lo.st = ST_FUT; }
    if (idx >= 32) {
      // the edit goes to the upper half, which becomes the cached leaf; the lower half is written out
      sp_write(t, L, Lo);
      sp_set_loc_lanes(t, Lo, (lp != 0) & (lane < 32), L);
      sd_refresh(t, p, L, Lo);
      if (pre != NONE) pre += lmw::first(t.db[p]);   // the new leaf starts behind the lower half
      sd_insert_after(t, p, sa_make(NL, nu, sp_nf(U)), lmw::reduce_add(sp_alen(U)));
      if (t.err) { t.cache_leaf = NONE; t.cr.n = 255; t.dirty = false; t.loc_pend = 0; return; }
      p = p + 1; idx -= 32; R = U; L = NL;
      t.cache_leaf = NL;
      lp = (uint32_t)lane < nu ? 1u : 0u;   // every item of the upper half moved: loc[] := NL at the next flush
      moved = true;
    } else {
      sp_write(t, NL, U);
      sp_set_loc_lanes(t, U, (uint32_t)lane < nu, NL);
      sd_insert_after(t, p, sa_make(NL, nu, sp_nf(U)), lmw::reduce_add(sp_alen(U)));
      if (t.err) { t.cache_leaf = NONE; t.cr.n = 255; t.dirty = false; t.loc_pend = 0; return; }
      R = Lo;
      lp = lane < 32 ? lp : 0u;
    }
  }
  SpanRegs N = sp_shift_in(R, lp, idx, A, B, cnt, newA || moved, newB || moved);
  t.cache_p = p; t.cache_pre = pre; t.cr = N; t.dirty = true; t.loc_pend = lp;
  sd_refresh(t, p, L, N);
  R = N;
}
LM_DEV void sp_insert_item(Ts& t, uint32_t& p, SpanRegs& R, uint32_t& idx, const SpanItem& it, bool new_elems, uint32_t pre) {
  sp_insert_items(t, p, R, idx, it, it, 1, new_elems, false, pre);
}

// ---- the common insert, instruction-lean (the kernel is issue-bound: one wave alone needs half the time of a full machine):
// the cursor lies inside the cached leaf, which has room, and origin_right is the very next item (or the cursor is
// inside an active run) — no directory search, no leaf traffic, no sibling scan; the directory entry is patched in place.
// Returns false without touching anything when the general path is needed.
LM_DEV bool ts_insert_fast(Ts& t, uint32_t pos, uint32_t pid0, uint32_t len) {
  int lane = lmw::lane();
  if (t.cr.n > 62 || pos <= t.cache_pre) {   // no cached leaf: n = 255; unknown prefix: cache_pre = NONE
#ifndef LM_NO_POS0_FAST
    // the very start of the sequence while its first leaf is cached (edits at the top of a document: ≈120 of a configs[1]
    // document's 1,800 insert rows, ≈870 instructions each on the general path): no origin_left, origin_right = item 0 when
    // that one is not future — nothing lies in between, nothing to merge with
    if ((pos | t.cache_p) != 0 || t.cr.n > 62) return false;
    SpanItem A;
#ifdef LM_NO_COLLIDE_FAST
    if (!(lmw::ballot(!(t.cr.st & ST_FUT)) & 1)) return false;
    A.orr = lmw::bcast(t.cr.id, 0);
#else
    {
      uint64_t nf0 = lmw::ballot(!(t.cr.st & ST_FUT));
      if (!nf0) return false;
      int rs = lmw::ffs64(nf0);
      A.orr = lmw::bcast(t.cr.id, rs);
      if (rs != 0) {   // future items in front of origin_right: see the in-leaf case below
        uint32_t o_id = lmw::bcast(t.cr.id, 0), o_ol = lmw::bcast(t.cr.ol, 0), o_or = lmw::bcast(t.cr.orr, 0);
        if (o_ol == NONE && (o_or != A.orr || pid_peer(o_id) <= pid_peer(pid0))) return false;
      }
    }
#endif
    A.id = pid0; A.len = len; A.ol = NONE; A.st = 0;
    uint32_t n0 = t.cr.n;
    t.cr = sp_shift_in(t.cr, t.loc_pend, 0, A, A, 1, true, false);
    t.dirty = true;
    t.cache_pre = 0;
    lmw::wave_sync();
    if (lane == 0) { t.da[0] = sa_make(t.cache_leaf, n0 + 1, true); lmw::lds_add(&t.db[0], len); }
    t.tot_active += len;
    lmw::wave_sync();
    return true;
#else
    return false;
#endif
  }
  uint32_t p = t.cache_p, k = pos - t.cache_pre;
  lmw::wave_sync();
  if (k > lmw::first(t.db[p])) return false;
  uint32_t n = t.cr.n;
  uint32_t al = sp_alen(t.cr);
  uint32_t inc = lmw::scan_incl_add(al);
  uint64_t hit = lmw::ballot((al != 0) & (inc >= k));
  if (!hit) return false;
  uint32_t slot = (uint32_t)lmw::ffs64(hit), idx = slot + 1;
  uint32_t sln = lmw::bcast(al, (int)slot), sid = lmw::bcast(t.cr.id, (int)slot);
  uint32_t off = k - (lmw::bcast(inc, (int)slot) - sln);   // 1..sln: cursor right after element off-1 of the run
  uint32_t v_or = lmw::bcast(t.cr.orr, (int)slot), v_st = lmw::bcast(t.cr.st, (int)slot);
  SpanItem A, B;
  A.id = pid0; A.len = len; A.ol = sid + off - 1; A.st = 0;
  uint32_t cnt;
  if (off < sln) {
    // inside an active run: cut, the new run goes between the halves
    A.orr = sid + off;
    B.id = sid + off; B.len = sln - off; B.ol = sid + off - 1; B.orr = v_or; B.st = v_st;
    t.cr.len = (uint32_t)lane == slot ? off : t.cr.len;
    cnt = 2;
  } else {
    uint64_t nf = lmw::ballot(((uint32_t)lane >= idx) & !(t.cr.st & ST_FUT));
#ifdef LM_NO_COLLIDE_FAST
    if (!((nf >> idx) & 1)) return false;   // origin_right in another leaf, or future items in between (nf has no bit below idx)
    A.orr = lmw::bcast(t.cr.id, (int)idx);
#else
    if (!nf) return false;                  // origin_right in another leaf
    int rs = lmw::ffs64(nf);
    A.orr = lmw::bcast(t.cr.id, rs);
    if ((uint32_t)rs != idx) {
      // future items lie between the cursor and origin_right — concurrent runs of other peers at this very place.  The sibling scan
      // (crdt_rope.rs:156-237) ends at its FIRST item when that item is not a sibling (its origin_left is not ours: nothing has
      // been passed yet, so it cannot be a descendant of a passed item) or is a sibling with our origin_right and a larger peer:
      // the new run goes right here.  With concurrent branches replayed in descending peer order (k_dag_b) this is what every
      // colliding insert meets; anything else takes the general path.
      uint32_t o_id = lmw::bcast(t.cr.id, (int)idx), o_ol = lmw::bcast(t.cr.ol, (int)idx), o_or = lmw::bcast(t.cr.orr, (int)idx);
      if (o_ol == A.ol && (o_or != A.orr || pid_peer(o_id) <= pid_peer(pid0))) return false;
    }
#endif
    if ((v_st | (sid + sln - pid0) | ((sid ^ pid0) >> 24) | (v_or ^ A.orr)) == 0) {
      // run merging (FugueSpan::is_mergeable): the item grows
      t.cr.len = (uint32_t)lane == slot ? sln + len : t.cr.len;
      t.loc_pend = (uint32_t)lane == slot ? 1u : t.loc_pend;
      t.dirty = true;
      if (lane == 0) lmw::lds_add(&t.db[p], len);
      t.tot_active += len;
      lmw::wave_sync();
      PROF_CNT(t, PF_NDHIT, 1);
      return true;
    }
    B = A;
    cnt = 1;
  }
  t.cr = sp_shift_in(t.cr, t.loc_pend, idx, A, B, cnt, true, false);
  t.dirty = true;
  if (lane == 0) { t.da[p] = sa_make(t.cache_leaf, n + cnt, true); lmw::lds_add(&t.db[p], len); }
  t.tot_active += len;
  lmw::wave_sync();
  if (cnt == 2) PROF_CNT(t, PF_NHIT, 1);
  return true;
}

// An insert whose position lies beyond the active length (a damaged row: no writer emits one).  The reference's position query
// comes back with the END of the rope (crdt_rope.rs:83 `query::<ActiveLenQueryPreferLeft>`, "missing": behind the last element
// whatever its status) — behind whatever concurrent branches its iteration happened to replay before, which depends on a hash map
// (dag/iter.rs:268-274): such a row has no value that is independent of the replay order.  LM_DATA_CORRUPTION, not a guess.  (The
// linear prefix — nothing concurrent exists there — keeps the clamp: the end of the rope is the end of the text.)
// ---- insert (Fugue integrate, crdt_rope.rs:63-247) of run [pid0, pid0+len) at active position pos
template <bool POS = false>
LM_DEV void ts_insert(Ts& t, uint32_t pos, uint32_t pid0, uint32_t len) {
  int lane = lmw::lane();
  PROF_T0();
  PROF_CNT(t, PF_NINS, 1);
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_INS") && lane == 0) fprintf(stderr, "INS pos %u id %u:%u len %u\n", pos, pid0 >> 24, pid0 & 0xffffff, len);
#endif
  // (POS: the kernel that replays damaged documents rejects the row — see above; the others clamp it as they always did: the check
  // was measured at 2 % of the replay of healthy documents, an exit edge in front of the in-leaf path)
  if (POS && pos > t.tot_active) { if (!t.err) LM_SETERR(t.err, ST_DATA_CORRUPTION); return; }   // (an earlier verdict — a directory overflow during the move in front of the row — stays)
  // (an insert BEYOND the end is placed by the reference behind everything its tree holds — trailing tombstones and future items
  // included (the B-tree query misses and returns the end of the tree, crdt_rope.rs:81-82), not behind the last ACTIVE element, where
  // the clamp below puts it: one damaged document in 3,600 was rendered with two list items elsewhere.  No writer emits such a row;
  // the row is noted — an OR into a register, no exit edge — and the document closed with LM_DATA_CORRUPTION)
  t.beyond |= pos > t.tot_active ? 1u : 0u;
  if (pos > t.tot_active) pos = t.tot_active;
  t.n_alive += len;
  if (ts_insert_fast(t, pos, pid0, len)) { PROF_ADD(t, PF_PLACE); PROF_CNT(t, PF_LEAF, 1u << 20); return; }
  uint32_t p = 0, idx = 0, origin_left = NONE, pre_p = 0;
  SpanRegs R;
  SpanItem nw;
  nw.id = pid0; nw.len = len; nw.st = 0;
  if (pos == 0) {
    R = sp_load(t, sa_leaf(lmw::first(t.da[0])), sa_n(lmw::first(t.da[0])));
  } else {
    uint32_t k = pos;
    lmw::wave_sync();
    if (t.cache_leaf != NONE && t.cache_pre != NONE && pos > t.cache_pre && pos - t.cache_pre <= lmw::first(t.db[t.cache_p])) {
      p = t.cache_p; k = pos - t.cache_pre;        // the position lies inside the cached leaf: no directory search
    } else p = sd_find_kth(t, k);
    if (p == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
    pre_p = pos - k;
    lmw::wave_sync();
    uint32_t a = lmw::first(t.da[p]);
    R = sp_load(t, sa_leaf(a), sa_n(a));
    uint32_t al = sp_alen(R);
    uint32_t inc = lmw::scan_incl_add(al);
    uint64_t hit = lmw::ballot((al != 0) & (inc >= k));
    if (!hit) { LM_SETERR(t.err, ST_INTERNAL); return; }
    uint32_t slot = (uint32_t)lmw::ffs64(hit);
    uint32_t off = k - (lmw::bcast(inc, (int)slot) - lmw::bcast(al, (int)slot));   // 1..len: cursor right after element off-1
    uint32_t sid = lmw::bcast(R.id, (int)slot), sln = lmw::bcast(R.len, (int)slot);
    origin_left = sid + off - 1;
    if (off < sln) {
      // cursor inside an active run: the run's next element is the (non-future) origin_right and nothing lies in between —
      // the run is cut and the new run dropped between the halves in one rewrite of the leaf
      SpanItem rt;
      rt.id = sid + off; rt.len = sln - off; rt.ol = sid + off - 1; rt.orr = lmw::bcast(R.orr, (int)slot); rt.st = lmw::bcast(R.st, (int)slot);
      if ((uint32_t)lane == slot) R.len = off;
      nw.ol = origin_left; nw.orr = sid + off;
      idx = slot + 1;
      PROF_ADD(t, PF_FIND);
      sp_insert_items(t, p, R, idx, nw, rt, 2, true, false, pre_p);
      PROF_ADD(t, PF_PLACE);
      PROF_CNT(t, PF_NHIT, 1);
      return;
    }
    idx = slot + 1;
  }
  PROF_ADD(t, PF_FIND);
  // origin_right = first non-future item at/after the cursor; items before it are "in between"
  uint32_t origin_right = NONE, r_ol = NONE, r_p = NONE, r_slot = 0;
  bool between = false;
  SpanRegs RR = R;
  {
    uint64_t nf = lmw::ballot(((uint32_t)lane >= idx) & !(R.st & ST_FUT));
    if (nf) {
      r_p = p; r_slot = (uint32_t)lmw::ffs64(nf);
      if (r_slot > idx) between = true;
    } else {
      if (R.n > idx) between = true;
      lmw::wave_sync();
      for (uint32_t q0 = p + 1; q0 < t.n_dir && r_p == NONE; q0 += 64) {
        uint32_t q = q0 + (uint32_t)lane;
        uint64_t hm = lmw::ballot(q < t.n_dir && sa_nf(t.da[q]));
        if (hm) r_p = q0 + (uint32_t)lmw::ffs64(hm);
      }
      if (r_p != NONE) {
        if (r_p > p + 1) between = true;
        uint32_t a = lmw::first(t.da[r_p]);
        RR = sp_load(t, sa_leaf(a), sa_n(a));
        uint64_t nf2 = lmw::ballot(!(RR.st & ST_FUT));
        if (!nf2) { LM_SETERR(t.err, ST_INTERNAL); return; }
        r_slot = (uint32_t)lmw::ffs64(nf2);
        if (r_slot > 0) between = true;
      } else if (p + 1 < t.n_dir) between = true;
    }
    if (r_p != NONE) { origin_right = lmw::bcast(RR.id, (int)r_slot); r_ol = lmw::bcast(RR.ol, (int)r_slot); }
  }
  PROF_ADD(t, PF_ORIGHT);
  uint32_t ins_p = p, ins_idx = idx;
  if (between) {
    // (the scan reads other leaves through sp_load, which serves the cached leaf from its registers; where it reads loc[],
    // the cached leaf's pending entries are written first — the leaf itself stays cached across the scan)
    // sibling scan over the future ITEMS between the cursor and origin_right (crdt_rope.rs:156-237); the elements inside a
    // run are continuations by construction, so every item is examined exactly once through its first element
    bool parent_right = origin_right != NONE && r_ol == origin_left;
    bool scanning = false, stop = false;
    uint32_t my_peer = pid_peer(pid0);
    uint32_t cp = p, ci = idx;
    SpanRegs C = R;
    for (uint32_t guard = 0; guard <= t.n_dir && !t.err && !stop; guard++) {
      uint32_t limit = (origin_right != NONE && cp == r_p) ? r_slot : C.n;
      // continuation items: origin_left = last element of the item right before them in this leaf, which the scan has
      // already passed (a run cut by deletes or by an insertion stays a chain of such items).  They are visited
      // non-siblings by construction — one ballot finds them all, and the scan hops over whole chains
      // (the same holds two to four items back — the right part of a run that an insertion cut follows what was inserted: an
      // item whose origin_left is the last element of one of the eight items in front of it, all inside the range, is passed like
      // a continuation; one at a time these were most of the ≈28 items a colliding edit of the other peer puts in the way)
      uint64_t contm;
      {
        uint32_t last = C.id + C.len - 1;
        uint32_t back = (uint32_t)lane - ci;   // items of the range in front of this one (huge for lanes in front of the range)
        bool inr = ((uint32_t)lane > ci) & ((uint32_t)lane < limit) & (C.ol != origin_left) & (C.ol != NONE);
        uint32_t l1 = lmw::shift_up(last, 1), l2 = lmw::shift_up(l1, 1), l3 = lmw::shift_up(l2, 1), l4 = lmw::shift_up(l3, 1);
        uint32_t l5 = lmw::shift_up(l4, 1), l6 = lmw::shift_up(l5, 1), l7 = lmw::shift_up(l6, 1), l8 = lmw::shift_up(l7, 1);
        bool found = (C.ol == l1) | ((back >= 2) & (C.ol == l2)) | ((back >= 3) & (C.ol == l3)) | ((back >= 4) & (C.ol == l4)) |
                     ((back >= 5) & (C.ol == l5)) | ((back >= 6) & (C.ol == l6)) | ((back >= 7) & (C.ol == l7)) | ((back >= 8) & (C.ol == l8));
        contm = lmw::ballot(inr & found);
      }
      for (uint32_t h = ci; h < limit; h++) {   // (every `stop` / error leaves through a break)
        if ((contm >> h) & 1) {
          uint64_t rest = ~contm >> h;
          uint32_t e = rest ? h + (uint32_t)lmw::ffs64(rest) : limit;   // first item at/after h that is not a continuation (bits at and beyond `limit` are clear)
          if (!scanning) { ins_p = cp; ins_idx = e; }
          h = e - 1;
          continue;
        }
        uint32_t o_id = lmw::bcast(C.id, (int)h), o_ol = lmw::bcast(C.ol, (int)h), o_or = lmw::bcast(C.orr, (int)h);
        PROF_CNT(t, PF_NHEAD, 1);   // in-between items examined by the sibling scan
        if (o_ol != origin_left) {
          // is o_ol one of the in-between elements already passed?  (inside an item at a position in [cursor, (cp,h)))
          bool visited = false;
          uint64_t here = sp_hit(C, o_ol);
          uint64_t in_r = cp != p ? sp_hit(R, o_ol) : 0ull;
          if (o_ol == NONE) visited = false;
          else if (here) { uint32_t xs = (uint32_t)lmw::ffs64(here); visited = xs < h && (cp != p || xs >= idx); }
          else if (in_r) visited = (uint32_t)lmw::ffs64(in_r) >= idx;
          else if (pid_ctr(o_ol) < t.cur[pid_peer(o_ol)]) visited = false;   // inside the tracker's version: not future
          else {
            sp_flush_loc(t);
            lmw::wave_sync();
#ifdef LM_LOC16
            uint32_t xl = ts_loc_find(t, o_ol);
#else
            uint32_t xl = t.loc[ts_g(t, o_ol)];
#endif
            if (xl < t.n_leaf) { uint32_t xp = sd_find_leaf(t, xl); visited = xp != NONE && xp > p && xp < cp; }
          }
          if (!visited) { stop = true; break; }
        } else {
          if (o_or == origin_right) {
            if (pid_peer(o_id) > my_peer) { stop = true; break; }
            scanning = false;
          } else {
            // the other item's right parent (crdt_rope.rs:205-216): its origin_right if that element is a sibling too
            uint32_t opr = NONE, o_p = NONE, o_s = 0;
            if (o_or != NONE) {
              uint32_t x_ol = NONE;
              uint64_t hc = sp_hit(C, o_or);
              uint64_t hr = (!hc && cp != p) ? sp_hit(R, o_or) : 0ull;
              uint64_t hq = (!hc && !hr && r_p != NONE && r_p != cp && r_p != p) ? sp_hit(RR, o_or) : 0ull;
              if (hc) { o_s = (uint32_t)lmw::ffs64(hc); o_p = cp; x_ol = lmw::bcast(C.id, (int)o_s) == o_or ? lmw::bcast(C.ol, (int)o_s) : o_or - 1; }
              else if (hr) { o_s = (uint32_t)lmw::ffs64(hr); o_p = p; x_ol = lmw::bcast(R.id, (int)o_s) == o_or ? lmw::bcast(R.ol, (int)o_s) : o_or - 1; }
              else if (hq) { o_s = (uint32_t)lmw::ffs64(hq); o_p = r_p; x_ol = lmw::bcast(RR.id, (int)o_s) == o_or ? lmw::bcast(RR.ol, (int)o_s) : o_or - 1; }
              else {
                sp_flush_loc(t);
                lmw::wave_sync();
#ifdef LM_LOC16
                uint32_t xl = ts_loc_find(t, o_or);
#else
                uint32_t xl = t.loc[ts_g(t, o_or)];
#endif
                if (xl >= t.n_leaf) { LM_SETERR(t.err, ST_INTERNAL); break; }
                uint32_t xp = sd_find_leaf(t, xl);
                if (xp == NONE) { LM_SETERR(t.err, ST_INTERNAL); break; }
                SpanRegs X = sp_load(t, xl, sa_n(lmw::first(t.da[xp])));
                uint64_t xm = sp_hit(X, o_or);
                if (!xm) { LM_SETERR(t.err, ST_INTERNAL); break; }
                o_s = (uint32_t)lmw::ffs64(xm); o_p = xp;
                x_ol = lmw::bcast(X.id, (int)o_s) == o_or ? lmw::bcast(X.ol, (int)o_s) : o_or - 1;
              }
              if (x_ol == origin_left) opr = o_or;
            }
            int c;
            if (opr != NONE && parent_right) c = (o_p < r_p || (o_p == r_p && o_s < r_slot)) ? -1 : 1;
            else if (opr != NONE) c = -1;
            else if (parent_right) c = 1;
            else c = 0;
            if (c < 0) scanning = true;
            else if (c == 0 && pid_peer(o_id) > my_peer) { stop = true; break; }
            else scanning = false;
          }
        }
        if (!scanning) { ins_p = cp; ins_idx = h + 1; }
      }
      if (stop || t.err) break;
      if (origin_right != NONE && cp == r_p) break;
      if (cp + 1 >= t.n_dir) break;
      cp++; ci = 0;
      if (origin_right != NONE && cp == r_p) C = RR;
      else { uint32_t a = lmw::first(t.da[cp]); C = sp_load(t, sa_leaf(a), sa_n(a)); }
    }
    if (t.err) return;
  }
  PROF_ADD(t, PF_BETWEEN);
  nw.ol = origin_left; nw.orr = origin_right;
  // run merging (FugueSpan::is_mergeable, fugue_span.rs:281-300): the new run continues the item right before the
  // insertion point — next id of the same peer, origin_left = that item's last element, same origin_right, same (clean)
  // status — so the item just grows.  Keystroke-per-change histories stay run-granular this way.
  if (ins_p == p && idx > 0 && ins_idx == idx) {
    uint32_t pv = idx - 1;
    uint32_t v_id = lmw::bcast(R.id, (int)pv), v_len = lmw::bcast(R.len, (int)pv), v_or = lmw::bcast(R.orr, (int)pv), v_st = lmw::bcast(R.st, (int)pv);
    if (v_st == 0 && v_id + v_len == pid0 && pid_peer(v_id) == pid_peer(pid0) && origin_left == pid0 - 1 && v_or == origin_right) {
      lmw::wave_sync();
      uint32_t a = lmw::first(t.da[p]);
      uint32_t L = sa_leaf(a);
      sp_take(t, L);
      if ((uint32_t)lane == pv) R.len = v_len + len;
      t.cache_p = p; t.cache_pre = pre_p; t.cr = R; t.dirty = true;
      t.loc_pend = (uint32_t)lane == pv ? 1u : t.loc_pend;   // (the whole item's loc[] is rewritten at the flush — the appended elements are what is new)
      sd_set(t, p, a, lmw::first(t.db[p]) + len);
      PROF_ADD(t, PF_PLACE);
      PROF_CNT(t, PF_NDHIT, 1);
      return;
    }
  }
  SpanRegs D;
  if (ins_p == p) D = R;
  else if (r_p != NONE && ins_p == r_p) D = RR;
  else { lmw::wave_sync(); uint32_t a = lmw::first(t.da[ins_p]); D = sp_load(t, sa_leaf(a), sa_n(a)); }
  sp_insert_item(t, ins_p, D, ins_idx, nw, true, ins_p == p ? pre_p : NONE);
  PROF_ADD(t, PF_PLACE);
}

// ---- the common status update, instruction-lean: the run holding element (peer, c) sits in the cached leaf, which has room
// for a cut; the run (or its part inside [c, c1)) gets the new status, c advances.  false = general path.
template <bool POS = false>
LM_DEV bool ts_update_fast(Ts& t, uint32_t peer, uint32_t& c, uint32_t c1, int mode, uint32_t hint_k = 0) {
  if (t.cr.n > 62) return false;   // (no cached leaf: n = 255)
  int lane = lmw::lane();
  uint32_t x = pid_make(peer, c);
  uint64_t hm = sp_hit(t.cr, x);
  if (!hm) return false;
  uint32_t slot = (uint32_t)lmw::ffs64(hm);
  uint32_t id0 = lmw::bcast(t.cr.id, (int)slot), ln = lmw::bcast(t.cr.len, (int)slot);
  uint32_t st0 = lmw::bcast(t.cr.st, (int)slot), st1 = st0;
  if (mode == UPD_SET_FUT) st1 |= ST_FUT;
  else if (mode == UPD_CLR_FUT) st1 &= ~ST_FUT;
  else if (mode == UPD_DEL_INC) st1 = (st1 + ST_DEL1) | ST_EVER;
  else if (st1 & (ST_DELMASK & ~ST_DEAD)) st1 -= ST_DEL1;
  uint32_t s_off = x - id0;
  uint32_t endc = pid_ctr(id0) + ln < c1 ? pid_ctr(id0) + ln : c1;
  uint32_t tail = pid_ctr(id0) + ln - endc, mid = endc - c;
  uint32_t n = t.cr.n;
#if defined(LM_NO_DEL_CHECK)
  (void)hint_k;
#elif defined(LM_DEL_CHECK_LIGHT)
  if (hint_k && t.cache_pre != NONE && !ts_del_pos_ok(t, t.cr, slot, s_off, st0, hint_k)) { pd_stop<POS>(t, hint_k, c1 - c, NONE, c); c = c1; return true; }
#else
  if (hint_k && !ts_del_pos_ok(t, t.cr, slot, s_off, st0, hint_k)) { pd_stop<POS>(t, hint_k, c1 - c, NONE, c); c = c1; return true; }
#endif
  if (mode == UPD_DEL_INC && !(st0 & ST_EVER)) t.n_alive -= mid;
  if ((s_off | tail) == 0) {
    t.cr.st = (uint32_t)lane == slot ? st1 : t.cr.st;
  } else {
    uint32_t orr0 = lmw::bcast(t.cr.orr, (int)slot);
    SpanItem A, B;
    uint32_t cnt;
    if (s_off > 0) {
      t.cr.len = (uint32_t)lane == slot ? s_off : t.cr.len;
      A.id = x; A.len = mid; A.ol = x - 1; A.orr = orr0; A.st = st1;
      B.id = pid_make(peer, endc); B.len = tail; B.ol = B.id - 1; B.orr = orr0; B.st = st0;
      cnt = tail ? 2u : 1u;
    } else {
      t.cr.len = (uint32_t)lane == slot ? mid : t.cr.len; t.cr.st = (uint32_t)lane == slot ? st1 : t.cr.st;
      A.id = pid_make(peer, endc); A.len = tail; A.ol = A.id - 1; A.orr = orr0; A.st = st0;
      B = A;
      cnt = 1;
    }
    t.cr = sp_shift_in(t.cr, t.loc_pend, slot + 1, A, B, cnt, false, false);
    n += cnt;
  }
  t.dirty = true;
  uint32_t d_act = (st_active(st1) ? mid : 0u) - (st_active(st0) ? mid : 0u);
  bool nf = sp_nf(t.cr);
  lmw::wave_sync();
  if (lane == 0) { t.da[t.cache_p] = sa_make(t.cache_leaf, n, nf); if (d_act) lmw::lds_add(&t.db[t.cache_p], d_act); }
  t.tot_active += d_act;
  lmw::wave_sync();
  c = endc;
  return true;
}

// ---- status update of the elements with ids [c0,c1) of `peer` (crdt_rope.rs:345-381 by id): walk run by run
// hint_k != 0: the first target is the hint_k-th active element (1-based) of the tracker's current version — a delete row
// carries its position — so the leaf is found through the LDS directory and only verified by id; loc[] is the fallback
template <bool POS = false>
LM_DEV void ts_update_range(Ts& t, uint32_t peer, uint32_t c0, uint32_t c1, int mode, uint32_t hint_k = 0) {
  int lane = lmw::lane();
  uint32_t c = c0;
  // items only hold applied elements, so the in-leaf path needs neither the peer's element base nor its end (two LDS round trips)
  // (one attempt outside any loop: a row's range mostly lies in one run of the cached leaf, and a loop around the attempt carried
  // the whole cached leaf through its phi nodes — 70 instructions per row, most of them register moves)
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_INS") && lane == 0) fprintf(stderr, "UPD mode %d hint %u peer %u [%u, %u) POS %d\n", mode, hint_k, peer, c0, c1, (int)POS);
#endif
  if (c >= c1) return;
  bool tried = true;   // the in-leaf path has just declined this very element
  if (ts_update_fast<POS>(t, peer, c, c1, mode, hint_k)) {
    PROF_CNT(t, PF_LEAF, 1);
    if (c >= c1) { if (POS && t.err == ST_POSDEL && lmw::lane() == 0) { uint32_t* w = pd_w(t); w[2] = pid_make(peer, c0); w[3] -= c0; } return; }   // (the in-leaf path leaves the counter it stopped at in word 3)
    tried = false;
  }
  uint32_t eb = t.ebase[peer];
  if (c1 > t.end[peer]) {   // a damaged target range cannot make the walk longer than the peer's history
    // a delete row whose targets lie beyond what its peer has inserted (a damaged peer table, ...): the reference deletes whatever sits
    // at the row's position (crdt_rope.rs:256-335) — not a value this engine reproduces (see ts_del_pos_ok)
    if (hint_k) { pd_stop<POS>(t, hint_k, c1 - c, pid_make(peer, c0), c - c0); return; }
    c1 = t.end[peer];
  }
  for (uint32_t guard = 0; c < c1 && !t.err && guard < (1u << 26); guard++) {
    lmw::wave_sync();
    if (!tried && ts_update_fast<POS>(t, peer, c, c1, mode, hint_k)) { PROF_CNT(t, PF_LEAF, 1); if (POS && t.err == ST_POSDEL && lmw::lane() == 0) { uint32_t* w = pd_w(t); w[2] = pid_make(peer, c0); w[3] -= c0; } continue; }
    tried = false;
    uint32_t x = pid_make(peer, c);
    uint32_t p;
    SpanRegs R;
    uint64_t hm = 0;
    if (t.cache_leaf != NONE) hm = sp_hit(t.cr, x);
    if (hm) { R = t.cr; p = t.cache_p; }
    else if (hint_k && hint_k <= t.tot_active) {
      uint32_t k = hint_k;
      lmw::wave_sync();
      if (t.cache_leaf != NONE && t.cache_pre != NONE && hint_k > t.cache_pre && hint_k - t.cache_pre <= lmw::first(t.db[t.cache_p])) { p = t.cache_p; k = hint_k - t.cache_pre; }
      else p = sd_find_kth(t, k);
      if (p != NONE) {
        lmw::wave_sync();
        uint32_t a = lmw::first(t.da[p]);
        R = sp_load(t, sa_leaf(a), sa_n(a));
        hm = sp_hit(R, x);
        if (hm) { sp_take(t, sa_leaf(a)); t.cache_p = p; t.cache_pre = hint_k - k; t.cr = R; }
      }
    }
    if (!hm) {
      if (hint_k) { pd_stop<POS>(t, hint_k, c1 - c, pid_make(peer, c0), c - c0); return; }   // the element at the row's position is not the row's target (see ts_del_pos_ok)
#ifdef LM_LOC16
      uint32_t lf = ts_loc_find(t, x);
#else
      uint32_t lf = lmw::first(t.loc[eb + c]);
#endif
      PROF_CNT(t, 15, 1);   // status updates that went through loc[] (the target was not in the cached leaf)
      if (lf >= t.n_leaf || lf == t.cache_leaf) { c++; continue; }   // not an element of this container (malformed target): ignored
      // all five arrays are requested for all 64 slots right away; the directory lookup (LDS) runs while they are in
      // flight, and the item count then masks the unused slots.  The leaf becomes the cached leaf: a delete run arrives
      // as one row per contiguous id span, and the following rows address its neighbours
      const uint32_t* rec = t.it + (uint64_t)lf * SP_REC;
      uint32_t xid = rec[lane], xln = rec[64 + lane], xol = rec[128 + lane], xor_ = rec[192 + lane], xst = rec[256 + lane];
      p = sd_find_leaf(t, lf);
      if (p == NONE) { c++; continue; }                // leaf of another container of the same document
      R.n = sa_n(lmw::first(t.da[p]));
      bool in = (uint32_t)lane < R.n;
      R.id = in ? xid : NONE; R.len = in ? xln : 0u; R.ol = in ? xol : NONE; R.orr = in ? xor_ : NONE; R.st = in ? xst : ST_FUT;
      sp_take(t, lf); t.cache_p = p; t.cache_pre = NONE; t.cr = R;
      hm = sp_hit(R, x);
      if (!hm) { c++; continue; }
    }
    if (p == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
    uint32_t slot = (uint32_t)lmw::ffs64(hm);
    uint32_t id0 = lmw::bcast(R.id, (int)slot), ln = lmw::bcast(R.len, (int)slot);
    uint32_t st0 = lmw::bcast(R.st, (int)slot), st1 = st0, orr0 = lmw::bcast(R.orr, (int)slot);
    if (mode == UPD_SET_FUT) st1 |= ST_FUT;
    else if (mode == UPD_CLR_FUT) st1 &= ~ST_FUT;
    else if (mode == UPD_DEL_INC) st1 = (st1 + ST_DEL1) | ST_EVER;
    else if (st1 & (ST_DELMASK & ~ST_DEAD)) st1 -= ST_DEL1;
    uint32_t s_off = x - id0;                                   // elements of the run before the range
    uint32_t endc = pid_ctr(id0) + ln < c1 ? pid_ctr(id0) + ln : c1;
    uint32_t tail = pid_ctr(id0) + ln - endc;                   // elements of the run beyond the range
#ifndef LM_NO_DEL_CHECK
    if (hint_k && !ts_del_pos_ok(t, R, slot, s_off, st0, hint_k)) { pd_stop<POS>(t, hint_k, c1 - c, pid_make(peer, c0), c - c0); return; }
#endif
    if (mode == UPD_DEL_INC && !(st0 & ST_EVER)) t.n_alive -= endc - c;
    lmw::wave_sync();
    uint32_t L = sa_leaf(lmw::first(t.da[p]));
    if ((s_off | tail) == 0) {
      // the whole run: one status word
      R.st = (uint32_t)lane == slot ? st1 : R.st;
      t.cr.st = R.st; t.dirty = true;   // (R is the cached leaf: either it was, or the lookup above made it so)
      sd_refresh(t, p, L, R);
    } else {
      // the run is cut at the range's ends and the affected part gets the new status: one rewrite of the leaf
      SpanItem A, B;
      uint32_t cnt, idx = slot + 1;
      if (s_off > 0) {
        R.len = (uint32_t)lane == slot ? s_off : R.len;
        A.id = x; A.len = endc - c; A.ol = x - 1; A.orr = orr0; A.st = st1;
        B.id = pid_make(peer, endc); B.len = tail; B.ol = B.id - 1; B.orr = orr0; B.st = st0;
        cnt = tail ? 2u : 1u;
      } else {
        R.len = (uint32_t)lane == slot ? endc - c : R.len; R.st = (uint32_t)lane == slot ? st1 : R.st;
        A.id = pid_make(peer, endc); A.len = tail; A.ol = A.id - 1; A.orr = orr0; A.st = st0;
        B = A;
        cnt = 1;
      }
      sp_insert_items(t, p, R, idx, A, B, cnt, false, false, t.cache_pre);
      if (t.err) return;
    }
    c = endc;
  }
}

// ---- a delete row applied BY POSITION (crdt_rope.rs:256-335: the reference never looks at a delete's target ids while it applies
// it — it removes the `len` active elements from the row's position on and remembers their ids for later retreats / forwards,
// tracker.rs:193-252).  Every writer's rows name exactly those elements, and the replay applies rows by id and only compares; a
// DAMAGED row (a flipped byte in a peer table, a target span that is off) still has a value in the reference, which this gives: the
// part of the row that matched (n_match ids from `first` on, pieces verified at the position) stays as it is, the pd_n elements
// that follow at position pd_k are deleted whatever their ids, and all pieces go into the document's list (row, id, length), which
// ts_move_ops consults instead of the row's own target span.  Cold: the row loop comes here only on ST_POSDEL.
LM_DEV void ts_del_positional(Ts& t, uint32_t row) {
  int lane = lmw::lane();
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_BASE") && lane == 0 && t.err != ST_POSDEL) fprintf(stderr, "ts_del_positional with err %d\n", (int)t.err);
#endif
  t.err = 0;
  lmw::wave_sync();
  uint32_t* w = pd_w(t);
  const uint32_t k0 = lmw::first(w[0]), first = lmw::first(w[2]), n_match = lmw::first(w[3]);
  uint32_t n = lmw::first(w[1]), n_pos = lmw::first(w[4]);
  uint32_t* pos_list = (uint32_t*)(uintptr_t)(((uint64_t)lmw::first(w[6]) << 32) | lmw::first(w[5]));
  if (!pos_list) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return; }
  const uint32_t pd_cap = lmw::first(w[7]);
  uint32_t* row_idx = (uint32_t*)(uintptr_t)(((uint64_t)lmw::first(w[9]) << 32) | lmw::first(w[8]));
  if (row_idx && lane == 0 && (n_match || n > 0)) row_idx[row] = n_pos;   // (the row's pieces follow one another from here)
  if (n_match) {
    if (n_pos >= pd_cap) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return; }
    if (lane == 0) { pos_list[3 * n_pos] = row; pos_list[3 * n_pos + 1] = first; pos_list[3 * n_pos + 2] = n_match; }
    n_pos++;
  }
  for (uint32_t guard = 0; n > 0 && !t.err && guard < (1u << 24); guard++) {
    if (k0 > t.tot_active) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return; }   // delete beyond the end: the reference's query fails
    uint32_t k = k0;
    uint32_t p = sd_find_kth(t, k);
    if (p == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
    lmw::wave_sync();
    uint32_t a = lmw::first(t.da[p]);
    SpanRegs R = sp_load(t, sa_leaf(a), sa_n(a));
    uint32_t al = sp_alen(R);
    uint32_t inc = lmw::scan_incl_add(al);
    uint64_t hit = lmw::ballot((al != 0) & (inc >= k));
    if (!hit) { LM_SETERR(t.err, ST_INTERNAL); return; }
    int slot = lmw::ffs64(hit);
    uint32_t off = k - (lmw::bcast(inc, slot) - lmw::bcast(al, slot)) - 1;   // 0-based offset of the k-th active element inside its item
    uint32_t id0 = lmw::bcast(R.id, slot), ln = lmw::bcast(R.len, slot);
    uint32_t piece = ln - off < n ? ln - off : n;
    uint32_t x = id0 + off;
    if (n_pos >= pd_cap) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return; }
    if (lane == 0) { pos_list[3 * n_pos] = row; pos_list[3 * n_pos + 1] = x; pos_list[3 * n_pos + 2] = piece; w[4] = n_pos + 1; }
    n_pos++;
#ifdef LM_EMU_TRACE
    if (getenv("LM_EMU_BASE") && lane == 0) fprintf(stderr, "POSDEL row %u: position %u, %u left, matched %u from %u:%u; piece %u:%u+%u\n", row, k0, n, n_match, first >> 24, first & 0xffffff, x >> 24, x & 0xffffff, piece);
#endif
    // the leaf becomes the cached leaf: the by-id update below then finds the item without loc[] (a plain replay may not keep it yet)
    sp_take(t, sa_leaf(a)); t.cache_p = p; t.cache_pre = k0 - k; t.cr = R;
    ts_update_range(t, pid_peer(x), pid_ctr(x), pid_ctr(x) + piece, UPD_DEL_INC);
    n -= piece;
  }
  if (lane == 0) w[4] = n_pos;
  lmw::mem_fence();
  lmw::wave_sync();
}
// the pieces of a row that was applied by position (ts_move_ops: retreat / forward of a delete row); false = not such a row
LM_DEV bool ts_move_positional(Ts& t, uint32_t row, bool whole, int mode) {
  lmw::wave_sync();
  const uint32_t* w = pd_w(t);
  const uint32_t n_pos = lmw::first(w[4]);
  if (!n_pos) return false;
  const uint32_t* pos_list = (const uint32_t*)(uintptr_t)(((uint64_t)lmw::first(w[6]) << 32) | lmw::first(w[5]));
  uint32_t lane = (uint32_t)lmw::lane();
  const uint32_t* row_idx = (const uint32_t*)(uintptr_t)(((uint64_t)lmw::first(w[9]) << 32) | lmw::first(w[8]));
  if (row_idx) {   // (a long list: the row's pieces are found through the row index and follow one another)
    uint32_t j = lmw::first(row_idx[row]);
    if (j == NONE || j >= n_pos || lmw::first(pos_list[3 * j]) != row) return false;
    if (!whole) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return true; }
    for (; j < n_pos && !t.err && lmw::first(pos_list[3 * j]) == row; j++) {
      uint32_t x = lmw::first(pos_list[3 * j + 1]), ln = lmw::first(pos_list[3 * j + 2]);
      ts_update_range(t, pid_peer(x), pid_ctr(x), pid_ctr(x) + ln, mode);
    }
    return true;
  }
  uint64_t pm = lmw::ballot(lane < n_pos && pos_list[3 * lane] == row);
  if (!pm) return false;
  // (a version that ends inside such a row would need the op-offset → piece mapping of tracker.rs:193-252; not reproduced)
  if (!whole) { LM_SETERR(t.err, ST_DATA_CORRUPTION); return true; }
  while (pm && !t.err) {
    int j = lmw::ffs64(pm);
    pm &= pm - 1;
    uint32_t x = lmw::first(pos_list[3 * j + 1]), ln = lmw::first(pos_list[3 * j + 2]);
    ts_update_range(t, pid_peer(x), pid_ctr(x), pid_ctr(x) + ln, mode);
  }
  return true;
}

// ---- op rows are read eight at a time: one 256-byte wave load per window (lane = dword of the window), and the next
// window is requested while the current one is consumed.  (A lone wave replaying configs[1] took 2 µs per op row — one
// dependent HBM round trip for the row itself — before this.)
struct RowWin { uint32_t base, lim, cur, nxt; };
LM_DEV uint32_t rw_load(const uint32_t* op_w, uint32_t base, uint32_t lim) {
  uint32_t lane = (uint32_t)lmw::lane();
  return base + (lane >> 3) < lim ? op_w[(uint64_t)base * 8 + lane] : 0u;
}
LM_DEV void rw_open(RowWin& w, const uint32_t* op_w, uint32_t first, uint32_t lim) {   // rows [first, lim) will be read in ascending order
  w.base = first; w.lim = lim;
  w.cur = rw_load(op_w, first, lim);
  w.nxt = rw_load(op_w, first + 8, lim);
}
LM_DEV OpRow rw_get(RowWin& w, const uint32_t* op_w, uint32_t row) {
  uint32_t ahead = row - w.base;   // (a row below the window wraps to a huge value)
  if (ahead >= 16) rw_open(w, op_w, row, w.lim);
  else if (ahead >= 8) { w.base += 8; w.cur = w.nxt; w.nxt = rw_load(op_w, w.base + 8, w.lim); }
  int j = (int)(row - w.base) * 8;
  OpRow r;
  r.cidx_kind = lmw::bcast(w.cur, j); r.prop = (int32_t)lmw::bcast(w.cur, j + 1); r.len = lmw::bcast(w.cur, j + 2); r.ctr = lmw::bcast(w.cur, j + 3);
  r.a0 = lmw::bcast(w.cur, j + 4); r.a1 = lmw::bcast(w.cur, j + 5); r.a2 = (int32_t)lmw::bcast(w.cur, j + 6); r.chg = lmw::bcast(w.cur, j + 7);
  return r;
}

// id of the active element at position `pos` (0-based), NONE beyond the end — MovableList moves name their source item by
// position only (tracker.rs:289-347, crdt_rope.rs:286-312); not on any hot path
LM_DEV uint32_t ts_active_id_at(Ts& t, uint32_t pos) {
  if (pos >= t.tot_active) return NONE;
  uint32_t k = pos + 1;
  uint32_t p = sd_find_kth(t, k);
  if (p == NONE) return NONE;
  lmw::wave_sync();
  uint32_t a = lmw::first(t.da[p]);
  SpanRegs R = sp_load(t, sa_leaf(a), sa_n(a));
  uint32_t al = sp_alen(R);
  uint32_t inc = lmw::scan_incl_add(al);
  uint64_t hit = lmw::ballot((al != 0) & (inc >= k));
  if (!hit) return NONE;
  int slot = lmw::ffs64(hit);
  uint32_t off = k - (lmw::bcast(inc, slot) - lmw::bcast(al, slot));   // 1..len
  return lmw::bcast(R.id, slot) + off - 1;
}

// retreat (dir < 0) / forward (dir > 0) every op of `peer` with id in [c0,c1) that belongs to container `cidx`
// (ML: the instantiation for documents that hold a MovableList — see k_integrate_span_ml below)
// SWEEP (k_integrate_span_plain_sweep, the default kernel of DF_PLAIN documents; LM_PLAIN=1 / 0 switch it off): a long range toggles the future flag of its INSERT rows'
// items in one pass over the leaves — every item of `peer` with ids inside [c0,c1) belongs to an insert row of this container —
// instead of one by-id update per row; the runs straddling c0 / c1 are cut first by two single-element updates.
template <bool ML, bool SWEEP, bool POS = false>
LM_DEV void ts_move_ops(Ts& t, const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t peer, uint32_t c0, uint32_t c1, int dir) {
  uint32_t ci = find_change(d, m, peer, c0);
  if (ci == NONE) return;
  bool swept = false;
#ifdef LM_SWEEP_EAGER   // tests: every range of three or more ids goes through the sweep (the fuzz corpora's ranges are short)
  if (SWEEP && c1 - c0 > 2) {
#else
  if (SWEEP && c1 - c0 > 8 * t.n_dir + 64) {
#endif
    int lane = lmw::lane();
    int mode = dir < 0 ? UPD_SET_FUT : UPD_CLR_FUT;
    ts_update_range(t, peer, c0, c0 + 1, mode);
    if (!t.err) ts_update_range(t, peer, c1 - 1, c1, mode);
    if (t.err) return;
    sp_flush(t);                                                     // the pass works on the leaf records in HBM
    sd_sync_cached(t);
    t.cache_leaf = NONE; t.cr.n = 255; t.cache_pre = NONE; t.dirty = false; t.loc_pend = 0;
    lmw::wave_sync();
    uint32_t lo = pid_make(peer, c0), span = c1 - c0;
    for (uint32_t q = 0; q < t.n_dir; q++) {
      uint32_t a = lmw::first(t.da[q]);
      uint32_t L = sa_leaf(a), n = sa_n(a);
      uint32_t* rec = t.it + (uint64_t)L * SP_REC;
      bool in = (uint32_t)lane < n;
      uint32_t id = in ? rec[lane] : NONE, ln = in ? rec[64 + lane] : 0u, st = in ? rec[256 + lane] : ST_FUT;
      bool hit = in & (id - lo < span);                              // whole items: the ends of the range were cut above
      if (!lmw::any(hit)) continue;
      uint32_t st1 = hit ? (dir < 0 ? (st | ST_FUT) : (st & ~ST_FUT)) : st;
      if (hit) rec[256 + lane] = st1;
      uint32_t act = lmw::reduce_add((in && st_active(st1)) ? ln : 0u);
      bool nf = lmw::ballot(in && !(st1 & ST_FUT)) != 0;
      sd_set(t, q, sa_make(L, n, nf), act);
    }
    swept = true;
  }
  uint32_t hi = d.peer_chg1[m.praw0 + peer];
  for (; ci < hi && !t.err; ci++) {
    uint32_t crow = d.chg_sorted[m.chg0 + ci];
    const ChangeRow ch = d.chg[crow];
    if (ch.ctr >= c1) break;
    if (!((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1)) continue;
    uint32_t lo = ch.op0, hr = ch.op0 + ch.n_op;
    while (lo < hr) { uint32_t mid = (lo + hr) >> 1; if (d.op[mid].ctr + d.op[mid].len <= c0) lo = mid + 1; else hr = mid; }
    RowWin w;
    rw_open(w, (const uint32_t*)d.op, lo, ch.op0 + ch.n_op);
    for (uint32_t row = lo; row < ch.op0 + ch.n_op && !t.err; row++) {
      OpRow r = rw_get(w, (const uint32_t*)d.op, row);
      if (r.ctr >= c1) break;
      if ((r.cidx_kind & 0xffff) != cidx) continue;
      uint32_t kind = (r.cidx_kind >> 16) & 0xff;
      uint32_t a = (c0 > r.ctr ? c0 : r.ctr) - r.ctr, b = (c1 < r.ctr + r.len ? c1 : r.ctr + r.len) - r.ctr;
      if (a >= b) continue;
      // a MovableList move is two halves (id_to_cursor.rs Cursor::Move): its own item, then the item it deleted, whose id the
      // first application left in the move item's payload slot.  Both go through the call sites below (one more trip of this
      // loop) — a second inlined copy of ts_update_range would grow the kernel by half
      if (SWEEP && swept && (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END)) continue;
      uint32_t mv_tgt = NONE;
      if (ML && kind == OK_LIST_MOVE) {
        lmw::wave_sync();
        mv_tgt = lmw::first((d.cp + (((uint64_t)m.elem0_hi << 32) | m.elem0_lo))[ts_g(t, pid_make(peer, r.ctr))]);
        kind = OK_LIST_INS;
      }
      for (;;) {
        if (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END) {
          ts_update_range(t, peer, r.ctr + a, r.ctr + b, dir < 0 ? UPD_SET_FUT : UPD_CLR_FUT);
        } else if (kind == OK_DEL) {
          uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
          uint32_t t0, t1;
          if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
          else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }
          if (!POS || !ts_move_positional(t, row, a == 0 && b == r.len, dir < 0 ? UPD_DEL_DEC : UPD_DEL_INC))
            ts_update_range(t, r.a0, t0, t1, dir < 0 ? UPD_DEL_DEC : UPD_DEL_INC);
        }
        if (!ML || mv_tgt == NONE || pid_peer(mv_tgt) >= m.n_peers) break;
        r.a0 = pid_peer(mv_tgt); r.a1 = pid_ctr(mv_tgt); r.a2 = 1; kind = OK_DEL; mv_tgt = NONE;   // (a, b) = (0, 1)
      }
    }
  }
}

// ---- the tracker's BASE version (batch replays only; resident trackers keep no base — a later import may be concurrent with
// anything they hold).  k_dag_b flags the node in front of which the replayed history is a CRITICAL version — every node still to
// come depends on all of it — and behind which a concurrent section begins (the reference starts its tracker at such a version
// with everything before it collapsed into one unknown span, tracker.rs:40-60 `new_with_unknown`; Eg-walker's "critical
// versions").  The tracker will never be moved below it again, so when it stands exactly there
//   * ts_convert_base: every item deleted by then becomes ST_DEAD (the counted part of the status returns to zero): one pass over
//     the status words;
//   * ts_reset_to_base: a later move BACK to exactly that version — the replay switches to a concurrent branch — undoes
//     everything applied since in ONE pass over the leaves: items of ops beyond the base become future, every counted delete is
//     dropped (all of them were applied beyond the base).  Row by row this was the retreat of a whole branch: one by-id update
//     per delete row and a sweep for the insert rows (configs[1]: ≈9 % of the kernel).
LM_DEV void ts_convert_base(Ts& t) {
  int lane = lmw::lane();
  lmw::wave_sync();
  for (uint32_t q = 0; q < t.n_dir; q++) {
    uint32_t a = lmw::first(t.da[q]);
    uint32_t L = sa_leaf(a), n = sa_n(a);
    if (L == t.cache_leaf) {
      uint32_t st = t.cr.st;
      t.cr.st = (st & ST_DELMASK) ? ((st & ~ST_DELMASK) | ST_DEAD) : st;
      t.dirty = true;
      continue;
    }
    uint32_t* rec = t.it + (uint64_t)L * SP_REC;
    if ((uint32_t)lane < n) {
      uint32_t st = rec[256 + lane];
      if (st & ST_DELMASK) rec[256 + lane] = (st & ~ST_DELMASK) | ST_DEAD;
    }
  }
}
LM_DEV void ts_reset_to_base(Ts& t, const uint32_t* s_base) {
  int lane = lmw::lane();
  sp_flush(t);                                                     // the pass works on the leaf records in HBM
  sd_sync_cached(t);
  t.cache_leaf = NONE; t.cr.n = 255; t.cache_pre = NONE; t.dirty = false; t.loc_pend = 0;
  lmw::wave_sync();
  for (uint32_t q = 0; q < t.n_dir; q++) {
    uint32_t a = lmw::first(t.da[q]);
    uint32_t L = sa_leaf(a), n = sa_n(a);
    uint32_t* rec = t.it + (uint64_t)L * SP_REC;
    bool in = (uint32_t)lane < n;
    uint32_t id = in ? rec[lane] : 0u, ln = in ? rec[64 + lane] : 0u, st = in ? rec[256 + lane] : ST_FUT;
    bool post = in && pid_ctr(id) >= s_base[pid_peer(id)];
    uint32_t st1 = (st & ~(ST_DELMASK & ~ST_DEAD)) | (post ? ST_FUT : 0u);
    if (!lmw::any(in && st1 != st)) continue;
    if (in && st1 != st) rec[256 + lane] = st1;
    uint32_t act = lmw::reduce_add((in && st_active(st1)) ? ln : 0u);
    bool nf = lmw::ballot(in && !(st1 & ST_FUT)) != 0;
    sd_set(t, q, sa_make(L, n, nf), act);
  }
}
// The tracker moves from s_cur to the version vv (the dependencies of the node about to be replayed): Tracker::checkout
// (tracker.rs:354-461), peer by peer — or in one pass when vv is the tracker's base (above).  `conv`: k_dag_b flagged this node.
// loc[] of every item the tracker holds, in one pass over its leaves.  A plain document's replay starts WITHOUT loc[]: while
// its history is one chain nothing is ever looked up by id — delete rows carry their position, there is no future item for a
// sibling scan to resolve, no op to retreat — so the flushes of the cached leaf write no loc[] entry (configs[1]: the 50k-op
// base, 60 % of its rows; a document imported sequentially never pays for loc[] at all).  The first move of the tracker
// (ts_goto) brings it up to date; from then on it is maintained as always.
LM_DEV void ts_build_loc(Ts& t, uint32_t* loc_real) {
  int lane = lmw::lane();
  t.loc = loc_real;
  lmw::wave_sync();
  for (uint32_t q = 0; q < t.n_dir; q++) {
    uint32_t a = lmw::first(t.da[q]);
    uint32_t L = sa_leaf(a), n = sa_n(a);
    if (L == t.cache_leaf) { t.loc_pend = (uint32_t)lane < t.cr.n ? 1u : 0u; continue; }   // (written with the leaf's next flush)
    const uint32_t* rec = t.it + (uint64_t)L * SP_REC;
    bool in = (uint32_t)lane < n;
    SpanRegs R;
    R.n = n; R.id = in ? rec[lane] : NONE; R.len = in ? rec[64 + lane] : 0u; R.ol = NONE; R.orr = NONE; R.st = 0;
    sp_set_loc_lanes(t, R, in, L);
  }
}
// ---- the tracker moves to ANY version in one pass over the document's delete rows and one over its leaves (resident documents,
// Tracker::checkout / forward, tracker.rs:354-546).  ts_move_ops undoes / redoes delete rows one by one — a by-id lookup, a leaf
// switch and a directory update per row: ≈170 ms for a move across a sixteenth of a 1M-op document's history, 17 % of a configs[1]
// import run (the retreat of the branch the import is concurrent with).  An element's status at a version V is a function of V
// alone: future unless its insert lies in V, deleted as many times as delete ops of V target it.  So
//   1. items that straddle the moved ends of V are cut there (two by-id updates per moved peer — as the insert sweep does);
//   2. dcnt[] (a word per element slot of the document) := 0, then every delete row of the container whose op lies in V adds one
//      to each of its targets — lane = row, 64 rows per step, global atomics (two rows may target one element);
//   3. every leaf: status := (kept bits) | future? | count — the count is read at the item's first element: an item never straddles
//      a delete's range, every delete of V has been applied to this tracker before and cut it there (items are never merged).
// target(p) = ov for p == ov_peer, vv[p] otherwise (the call sites' override for the replayed node's own peer).
LM_DEV uint32_t ts_vs_target(const uint32_t* vv, uint32_t ov_peer, uint32_t ov, uint32_t p) { return p == ov_peer ? ov : vv[p]; }
LM_DEV void ts_sweep_version(Ts& t, const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t P, const uint32_t* vv, uint32_t ov_peer, uint32_t ov,
                             const uint32_t* s_cur, uint32_t* dcnt) {
  int lane = lmw::lane();
  // 1. cuts
  for (uint32_t p = 0; p < P && !t.err; p++) {
    uint32_t cur = s_cur[p], tgt = ts_vs_target(vv, ov_peer, ov, p);
    if (cur == tgt) continue;
    uint32_t lo = cur < tgt ? cur : tgt, hi = cur < tgt ? tgt : cur;
    int mode = cur > tgt ? UPD_SET_FUT : UPD_CLR_FUT;
    ts_update_range(t, p, lo, lo + 1, mode);
    if (!t.err && hi - lo > 1) ts_update_range(t, p, hi - 1, hi, mode);
    // a version may end INSIDE a delete row (a checkout between two atoms of a run of backspaces): only a part of the row's
    // targets is deleted there — the item is cut where that part ends (its status is whatever pass 3 decides)
    uint32_t ci = t.err ? NONE : find_change(d, m, p, tgt);
    if (ci != NONE) {
      uint32_t crow = d.chg_sorted[m.chg0 + ci];
      const ChangeRow ch = d.chg[crow];
      if (ch.ctr < tgt && ((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1)) {
        uint32_t rl = ch.op0, rh = ch.op0 + ch.n_op;
        while (rl < rh) { uint32_t mid = (rl + rh) >> 1; if (d.op[mid].ctr + d.op[mid].len <= tgt) rl = mid + 1; else rh = mid; }
        if (rl < ch.op0 + ch.n_op) {
          const OpRow r = d.op[rl];
          if (r.ctr < tgt && (r.cidx_kind & 0xffff) == cidx && ((r.cidx_kind >> 16) & 0xff) == OK_DEL && r.a0 < P) {
            uint32_t b = tgt - r.ctr, Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
            uint32_t x = r.a2 > 0 ? r.a1 + b : r.a1 + (Ln - b);   // the element whose left edge is the end of the deleted part
            if (b < Ln && x < t.end[r.a0]) ts_update_range(t, r.a0, x, x + 1, UPD_SET_FUT);
          }
        }
      }
    }
  }
  if (t.err) return;
  sp_flush(t);                                                     // the passes work on the leaf records in HBM
  sd_sync_cached(t);
  t.cache_leaf = NONE; t.cr.n = 255; t.cache_pre = NONE; t.dirty = false; t.loc_pend = 0;
  lmw::wave_sync();
  // 2. delete counts at the target version
  {
    struct alignas(16) U4 { uint32_t x, y, z, w; };
    U4* c4 = (U4*)dcnt;
    const U4 zero4 = {0, 0, 0, 0};
    for (uint32_t i = (uint32_t)lane; i < (m.atoms + 3) / 4; i += 64) c4[i] = zero4;
  }
  lmw::mem_fence();
  lmw::wave_sync();
  for (uint32_t i0 = 0; i0 < m.n_op; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    uint32_t g0 = 0, n = 0;
    if (i < m.n_op) {
      const OpRow& r = d.op[m.op0 + i];
      uint32_t ck = r.cidx_kind;
      if ((ck & 0xffff) == cidx && ((ck >> 16) & 0xff) == OK_DEL && (d.chg_flag[r.chg] & 1u)) {
        const ChangeRow& ch = d.chg[r.chg];
        uint32_t peer = ch.peer;
        uint32_t lo = ch.ctr + d.chg_skip[r.chg], hi = ts_vs_target(vv, ov_peer, ov, peer);
        if (lo < r.ctr) lo = r.ctr;
        if (hi > r.ctr + r.len) hi = r.ctr + r.len;
        if (lo < hi && r.a0 < P) {
          uint32_t a = lo - r.ctr, b = hi - r.ctr;
          uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
          uint32_t t0, t1;
          if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
          else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }
          uint32_t e = t.end[r.a0];
          if (t1 > e) t1 = e;                                      // (a damaged target range: what ts_update_range clamps)
          if (t0 < t1 && b <= Ln) { g0 = t.ebase[r.a0] + t0; n = t1 - t0; }
        }
      }
    }
    for (uint32_t k = 0; k < n && k < 16; k++) lmw::atomic_add(&dcnt[g0 + k], 1u);
    // (a long range — a selection deleted at once — is counted by the whole wave)
    uint64_t lm_ = lmw::ballot(n > 16);
    while (lm_) {
      int src = lmw::ffs64(lm_);
      lm_ &= lm_ - 1;
      uint32_t bg = lmw::bcast(g0, src), bn = lmw::bcast(n, src);
      for (uint32_t k = 16 + (uint32_t)lane; k < bn; k += 64) lmw::atomic_add(&dcnt[bg + k], 1u);
    }
  }
  lmw::mem_fence();
  lmw::wave_sync();
  // 3. the leaves
  for (uint32_t q = 0; q < t.n_dir; q++) {
    uint32_t a = lmw::first(t.da[q]);
    uint32_t L = sa_leaf(a), n = sa_n(a);
    uint32_t* rec = t.it + (uint64_t)L * SP_REC;
    bool in = (uint32_t)lane < n;
    uint32_t id = in ? rec[lane] : 0u, ln = in ? rec[64 + lane] : 0u, st = in ? rec[256 + lane] : ST_FUT;
    uint32_t st1 = st;
    bool over = false, ragged = false;
    if (in) {
      uint32_t peer = pid_peer(id), ctr = pid_ctr(id);
      uint32_t cnt = dcnt[t.ebase[peer] + ctr];
      over = cnt > 0x7fffu;                                        // (the status word's counter; bit 23 is ST_DEAD)
      if (over) cnt = 0x7fffu;
      st1 = (st & ~(ST_FUT | ST_DELMASK)) | (ctr >= ts_vs_target(vv, ov_peer, ov, peer) ? ST_FUT : 0u) | (cnt << 8) | (cnt ? ST_EVER : 0u);
#ifdef LM_EMU_CHECK
      for (uint32_t k = 1; k < ln; k++)
        if (dcnt[t.ebase[peer] + ctr + k] != cnt || ((ctr + k >= ts_vs_target(vv, ov_peer, ov, peer)) != ((st1 & ST_FUT) != 0))) {
          fprintf(stderr, "CHECK version sweep: item %u:%u+%u is not uniform at element %u (count %u vs %u)\n", peer, ctr, ln, k, dcnt[t.ebase[peer] + ctr + k], cnt);
          ragged = true;
        }
#endif
    }
    if (lmw::any(over)) { LM_SETERR(t.err, ST_UNSUPPORTED); return; }
    if (lmw::any(ragged)) { LM_SETERR(t.err, ST_INTERNAL); return; }
    if (!lmw::any(in && st1 != st)) continue;
    if (in && st1 != st) rec[256 + lane] = st1;
    uint32_t act = lmw::reduce_add((in && st_active(st1)) ? ln : 0u);
    bool nf = lmw::ballot(in && !(st1 & ST_FUT)) != 0;
    sd_set(t, q, sa_make(L, n, nf), act);
  }
}
// is the move s_cur → target long enough for the pass?  (its cost: the document's op rows / 64 + its element slots / 256 + its
// leaves; a row-by-row move costs ≈ a by-id update per row of the moved range)
LM_DEV bool ts_sweep_pays(const Ts& t, const DocMeta& m, uint32_t P, const uint32_t* vv, uint32_t ov_peer, uint32_t ov, const uint32_t* s_cur) {
  uint32_t dist = 0;
  for (uint32_t p = (uint32_t)lmw::lane(); p < P; p += 64) { uint32_t c = s_cur[p], g = ts_vs_target(vv, ov_peer, ov, p); dist += c > g ? c - g : g - c; }
  dist = lmw::reduce_add(dist);
#ifdef LM_SWEEP_EAGER   // tests: every move takes the pass
  return dist > 0;
#else
  return dist > 256 && (uint64_t)dist * 16 > (uint64_t)m.atoms / 32 + m.n_op + 40ull * t.n_dir + 2048;
#endif
}
// … and for a BATCH replay (the common kernel: documents with style anchors, small documents of several peers that sync every few
// dozen ops — configs[3]): the pass costs ≈ 1.25 instructions per op row of the document + 60 per leaf + ≈600 per moved peer (the
// two cuts) + ≈500; row by row a move costs ≈300 per ROW — rows ≈ moved ids x rows per id of the document.  configs[3] (≈1,000 op
// rows, ≈3 ids per row): a move across 100 ids is ≈35 rows ≈ 10k instructions against ≈4k; the cost model had 74 % of that
// config's integrate stage in ts_move_ops.
LM_DEV bool ts_sweep_pays_batch(const Ts& t, const Dev& d, const DocMeta& m, uint32_t P, const uint32_t* vv, const uint32_t* s_cur) {
  uint32_t dist = 0, np = 0;
  for (uint32_t p = (uint32_t)lmw::lane(); p < P; p += 64) { uint32_t c = s_cur[p], g = vv[p]; dist += c > g ? c - g : g - c; np += c != g ? 1u : 0u; }
  dist = lmw::reduce_add(dist);
  np = lmw::reduce_add(np);
#ifdef LM_SWEEP_EAGER   // tests: every move takes the pass
  return dist > 0;
#else
  const uint64_t rows = (uint64_t)dist * m.n_op / (m.atoms ? m.atoms : 1u);
  return dist > 16 && rows * d.vs_row_cost > 5ull * m.n_op / 4 + m.atoms / 40 + 60ull * t.n_dir + 600ull * np + 500;
#endif
}
#ifndef LM_BATCH_VSWEEP
#define LM_BATCH_VSWEEP 0   // 1: the common batch kernel moves its tracker by version passes when ts_sweep_pays_batch says so — measured on configs[3] (profiles/r04_batch_version_sweep.log): -9 % of the kernel at best, a slower step (52 B of scratch instead of 24); kept for the test build
#endif
template <bool ML, bool SWEEP, bool VS, bool POS = false>
LM_DEV void ts_goto(Ts& t, const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t P, const uint32_t* vv, uint32_t* s_cur, uint32_t* s_base,
                    bool& base_on, bool conv, uint32_t* loc_real, uint32_t* dcnt_real) {
  int lane = lmw::lane();
  if (!t.loc) {
    bool mv = false;
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) mv |= s_cur[p] != vv[p];
    if (conv || lmw::any(mv)) ts_build_loc(t, loc_real);
  }
  bool reset = false;
  if (base_on) {
    bool same = true;
    uint32_t back = 0;
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) { uint32_t v = vv[p], c = s_cur[p]; same &= v == s_base[p]; back += c > v ? c - v : 0u; }
    back = lmw::reduce_add(back);
#ifdef LM_SWEEP_EAGER   // tests: every move back to the base takes the pass
    reset = !lmw::any(!same) && back > 0;
#else
    reset = !lmw::any(!same) && back > 4 * t.n_dir + 32;   // (a short way back is cheaper row by row than a pass over every leaf)
#endif
  }
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_BASE") && lane == 0) fprintf(stderr, "BASE cont %u: %s%s (n_dir %u)\n", cidx, reset ? "reset" : "move", conv ? " +convert" : "", t.n_dir);
#endif
  if (reset) ts_reset_to_base(t, s_base);
  // (VS: the common batch kernel only — the plain instantiations move by leaf sweeps and base resets and stay as they are.  Not
  // beside a base: the sticky ST_DEAD bits of the base and the counts of the pass would both hold the deletes below it)
  else if (VS && !base_on && !conv && dcnt_real && (lmw::block_sync(), ts_sweep_pays_batch(t, d, m, P, vv, s_cur))) ts_sweep_version(t, d, m, cidx, P, vv, NONE, 0, s_cur, dcnt_real);
  else
    for (uint32_t p = 0; p < P && !t.err; p++) {
      uint32_t cur = s_cur[p], tgt = vv[p];
      if (cur > tgt) ts_move_ops<ML, SWEEP, POS>(t, d, m, cidx, p, tgt, cur, -1);
      else if (cur < tgt) ts_move_ops<ML, SWEEP, POS>(t, d, m, cidx, p, cur, tgt, +1);
    }
  lmw::block_sync();
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_cur[p] = vv[p]; if (conv) s_base[p] = vv[p]; }
  if (conv && !t.err) { ts_convert_base(t); base_on = true; }
  lmw::block_sync();
}

#ifdef LM_EMU_CHECK
// debug-only (kernel-logic harness built with -DLM_EMU_CHECK): the directory against the leaves and loc[] against both
inline bool ts_check(Ts& t, const char* what, uint32_t row) {
  bool ok = true;
  sp_flush(t);   // the checker reads HBM
  lmw::wave_sync();
  if (lmw::lane() == 0) {
    uint32_t tot = 0;
    for (uint32_t q = 0; q < t.n_dir && ok; q++) {
      uint32_t a = t.da[q], L = sa_leaf(a), n = sa_n(a), act = 0;
      bool nf = false;
      const uint32_t* rec = t.it + (uint64_t)L * SP_REC;
      for (uint32_t i = 0; i < n; i++) {
        uint32_t id0 = rec[i], ln = rec[64 + i], st = rec[256 + i];
        if (ln == 0) { fprintf(stderr, "CHECK %s row=%u: dir[%u] leaf %u item %u has length 0\n", what, row, q, L, i); ok = false; }
        if (st_active(st)) act += ln;
        nf |= !(st & ST_FUT);
        for (uint32_t k = 0; k < ln && t.loc; k++) {
#ifdef LM_LOC16
          uint32_t expect = (k == 0 || ((id0 + k) & (LOC_W - 1)) == 0) ? L : NONE;   // kept entries only: heads and multiples of LOC_W
#else
          uint32_t expect = L;
#endif
          if (t.loc[ts_g(t, id0 + k)] != expect) { fprintf(stderr, "CHECK %s row=%u: loc of %u:%u is %u, item lives in leaf %u\n", what, row, id0 >> 24, (id0 & 0xffffff) + k, t.loc[ts_g(t, id0 + k)], L); ok = false; break; }
        }
      }
      if (act != t.db[q] || nf != sa_nf(a)) { fprintf(stderr, "CHECK %s row=%u: dir[%u] leaf %u active %u (cached %u) nf %d (cached %d)\n", what, row, q, L, act, t.db[q], (int)nf, (int)sa_nf(a)); ok = false; }
      tot += act;
    }
    if (ok && t.ds_on)   // the block sums, except the cached leaf's block (re-summed when the leaf leaves the cache / before a search)
      for (uint32_t j = 0; (j << SD_BSH) < t.n_dir && ok; j++) {
        if (t.cache_leaf != NONE && j == (t.cache_p >> SD_BSH)) continue;
        uint32_t sum = 0;
        for (uint32_t q = j << SD_BSH; q < ((j + 1) << SD_BSH) && q < t.n_dir; q++) sum += t.db[q];
        if (sum != t.ds[j]) { fprintf(stderr, "CHECK %s row=%u: block %u sums to %u, cached sum %u\n", what, row, j, sum, t.ds[j]); ok = false; }
      }
    if (ok && tot != t.tot_active) { fprintf(stderr, "CHECK %s row=%u: total active %u, cached %u\n", what, row, tot, t.tot_active); ok = false; }
    if (ok && t.cache_leaf != NONE) {
      if (t.cache_p >= t.n_dir || sa_leaf(t.da[t.cache_p]) != t.cache_leaf) { fprintf(stderr, "CHECK %s row=%u: cached leaf %u is not at directory position %u\n", what, row, t.cache_leaf, t.cache_p); ok = false; }
      uint32_t pre = 0;
      for (uint32_t q = 0; q < t.cache_p && q < t.n_dir; q++) pre += t.db[q];
      if (ok && t.cache_pre != NONE && pre != t.cache_pre) { fprintf(stderr, "CHECK %s row=%u: %u active elements before the cached leaf, cached prefix %u\n", what, row, pre, t.cache_pre); ok = false; }
    }
  }
  return lmw::any(!ok) ? false : true;
}
#define TS_CHECK(what, row) do { if (!t.err && !ts_check(t, what, row)) t.err = ST_INTERNAL; } while (0)
#else
#define TS_CHECK(what, row) do {} while (0)
#endif

}  // namespace lm
#include "lm_k_integrate_linear.h"
namespace lm {

// ---- resident documents (DiffCalculatorRetainMode::Persist, diff_calc.rs:62-68,1371-1376; Tracker::checkout /
// forward, tracker.rs:350-546).  A context in resident mode keeps every document's trackers in HBM between runs: the leaf pool,
// each sequence container's leaf directory (both words), the tracker's version per (container, peer) and what it has applied
// per peer.  The next run — more blobs imported, or another checkout — starts from that state: it applies only the changes
// beyond the applied version (each after moving the tracker to the change's dependencies, as in the replay from the empty
// version) and finally moves the tracker to the version being rendered; visibility is then "active" (not future, delete count
// 0), not "never deleted".  Element ids are packed with the document's peer INDEX, which follows PeerID order: when a new
// peer sorts in front of an old one the stored leaves are renumbered first; loc[] is rebuilt from the leaves in every run.
struct ResDoc {            // one per document, written by the host for every run (lm_pipeline.h)
  uint64_t tk_off;         // first word of the document's record in DevRes::tk
  uint32_t pcap, ccap;     // peers / containers the record has room for
  uint32_t reset;          // 1: the stored tracker is not to be used (first run, capacity grown, a failed run before): replay from the empty version
  uint32_t elem_cap;       // element slots of the document's slice of the element arena (cp[] / loc[])
};
struct DevRes {
  const ResDoc* doc;
  uint32_t* tk;                  // per-document tracker records (tk_* layout below)
  const uint32_t* dir_a_prev;    // leaf directories as the previous run left them: word A / word B, [leaf0 + i]
  const uint32_t* dir_b_prev;
  uint32_t* dir_b;               // this run's word B (word A goes to Dev::dir_out, which the emit stage reads)
};
// record: [0] tracker valid  [1] peers  [2] containers  [3] leaves used  [4,5] first element slot of the document  [6] element slots cleared so far  [7] != 0: that run left changes pending
//         [8] element slots handed out in the document's slice (bases + capacities below)  [9..11] reserved
//         peers × (PeerID lo, hi) | peers × applied end | peers × element base | peers × element capacity | containers × (TK_CW + pcap words)
static constexpr uint32_t TK_HDR = 12, TK_CW = 8;   // container record: root0, n_dir, n_alive, exists (sticky, k_res_exists), 4 reserved, then cur[pcap]
LM_DEV uint32_t tk_words(uint32_t pcap, uint32_t ccap) { return TK_HDR + 5 * pcap + ccap * (TK_CW + pcap); }
LM_DEV uint32_t* tk_peers(uint32_t* tk) { return tk + TK_HDR; }
LM_DEV uint32_t* tk_applied(uint32_t* tk, uint32_t pcap) { return tk + TK_HDR + 2 * pcap; }
LM_DEV uint32_t* tk_ebase(uint32_t* tk, uint32_t pcap) { return tk + TK_HDR + 3 * pcap; }
LM_DEV uint32_t* tk_ecap(uint32_t* tk, uint32_t pcap) { return tk + TK_HDR + 4 * pcap; }
LM_DEV uint32_t* tk_cont(uint32_t* tk, uint32_t pcap, uint32_t c) { return tk + TK_HDR + 5 * pcap + c * (TK_CW + pcap); }

// K9 (span-granular): one wave per document.  Dynamic LDS: [dir_cap] word A, [dir_cap] word B, then 3 × pmax.
// Two kernels share this body.  ML = false (k_integrate_span) is the replay of Text / List containers and takes the documents
// WITHOUT a MovableList; ML = true (k_integrate_span_ml) also knows the move rows and takes the documents WITH one (DF_MOVABLE);
// it is launched only for batches that hold such a document.  The split keeps the move handling out of the scalar-issue-bound
// loop of the common kernel: folded into one kernel — a kind test per row plus one more trip through the insert / delete call
// sites for a move — it cost configs[1] 13 % of the kernel's time (15.3 → 17.3 ms per 5,000-document launch) for rows that
// never occur there.
#ifndef LM_INTEGRATE_WAVES
#define LM_INTEGRATE_WAVES 5
#endif
#if defined(LM_LOC_FULL) && !defined(LM_LAZY_LOC)
#define LM_LAZY_LOC 0     // (the per-element layout reads loc[] without going through ts_loc_find)
#endif
#ifndef LM_LAZY_LOC
#define LM_LAZY_LOC 1     // 0: loc[] is kept from the first op on (rounds 1-3; A/B builds)
#endif
#ifndef LM_BASE_RESET
#define LM_BASE_RESET 1   // 0: no base version — every move of the tracker goes row by row / by the insert sweep (rounds 1-3; A/B builds)
#endif
// PLAIN = true (k_integrate_span_plain_sweep by default, k_integrate_span_plain under LM_PLAIN=1; LM_PLAIN=0 = common kernel): the
// documents flagged DF_PLAIN — no checkout, no sliced change, no style anchor — whose row loop needs neither the slicing of a row
// against the known prefix / the rendered version nor the style branches.
// POS (k_integrate_span_pos): the replay of the documents an earlier launch left with ST_POSDEL — a delete row that does not match
// its position — with ts_del_positional / ts_move_positional compiled in.  A kernel of its own: inlined into the other
// instantiations, the two routines' copies of ts_update_range grew every kernel by a sixth and cost the replay of HEALTHY documents
// 15 % through the instruction cache (profiles/r05_posdel_ab.log: 7.8 -> 9.0 ms per 5,000 configs[1] documents).
template <bool ML, bool PLAIN, bool SWEEP = false, bool RES = false, bool FUSE = false, bool POS = false>
LM_DEV void integrate_span_body(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count, DevRes rs = DevRes{}) {
  uint32_t doc = d.doc_order[(uint32_t)lmw::bid()];
  int lane = lmw::lane();
  LM_DYN_SHARED(uint32_t, s_mem);
  uint32_t* s_da = s_mem;
  uint32_t* s_db = s_mem + dir_cap;
  uint32_t* s_ds = s_db + dir_cap;                 // block sums
  uint32_t* s_ebase = s_ds + ((dir_cap >> SD_BSH) + 2) + PD_LDS;   // (PD_LDS words of positional-delete bookkeeping in front of the peer tables)
  uint32_t* s_cur = s_ebase + pmax;
  uint32_t* s_end = s_cur + pmax;
  uint32_t* s_tgt = s_end + pmax;    // RES only: the version being rendered (s_end is the latest applied version there)
  uint32_t* s_app = s_tgt + pmax;    // RES only: what the stored tracker has applied, per peer
  uint32_t* s_pmap = s_app + pmax;   // RES only: peer index of the stored tracker → peer index of this run
  DocMeta m = d.doc[doc];
  if (RES) {
    // the document's record is needed again when the tracker record is written at the very end: as wave-uniform scalars it
    // costs SGPRs (spilled to VGPR lanes at worst), as loaded vector values it cost 70+ bytes of scratch per lane
    uint32_t* mw = (uint32_t*)&m;
    for (uint32_t i = 0; i < sizeof(DocMeta) / 4; i++) mw[i] = lmw::first(mw[i]);
  }
  uint64_t elem0 = ((uint64_t)m.elem0_hi << 32) | m.elem0_lo;
  {
    uint32_t fl = m.flags;
    if (RES && (fl & DF_PLAIN) && d.front_off[doc + 1] > d.front_off[doc]) fl &= ~DF_PLAIN;   // resident, rendered at a checked-out version: the general instantiation's
    // (POS: plain documents — fused ones too, row by row — are k_integrate_span_pos_plain's, the others k_integrate_span_pos's)
    if (POS ? ((fl & DF_MOVABLE) != 0 || ((fl & DF_PLAIN) != 0) != PLAIN) : (fl & (DF_MOVABLE | DF_PLAIN)) != ((ML ? DF_MOVABLE : 0u) | (PLAIN ? DF_PLAIN : 0u))) return;   // another kernel's document
    if (PLAIN && !RES && !POS && ((fl & DF_FUSED) != 0 && d.fuse != nullptr) != FUSE) return;                      // (k_integrate_span_plain_fuse's / the plain kernel's)
  }
  (void)SWEEP;
  if (retry_pass && m.status != (POS ? ST_POSDEL : ST_RETRY)) return;
  if (status_fatal(m.status) && !retry_pass) return;
  // RES: the element layout is the stored tracker's (k_res_layout) — loc[] is kept, only the slots new to this run are cleared
  const bool keep_loc = RES && !ML && !retry_pass && (m.flags & DF_LAYOUT_SAME) != 0;
  {   // loc[] of the document := NONE, four entries per store (the slice is 16-byte aligned and padded to a multiple of four)
    struct alignas(16) U4 { uint32_t x, y, z, w; };
    U4* l4 = (U4*)(d.loc + elem0);
    const U4 none4 = {NONE, NONE, NONE, NONE};
    uint32_t from4 = 0;
    if (keep_loc) from4 = lmw::first((rs.tk + rs.doc[doc].tk_off)[6]) / 4;   // (slices are padded to multiples of four: the boundary group is cleared again only if it was never used — it holds no kept entry, see k_res_layout)
    if (RES || !d.loc_cleared || retry_pass)
      for (uint32_t i = from4 + (uint32_t)lane; i < (m.atoms + 3) / 4; i += 64) l4[i] = none4;
  }
  if (retry_pass) {
    for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) {   // sequence containers only: a Map's flag belongs to k_map_lww
      uint32_t ck = d.cont[m.cid0 + c].kind_root & 0xff;
      if (ck == CK_TEXT || ck == CK_LIST || (ML && ck == CK_MOVABLE)) d.cont[m.cid0 + c].touched = 0;
    }
    lmw::block_sync();
    if (lane == 0) d.doc[doc].status = ST_OK;
    m.status = ST_OK;
    lmw::block_sync();
  }
  if (status_fatal(m.status)) return;
  uint32_t P = m.n_peers;
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_ebase[p] = d.elem_base[m.praw0 + p]; s_end[p] = d.peer_end[m.praw0 + p]; }
  // (POS: the peers' ends in the document's BASE version — a document staged on a snapshot's state; ids below it are content the tracker
  // holds only under the synthetic peer's ids: a delete row that names them is applied by position straight away, see the row loop)
  if (POS && !RES) for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_app[p] = d.peer_base[m.praw0 + p];
  // RES: the stored tracker of this document, if it can be used
  uint32_t* tk = nullptr;
  uint32_t tk_pcap = 0, P0 = 0, C0 = 0;
  bool fresh = true, renumber = false;
  // RES: rendered at a checked-out version (the frontiers were accepted): the tracker ends there and "active" is what shows
  const bool to_version = RES && !PLAIN && d.front_off[doc + 1] > d.front_off[doc] && !(m.flags & DF_FRONT_ERR);
  if (RES) {
    // (the record is read with vector loads: its words are made wave-uniform scalars explicitly, or the record pointer and the
    // capacities live in VGPRs for the whole kernel)
    ResDoc rd;
    {
      const uint32_t* rw = (const uint32_t*)(rs.doc + doc);
      rd.tk_off = ((uint64_t)lmw::first(rw[1]) << 32) | lmw::first(rw[0]);
      rd.pcap = lmw::first(rw[2]); rd.ccap = lmw::first(rw[3]); rd.reset = lmw::first(rw[4]); rd.elem_cap = lmw::first(rw[5]);
    }
    tk = rs.tk + rd.tk_off;
    tk_pcap = rd.pcap;
    if (P > pmax || P > rd.pcap || m.n_cont > rd.ccap) { if (lane == 0) { LM_SETERR(d.doc[doc].status, ST_INTERNAL); tk[0] = 0; } return; }   // (the host sizes the record)
    // every applied op is replayed; the version being rendered only decides where the tracker is left (peer_end_all = the
    // applied end at the latest version, written by k_dag_b for every resident document)
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_tgt[p] = d.peer_end[m.praw0 + p]; s_end[p] = d.peer_end_all[m.praw0 + p]; s_app[p] = 0; }
    lmw::block_sync();
    P0 = lmw::first(tk[1]); C0 = lmw::first(tk[2]);
    fresh = rd.reset != 0 || retry_pass != 0 || lmw::first(tk[0]) != 1u || P0 > P || C0 > m.n_cont || P0 > rd.pcap;
    if (!fresh) {
      bool lost = false;
      for (uint32_t q = (uint32_t)lane; q < P0; q += 64) {
        uint64_t id = ((uint64_t)tk_peers(tk)[2 * q + 1] << 32) | tk_peers(tk)[2 * q];
        uint32_t lo = 0, hi = P;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < id) lo = mid + 1; else hi = mid; }
        bool ok = lo < P && d.peer_uniq[m.praw0 + lo] == id;
        s_pmap[q] = ok ? lo : 0u;
        lost |= !ok;
        renumber |= ok && lo != q;
        // a MovableList move keeps the id of the item it deleted in its element slot (cp[]): the stored tracker is only usable
        // while the element layout is the one it was written under
        if (ML && ok && tk_ebase(tk, tk_pcap)[q] != d.elem_base[m.praw0 + lo]) lost = true;
      }
      if (ML && (lmw::first(tk[4]) != m.elem0_lo || lmw::first(tk[5]) != m.elem0_hi)) lost = true;   // (the document's element slice moved)
      fresh = lmw::any(lost);
      renumber = lmw::any(renumber);
      lmw::block_sync();
      if (!fresh) for (uint32_t q = (uint32_t)lane; q < P0; q += 64) s_app[s_pmap[q]] = tk_applied(tk, tk_pcap)[q];
    }
    if (lane == 0) tk[0] = 0;   // the record is being rewritten: valid again only when this run completes
    if (fresh && !retry_pass && lane == 0) lmw::atomic_add(retry_count + 1, 1u);   // diagnostics: documents replayed from the empty version
  }
  lmw::block_sync();
  uint64_t vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
  Ts t;
  t.it = d.it + (uint64_t)m.leaf0 * SP_REC;
  t.loc = d.loc + elem0;
  t.ebase = s_ebase; t.cur = s_cur; t.end = s_end; t.da = s_da; t.db = s_db; t.ds = s_ds; t.ds_on = false;
  t.dir_cap = dir_cap; t.leaf_cap = m.leaf_cap; t.n_leaf = 0; t.err = 0; t.beyond = 0;
  t.n_alive = 0;
  if (lane == 0) {   // (resident trackers keep the by-id verdict — LM_DATA_CORRUPTION: their lists would have to outlive the run)
    uint64_t pl = 0, ri = 0, pcap = PD_CAP;
    if (POS && d.posdel) {
      const uint64_t p0 = d.posdel_off ? d.posdel_off[doc] : (uint64_t)doc * PD_CAP;
      if (d.posdel_off) pcap = d.posdel_off[doc + 1] - p0;
      pl = (uint64_t)(uintptr_t)(d.posdel + p0 * 3);
      if (d.pd_row_idx && pcap > PD_CAP) ri = (uint64_t)(uintptr_t)d.pd_row_idx;
    }
    uint32_t* w = s_ebase - PD_LDS;
    w[0] = w[1] = w[2] = w[3] = w[4] = 0; w[5] = (uint32_t)pl; w[6] = (uint32_t)(pl >> 32);
    w[7] = (uint32_t)(pcap > 0xffffffffull ? 0xffffffffull : pcap); w[8] = (uint32_t)ri; w[9] = (uint32_t)(ri >> 32); w[10] = w[11] = 0;
  }
#ifdef LM_PROF
  for (int i = 0; i < PF_N; i++) t.prof[i] = 0;
  uint64_t pf_begin = lmw::clock();
#endif
  uint32_t dir_used = 0;
  if (m.leaf_cap > MAX_LEAVES_PER_DOC || (retry_pass && retry_pass != 3u && m.leaf_cap > dir_cap) || P > pmax) { if (lane == 0) LM_SETERR(d.doc[doc].status, ST_UNSUPPORTED); return; }
  if (RES && !fresh) {
    // the stored leaves: renumbered when the peer order changed, and loc[] (cleared above) written again for every item —
    // unless the layout did not move (keep_loc: then nothing was cleared and nothing is renumbered)
    t.n_leaf = lmw::first(tk[3]);
    if (t.n_leaf > t.leaf_cap) { fresh = true; t.n_leaf = 0; }
    for (uint32_t c = 0; c < C0 && !fresh && (!keep_loc || renumber); c++) {
      const uint32_t* rec = tk_cont(tk, tk_pcap, c);
      uint32_t r0 = lmw::first(rec[0]), nr = lmw::first(rec[1]);
      for (uint32_t q = 0; q < nr; q++) {
        uint32_t a = lmw::first(rs.dir_a_prev[m.leaf0 + r0 + q]);
        uint32_t L = sa_leaf(a), n = sa_n(a);
        uint32_t* lr = t.it + (uint64_t)L * SP_REC;
        bool in = (uint32_t)lane < n;
        SpanRegs R;
        R.n = n;
        R.id = in ? lr[lane] : NONE; R.len = in ? lr[64 + lane] : 0u;
        if (renumber) {
          uint32_t ol = in ? lr[128 + lane] : NONE, orr = in ? lr[192 + lane] : NONE;
          if (in) {
            R.id = pid_make(s_pmap[pid_peer(R.id)], pid_ctr(R.id));
            if (ol != NONE) ol = pid_make(s_pmap[pid_peer(ol)], pid_ctr(ol));
            if (orr != NONE) orr = pid_make(s_pmap[pid_peer(orr)], pid_ctr(orr));
            lr[lane] = R.id; lr[128 + lane] = ol; lr[192 + lane] = orr;
          }
        }
        if (!keep_loc) sp_set_loc_lanes(t, R, in, L);
      }
    }
    lmw::mem_fence();
    lmw::block_sync();
  }
  for (uint32_t cidx = 0; cidx < m.n_cont && !t.err; cidx++) {
    uint32_t ckind = d.cont[m.cid0 + cidx].kind_root & 0xff;
    if (ckind != CK_TEXT && ckind != CK_LIST && !(ML && ckind == CK_MOVABLE)) continue;
    const bool resume = RES && !fresh && cidx < C0;
    if (!resume && t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); break; }
    lmw::block_sync();
    t.cache_leaf = NONE; t.cr.n = 255; t.cache_pre = NONE; t.dirty = false; t.loc_pend = 0;
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = 0;
    if (resume) {
      // the container's tracker as the previous run left it: directory, element counts, version
      const uint32_t* rec = tk_cont(tk, tk_pcap, cidx);
      uint32_t r0 = lmw::first(rec[0]), nr = lmw::first(rec[1]);
      if (nr == 0 || nr > dir_cap) { if (nr) t.err = ST_RETRY; else LM_SETERR(t.err, ST_INTERNAL); break; }   // (optimistic LDS directory: the retry launch replays from the empty version)
      uint32_t act = 0;
      for (uint32_t i = (uint32_t)lane; i < nr; i += 64) { uint32_t b = rs.dir_b_prev[m.leaf0 + r0 + i]; s_da[i] = rs.dir_a_prev[m.leaf0 + r0 + i]; s_db[i] = b; act += b; }
      t.n_dir = nr; t.tot_active = lmw::reduce_add(act); t.n_alive = lmw::first(rec[2]);
      lmw::block_sync();
      t.ds_on = nr > SD_LINEAR;
      if (t.ds_on) sd_sums_from(t, 0);
      for (uint32_t q = (uint32_t)lane; q < P0; q += 64) s_cur[s_pmap[q]] = rec[TK_CW + q];
    } else {
      uint32_t L0 = t.n_leaf++;
      if (lane == 0) { s_da[0] = sa_make(L0, 0, false); s_db[0] = 0; }
      t.n_dir = 1; t.tot_active = 0; t.n_alive = 0; t.ds_on = false;
    }
    lmw::block_sync();
    bool touched = false;
    t.loc = (PLAIN && !RES && LM_LAZY_LOC) ? nullptr : d.loc + elem0;   // (ts_build_loc)
    bool base_on = false;              // the tracker has a base version (s_base), ts_goto
    uint32_t* s_base = s_tgt;          // (not RES: the slot of the resident kernels' rendered version)
    // ---- the linear prefix (lm_k_integrate_linear.h): the nodes in front of the first "a concurrent section begins here" flag —
    // every version up to there is critical — are replayed as a positional rope, not through the tracker.  (DF_CUT: k_dag_b
    // computed the flags; a document that is one chain has none and is replayed here to its end.)
    uint32_t oi0 = 0;
    if (PLAIN && !RES && LM_LINEAR && LM_LAZY_LOC && (m.flags & DF_CUT) && !d.no_linear) {
      Tl c;
      tl_none(c);
      bool emptied = false;
      const uint32_t* op_w = (const uint32_t*)op_ro;
      for (; oi0 < m.n_nodes && !t.err; oi0++) {
        uint32_t n = d.node_order[m.chg0 + oi0];
        if (lmw::first(g.node_done[m.chg0 + n]) & 2u) break;
        uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
        uint32_t node_peer = chg_ro[sorted_ro[m.chg0 + first]].peer;
        const ChangeRow lc = chg_ro[sorted_ro[m.chg0 + last]];
        RowWin w;
        w.base = NONE - 64; w.lim = m.op0 + m.n_op; w.cur = 0; w.nxt = 0;
        bool node_contig = false, node_mine = true;
        if (FUSE && last > first) {   // (see the tracker's loop below)
          bool bad = false, mine = false;
          for (uint32_t i = first + (uint32_t)lane; i <= last; i += 64) {
            uint32_t cr0 = sorted_ro[m.chg0 + i];
            if (i < last) { const ChangeRow c0 = chg_ro[cr0], c1 = chg_ro[sorted_ro[m.chg0 + i + 1]]; bad |= c0.op0 + c0.n_op != c1.op0; }
            mine |= ((d.chg_mask[2 * (uint64_t)cr0 + ((cidx >> 5) & 1)] >> (cidx & 31)) & 1) != 0;
          }
          node_contig = !lmw::any(bad);
          if (node_contig) node_mine = lmw::any(mine);
        }
        for (uint32_t ci = first; ci <= last && node_mine && !t.err; ci++) {
          uint32_t crow = sorted_ro[m.chg0 + ci];
          const ChangeRow ch = chg_ro[crow];
          uint32_t n_rows = ((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1) ? ch.n_op : 0u;
          if (FUSE && node_contig) { n_rows = lc.op0 + lc.n_op - ch.op0; ci = last; }
          // op rows 64 at a time, lane = row (two 16-byte loads per lane), the next 64 requested while these are replayed; the rows
          // of this container (FUSE: that head a run) are picked by one ballot, a row's words come out of the registers with ONE lane
          // index.  (The tracker's 8-row window — lane = word of the window — costs eight readlanes with eight indices and two window
          // tests per row: 0.55 µs per row by the -DLM_PROF clock, a quarter of a delete.)
          struct alignas(16) U4 { uint32_t x, y, z, w; };
          const uint32_t row_end = ch.op0 + n_rows;
          const U4 z4 = {0u, 0u, 0u, 0u};
          U4 na = z4, nb = z4;
          if (ch.op0 + (uint32_t)lane < row_end) { const U4* p = (const U4*)(op_ro + ch.op0 + (uint32_t)lane); na = p[0]; nb = p[1]; }
          for (uint32_t base = ch.op0; base < row_end && !t.err; base += 64) {
            U4 a4 = na, b4 = nb;
            if (base + 64 + (uint32_t)lane < row_end) { const U4* p = (const U4*)(op_ro + base + 64 + (uint32_t)lane); na = p[0]; nb = p[1]; }
            const bool in = base + (uint32_t)lane < row_end;
            // what a row needs beyond its words — kind, the delete's leftmost position and length, its checks, the insert's first id — is
            // worked out for all 64 rows at once on the vector unit (the kernel is bound by SCALAR issue: ≈25 scalar instructions per row
            // as wave-uniform arithmetic, ≈25 vector instructions per 64 rows here); the row loop reads four words per row
            if (FUSE && in && (a4.x & OPF_HEAD)) {   // a head row carries its run's extent (k_fuse_rows): total length; deletes: leftmost target, signed total
              const uint32_t row = base + (uint32_t)lane;
              uint32_t fa1 = d.fuse[2 * (uint64_t)row];
              int32_t fa2 = (int32_t)d.fuse[2 * (uint64_t)row + 1];
              a4.z = (uint32_t)(fa2 < 0 ? -fa2 : fa2);
              if (((a4.x >> 16) & 0xff) == OK_DEL) { b4.y = fa1; b4.z = (uint32_t)fa2; }
            }
            const uint32_t kindv = (a4.x >> 16) & 0xff;
            const bool insv = (kindv == OK_TEXT_INS) | (kindv == OK_LIST_INS), delv = kindv == OK_DEL;
            const int32_t a2v = (int32_t)b4.z;
            const uint32_t Lnv = (uint32_t)(a2v < 0 ? -a2v : a2v);
            // (the row's checks are the tracker's, below: a span as long as its op, a position inside the sequence)
            const uint32_t badv = (Lnv ^ a4.z) | (a4.y >> 31) | ((b4.z >> 31) & ((a4.y + 1u - Lnv) >> 31));
            uint32_t codev = insv ? 1u : (delv ? (badv ? 3u : 2u) : 0u);               // 1 insert, 2 delete, 3 a delete row that fails its checks
            uint32_t posv = delv ? (a2v > 0 ? a4.y : a4.y + 1u - Lnv) : a4.y;            // active position (a delete: of its leftmost target)
            uint32_t lenv = delv ? Lnv : a4.z;
            const uint32_t pidv = pid_make(node_peer, a4.w);
            uint64_t mine = lmw::ballot(in & ((a4.x & 0xffff) == cidx) & (!FUSE | !(a4.x & OPF_CONT)) & (codev != 0));
            while (mine && !t.err) {
              const int j = lmw::ffs64(mine);
              mine &= mine - 1;
              const uint32_t code = lmw::bcast(codev, j), pos = lmw::bcast(posv, j), n = lmw::bcast(lenv, j);
              touched = true;
              if (code == 1) tl_insert(t, c, pos, lmw::bcast(pidv, j), n);
              else if (code == 2) {
                const uint32_t left = tl_delete(t, c, pos, n, emptied);
                if (left) { lenv = lane == j ? left : lenv; mine |= 1ull << j; }   // the range runs on into the next leaf: the rest is this lane's row again
              } else LM_SETERR(t.err, ST_DATA_CORRUPTION);
            }
          }
        }
        if (lane == 0) s_cur[node_peer] = lc.ctr + lc.len;
      }
      tl_finish(t, c, emptied);
      lmw::block_sync();
#ifdef LM_EMU_TRACE
      if (getenv("LM_EMU_BASE") && lane == 0) fprintf(stderr, "LINEAR cont %u: %u of %u nodes, %u elements in %u leaves%s\n", cidx, oi0, m.n_nodes, t.tot_active, t.n_dir, emptied ? " (compacted)" : "");
#endif
      TS_CHECK("linear prefix", oi0);
    }
    for (uint32_t oi = oi0; oi < m.n_nodes && !t.err; oi++) {
      uint32_t n = d.node_order[m.chg0 + oi];
      uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
      const uint32_t* vv = vvh_ro + vvh0 + (uint64_t)n * P;
      uint32_t node_peer = chg_ro[sorted_ro[m.chg0 + first]].peer;
      // (k_dag_b: the replayed history in front of this node is a critical version and a concurrent section begins behind it)
      const bool conv_node = !RES && LM_BASE_RESET && (lmw::first(g.node_done[m.chg0 + n]) & 2u) != 0;
      if (RES) {   // a node the stored tracker has applied to its end
        const ChangeRow lc = chg_ro[sorted_ro[m.chg0 + last]];
        if (lc.ctr + lc.len <= s_app[node_peer]) continue;
      }
      bool checked_out = false;
      const uint32_t* op_w = (const uint32_t*)op_ro;
      RowWin w;
      w.base = NONE - 64; w.lim = m.op0 + m.n_op; w.cur = 0; w.nxt = 0;   // opened by the first row; stays open across the node's changes
      // FUSE (one change per keystroke): when the rows of the node's changes lie back to back in the op table — consecutive changes of
      // consecutive blocks — the node is ONE row range: the per-change work (two dependent gathers and ≈60 instructions for each of a
      // 40,000-change document's changes, most of which hold one continuation row that is skipped anyway) is paid once per node
      bool node_contig = false;
      uint32_t node_row_hi = 0, node_end_ctr = 0;
      if (FUSE && last > first) {
        bool bad = false, mine = false;
        for (uint32_t i = first + (uint32_t)lane; i <= last; i += 64) {
          uint32_t cr0 = sorted_ro[m.chg0 + i];
          if (i < last) { const ChangeRow c0 = chg_ro[cr0], c1 = chg_ro[sorted_ro[m.chg0 + i + 1]]; bad |= c0.op0 + c0.n_op != c1.op0; }
          mine |= ((d.chg_mask[2 * (uint64_t)cr0 + ((cidx >> 5) & 1)] >> (cidx & 31)) & 1) != 0;
        }
        node_contig = !lmw::any(bad);
        if (node_contig) {
          if (!lmw::any(mine)) continue;   // no change of the node holds a row of this container
          const ChangeRow lc = chg_ro[sorted_ro[m.chg0 + last]];
          node_row_hi = lc.op0 + lc.n_op; node_end_ctr = lc.ctr + lc.len;
        }
      }
      for (uint32_t ci = first; ci <= last && !t.err; ci++) {
        uint32_t crow = sorted_ro[m.chg0 + ci];
        const ChangeRow ch = chg_ro[crow];
        uint32_t skip_to = PLAIN ? ch.ctr : ch.ctr + skip_ro[crow];
        if (RES) { uint32_t ap = s_app[node_peer]; if (ap > skip_to) skip_to = ap; }   // what the stored tracker has applied is skipped like a known prefix
        uint32_t pe = PLAIN ? ch.ctr + ch.len : s_end[node_peer];   // (PLAIN: every applied change lies inside the rendered version)
        uint32_t n_rows = ((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1) ? ch.n_op : 0u;
        if (RES && ch.ctr + ch.len <= (PLAIN ? s_app[node_peer] : skip_to)) n_rows = 0;   // (PLAIN: the applied end lies on a change boundary — no sliced change)
        if (FUSE && node_contig) { n_rows = node_row_hi - ch.op0; pe = node_end_ctr; ci = last; }   // the whole node in this trip
        if (PLAIN && !RES && n_rows && !checked_out) {
          // the tracker moves to the node's dependencies before its first row.  A plain document has no sliced change: a change
          // whose mask names the container holds a row for it, so the move is decided here, once per change, and the row loop
          // carries no test for it (7 instructions per op row, 4 of them register moves in front of the branch)
          checked_out = true;
          PROF_T0();
          ts_goto<ML, SWEEP, false, POS>(t, d, m, cidx, P, vv, s_cur, s_base, base_on, conv_node, d.loc + elem0, nullptr);
          PROF_ADD(t, PF_CHECKOUT);
          TS_CHECK("checkout", ch.op0);
        }
#ifndef LM_ROWS64
#define LM_ROWS64 1
#endif
#ifndef LM_ROWS64_VEC
#define LM_ROWS64_VEC 1
#endif
        // V64 (the plain batch kernels — no sliced change, no style anchor, no move, no checkout: a row is an insert or a delete): op rows
        // 64 at a time, lane = row, and what the replay needs of a row is worked out for all 64 at once on the vector unit — the kind,
        // the delete's target range [t0, t1), its position hint and its checks, the insert's first id.  The kernel is bound by SCALAR
        // issue: as wave-uniform arithmetic that was ≈30 scalar instructions per row, here ≈30 vector instructions per 64
        // rows; the row loop reads five words per row with one lane index.
        constexpr bool V64 = PLAIN && !RES && !POS && LM_ROWS64 && LM_ROWS64_VEC;
        if (V64) {
          struct alignas(16) V4 { uint32_t x, y, z, w; };
          const uint32_t v_end = ch.op0 + n_rows;
          for (uint32_t base = ch.op0; base < v_end && !t.err; base += 64) {
            V4 a4 = {0u, 0u, 0u, 0u}, b4 = {0u, 0u, 0u, 0u};
            const bool in = base + (uint32_t)lane < v_end;
            if (in) { const V4* p = (const V4*)(op_ro + base + (uint32_t)lane); a4 = p[0]; b4 = p[1]; }
            if (FUSE && in && (a4.x & OPF_HEAD)) {   // a head row carries its run's extent (k_fuse_rows): total length; deletes: leftmost target, signed total
              const uint32_t row = base + (uint32_t)lane;
              uint32_t fa1 = d.fuse[2 * (uint64_t)row];
              int32_t fa2 = (int32_t)d.fuse[2 * (uint64_t)row + 1];
              a4.z = (uint32_t)(fa2 < 0 ? -fa2 : fa2);
              if (((a4.x >> 16) & 0xff) == OK_DEL) { b4.y = fa1; b4.z = (uint32_t)fa2; }
            }
            const uint32_t kindv = (a4.x >> 16) & 0xff;
            const bool insv = (kindv == OK_TEXT_INS) | (kindv == OK_LIST_INS), delv = kindv == OK_DEL;
            const int32_t a2v = (int32_t)b4.z;
            const uint32_t Lnv = (uint32_t)(a2v < 0 ? -a2v : a2v);
            // a delete span as long as its op and with a position inside the sequence (see the general loop below for the why)
            const uint32_t badv = (Lnv ^ a4.z) | (a4.y >> 31) | ((b4.z >> 31) & ((a4.y + 1u - Lnv) >> 31));
            const uint32_t codev = insv ? 1u : (delv ? (badv ? 3u : 2u) : 0u);   // (a row that is neither is no row of a plain document's replay)
            // insert: position, first id, length | delete: position hint (1-based, leftmost target), [t0, t1) of peer a0
            const uint32_t w0v = insv ? a4.y : (a2v > 0 ? a4.y + 1u : a4.y + 2u - Lnv);
            const uint32_t w1v = insv ? pid_make(node_peer, a4.w) : (a2v > 0 ? b4.y : b4.y + (Lnv - a4.z));
            const uint32_t w2v = insv ? a4.z : (a2v > 0 ? b4.y + a4.z : b4.y + Lnv);
            uint64_t mine = lmw::ballot(in & ((a4.x & 0xffff) == cidx) & (!FUSE | !(a4.x & OPF_CONT)) & (codev != 0));
            while (mine && !t.err) {
              PROF_T0();
              const int j = lmw::ffs64(mine);
              mine &= mine - 1;
              const uint32_t code = lmw::bcast(codev, j), w0 = lmw::bcast(w0v, j), w1 = lmw::bcast(w1v, j), w2 = lmw::bcast(w2v, j);
              touched = true;
              PROF_ADD(t, PF_ROW);
              if (code == 1) {
                ts_insert(t, w0, w1, w2);
                TS_CHECK("insert", base + (uint32_t)j);
              } else if (code == 2) {
                ts_update_range(t, lmw::bcast(b4.x, j), w1, w2, UPD_DEL_INC, w0);
                PROF_ADD(t, PF_DELETE);
                PROF_CNT(t, PF_NDEL, 1);
                TS_CHECK("delete", base + (uint32_t)j);
              } else LM_SETERR(t.err, ST_DATA_CORRUPTION);
            }
          }
        } else {
        uint32_t row0 = ch.op0, last_row = 0;   // (POS only: where the row loop is entered again behind a row that was finished by position)
      rows_again:
        // R64 (the plain batch kernels): op rows 64 at a time, lane = row — two 16-byte loads per lane, one ballot picks the rows of this
        // container (FUSE: that head a run), a row's words come out of the registers with one lane index; the other instantiations keep
        // the 8-row window (rw_get: lane = word of the window) — their register budgets have no room for eight more vector registers
#ifndef LM_ROWS64
#define LM_ROWS64 1
#endif
#ifndef LM_ROWS64_ALL
#define LM_ROWS64_ALL 0
#endif
        constexpr bool R64 = LM_ROWS64 && ((PLAIN && !RES && !POS) || (LM_ROWS64_ALL && !ML));   // (PLAIN && !RES && !POS: only in -DLM_ROWS64_VEC=0 builds — V64 above otherwise)
        struct alignas(16) U4 { uint32_t x, y, z, w; };
        const uint32_t row_end = ch.op0 + n_rows;
        for (uint32_t base = row0; base < row_end && !t.err; base += (R64 ? 64u : 1u)) {
          U4 a4 = {0u, 0u, 0u, 0u}, b4 = {0u, 0u, 0u, 0u};
          uint64_t mine = 1;
          if (R64) {
            const bool in = base + (uint32_t)lane < row_end;
            if (in) { const U4* p = (const U4*)(op_ro + base + (uint32_t)lane); a4 = p[0]; b4 = p[1]; }
            mine = lmw::ballot(in & ((a4.x & 0xffff) == cidx) & (!FUSE | !(a4.x & OPF_CONT)));
          }
          while (mine && !t.err) {
          PROF_T0();
          uint32_t row;
          OpRow r;
          if (R64) {
            const int j = lmw::ffs64(mine);
            mine &= mine - 1;
            row = base + (uint32_t)j;
            r.cidx_kind = lmw::bcast(a4.x, j); r.prop = (int32_t)lmw::bcast(a4.y, j); r.len = lmw::bcast(a4.z, j); r.ctr = lmw::bcast(a4.w, j);
            r.a0 = lmw::bcast(b4.x, j); r.a1 = lmw::bcast(b4.y, j); r.a2 = (int32_t)lmw::bcast(b4.z, j); r.chg = 0;
          } else {
            mine = 0;
            row = base;
            r = rw_get(w, op_w, row);
            if ((r.cidx_kind & 0xffff) != cidx) continue;
            if (FUSE && (r.cidx_kind & OPF_CONT)) continue;
          }
          if (POS) last_row = row;
          if (FUSE) {
            // k_fuse_rows (lm_k_fuse.h): a row that continues the run of the row in front of it was replayed with its head (skipped
            // above); a head carries its run's extent (inserts: the total length; deletes: the leftmost target and the signed total)
            if (r.cidx_kind & OPF_HEAD) {
              lmw::wave_sync();
              uint32_t fa1 = lmw::first(d.fuse[2 * (uint64_t)row]);
              int32_t fa2 = (int32_t)lmw::first(d.fuse[2 * (uint64_t)row + 1]);
              r.len = (uint32_t)(fa2 < 0 ? -fa2 : fa2);
              if (((r.cidx_kind >> 16) & 0xff) == OK_DEL) { r.a1 = fa1; r.a2 = fa2; }
            }
          }
          if (!PLAIN && r.ctr + r.len <= skip_to) continue;
          uint32_t kind = (r.cidx_kind >> 16) & 0xff;
          uint32_t a = PLAIN ? 0u : (skip_to > r.ctr ? skip_to - r.ctr : 0);
          touched = true;
          if (!PLAIN && r.ctr + a >= pe) continue;
          uint32_t b = PLAIN ? r.len : (r.ctr + r.len <= pe ? r.len : pe - r.ctr);
          if (!(PLAIN && !RES) && !checked_out) {
            checked_out = true;
            if (!RES) ts_goto<ML, SWEEP, (!ML && !PLAIN && !RES && !POS && LM_BATCH_VSWEEP), POS>(t, d, m, cidx, P, vv, s_cur, s_base, base_on, conv_node, d.loc + elem0, d.dcnt ? d.dcnt + elem0 : nullptr);
            else {
            lmw::block_sync();
            if (RES && !ML && d.dcnt && ts_sweep_pays(t, m, P, vv, node_peer, r.ctr + a, s_cur)) ts_sweep_version(t, d, m, cidx, P, vv, node_peer, r.ctr + a, s_cur, d.dcnt + elem0);
            else
            for (uint32_t p = 0; p < P && !t.err; p++) {
              uint32_t cur = s_cur[p], tgt = vv[p];
              // RES: the stored tracker may stand below this peer's own earlier ops (a previous run left it at an older
              // version): everything of the peer in front of this op belongs to the op's version
              if (RES && p == node_peer) tgt = r.ctr + a;
              if (cur > tgt) ts_move_ops<ML, SWEEP>(t, d, m, cidx, p, tgt, cur, -1);
              else if (cur < tgt) ts_move_ops<ML, SWEEP>(t, d, m, cidx, p, cur, tgt, +1);
            }
            lmw::block_sync();
            for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = (RES && p == node_peer) ? r.ctr + a : vv[p];
            lmw::block_sync();
            }
            PROF_ADD(t, PF_CHECKOUT);
            TS_CHECK("checkout", row);
          }
          PROF_ADD(t, PF_ROW);
          // MovableList move (diff_calc.rs:1917-1941 → tracker.rs:289-347): the active item at `from` is deleted — its id is kept
          // in the move item's own payload slot (cp[], otherwise unused for a move) for later retreats / forwards — and a new
          // item with the op's id is inserted at `to`, evaluated after the deletion.  The row is rewritten as that delete and
          // then as that insert, each taking the ordinary path below (no second inlined copy of the two big routines)
          uint32_t mv_to = NONE;
          if (ML && kind == OK_LIST_MOVE) {
            uint32_t tgt = ts_active_id_at(t, (uint32_t)r.a2);
            if (tgt == NONE || (uint32_t)r.prop >= t.tot_active) { LM_SETERR(t.err, ST_DATA_CORRUPTION); break; }   // (after the deletion `to` may equal the new length)
            if (lane == 0) (d.cp + elem0)[ts_g(t, pid_make(node_peer, r.ctr))] = tgt;
            mv_to = (uint32_t)r.prop;
            r.prop = r.a2; r.a0 = pid_peer(tgt); r.a1 = pid_ctr(tgt); r.a2 = 1; kind = OK_DEL;
          }
          for (;;) {
            if (kind == OK_TEXT_INS || kind == OK_LIST_INS) {
              ts_insert<POS>(t, (uint32_t)r.prop + a, pid_make(node_peer, r.ctr + a), b - a);
              TS_CHECK("insert", row);
            } else if (PLAIN || kind == OK_DEL) {
              // (PLAIN: a two-way dispatch — a row that is neither an insert nor a delete becomes an empty range below; a third
              // edge to the end of the row cost 17 register moves per delete row, hoisted in front of its branch)
              uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
              uint32_t t0, t1;
              if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
              else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }
              // a delete span as long as its op and with a position inside the sequence (what every writer emits; the reference's
              // decoder does not compare the two lengths, block_encode.rs:651-704, and would delete |span| elements at the position
              // while the counters advance by the op's length — not a value this engine reproduces: LM_DATA_CORRUPTION)
              // (no branch out of the row for it: the finding empties the range, and the row loop ends on t.err as for any other error —
              // an extra exit edge here cost 46 register moves per delete row, hoisted in front of the branch)
              // (as integer arithmetic: a boolean expression of wave-uniform compares becomes a chain of 64-bit lane-mask selects.
              // Ln <= 2^31 and, where the last term counts, 0 <= prop: the sign bit of prop + 1 - Ln says prop + 1 < Ln)
              const uint32_t not_del = PLAIN ? (kind ^ OK_DEL) : 0u;
              const uint32_t bad_bits = (Ln ^ r.len) | ((uint32_t)r.prop >> 31) | (((uint32_t)r.a2 >> 31) & (((uint32_t)r.prop + 1u - Ln) >> 31));
              // unsliced row: its position addresses the leftmost target (forward: prop; backward: prop + 1 - len)
              uint32_t hint = 0;
              if (a == 0 && b == r.len) hint = r.a2 > 0 ? (uint32_t)r.prop + 1 : (uint32_t)r.prop + 2 - Ln;
              if (not_del | bad_bits) { if (!not_del) LM_SETERR(t.err, ST_DATA_CORRUPTION); t1 = t0; hint = 0; }
              if (POS && !RES && hint && !t.err && t1 > t0 && t1 <= lmw::first(s_app[r.a0])) {
                // every target lies below the base version (lm_snapshot_base.h): no item carries these ids — the row is applied by position,
                // as ts_update_range would find out by failing to match it (crdt_rope.rs:256-335; tracker.rs:40-60: content the tracker
                // knows as a placeholder) — without the mismatch, the unwinding and the restart of the row loop
                lmw::wave_sync();
                if (lane == 0) { uint32_t* w = pd_w(t); w[0] = hint; w[1] = t1 - t0; w[2] = pid_make(r.a0, t0); w[3] = 0; }
                ts_del_positional(t, row);
              } else
              ts_update_range<POS>(t, r.a0, t0, t1, UPD_DEL_INC, hint);
              PROF_ADD(t, PF_DELETE);
              PROF_CNT(t, PF_NDEL, 1);
              TS_CHECK("delete", row);
            } else if (!PLAIN && kind == OK_STYLE_START) {
              ts_insert<POS>(t, (uint32_t)r.prop, pid_make(node_peer, r.ctr), 1);
            } else if (!PLAIN && kind == OK_STYLE_END) {
              uint32_t end_pos = NONE;
              if (row > ch.op0) {
                const OpRow pr = op_ro[row - 1];
                if (((pr.cidx_kind >> 16) & 0xff) == OK_STYLE_START && pr.ctr + 1 == r.ctr && (pr.cidx_kind & 0xffff) == cidx)
                  end_pos = (uint32_t)pr.prop + pr.a0;
              }
              if (end_pos == NONE) { LM_SETERR(t.err, ST_UNSUPPORTED); break; }
              uint32_t pos = end_pos + 1 < t.tot_active ? end_pos + 1 : t.tot_active;
              ts_insert<POS>(t, pos, pid_make(node_peer, r.ctr), 1);
            }
            if (!ML || mv_to == NONE || t.err) break;
            r.prop = (int32_t)mv_to; kind = OK_LIST_INS; mv_to = NONE;   // second half of a move: the new item
          }
          }   // (rows of this step)
        }
        if (POS && t.err == ST_POSDEL) {   // a delete row that does not match its position: finished by position, then on with the next row
          ts_del_positional(t, last_row);
          row0 = last_row + 1;
          goto rows_again;
        }
        }   // (!V64)
        if (checked_out && lane == 0) s_cur[node_peer] = (FUSE && node_contig) ? pe : (ch.ctr + ch.len < pe ? ch.ctr + ch.len : pe);
      }
      lmw::block_sync();
    }
    if (RES && !PLAIN && to_version && !t.err) {
      // the tracker moves to the version being rendered (Tracker::checkout, tracker.rs:354-461).  A document rendered at the
      // latest version needs no move: every applied op has been replayed, "never deleted" is what shows (as in a batch), and the
      // tracker stays where the last change left it — the next run moves it wherever its first change needs it
      lmw::block_sync();
      if (!ML && d.dcnt && ts_sweep_pays(t, m, P, s_tgt, NONE, 0, s_cur)) ts_sweep_version(t, d, m, cidx, P, s_tgt, NONE, 0, s_cur, d.dcnt + elem0);
      else
      for (uint32_t p = 0; p < P && !t.err; p++) {
        uint32_t cur = s_cur[p], tgt = s_tgt[p];
        if (cur > tgt) ts_move_ops<ML, SWEEP>(t, d, m, cidx, p, tgt, cur, -1);
        else if (cur < tgt) ts_move_ops<ML, SWEEP>(t, d, m, cidx, p, cur, tgt, +1);
      }
      lmw::block_sync();
      for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = s_tgt[p];
      lmw::block_sync();
      TS_CHECK("closing checkout", 0);
      if (t.err) break;
    }
    if (RES) touched = true;
    sp_flush(t);
    lmw::block_sync();
    if (dir_used + t.n_dir > m.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); break; }
    for (uint32_t i = (uint32_t)lane; i < t.n_dir; i += 64) d.dir_out[m.leaf0 + dir_used + i] = s_da[i];
    if (RES) {
      for (uint32_t i = (uint32_t)lane; i < t.n_dir; i += 64) rs.dir_b[m.leaf0 + dir_used + i] = s_db[i];
      uint32_t* rec = tk_cont(tk, tk_pcap, cidx);
      for (uint32_t p = (uint32_t)lane; p < P; p += 64) rec[TK_CW + p] = s_cur[p];
      if (lane == 0) { rec[0] = dir_used; rec[1] = t.n_dir; rec[2] = t.n_alive; }
    }
    if (lane == 0) {
      d.cont_root0[m.cid0 + cidx] = dir_used;
      d.cont_nroot[m.cid0 + cidx] = t.n_dir;
      // the state store holds a root sequence once something is visible in it (diff_calc.rs:299: a container state is
      // created by a non-empty diff; from the empty version the diff is empty exactly when nothing is visible)
      // (a MovableList exists once an element was ever inserted — k_mlist_post adds that case after this stage)
      // (RES: visible at the latest version — n_alive counts every applied op — or at the version the tracker was just moved to;
      // earlier runs' verdicts are OR-ed in by k_res_exists)
      if (touched && (t.n_alive > 0 || (RES && !PLAIN && to_version && t.tot_active > 0))) d.cont[m.cid0 + cidx].touched = 1;
    }
    dir_used += t.n_dir;
    lmw::block_sync();
  }
  // a delete row that does not match its position: the document is replayed by k_integrate_span_pos (lm_pipeline.h) — unless this IS
  // that kernel, the document is resident (its list would have to outlive the run) or holds a MovableList: LM_DATA_CORRUPTION
  if (!t.err && t.beyond) t.err = ST_DATA_CORRUPTION;   // an insert row beyond the end (ts_insert / tl_insert)
  // (RES: the context replays such a document of a folded batch through the batch kernels, lm_capi_impl.h redo — DF_REDO)
  if (t.err == ST_POSDEL && RES && !ML && d.posdel_redo && lane == 0) lmw::atomic_or(&d.doc[doc].flags, DF_REDO);
  if (t.err == ST_POSDEL && (POS || RES || ML || !d.posdel)) t.err = ST_DATA_CORRUPTION;
  // (retry_pass 3: the by-position kernel launched with the OPTIMISTIC directory — a batch whose documents all take this path, staged on
  // snapshot states, would otherwise run at the occupancy the worst-case directory leaves; a document that overflows it is replayed once
  // more by the same kernel with the worst-case directory, retry_pass 2)
  if (POS && retry_pass == 3u && t.err == ST_RETRY) t.err = ST_POSDEL;
  if (t.err && lane == 0) {
    d.doc[doc].status = t.err;
    if (t.err == ST_RETRY) lmw::atomic_add(retry_count, 1u);
    if (t.err == ST_POSDEL) lmw::atomic_add(retry_count + 3, 1u);
  }
  if (RES && !t.err) {
    // the record of this run: peers (the numbering the leaves are packed with), what has been applied, the element layout
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) {
      uint64_t id = d.peer_uniq[m.praw0 + p];
      tk_peers(tk)[2 * p] = (uint32_t)id; tk_peers(tk)[2 * p + 1] = (uint32_t)(id >> 32);
      tk_applied(tk, tk_pcap)[p] = s_end[p];
      tk_ebase(tk, tk_pcap)[p] = s_ebase[p];
      tk_ecap(tk, tk_pcap)[p] = d.elem_cap[m.praw0 + p];
    }
    // containers that are not sequences keep an empty tracker record (their sticky `exists` word is k_res_exists's)
    for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) {
      uint32_t ck = d.cont[m.cid0 + c].kind_root & 0xff;
      if (ck != CK_TEXT && ck != CK_LIST && !(ML && ck == CK_MOVABLE)) { uint32_t* rec = tk_cont(tk, tk_pcap, c); rec[0] = 0; rec[1] = 0; rec[2] = 0; }
    }
    lmw::mem_fence();
    lmw::block_sync();
    if (lane == 0) { tk[1] = P; tk[2] = m.n_cont; tk[3] = t.n_leaf; tk[4] = m.elem0_lo; tk[5] = m.elem0_hi; tk[6] = (m.atoms + 3) & ~3u; tk[7] = m.pending_lo | m.pending_hi; tk[8] = m.atoms; tk[0] = 1; }
  }
  if (lane == 0) d.doc[doc].pad0 = dir_used;
#ifdef LM_PROF
  t.prof[PF_TOTAL] = lmw::clock() - pf_begin;
  if (lane == 0) for (int i = 0; i < PF_N; i++) d.prof[(uint64_t)doc * PF_N + i] = t.prof[i];
#endif
}
LM_KERNEL LM_WAVES_PER_SIMD(LM_INTEGRATE_WAVES) void k_integrate_span(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, false>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
LM_KERNEL LM_WAVES_PER_SIMD(LM_INTEGRATE_WAVES) void k_integrate_span_plain(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
LM_KERNEL LM_WAVES_PER_SIMD(LM_INTEGRATE_WAVES) void k_integrate_span_plain_sweep(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, true, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
// the plain documents k_dag_a flagged DF_FUSED: one change per keystroke, rows chained into runs by k_fuse_rows
LM_KERNEL LM_WAVES_PER_SIMD(LM_INTEGRATE_WAVES) void k_integrate_span_plain_fuse(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, true, true, false, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
LM_KERNEL LM_WAVES_PER_SIMD(LM_INTEGRATE_WAVES) void k_integrate_span_ml(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<true, false>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
// the documents an earlier launch left with ST_POSDEL (a damaged delete row), whatever their DF_PLAIN flag: the common body + POS
LM_KERNEL LM_WAVES_PER_SIMD(4) LM_ONE_WAVE_GROUPS void k_integrate_span_pos(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, false, false, false, false, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
// … and the PLAIN ones among them (Text / List inserts and deletes only, rendered at the latest version): the plain body + POS — linear prefix,
// lazy loc[], no style / checkout paths.  A batch staged on snapshot states (lm_snapshot_base.h) sends EVERY document here: the common body
// replays such a document six times slower than the plain one (tests/tools/gpu_snapbase.py).  Rows are read through the 8-row window (no
// V64 / R64: the row loop is entered again behind a row finished by position)
LM_KERNEL LM_WAVES_PER_SIMD(4) LM_ONE_WAVE_GROUPS void k_integrate_span_pos_plain(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count) {
  integrate_span_body<false, true, true, false, false, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count);
}
// Resident documents (DevRes above): the common body with PLAIN = false (sliced rows: the applied prefix), SWEEP, RES.
// (128 VGPRs: the prologue / epilogue state of a resident document does not fit the 96 of five waves per SIMD without scratch)
#ifndef LM_RES_WAVES
#define LM_RES_WAVES 4
#endif
LM_KERNEL LM_WAVES_PER_SIMD(LM_RES_WAVES) LM_ONE_WAVE_GROUPS void k_integrate_span_res(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count, DevRes rs) {
  integrate_span_body<false, false, true, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count, rs);
}
// the documents flagged DF_PLAIN (no sliced change, no style anchor, no MovableList, rendered at the latest version)
#ifndef LM_RES_PLAIN_WAVES
#define LM_RES_PLAIN_WAVES LM_INTEGRATE_WAVES
#endif
LM_KERNEL LM_WAVES_PER_SIMD(LM_RES_PLAIN_WAVES) LM_ONE_WAVE_GROUPS void k_integrate_span_res_plain(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count, DevRes rs) {
  integrate_span_body<false, true, true, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count, rs);
}
LM_KERNEL LM_WAVES_PER_SIMD(LM_RES_WAVES) LM_ONE_WAVE_GROUPS void k_integrate_span_res_ml(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                                const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                                const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro, uint32_t retry_pass,
                                uint32_t* retry_count, DevRes rs) {
  integrate_span_body<true, false, true, true>(d, g, dir_cap, pmax, op_ro, chg_ro, sorted_ro, skip_ro, vvh_ro, retry_pass, retry_count, rs);
}

// Before the payload fill and the integrate stage of a run: the element layout of a resident document.  A batch lays a
// document's elements out peer after peer (k_dag_a: base = Σ extents of the peers in front); a resident document would then move
// every later peer's elements whenever an earlier peer grows — and with them loc[], the payload slots and everything else that
// is indexed by element.  Here every peer owns a region with room to grow (extent × 1.5 + 64), handed out at the end of the
// document's slice when the peer is first seen; a known peer keeps its base as long as its extent fits its region, whatever
// index it has now.  Only when a region (or the slice) is outgrown the document is laid out anew — its stored tracker is then
// not used (DF_LAYOUT_SAME off: k_elem_fill fills everything, the integrate stage rebuilds loc[] — or replays, for what cannot
// be rebuilt).  DocMeta.atoms becomes the extent of the layout (slots up to it are cleared / bounded).  One wave per document.
LM_KERNEL void k_res_layout(Dev d, DevRes rs) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  LM_SHARED(uint32_t, s_base, MAX_PEERS);
  LM_SHARED(uint32_t, s_cap, MAX_PEERS);
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  const ResDoc rd = rs.doc[doc];
  uint32_t* tk = rs.tk + rd.tk_off;
  const uint32_t P = m.n_peers;
  if (P > MAX_PEERS) return;
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_base[p] = NONE; s_cap[p] = 0; }
  lmw::block_sync();
  bool same = !rd.reset && tk[0] == 1u && tk[4] == m.elem0_lo && tk[5] == m.elem0_hi && tk[1] <= P && tk[1] <= rd.pcap &&
              tk[2] <= m.n_cont && tk[3] <= m.leaf_cap;   // (everything else that makes the integrate stage start from the empty version)
  uint32_t top = 0;
  if (same) {
    uint32_t P0 = tk[1];
    bool bad = false;
    for (uint32_t q = (uint32_t)lane; q < P0; q += 64) {
      uint64_t id = ((uint64_t)tk_peers(tk)[2 * q + 1] << 32) | tk_peers(tk)[2 * q];
      uint32_t lo = 0, hi = P;
      while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < id) lo = mid + 1; else hi = mid; }
      if (lo < P && d.peer_uniq[m.praw0 + lo] == id && d.peer_ext[m.praw0 + lo] <= tk_ecap(tk, rd.pcap)[q]) { s_base[lo] = tk_ebase(tk, rd.pcap)[q]; s_cap[lo] = tk_ecap(tk, rd.pcap)[q]; }
      else bad = true;
    }
    same = !lmw::any(bad);
    top = tk[8];
  }
  lmw::block_sync();
  if (!same) { top = 0; for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_base[p] = NONE; }
  lmw::block_sync();
  // peers without a region (all of them when the document is laid out anew), in index order
  for (uint32_t p0 = 0; p0 < P; p0 += 64) {
    uint32_t p = p0 + (uint32_t)lane;
    bool need = p < P && s_base[p] == NONE;
    uint32_t ext = p < P ? d.peer_ext[m.praw0 + p] : 0u;
    uint32_t cap = need ? ((ext + ext / 2 + 64 + 3) & ~3u) : 0u;
    uint32_t inc = lmw::scan_incl_add(cap);
    if (need) { s_base[p] = top + inc - cap; s_cap[p] = cap; }
    top += lmw::bcast(inc, 63);
  }
  lmw::block_sync();
  if (top > rd.elem_cap) {
    if (same) {   // the new peers do not fit behind the old ones: lay the document out anew (the host sized the slice for that)
      same = false; top = 0;
      for (uint32_t p0 = 0; p0 < P; p0 += 64) {
        uint32_t p = p0 + (uint32_t)lane;
        uint32_t ext = p < P ? d.peer_ext[m.praw0 + p] : 0u;
        uint32_t cap = p < P ? ((ext + ext / 2 + 64 + 3) & ~3u) : 0u;
        uint32_t inc = lmw::scan_incl_add(cap);
        if (p < P) { s_base[p] = top + inc - cap; s_cap[p] = cap; }
        top += lmw::bcast(inc, 63);
      }
      lmw::block_sync();
    }
    if (top > rd.elem_cap) { if (lane == 0) LM_SETERR(d.doc[doc].status, ST_INTERNAL); return; }
  }
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { d.elem_base[m.praw0 + p] = s_base[p]; d.elem_cap[m.praw0 + p] = s_cap[p]; }
  if (lane == 0) {
    uint32_t fl = d.doc[doc].flags & ~(DF_LAYOUT_SAME | DF_FILL_KEPT);
    if (same) fl |= DF_LAYOUT_SAME | (tk[7] == 0u ? DF_FILL_KEPT : 0u);   // [7]: atoms the stored run left pending — their rows were not filled
    d.doc[doc].flags = fl;
    d.doc[doc].atoms = top;
  }
}

// After every stage that decides which containers the state store holds (integrate, k_map_lww, k_mlist_post, k_state_roots):
// a container state, once created, stays (state.rs:621-849 never drops one) — `touched` of a resident document is the OR over
// all its runs.  One wave per document.
// a rendering run of a shared replay reuses the tables of the import run: what the previous rendering's checkout showed must not
// leak into this one's view of the state store (the containers' `touched` words live in those tables)
LM_KERNEL void k_cont_untouch(Dev d, uint32_t n_cid) {
  uint32_t t = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  if (t < n_cid) d.cont[t].touched = 0;
}
LM_KERNEL void k_res_exists(Dev d, DevRes rs, uint32_t mode) {
  // mode 0: a resident document — OR over all its runs.  Documents staged once for several renderings (lm_capi_impl.h, "shared
  // replay": entries of one batch that name the same blobs and differ in their checkout): every rendering is import_batch + ONE
  // checkout of its own, so what the state store holds is what the import created (mode 1: this run IS that import — the verdict is
  // kept in the record's word 4) plus what this rendering's checkout shows (mode 2: word 4 | this run), never what another
  // rendering's checkout showed.
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  const ResDoc rd = rs.doc[doc];
  if (m.n_cont > rd.ccap) return;
  uint32_t* tk = rs.tk + rd.tk_off;
  for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) {
    uint32_t* rec = tk_cont(tk, rd.pcap, c);
    uint32_t t = d.cont[m.cid0 + c].touched ? 1u : 0u;
    uint32_t e = mode == 1 ? t : ((mode == 2 ? rec[4] == 1u : rec[3] == 1u) ? 1u : 0u) | t;
    rec[3] = e;
    if (mode == 1) rec[4] = e;
    d.cont[m.cid0 + c].touched = e;
  }
}

}  // namespace lm
