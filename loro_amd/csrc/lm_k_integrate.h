// Integrate stage: Event-Graph-Walker replay of a sequence container (Text / List), one wavefront per
// document.  Element-granular restatement of the reference tracker, shaped for a 64-lane wave:
//   * elements live in 64-slot leaves in HBM (one lane per slot: id, origin_left, origin_right, status —
//     FugueSpan per element, container/richtext/fugue_span.rs:191-207,257-279);
//   * the ORDER of the leaves and their cached (count, active count) live in an LDS directory, one 32-bit
//     entry per leaf.  Finding the k-th active element is an LDS chunk-sum + one wave prefix scan, so an
//     insert costs a single dependent HBM round trip (the leaf itself) instead of a tree descent;
//   * `loc[element] → leaf` replaces IdToCursor (container/richtext/tracker/id_to_cursor.rs);
//   * deletes / retreat / forward address elements by id (the wire carries the target id span,
//     container/list/list_op.rs:132-142), so they never search by position.
// Reference algorithm (paths relative to /root/reference/crates/loro-internal/src):
//   Tracker::insert/delete/checkout/forward   container/richtext/tracker.rs:88-160,193-252,354-546
//   CrdtRope::insert (Fugue integrate)        container/richtext/tracker/crdt_rope.rs:63-247
//   ActiveLenQueryPreferLeft                  container/richtext/tracker/crdt_rope.rs:564-615
//   replay_container_ops_from_empty           diff_calc.rs:445-485
//   apply_crdt_op_to_tracker                  diff_calc.rs:993-1137
#pragma once
#include "lm_k_dag.h"

namespace lm {

static constexpr uint32_t ST_FUT = 1u, ST_EVER = 2u, ST_DEL1 = 0x100u, ST_DELMASK = 0x00FFFF00u;
// span-granular kernel: deleted at the tracker's base version (a critical version of the history, ts_convert_base) — the top bit
// of the delete count, so every "is it active" test sees it; the counted part below it then holds deletes applied since
static constexpr uint32_t ST_DEAD = 0x00800000u;
LM_DEV bool st_active(uint32_t st) { return (st & (ST_FUT | ST_DELMASK)) == 0; }

// directory entry: leaf id (17 bits) | "holds a non-future element" (1 bit) | element count (7 bits) | active count (7 bits)
static constexpr uint32_t DIR_LEAF_MASK = 0x1FFFFu;
static constexpr uint32_t MAX_LEAVES_PER_DOC = 1u << 17;
static constexpr uint32_t DIR_NF = 1u << 17;
LM_DEV uint32_t de_make(uint32_t leaf, uint32_t n, uint32_t act, bool nf) { return leaf | (nf ? DIR_NF : 0u) | (n << 18) | (act << 25); }
LM_DEV uint32_t de_leaf(uint32_t e) { return e & DIR_LEAF_MASK; }
LM_DEV bool de_nf(uint32_t e) { return (e & DIR_NF) != 0; }
LM_DEV uint32_t de_n(uint32_t e) { return (e >> 18) & 0x7f; }
LM_DEV uint32_t de_act(uint32_t e) { return e >> 25; }

// optional cycle accounting (compile with -DLM_PROF): slots of the per-document profile record
enum { PF_ROW = 0, PF_FIND, PF_LEAF, PF_ORIGHT, PF_BETWEEN, PF_PLACE, PF_DELETE, PF_CHECKOUT, PF_NINS, PF_NDEL, PF_NEXTRA, PF_NHEAD, PF_TOTAL, PF_NHIT, PF_NDHIT, PF_N = 16 };
#ifdef LM_PROF
#define PROF_T0() uint64_t pf_t0_ = lmw::clock()
#define PROF_ADD(t, slot) do { uint64_t pf_t1_ = lmw::clock(); (t).prof[slot] += pf_t1_ - pf_t0_; pf_t0_ = pf_t1_; } while (0)
#define PROF_CNT(t, slot, v) do { (t).prof[slot] += (v); } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(t, slot) do {} while (0)
#define PROF_CNT(t, slot, v) do {} while (0)
#endif

struct LeafRegs { uint32_t n, id, ol, orr, st; };   // one leaf in registers: lane i holds slot i

struct Tr {  // wave-uniform context of one (document, sequence container) replay
  uint32_t* it;                   // HBM leaves, 1 KiB records: [leaf*256 + {0 id, 64 origin_left, 128 origin_right, 192 status} + slot]
  uint32_t* loc;                  // doc element → leaf
  const uint32_t* ebase;          // LDS: element base per peer
  const uint32_t* cur;            // LDS: the tracker's version per peer at the head of the node being replayed
  const uint32_t* end;            // LDS: version being rendered per peer (no element exists at or beyond it)
  uint32_t* dir;                  // LDS leaf directory in document order
  uint8_t* lchunk;                // HBM: leaf → chunk (= owning lane) of its directory entry (kept out of LDS for occupancy)
  uint32_t n_dir, dir_cap, CH, inv_CH;    // lane c owns directory entries [c*CH, (c+1)*CH); CH is odd (bank-conflict free)
  uint32_t my_sum;                // PER-LANE: Σ active counts of this lane's chunk, maintained incrementally
  uint32_t n_leaf, leaf_cap;
  uint32_t tot_active;
  uint32_t cache_p;               // directory position of the cached leaf (positions only shift on splits, which drop the cache)
  uint32_t cache_leaf;            // leaf currently mirrored in `cr` (NONE = none): typing stays in one leaf for many ops
  LeafRegs cr;                    // write-through register copy of that leaf
  int32_t err;
  uint32_t beyond;   // an insert row named a position beyond the end
  uint32_t posmis;   // a delete row whose targets are not the elements at its position (DF_REDO: replayed by the span-granular kernels)
#ifdef LM_PROF
  uint64_t prof[PF_N];
#endif
};

LM_DEV uint32_t tr_g(const Tr& t, uint32_t pid) { return t.ebase[pid_peer(pid)] + pid_ctr(pid); }

// ---- directory primitives (LDS only)
// The directory is an array of entries in document order.  Lane c owns the chunk [c*CH, (c+1)*CH) and keeps
// the sum of its active counts in a register (my_sum), so locating the k-th active element is one DPP scan
// over the 64 chunk sums plus one scan inside the owning chunk.
// chunk of directory position p: floor(p / CH) through the precomputed reciprocal (p < 2^16·CH is far inside its exact range)
LM_DEV uint32_t dir_chunk_at(const Tr& t, uint32_t p) { return (uint32_t)(((uint64_t)p * t.inv_CH) >> 32); }

// k-th active element (1 <= k <= tot_active) → directory position; k becomes the rank inside that leaf
LM_DEV uint32_t dir_find_kth(const Tr& t, uint32_t& k) {
  int lane = lmw::lane();
  lmw::wave_sync();
  uint32_t inc = lmw::scan_incl_add(t.my_sum);
  uint64_t m = lmw::ballot(inc >= k);
  if (!m) return NONE;
  int owner = lmw::ffs64(m);
  k -= lmw::bcast(inc, owner) - lmw::bcast(t.my_sum, owner);
  uint32_t cbase = (uint32_t)owner * t.CH;
  for (uint32_t j0 = 0; j0 < t.CH; j0 += 64) {
    uint32_t j = j0 + (uint32_t)lane, i = cbase + j;
    uint32_t a = (j < t.CH && i < t.n_dir) ? de_act(t.dir[i]) : 0;
    uint32_t inc2 = lmw::scan_incl_add(a);
    uint64_t m2 = lmw::ballot(inc2 >= k);
    if (m2) {
      int s = lmw::ffs64(m2);
      k -= lmw::bcast(inc2, s) - lmw::bcast(a, s);
      return cbase + j0 + (uint32_t)s;
    }
    k -= lmw::bcast(inc2, 63);
  }
  return NONE;
}
// directory position of leaf L (searches only the chunk that holds it)
LM_DEV uint32_t dir_find_leaf_in(const Tr& t, uint32_t L, uint32_t chunk);
LM_DEV uint32_t dir_find_leaf(const Tr& t, uint32_t L) {
  lmw::wave_sync();
  return dir_find_leaf_in(t, L, lmw::first((uint32_t)t.lchunk[L]));
}
LM_DEV uint32_t dir_find_leaf_in(const Tr& t, uint32_t L, uint32_t chunk) {
  int lane = lmw::lane();
  lmw::wave_sync();
  uint32_t cbase = chunk * t.CH;
  for (uint32_t j0 = 0; j0 < t.CH; j0 += 64) {
    uint32_t j = j0 + (uint32_t)lane, i = cbase + j;
    bool hit = j < t.CH && i < t.n_dir && de_leaf(t.dir[i]) == L;
    uint64_t m = lmw::ballot(hit);
    if (m) return cbase + j0 + (uint32_t)lmw::ffs64(m);
  }
  return NONE;
}
// replace the entry at position p (same leaf, new counts); `chunk` is the chunk of p
LM_DEV void dir_update(Tr& t, uint32_t p, uint32_t chunk, uint32_t old_e, uint32_t new_e) {
  int lane = lmw::lane();
  if (lane == 0) t.dir[p] = new_e;
  if ((uint32_t)lane == chunk) t.my_sum += de_act(new_e) - de_act(old_e);
  t.tot_active += de_act(new_e) - de_act(old_e);
}
// insert entry e right after position p (`chunk` = chunk of p)
LM_DEV void dir_insert_after(Tr& t, uint32_t p, uint32_t chunk, uint32_t e) {
  int lane = lmw::lane();
  if (t.n_dir >= t.dir_cap) { t.err = ST_RETRY; return; }   // optimistic directory size exceeded
  if (t.n_dir >= 64 * t.CH) { LM_SETERR(t.err, ST_INTERNAL); return; }
  lmw::wave_sync();
  uint32_t q = p + 1, n_old = t.n_dir;
  uint32_t cq = q == (chunk + 1) * t.CH ? chunk + 1 : chunk;
  // chunk bookkeeping: every chunk right of q gains the last entry of its left neighbour and loses its own last
  {
    uint32_t c = (uint32_t)lane;
    uint32_t first = c * t.CH, lastp = first + t.CH - 1;
    uint32_t gain = NONE, lose = NONE;
    if (c == cq) gain = e;
    else if (first > q && first <= n_old) gain = t.dir[first - 1];
    if (lastp >= q && lastp < n_old) lose = t.dir[lastp];
    if (c == cq && lastp < q) lose = NONE;
    if (gain != NONE) { t.my_sum += de_act(gain); t.lchunk[de_leaf(gain)] = (uint8_t)c; }
    if (lose != NONE) t.my_sum -= de_act(lose);
  }
  t.tot_active += de_act(e);
  lmw::wave_sync();
  for (uint32_t hi = n_old; hi > q;) {
    uint32_t c0 = hi > q + 64 ? hi - 64 : q;
    uint32_t i = c0 + (uint32_t)lane;
    bool in = i < hi;
    uint32_t v = in ? t.dir[i] : 0;
    lmw::wave_sync();
    if (in) t.dir[i + 1] = v;
    lmw::wave_sync();
    hi = c0;
  }
  if (lane == 0) t.dir[q] = e;
  t.n_dir++;
  lmw::wave_sync();
}

LM_DEV LeafRegs tr_leaf_load(const Tr& t, uint32_t L, uint32_t n) {
  int lane = lmw::lane();
  lmw::wave_sync();
  LeafRegs r;
  r.n = n;
  bool in = (uint32_t)lane < n;
  r.id = in ? t.it[L * 256 + lane] : NONE;
  r.ol = in ? t.it[L * 256 + 64 + lane] : NONE;
  r.orr = in ? t.it[L * 256 + 128 + lane] : NONE;
  r.st = in ? t.it[L * 256 + 192 + lane] : ST_FUT;
  return r;
}

// position comparison of two elements (by packed id): -1 a before b, 0 same, +1 a after b
LM_DEV int tr_cmp_pos(Tr& t, uint32_t a, uint32_t b) {
  int lane = lmw::lane();
  if (a == b) return 0;
  lmw::wave_sync();
  uint32_t la = t.loc[tr_g(t, a)], lb = t.loc[tr_g(t, b)];
  if (la >= t.n_leaf || lb >= t.n_leaf) { LM_SETERR(t.err, ST_INTERNAL); return 0; }
  if (la == lb) {
    uint32_t id = t.it[la * 256 + lane];
    uint32_t pa = dir_find_leaf(t, la);
    if (pa == NONE) { LM_SETERR(t.err, ST_INTERNAL); return 0; }
    uint32_t n = de_n(lmw::first(t.dir[pa]));
    int sa = lmw::ffs64(lmw::ballot((uint32_t)lane < n && id == a)), sb = lmw::ffs64(lmw::ballot((uint32_t)lane < n && id == b));
    return sa < sb ? -1 : 1;
  }
  uint32_t pa = dir_find_leaf(t, la), pb = dir_find_leaf(t, lb);
  return pa < pb ? -1 : 1;
}

// write `cnt` consecutive items of the logical sequence Q into leaf `dst` starting at Q index q0.
// Q = old[0,ins) ++ new run[0,len) ++ old[ins,n).  `old` lives in registers (R), new items are synthesised.
LM_DEV uint32_t tr_write_items(Tr& t, uint32_t dst, uint32_t q0, uint32_t cnt, const LeafRegs& R, uint32_t ins, uint32_t len,
                               uint32_t pid0, uint32_t ol0, uint32_t orr, bool update_loc_old, uint32_t first_changed,
                               bool& nf, LeafRegs* out = nullptr) {
  int lane = lmw::lane();
  uint32_t q = q0 + (uint32_t)lane;
  bool in = (uint32_t)lane < cnt;
  bool is_new = q >= ins && q < ins + len;
  uint32_t src = q < ins ? q : (q >= ins + len ? q - len : 0);
  uint32_t vid = lmw::shfl(R.id, (int)(src & 63)), vol = lmw::shfl(R.ol, (int)(src & 63));
  uint32_t vor = lmw::shfl(R.orr, (int)(src & 63)), vst = lmw::shfl(R.st, (int)(src & 63));
  if (is_new) {
    uint32_t k = q - ins;
    vid = pid0 + k;
    vol = k == 0 ? ol0 : pid0 + k - 1;
    vor = orr;
    vst = 0;
  }
  if (in && (uint32_t)lane >= first_changed) {  // slots before the cursor keep their contents when rewriting in place
    t.it[dst * 256 + lane] = vid; t.it[dst * 256 + 64 + lane] = vol; t.it[dst * 256 + 128 + lane] = vor; t.it[dst * 256 + 192 + lane] = vst;
    if (is_new || update_loc_old) t.loc[tr_g(t, vid)] = dst;
  }
  if (out) {  // the new contents of `dst`, for the register-resident leaf cache
    out->n = cnt;
    out->id = in ? vid : NONE; out->ol = in ? vol : NONE; out->orr = in ? vor : NONE; out->st = in ? vst : ST_FUT;
  }
  nf = lmw::ballot(in && !(vst & ST_FUT)) != 0;
  return (uint32_t)lmw::popc64(lmw::ballot(in && st_active(vst)));
}

// Insert run [pid0, pid0+len) at (directory position p, index ins) with origins (ol0, orr).
// R holds the registers of that leaf when `have_R`.  Long runs are placed in pieces of <= 64.
LM_DEV void tr_place_run(Tr& t, uint32_t p, uint32_t ins, uint32_t pid0, uint32_t len, uint32_t ol0, uint32_t orr,
                         LeafRegs R, bool have_R) {
  uint32_t done = 0;
  while (done < len && !t.err) {
    uint32_t piece = len - done > 64 ? 64 : len - done;
    lmw::wave_sync();
    uint32_t e = lmw::first(t.dir[p]);
    uint32_t L = de_leaf(e), n = de_n(e), old_act = de_act(e);
    if (!have_R) { if (L == t.cache_leaf) R = t.cr; else R = tr_leaf_load(t, L, n); }
    have_R = false;
    uint32_t total = n + piece;
    uint32_t p_ol = done == 0 ? ol0 : pid0 + done - 1;
    uint32_t chunk = dir_chunk_at(t, p);
    (void)old_act;
    if (total <= 64) {
      bool nf;
      uint32_t na = tr_write_items(t, L, 0, total, R, ins, piece, pid0 + done, p_ol, orr, false, ins, nf, &t.cr);
      t.cache_leaf = L; t.cache_p = p;
      dir_update(t, p, chunk, e, de_make(L, total, na, nf));
      ins += piece;
    } else {
      // even split into two leaves (total <= 128); both halves keep >= 32 elements
      if (t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
      uint32_t NL = t.n_leaf++;
      t.cache_leaf = NONE;
      uint32_t left = (total + 1) / 2, right = total - left;
      LeafRegs outL, outR;
      bool nf_l, nf_r;
      uint32_t na_l = tr_write_items(t, L, 0, left, R, ins, piece, pid0 + done, p_ol, orr, false, ins < left ? ins : left, nf_l, &outL);
      uint32_t na_r = tr_write_items(t, NL, left, right, R, ins, piece, pid0 + done, p_ol, orr, true, 0, nf_r, &outR);
      dir_update(t, p, chunk, e, de_make(L, left, na_l, nf_l));
      dir_insert_after(t, p, chunk, de_make(NL, right, na_r, nf_r));
      uint32_t endq = ins + piece;  // Q index right after the piece
      if (endq <= left) { ins = endq; t.cr = outL; t.cache_leaf = L; t.cache_p = p; }
      else { p = p + 1; ins = endq - left; t.cr = outR; t.cache_leaf = NL; t.cache_p = p; }
      if (t.err) t.cache_leaf = NONE;
    }
    done += piece;
    lmw::wave_sync();
  }
}

// Fugue integrate of one insert run at active position `pos` (crdt_rope.rs:63-247)
LM_DEV void tr_insert(Tr& t, uint32_t pos, uint32_t pid0, uint32_t len) {
  int lane = lmw::lane();
  PROF_T0();
  PROF_CNT(t, PF_NINS, 1);
  uint32_t p, ins, origin_left = NONE;
  t.beyond |= pos > t.tot_active ? 1u : 0u;    // (lm_k_integrate_span.h ts_insert: the reference places such a row behind trailing tombstones too — LM_DATA_CORRUPTION when the document is closed)
  if (pos > t.tot_active) pos = t.tot_active;  // beyond the end: clamp (query "missing" case)
  if (pos == 0) { p = 0; ins = 0; }
  else {
    uint32_t k = pos;
    p = dir_find_kth(t, k);
    if (p == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
    ins = k;  // rank inside the leaf, resolved to a slot below
  }
  lmw::wave_sync();
  PROF_ADD(t, PF_FIND);
  uint32_t e0 = lmw::first(t.dir[p]);
  LeafRegs R;
  if (de_leaf(e0) == t.cache_leaf) { R = t.cr; PROF_CNT(t, PF_NHIT, 1); }
  else R = tr_leaf_load(t, de_leaf(e0), de_n(e0));
  if (pos != 0) {
    // slot of the k-th active element of this leaf; the cursor sits right after it
    uint64_t am = lmw::ballot(st_active(R.st));
    uint32_t below = (uint32_t)lmw::popc64(am & ((2ull << lane) - 1));
    uint64_t hit = lmw::ballot(((am >> lane) & 1) && below == ins);
    if (!hit) { LM_SETERR(t.err, ST_INTERNAL); return; }
    int slot = lmw::ffs64(hit);
    origin_left = lmw::bcast(R.id, slot);
    ins = (uint32_t)slot + 1;
  }
  PROF_ADD(t, PF_LEAF);
  // origin_right = first non-future element at/after the cursor; everything before it is "in between".
  // Leaves holding only future elements (a concurrent run being skipped) are passed over in the LDS directory
  // through the entries' non-future bit: at most ONE leaf is loaded here, and the sibling scan reuses it.
  uint32_t origin_right = NONE, r_ol = NONE, r_p = NONE, r_slot = 0;
  bool between = false;
  LeafRegs RR = R;   // the leaf that holds origin_right
  {
    uint64_t nf = lmw::ballot((uint32_t)lane >= ins && (uint32_t)lane < R.n && !(R.st & ST_FUT));
    if (nf) {
      int s = lmw::ffs64(nf);
      r_p = p; r_slot = (uint32_t)s;
      if ((uint32_t)s > ins) between = true;
    } else {
      if (R.n > ins) between = true;
      lmw::wave_sync();
      for (uint32_t q0 = p + 1; q0 < t.n_dir && r_p == NONE; q0 += 64) {
        uint32_t q = q0 + (uint32_t)lane;
        uint64_t hm = lmw::ballot(q < t.n_dir && de_nf(t.dir[q]));
        if (hm) r_p = q0 + (uint32_t)lmw::ffs64(hm);
      }
      if (r_p != NONE) {
        if (r_p > p + 1) between = true;
        uint32_t e = lmw::first(t.dir[r_p]);
        RR = tr_leaf_load(t, de_leaf(e), de_n(e));
        PROF_CNT(t, PF_NEXTRA, 1);
        uint64_t nf2 = lmw::ballot((uint32_t)lane < RR.n && !(RR.st & ST_FUT));
        if (!nf2) { LM_SETERR(t.err, ST_INTERNAL); return; }   // directory bit out of sync with the leaf
        r_slot = (uint32_t)lmw::ffs64(nf2);
        if (r_slot > 0) between = true;
      } else if (p + 1 < t.n_dir) between = true;
    }
    if (r_p != NONE) { origin_right = lmw::bcast(RR.id, (int)r_slot); r_ol = lmw::bcast(RR.ol, (int)r_slot); }
  }
  PROF_ADD(t, PF_ORIGHT);
  uint32_t ins_p = p, ins_idx = ins;
  if (between) {
    // Sibling scan over the future elements between the cursor and origin_right (crdt_rope.rs:156-237).
    // Only run HEADS need the sequential rule: an element whose origin_left is the element physically right
    // before it is a continuation of a visited run — it can neither break the scan nor be a sibling — so a
    // whole run is skipped with one ballot.  "visited" (crdt_rope.rs:161,177-181) is the set of in-between
    // elements already passed, i.e. membership is a position test against [cursor, current).
    bool parent_right = origin_right != NONE && r_ol == origin_left;
    bool scanning = false;
    uint32_t cp = p, ci = ins;
    LeafRegs C = R;
    uint32_t my_peer = pid_peer(pid0);
    uint32_t carry_id = NONE;   // id of the element physically before slot 0 of the current leaf (within the scan)
    bool stop = false;
    for (uint32_t guard = 0; guard <= t.n_dir && !t.err && !stop; guard++) {
      uint32_t limit = (origin_right != NONE && cp == r_p) ? r_slot : C.n;
      uint32_t prev_id = lmw::shfl_up(C.id, 1);
      if (lane == 0) prev_id = carry_id;
      bool inr = (uint32_t)lane >= ci && (uint32_t)lane < limit;
      // (an element whose origin_left is OUR origin_left is a sibling and always examined — the element right
      //  after the cursor is the usual case, also when it is the first slot of the next leaf)
      bool is_head = inr && (C.ol != prev_id || C.ol == origin_left);
      uint64_t hm = lmw::ballot(is_head);
      while (hm && !stop && !t.err) {
        int h = lmw::ffs64(hm);
        hm &= hm - 1;
        PROF_CNT(t, PF_NHEAD, 1);
        // continuation elements in [ci, h) extend the visited runs
        if (!scanning && (uint32_t)h > ci) { ins_p = cp; ins_idx = (uint32_t)h; }
        uint32_t o_id = lmw::bcast(C.id, h), o_ol = lmw::bcast(C.ol, h), o_or = lmw::bcast(C.orr, h);
        if (o_ol != origin_left) {
          // is o_ol one of the in-between elements already visited?  (position in [cursor, (cp,h)))
          bool visited = false;
          uint64_t here = lmw::ballot((uint32_t)lane < C.n && C.id == o_ol);
          uint64_t in_r = cp != p ? lmw::ballot((uint32_t)lane < R.n && R.id == o_ol) : 0ull;   // cursor leaf (still in registers)
          if (here) {
            uint32_t xs = (uint32_t)lmw::ffs64(here);
            visited = xs < (uint32_t)h && (cp != p || xs >= ins);
          } else if (in_r) {
            visited = (uint32_t)lmw::ffs64(in_r) >= ins;
          } else if (o_ol != NONE && pid_ctr(o_ol) < t.cur[pid_peer(o_ol)]) {
            visited = false;   // inside the tracker's version ⇒ not future ⇒ cannot be one of the in-between elements
          } else if (o_ol != NONE) {
            lmw::wave_sync();
            uint32_t xl = t.loc[tr_g(t, o_ol)];
            if (xl < t.n_leaf) {
              uint32_t xp = dir_find_leaf(t, xl);
              if (xp != NONE && xp >= p && xp <= cp) {
                if (xp > p && xp < cp) visited = true;
                else {
                  uint32_t xn = de_n(lmw::first(t.dir[xp]));
                  uint32_t xid = xp == cp ? C.id : ((uint32_t)lane < xn ? t.it[xl * 256 + lane] : NONE);
                  uint64_t xm = lmw::ballot((uint32_t)lane < xn && xid == o_ol);
                  if (xm) {
                    uint32_t xs = (uint32_t)lmw::ffs64(xm);
                    visited = (xp != p || xs >= ins) && (xp != cp || xs < (uint32_t)h);
                  }
                }
              }
            }
          }
          if (!visited) { stop = true; break; }
        } else {
          if (o_or == origin_right) {
            if (pid_peer(o_id) > my_peer) { stop = true; break; }
            scanning = false;
          } else {
            // the other element's right parent (crdt_rope.rs:205-216): its origin_right if that is a sibling too.
            // Its position (directory position, slot) comes with the lookup, so no second search is needed to
            // compare it with our origin_right at (r_p, r_slot).  The two leaves held in registers are tried first.
            uint32_t opr = NONE, o_p = NONE, o_s = 0;
            if (o_or != NONE) {
              uint32_t x_ol;
              uint64_t hc = lmw::ballot((uint32_t)lane < C.n && C.id == o_or);
              uint64_t hr = (!hc && cp != p) ? lmw::ballot((uint32_t)lane < R.n && R.id == o_or) : 0ull;
              if (hc) { o_s = (uint32_t)lmw::ffs64(hc); o_p = cp; x_ol = lmw::bcast(C.ol, (int)o_s); }
              else if (hr) { o_s = (uint32_t)lmw::ffs64(hr); o_p = p; x_ol = lmw::bcast(R.ol, (int)o_s); }
              else {
                lmw::wave_sync();
                uint32_t xl = t.loc[tr_g(t, o_or)];
                if (xl >= t.n_leaf) { LM_SETERR(t.err, ST_INTERNAL); break; }
                uint32_t xp = dir_find_leaf(t, xl);
                if (xp == NONE) { LM_SETERR(t.err, ST_INTERNAL); break; }
                uint32_t xn = de_n(lmw::first(t.dir[xp]));
                uint32_t xid = (uint32_t)lane < xn ? t.it[xl * 256 + lane] : NONE;
                uint32_t xol = (uint32_t)lane < xn ? t.it[xl * 256 + 64 + lane] : NONE;
                uint64_t xm = lmw::ballot(xid == o_or);
                if (!xm) { LM_SETERR(t.err, ST_INTERNAL); break; }
                o_s = (uint32_t)lmw::ffs64(xm); o_p = xp; x_ol = lmw::bcast(xol, (int)o_s);
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
        if (!scanning) { ins_p = cp; ins_idx = (uint32_t)h + 1; }
        ci = (uint32_t)h + 1;
      }
      if (stop || t.err) break;
      // trailing continuation elements of this leaf
      if (!scanning && limit > ci) { ins_p = cp; ins_idx = limit; }
      if (origin_right != NONE && cp == r_p) break;   // reached origin_right
      if (cp + 1 >= t.n_dir) break;
      carry_id = C.n ? lmw::bcast(C.id, (int)(C.n - 1)) : carry_id;
      cp++; ci = 0;
      if (origin_right != NONE && cp == r_p) C = RR;
      else { uint32_t e = lmw::first(t.dir[cp]); C = tr_leaf_load(t, de_leaf(e), de_n(e)); }
    }
  }
  PROF_ADD(t, PF_BETWEEN);
  tr_place_run(t, ins_p, ins_idx, pid0, len, origin_left, origin_right, R, ins_p == p);
  PROF_ADD(t, PF_PLACE);
}

// status update of the elements with ids [c0,c1) of `peer` (crdt_rope.rs:345-381 by id instead of by cursor)
enum { UPD_SET_FUT = 0, UPD_CLR_FUT = 1, UPD_DEL_INC = 2, UPD_DEL_DEC = 3 };
// (locating a delete's targets by the op's position through the LDS directory, with the ids only verified, was measured:
//  the two directory scans cost more issue slots than the loc[] gather they save — 66.6 → 76.7 ms per configs[1] step)
LM_DEV void tr_update_range(Tr& t, uint32_t peer, uint32_t c0, uint32_t c1, int mode) {
  int lane = lmw::lane();
  uint32_t eb = t.ebase[peer];
  if (c1 > t.end[peer]) c1 = t.end[peer];   // a damaged target range cannot make the walk longer than the peer's history
  for (uint32_t cb = c0; cb < c1 && !t.err; cb += 64) {
    lmw::wave_sync();
    uint32_t c = cb + (uint32_t)lane;
    bool valid = c < c1;
    uint32_t chi_ = cb + 64 < c1 ? cb + 64 : c1;
    if (t.cache_leaf != NONE) {
      // every target of this step inside the cached leaf?  then no gather and no leaf load are needed
      uint32_t lo_ = pid_make(peer, cb), hi_ = pid_make(peer, chi_ - 1);
      bool chit = (uint32_t)lane < t.cr.n && t.cr.id >= lo_ && t.cr.id <= hi_;
      uint64_t cm = lmw::ballot(chit);
      if ((uint32_t)lmw::popc64(cm) == chi_ - cb) {
        uint32_t Lc = t.cache_leaf;
        uint32_t p = t.cache_p;
        {
          uint32_t st = t.cr.st;
          if (chit) {
            if (mode == UPD_SET_FUT) st |= ST_FUT;
            else if (mode == UPD_CLR_FUT) st &= ~ST_FUT;
            else if (mode == UPD_DEL_INC) st = (st + ST_DEL1) | ST_EVER;
            else if (st & ST_DELMASK) st -= ST_DEL1;
            t.it[Lc * 256 + 192 + lane] = st;
          }
          t.cr.st = st;
          uint32_t e = lmw::first(t.dir[p]);
          uint32_t new_act = (uint32_t)lmw::popc64(lmw::ballot((uint32_t)lane < t.cr.n && st_active(st)));
          uint32_t new_e = de_make(Lc, t.cr.n, new_act, lmw::ballot((uint32_t)lane < t.cr.n && !(st & ST_FUT)) != 0);
          if (new_e != e) dir_update(t, p, dir_chunk_at(t, p), e, new_e);
          PROF_CNT(t, PF_NDHIT, 1);
          continue;
        }
      }
    }
    uint32_t lf = valid ? t.loc[eb + c] : NONE;
    if (valid && lf >= t.n_leaf) lf = NONE;  // not an element of this container (malformed target): ignored
    uint64_t pend = lmw::ballot(lf != NONE);
    uint32_t chi = cb + 64 < c1 ? cb + 64 : c1;
    uint32_t lo_pid = pid_make(peer, cb), hi_pid = pid_make(peer, chi - 1);
    while (pend) {
      int l0 = lmw::ffs64(pend);
      uint32_t Lf = lmw::bcast(lf, l0);
      pend &= ~lmw::ballot(lf == Lf);
      // the leaf's chunk byte, ids and statuses are fetched together (one round trip); counts come from LDS
      uint32_t cbyte = t.lchunk[Lf];
      uint32_t id = t.it[Lf * 256 + lane];
      uint32_t st = t.it[Lf * 256 + 192 + lane];
      uint32_t p = dir_find_leaf_in(t, Lf, lmw::first(cbyte));
      if (p == NONE) continue;  // leaf of another container of the same document (malformed target)
      uint32_t e = lmw::first(t.dir[p]);
      uint32_t n = de_n(e);
      bool in = (uint32_t)lane < n;
      if (!in) { id = NONE; st = ST_FUT; }
      bool hit = in && id >= lo_pid && id <= hi_pid;
      if (hit) {
        if (mode == UPD_SET_FUT) st |= ST_FUT;
        else if (mode == UPD_CLR_FUT) st &= ~ST_FUT;
        else if (mode == UPD_DEL_INC) st = (st + ST_DEL1) | ST_EVER;
        else if (st & ST_DELMASK) st -= ST_DEL1;
        t.it[Lf * 256 + 192 + lane] = st;
      }
      if (Lf == t.cache_leaf) t.cr.st = in ? st : t.cr.st;   // keep the register copy coherent
      uint32_t new_act = (uint32_t)lmw::popc64(lmw::ballot(in && st_active(st)));
#ifdef LM_EMU_CHECK
      if (lane == 0 && getenv("LM_DBG")) fprintf(stderr, "  upd peer=%u [%u,%u) mode=%d leaf=%u p=%u chunk=%u act %u->%u CH=%u\n", peer, c0, c1, mode, Lf, p, (unsigned)t.lchunk[Lf], de_act(e), new_act, t.CH);
#endif
      uint32_t new_e = de_make(Lf, n, new_act, lmw::ballot(in && !(st & ST_FUT)) != 0);
      if (new_e != e) dir_update(t, p, dir_chunk_at(t, p), e, new_e);
    }
  }
}

// id of the active element at position `pos` (0-based), NONE beyond the end (MovableList moves, tracker.rs:289-347)
LM_DEV uint32_t tr_active_id_at(Tr& t, uint32_t pos) {
  if (pos >= t.tot_active) return NONE;
  int lane = lmw::lane();
  uint32_t k = pos + 1;
  uint32_t p = dir_find_kth(t, k);
  if (p == NONE) return NONE;
  lmw::wave_sync();
  uint32_t e0 = lmw::first(t.dir[p]);
  LeafRegs R;
  if (de_leaf(e0) == t.cache_leaf) R = t.cr;
  else R = tr_leaf_load(t, de_leaf(e0), de_n(e0));
  uint64_t am = lmw::ballot((uint32_t)lane < R.n && st_active(R.st));
  uint32_t below = (uint32_t)lmw::popc64(am & ((2ull << lane) - 1));
  uint64_t hit = lmw::ballot(((am >> lane) & 1) && below == k);
  if (!hit) return NONE;
  return lmw::bcast(R.id, lmw::ffs64(hit));
}

// retreat (dir < 0) / forward (dir > 0) every op of `peer` with id in [c0,c1) that belongs to container `cidx`
LM_DEV void tr_move_ops(Tr& t, const Dev& d, const DocMeta& m, uint32_t cidx, uint32_t peer, uint32_t c0, uint32_t c1, int dir) {
  uint32_t ci = find_change(d, m, peer, c0);
  if (ci == NONE) return;
  uint32_t hi = d.peer_chg1[m.praw0 + peer];
  for (; ci < hi && !t.err; ci++) {
    uint32_t crow = d.chg_sorted[m.chg0 + ci];
    const ChangeRow ch = d.chg[crow];
    if (ch.ctr >= c1) break;
    if (!((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1)) continue;   // no op of this change belongs to the container
    // first row whose end is past c0 (rows are counter-ordered inside a change)
    uint32_t lo = ch.op0, hr = ch.op0 + ch.n_op;
    while (lo < hr) { uint32_t mid = (lo + hr) >> 1; if (d.op[mid].ctr + d.op[mid].len <= c0) lo = mid + 1; else hr = mid; }
    for (uint32_t row = lo; row < ch.op0 + ch.n_op && !t.err; row++) {
      OpRow r = d.op[row];
      if (r.ctr >= c1) break;
      if ((r.cidx_kind & 0xffff) != cidx) continue;
      uint32_t kind = (r.cidx_kind >> 16) & 0xff;
      uint32_t a = (c0 > r.ctr ? c0 : r.ctr) - r.ctr, b = (c1 < r.ctr + r.len ? c1 : r.ctr + r.len) - r.ctr;
      if (a >= b) continue;
      // MovableList move: its own item, then (a second trip through the same call sites) the item it deleted, whose id the
      // first application left in the move item's payload slot
      uint32_t mv_tgt = NONE;
      if (kind == OK_LIST_MOVE) {
        lmw::wave_sync();
        mv_tgt = lmw::first((d.cp + (((uint64_t)m.elem0_hi << 32) | m.elem0_lo))[tr_g(t, pid_make(peer, r.ctr))]);
        kind = OK_LIST_INS;
      }
      for (;;) {
        if (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END) {
          tr_update_range(t, peer, r.ctr + a, r.ctr + b, dir < 0 ? UPD_SET_FUT : UPD_CLR_FUT);
        } else if (kind == OK_DEL) {
          uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
          uint32_t t0, t1;
          if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
          else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }  // op offset j deletes target + (L-1-j)
          tr_update_range(t, r.a0, t0, t1, dir < 0 ? UPD_DEL_DEC : UPD_DEL_INC);
        }
        if (mv_tgt == NONE || pid_peer(mv_tgt) >= m.n_peers) break;
        r.a0 = pid_peer(mv_tgt); r.a1 = pid_ctr(mv_tgt); r.a2 = 1; kind = OK_DEL; mv_tgt = NONE;   // (a, b) = (0, 1)
      }
    }
  }
}

#ifdef LM_EMU_CHECK
// debug-only (emulation): verify the directory against the leaves
inline bool tr_check(Tr& t, const char* what, uint32_t row) {
  bool ok = true;
  lmw::wave_sync();
  if (lmw::lane() == 0) {
    uint32_t tot = 0;
    for (uint32_t q = 0; q < t.n_dir && ok; q++) {
      uint32_t e = t.dir[q], L = de_leaf(e), a = 0;
      for (uint32_t i = 0; i < de_n(e); i++) a += st_active(t.it[L * 256 + 192 + i]) ? 1 : 0;
      bool nfb = false;
      for (uint32_t i = 0; i < de_n(e); i++) nfb |= !(t.it[L * 256 + 192 + i] & ST_FUT);
      if (nfb != de_nf(e)) { fprintf(stderr, "CHECK %s row=%u: dir[%u] leaf %u non-future bit %d, leaf says %d\n", what, row, q, L, (int)de_nf(e), (int)nfb); ok = false; }
      if (a != de_act(e)) { fprintf(stderr, "CHECK %s row=%u: dir[%u] leaf %u act=%u cached=%u n=%u\n", what, row, q, L, a, de_act(e), de_n(e)); ok = false; }
      tot += a;
    }
    if (ok && tot != t.tot_active) { fprintf(stderr, "CHECK %s row=%u: tot=%u cached=%u\n", what, row, tot, t.tot_active); ok = false; }
  }
  {
    uint32_t c = (uint32_t)lmw::lane(), sum = 0;
    for (uint32_t j = 0; j < t.CH; j++) {
      uint32_t i = c * t.CH + j;
      if (i < t.n_dir) {
        sum += de_act(t.dir[i]);
        if (t.lchunk[de_leaf(t.dir[i])] != c) { fprintf(stderr, "CHECK %s row=%u: lchunk[leaf %u]=%u expected %u\n", what, row, de_leaf(t.dir[i]), t.lchunk[de_leaf(t.dir[i])], c); ok = false; }
      }
    }
    if (sum != t.my_sum) {
      fprintf(stderr, "CHECK %s row=%u: lane %u my_sum=%u expected %u (CH=%u n_dir=%u)\n", what, row, c, t.my_sum, sum, t.CH, t.n_dir); ok = false;
      for (uint32_t q = 0; q < t.n_dir; q++) fprintf(stderr, " %u:%u/%u/%u", q, de_leaf(t.dir[q]), de_n(t.dir[q]), de_act(t.dir[q]));
      fprintf(stderr, "\n");
    }
  }
  return lmw::any(!ok) ? false : true;
}
#define TR_CHECK(what, row) do { if (!t.err && !tr_check(t, what, row)) t.err = ST_INTERNAL; } while (0)
#else
#define TR_CHECK(what, row) do {} while (0)
#endif

// K9: one wave per document — replay every sequence container from the empty version.
// Dynamic LDS: [dir_cap] directory entries, then MAX_PEERS element bases, then MAX_PEERS tracker versions.
LM_KERNEL LM_WAVES_PER_SIMD(6) void k_integrate(Dev d, DevDag g, uint32_t dir_cap, uint32_t pmax, const OpRow* __restrict__ op_ro,
                           const ChangeRow* __restrict__ chg_ro, const uint32_t* __restrict__ sorted_ro,
                           const uint32_t* __restrict__ skip_ro, const uint32_t* __restrict__ vvh_ro,
                           uint32_t retry_pass, uint32_t* retry_count) {
  uint32_t doc = d.doc_order[(uint32_t)lmw::bid()];
  int lane = lmw::lane();
  LM_DYN_SHARED(uint32_t, s_mem);
  uint32_t* s_dir = s_mem;                       // [dir_cap]
  uint32_t* s_ebase = s_mem + dir_cap;           // [pmax]
  uint32_t* s_cur = s_ebase + pmax;              // [pmax]
  uint32_t* s_end = s_cur + pmax;                // [pmax] version being rendered (applied end, or the checkout target)
  DocMeta m = d.doc[doc];
  uint64_t elem0 = ((uint64_t)m.elem0_hi << 32) | m.elem0_lo;
  if (retry_pass && m.status != ST_RETRY) return;  // second launch: only the documents whose optimistic directory overflowed
  if (status_fatal(m.status) && !retry_pass) return;
  // the element map starts empty (this wave owns it: no separate fill pass over the whole batch)
  for (uint32_t i = (uint32_t)lane; i < m.atoms; i += 64) d.loc[elem0 + i] = NONE;
  if (retry_pass) {
    for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) {   // sequence containers only: a Map's flag belongs to k_map_lww
      uint32_t ck = d.cont[m.cid0 + c].kind_root & 0xff;
      if (ck == CK_TEXT || ck == CK_LIST || ck == CK_MOVABLE) d.cont[m.cid0 + c].touched = 0;
    }
    lmw::block_sync();  // every lane has read the status before it is cleared
    if (lane == 0) d.doc[doc].status = ST_OK;
    m.status = ST_OK;
    lmw::block_sync();
  }
  if (status_fatal(m.status)) return;
  uint32_t P = m.n_peers;
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_ebase[p] = d.elem_base[m.praw0 + p]; s_end[p] = d.peer_end[m.praw0 + p]; }
  lmw::block_sync();
  uint64_t vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
  Tr t;
  t.it = d.it + (uint64_t)m.leaf0 * 256;
  t.loc = d.loc + elem0;
  t.ebase = s_ebase;
  t.cur = s_cur;
  t.end = s_end;
  t.dir = s_dir;
  t.lchunk = d.lf_chunk + m.leaf0;
  t.dir_cap = dir_cap;
  t.leaf_cap = m.leaf_cap;
  t.n_leaf = 0;
  t.err = 0; t.beyond = 0; t.posmis = 0;
#ifdef LM_PROF
  for (int i = 0; i < PF_N; i++) t.prof[i] = 0;
  uint64_t pf_begin = lmw::clock();
#endif
  uint32_t dir_used = 0;  // directory entries already flushed to HBM by earlier containers of this doc
  if (m.leaf_cap > MAX_LEAVES_PER_DOC || (retry_pass && m.leaf_cap > dir_cap) || P > pmax) { if (lane == 0) LM_SETERR(d.doc[doc].status, ST_UNSUPPORTED); return; }
  for (uint32_t cidx = 0; cidx < m.n_cont && !t.err; cidx++) {
    uint32_t kr = d.cont[m.cid0 + cidx].kind_root;
    uint32_t ckind = kr & 0xff;
    if (ckind != CK_TEXT && ckind != CK_LIST && ckind != CK_MOVABLE) continue;
    // fresh tracker: one empty leaf
    if (t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); break; }
    uint32_t L0 = t.n_leaf++;
    if (lane == 0) { s_dir[0] = de_make(L0, 0, 0, false); t.lchunk[L0] = 0; }
    t.n_dir = 1; t.tot_active = 0; t.my_sum = 0; t.cache_leaf = NONE;
    // chunk size: the leaves this container can still create spread over 64 lanes, kept odd
    t.CH = ((m.leaf_cap - L0 + 63) / 64) | 1u;
    if (t.CH < 3) t.CH = 3;  // keeps the 32-bit reciprocal below representable
    t.inv_CH = 0xFFFFFFFFu / t.CH + 1u;
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = 0;
    lmw::block_sync();
    bool touched = false;
    for (uint32_t oi = 0; oi < m.n_nodes && !t.err; oi++) {
      uint32_t n = d.node_order[m.chg0 + oi];
      uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
      const uint32_t* vv = vvh_ro + vvh0 + (uint64_t)n * P;
      uint32_t node_peer = chg_ro[sorted_ro[m.chg0 + first]].peer;
      bool checked_out = false;
      for (uint32_t ci = first; ci <= last && !t.err; ci++) {
        uint32_t crow = sorted_ro[m.chg0 + ci];
        const ChangeRow ch = chg_ro[crow];
        uint32_t skip_to = ch.ctr + skip_ro[crow];
        uint32_t pe = s_end[node_peer];
        // documents with many containers: a change that does not touch this container contributes no rows
        uint32_t n_rows = ((lmw::first(d.chg_mask[2 * (uint64_t)crow + ((cidx >> 5) & 1)]) >> (cidx & 31)) & 1) ? ch.n_op : 0u;
        // the next row is fetched one iteration ahead with a VECTOR load (lanes 0-7, one dword each): scalar loads
        // share their wait counter with LDS, so a scalar prefetch would be waited for at the first directory access
        const uint32_t* op_w = (const uint32_t*)op_ro;
        uint32_t nx = (lane < 8 && n_rows) ? op_w[(uint64_t)ch.op0 * 8 + (uint32_t)lane] : 0u;
        for (uint32_t row = ch.op0; row < ch.op0 + n_rows && !t.err; row++) {
          PROF_T0();
          OpRow r;
          r.cidx_kind = lmw::bcast(nx, 0); r.prop = (int32_t)lmw::bcast(nx, 1); r.len = lmw::bcast(nx, 2); r.ctr = lmw::bcast(nx, 3);
          r.a0 = lmw::bcast(nx, 4); r.a1 = lmw::bcast(nx, 5); r.a2 = (int32_t)lmw::bcast(nx, 6); r.chg = lmw::bcast(nx, 7);
          if (row + 1 < ch.op0 + n_rows) nx = lane < 8 ? op_w[(uint64_t)(row + 1) * 8 + (uint32_t)lane] : 0u;
          if ((r.cidx_kind & 0xffff) != cidx) continue;
          if (r.ctr + r.len <= skip_to) continue;
          uint32_t kind = (r.cidx_kind >> 16) & 0xff;
          uint32_t a = skip_to > r.ctr ? skip_to - r.ctr : 0;  // already-known prefix of a sliced change
          touched = true;
          if (r.ctr + a >= pe) continue;                        // past the version being rendered (checkout)
          uint32_t b = r.ctr + r.len <= pe ? r.len : pe - r.ctr; // op offsets [a, b) are replayed
          if (!checked_out) {
            // move the tracker to the version the node's first op sees (tracker.rs:354-461)
            checked_out = true;
            for (uint32_t p = 0; p < P && !t.err; p++) {
              uint32_t cur = s_cur[p], tgt = vv[p];
              if (cur > tgt) tr_move_ops(t, d, m, cidx, p, tgt, cur, -1);
              else if (cur < tgt) tr_move_ops(t, d, m, cidx, p, cur, tgt, +1);
            }
            lmw::block_sync();
            for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = vv[p];
            lmw::block_sync();
            PROF_ADD(t, PF_CHECKOUT);
          }
          PROF_ADD(t, PF_ROW);
          // MovableList move (tracker.rs:289-347): the row is replayed as the delete of the active item at `from` (its id
          // is remembered in the move item's payload slot) and then as the insert of the op's own item at `to`
          uint32_t mv_to = NONE;
          if (kind == OK_LIST_MOVE) {
            uint32_t tgt = tr_active_id_at(t, (uint32_t)r.a2);
            if (tgt == NONE || (uint32_t)r.prop >= t.tot_active) { LM_SETERR(t.err, ST_DATA_CORRUPTION); break; }
            if (lane == 0) (d.cp + elem0)[tr_g(t, pid_make(node_peer, r.ctr))] = tgt;
            mv_to = (uint32_t)r.prop;
            r.a0 = pid_peer(tgt); r.a1 = pid_ctr(tgt); r.a2 = 1; kind = OK_DEL;
          }
          for (;;) {
            if (kind == OK_TEXT_INS || kind == OK_LIST_INS) {
              tr_insert(t, (uint32_t)r.prop + a, pid_make(node_peer, r.ctr + a), b - a);
              TR_CHECK("insert", row);
            } else if (kind == OK_DEL) {
              uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
              uint32_t t0, t1;
              if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
              else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }  // op offset j deletes target + (L-1-j)
              // every target of a delete that is applied for the first time is an ACTIVE element of this container — the writer saw it
              // when it deleted it (handler.rs:2245-2288).  This kernel deletes by id and does not compare with the row's position (the
              // span-granular kernels do, and finish a damaged row by position: ts_del_positional); the cheap half of that comparison is
              // kept: the active length must drop by exactly the row's length — a row that names elements nobody inserted, elements of
              // another container, deleted or future ones is LM_DATA_CORRUPTION, never a value the reference would not have computed
              // from it.  The other half (round 5, late): the targets must BE the elements at the row's positions — the reference deletes
              // what stands there (crdt_rope.rs:256-335), so a row re-pointed at other active elements would render a value the reference
              // does not compute (9 of 400 damaged rich-text documents did, tests/test_emu_richtext.py) — checked element by element before
              // the delete is applied; a mismatch is LM_DATA_CORRUPTION here (this kernel has no positional path: the host's cue to fall
              // back).  A move's delete half found its target by position already.
              if (mv_to == NONE && !t.err) {
                const int64_t start = r.a2 > 0 ? (int64_t)r.prop : (int64_t)r.prop + 1 - (int64_t)Ln;   // DeleteSpan::start (list_op.rs:303-309)
                for (uint32_t j = a; j < b && !t.err; j++) {
                  // offset j: positive span — the rest of the row stands at prop, prop + 1, … once offsets [0, a) are gone; negative span —
                  // offset j deletes target + (L-1-j), which stands at start + (L-1-j) (the offsets in front of it stood above)
                  const int64_t pos = r.a2 > 0 ? (int64_t)r.prop + (int64_t)(j - a) : start + (int64_t)(Ln - 1 - j);
                  const uint32_t idj = r.a2 > 0 ? r.a1 + j : r.a1 + (Ln - 1 - j);
                  if (pos < 0 || pos > 0x7fffffff || tr_active_id_at(t, (uint32_t)pos) != pid_make(r.a0, idj)) { LM_SETERR(t.err, ST_DATA_CORRUPTION); t.posmis = 1; }
                }
              }
              const uint32_t act0 = t.tot_active;
              tr_update_range(t, r.a0, t0, t1, UPD_DEL_INC);
              if ((act0 - t.tot_active) != (t1 - t0) || Ln != r.len) { LM_SETERR(t.err, ST_DATA_CORRUPTION); t.posmis |= Ln == r.len ? 1u : 0u; }
              PROF_ADD(t, PF_DELETE);
              PROF_CNT(t, PF_NDEL, 1);
              TR_CHECK("delete", row);
            } else if (kind == OK_STYLE_START) {
              tr_insert(t, (uint32_t)r.prop, pid_make(node_peer, r.ctr), 1);
            } else if (kind == OK_STYLE_END) {
              // diff_calc.rs:1105-1119: the matching StyleStart is the op right before (same peer, counter-1)
              uint32_t end_pos = NONE;
              if (row > ch.op0) {
                const OpRow pr = op_ro[row - 1];
                if (((pr.cidx_kind >> 16) & 0xff) == OK_STYLE_START && pr.ctr + 1 == r.ctr && (pr.cidx_kind & 0xffff) == cidx)
                  end_pos = (uint32_t)pr.prop + pr.a0;
              }
              if (end_pos == NONE) { LM_SETERR(t.err, ST_UNSUPPORTED); break; }
              uint32_t pos = end_pos + 1 < t.tot_active ? end_pos + 1 : t.tot_active;
              tr_insert(t, pos, pid_make(node_peer, r.ctr), 1);
            }
            if (mv_to == NONE || t.err) break;
            r.prop = (int32_t)mv_to; kind = OK_LIST_INS; mv_to = NONE;   // second half of a move: the new item
          }
        }
        // the node's own ops advance the tracker version
        if (checked_out && lane == 0) s_cur[node_peer] = ch.ctr + ch.len < pe ? ch.ctr + ch.len : pe;
      }
      lmw::block_sync();
    }
    // flush the directory (leaf order + counts) for the emit stage
    lmw::block_sync();
    if (dir_used + t.n_dir > m.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); break; }
    for (uint32_t i = (uint32_t)lane; i < t.n_dir; i += 64) d.dir_out[m.leaf0 + dir_used + i] = s_dir[i];
    // the state store holds a root sequence once something is visible in it (diff_calc.rs:299): any element never deleted
    bool alive = false;
    for (uint32_t q = 0; q < t.n_dir && touched; q++) {
      uint32_t e = s_dir[q];
      uint32_t st = (uint32_t)lane < de_n(e) ? t.it[(uint64_t)de_leaf(e) * 256 + 192 + lane] : ST_EVER;
      if (lmw::ballot(!(st & ST_EVER))) { alive = true; break; }
    }
    if (lane == 0) {
      d.cont_root0[m.cid0 + cidx] = dir_used;
      d.cont_nroot[m.cid0 + cidx] = t.n_dir;
      if (alive) d.cont[m.cid0 + cidx].touched = 1;
    }
    dir_used += t.n_dir;
    lmw::block_sync();
  }
  if (!t.err && t.beyond) t.err = ST_DATA_CORRUPTION;
  // a delete row whose targets are not the elements at its position is applied BY POSITION by the reference (crdt_rope.rs:256-335) and by
  // the span-granular batch kernels (k_integrate_span_pos); this kernel has no such path: the context replays the document through
  // those kernels (lm_capi_impl.h redo) — the verdict of a document must not depend on which kernel its batch's statistics picked
  if (t.err == ST_DATA_CORRUPTION && t.posmis && d.posdel_redo && lane == 0) lmw::atomic_or(&d.doc[doc].flags, DF_REDO);
  if (t.err && lane == 0) {
    d.doc[doc].status = t.err;
    if (t.err == ST_RETRY) lmw::atomic_add(retry_count, 1u);
  }
  if (lane == 0) d.doc[doc].pad0 = dir_used;  // leaves actually used (sizing diagnostics)
#ifdef LM_PROF
  t.prof[PF_TOTAL] = lmw::clock() - pf_begin;
  if (lane == 0) for (int i = 0; i < PF_N; i++) d.prof[(uint64_t)doc * PF_N + i] = t.prof[i];
#endif
}

}  // namespace lm
