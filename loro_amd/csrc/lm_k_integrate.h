// Integrate stage: Event-Graph-Walker replay of a sequence container (Text / List), one wavefront per
// document.  Element-granular restatement of the reference tracker, shaped for a 64-lane wave:
//   * the rope is a 3-level order-statistics tree: root array → 64-ary group nodes → 64-slot leaves;
//     a lane owns one slot, so a node scan is one coalesced load + one wave prefix-sum / ballot;
//   * every element keeps (id, origin_left, origin_right, status) — FugueSpan per element
//     (container/richtext/fugue_span.rs:191-207,257-279);
//   * `loc[element] → leaf` replaces IdToCursor (container/richtext/tracker/id_to_cursor.rs);
//   * deletes / retreat / forward address elements by id (the wire carries the target id span,
//     container/list/list_op.rs:132-142), so they never descend the tree.
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
LM_DEV bool st_active(uint32_t st) { return (st & (ST_FUT | ST_DELMASK)) == 0; }

struct Tr {  // wave-uniform context of one (document, sequence container) replay
  uint32_t *it_id, *it_ol, *it_or, *it_st;
  uint32_t *lf_n, *lf_next, *lf_grp;
  uint32_t *gp_leaf, *gp_act, *gp_n;
  uint32_t *rt_grp, *rt_act;      // already offset to this container's root array
  uint32_t* loc;                  // doc element → leaf
  const uint32_t* ebase;          // LDS: element base per peer
  unsigned long long* vis;        // scratch runs for the sibling scan
  uint32_t n_leaf, leaf_cap, n_grp, grp_cap, n_root, root_cap;
  uint32_t tot_active;
  uint32_t first_leaf;
  int32_t err;
};

LM_DEV uint32_t tr_g(const Tr& t, uint32_t pid) { return t.ebase[pid_peer(pid)] + pid_ctr(pid); }

LM_DEV uint32_t tr_root_find(const Tr& t, uint32_t G) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  for (uint32_t c = 0; c < t.n_root; c += 64) {
    uint32_t i = c + (uint32_t)lane;
    uint64_t m = lmw::ballot(i < t.n_root && t.rt_grp[i] == G);
    if (m) return c + (uint32_t)lmw::ffs64(m);
  }
  return NONE;
}
LM_DEV uint32_t tr_grp_slot(const Tr& t, uint32_t G, uint32_t L) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  uint64_t m = lmw::ballot((uint32_t)lane < t.gp_n[G] && t.gp_leaf[G * 64 + lane] == L);
  return m ? (uint32_t)lmw::ffs64(m) : NONE;
}
LM_DEV void tr_add_active(Tr& t, uint32_t L, int32_t delta) {
  if (delta == 0) return;
  int lane = lmw::lane();
  uint32_t G = t.lf_grp[L];
  uint32_t s = tr_grp_slot(t, G, L);
  uint32_t ri = tr_root_find(t, G);
  if (s == NONE || ri == NONE) {
#ifdef LM_EMU_TRACE
    if (lane == 0) fprintf(stderr, "add_active L=%u G=%u s=%u ri=%u n_leaf=%u n_grp=%u n_root=%u gp_n=%u\n", L, G, s, ri, t.n_leaf, t.n_grp, t.n_root, G < t.n_grp ? t.gp_n[G] : 999);
    if (lane == 0 && G < t.n_grp) { for (uint32_t q = 0; q < t.gp_n[G]; q++) fprintf(stderr, " [%u]=%u", q, t.gp_leaf[G * 64 + q]); fprintf(stderr, "\n"); }
#endif
    LM_SETERR(t.err, ST_INTERNAL); return; }
  if (lane == 0) {
    t.gp_act[G * 64 + s] += (uint32_t)delta;
    t.rt_act[ri] += (uint32_t)delta;
  }
  t.tot_active += (uint32_t)delta;
}

// insert (grp G, act) into the root array right after position `after`
LM_DEV void tr_root_insert(Tr& t, uint32_t after, uint32_t G, uint32_t act) {
  int lane = lmw::lane();
  if (t.n_root >= t.root_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
  // shift [after+1, n_root) right by one, processed from the tail in 64-entry chunks
  uint32_t lo = after + 1;
  for (uint32_t hi = t.n_root; hi > lo;) {
    uint32_t c0 = hi > lo + 64 ? hi - 64 : lo;
    uint32_t i = c0 + (uint32_t)lane;
    uint32_t a = 0, b = 0;
    bool in = i < hi;
    if (in) { a = t.rt_grp[i]; b = t.rt_act[i]; }
    lmw::wave_sync();
    if (in) { t.rt_grp[i + 1] = a; t.rt_act[i + 1] = b; }
    lmw::wave_sync();
    hi = c0;
  }
  if (lane == 0) { t.rt_grp[lo] = G; t.rt_act[lo] = act; }
  t.n_root++;
}
// insert leaf `NL` (active count act) into its predecessor's group right after leaf `L`
LM_DEV void tr_group_insert(Tr& t, uint32_t L, uint32_t NL, uint32_t act) {
  int lane = lmw::lane();
  uint32_t G = t.lf_grp[L];
  uint32_t n = t.gp_n[G];
  uint32_t s = tr_grp_slot(t, G, L);
  if (s == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
  uint32_t lf = (uint32_t)lane < n ? t.gp_leaf[G * 64 + lane] : NONE;
  uint32_t ac = (uint32_t)lane < n ? t.gp_act[G * 64 + lane] : 0;
  lmw::wave_sync();
  if (n < 64) {
    if ((uint32_t)lane > s && (uint32_t)lane < n) { t.gp_leaf[G * 64 + lane + 1] = lf; t.gp_act[G * 64 + lane + 1] = ac; }
    uint32_t ri = tr_root_find(t, G);
    if (ri == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
    if (lane == 0) {
      t.gp_leaf[G * 64 + s + 1] = NL; t.gp_act[G * 64 + s + 1] = act; t.gp_n[G] = n + 1; t.lf_grp[NL] = G;
      t.rt_act[ri] += act;
    }
    return;
  }
  // split the full group: 65 entries → 33 stay, 32 move to a new group
  if (t.n_grp >= t.grp_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
  uint32_t NG = t.n_grp++;
  // logical sequence q ∈ [0,65): q<=s → old[q]; q==s+1 → new; q>s+1 → old[q-1]
  uint32_t keep = 33;
  // entries of the new group: q = keep + lane, lane < 32
  {
    uint32_t q = keep + (uint32_t)lane;
    uint32_t src = q <= s ? q : (q == s + 1 ? 64u : q - 1);
    uint32_t vlf = lmw::shfl(lf, (int)(src & 63)), vac = lmw::shfl(ac, (int)(src & 63));
    if (src == 64u) { vlf = NL; vac = act; }
    if (lane < 32) { t.gp_leaf[NG * 64 + lane] = vlf; t.gp_act[NG * 64 + lane] = vac; t.lf_grp[vlf] = NG; }
  }
  // entries staying: q = lane < 33
  {
    uint32_t q = (uint32_t)lane;
    uint32_t src = q <= s ? q : (q == s + 1 ? 64u : q - 1);
    uint32_t vlf = lmw::shfl(lf, (int)(src & 63)), vac = lmw::shfl(ac, (int)(src & 63));
    if (src == 64u) { vlf = NL; vac = act; }
    if ((uint32_t)lane < keep) { t.gp_leaf[G * 64 + lane] = vlf; t.gp_act[G * 64 + lane] = vac; t.lf_grp[vlf] = G; }
  }
  if (lane == 0) { t.gp_n[G] = keep; t.gp_n[NG] = 32; }
  // root: recompute both groups' active sums
  uint32_t a_old = lmw::reduce_add((uint32_t)lane < keep ? t.gp_act[G * 64 + lane] : 0);
  uint32_t a_new = lmw::reduce_add(lane < 32 ? t.gp_act[NG * 64 + lane] : 0);
  uint32_t ri = tr_root_find(t, G);
  if (ri == NONE) { LM_SETERR(t.err, ST_INTERNAL); return; }
  if (lane == 0) t.rt_act[ri] = a_old;
  tr_root_insert(t, ri, NG, a_new);
}

struct LeafRegs { uint32_t n, id, ol, orr, st; };
LM_DEV LeafRegs tr_leaf_load(const Tr& t, uint32_t L) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  LeafRegs r;
  r.n = t.lf_n[L];
  bool in = (uint32_t)lane < r.n;
  r.id = in ? t.it_id[L * 64 + lane] : NONE;
  r.ol = in ? t.it_ol[L * 64 + lane] : NONE;
  r.orr = in ? t.it_or[L * 64 + lane] : NONE;
  r.st = in ? t.it_st[L * 64 + lane] : ST_FUT;
  return r;
}

// k-th active element (k >= 1, k <= tot_active) → (leaf, slot)
LM_DEV void tr_find_kth(const Tr& t, uint32_t k, uint32_t& leaf, uint32_t& slot) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  uint32_t G = NONE;
  for (uint32_t c = 0; c < t.n_root; c += 64) {
    uint32_t i = c + (uint32_t)lane;
    uint32_t a = i < t.n_root ? t.rt_act[i] : 0;
    uint32_t inc = lmw::scan_incl_add(a);
    uint32_t tot = lmw::bcast(inc, 63);
    if (k <= tot) {
      uint64_t m = lmw::ballot(inc >= k);
      int s = lmw::ffs64(m);
      k -= lmw::bcast(inc, s) - lmw::bcast(a, s);
      G = t.rt_grp[c + (uint32_t)s];
      break;
    }
    k -= tot;
  }
  if (G == NONE) { leaf = NONE; slot = 0; return; }
  {
    uint32_t a = (uint32_t)lane < t.gp_n[G] ? t.gp_act[G * 64 + lane] : 0;
    uint32_t inc = lmw::scan_incl_add(a);
    uint64_t m = lmw::ballot(inc >= k);
    if (!m) { leaf = NONE; slot = 0; return; }
    int s = lmw::ffs64(m);
    k -= lmw::bcast(inc, s) - lmw::bcast(a, s);
    leaf = t.gp_leaf[G * 64 + (uint32_t)s];
  }
  {
    uint32_t n = t.lf_n[leaf];
    uint32_t st = (uint32_t)lane < n ? t.it_st[leaf * 64 + lane] : ST_FUT;
    uint64_t am = lmw::ballot(st_active(st));
    uint32_t below = (uint32_t)lmw::popc64(am & ((2ull << lane) - 1));
    uint64_t hit = lmw::ballot(((am >> lane) & 1) && below == k);
    if (!hit) { leaf = NONE; slot = 0; return; }
    slot = (uint32_t)lmw::ffs64(hit);
  }
}

// position comparison of two elements (by packed id): -1 a before b, 0 same, +1 a after b
LM_DEV int tr_cmp_pos(Tr& t, uint32_t a, uint32_t b) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  if (a == b) return 0;
  uint32_t la = t.loc[tr_g(t, a)], lb = t.loc[tr_g(t, b)];
  if (la >= t.n_leaf || lb >= t.n_leaf) { LM_SETERR(t.err, ST_INTERNAL); return 0; }
  if (la == lb) {
    uint32_t id = (uint32_t)lane < t.lf_n[la] ? t.it_id[la * 64 + lane] : NONE;
    int sa = lmw::ffs64(lmw::ballot(id == a)), sb = lmw::ffs64(lmw::ballot(id == b));
    return sa < sb ? -1 : 1;
  }
  uint32_t ga = t.lf_grp[la], gb = t.lf_grp[lb];
  if (ga == gb) { uint32_t sa = tr_grp_slot(t, ga, la), sb = tr_grp_slot(t, ga, lb); return sa < sb ? -1 : 1; }
  uint32_t ra = tr_root_find(t, ga), rb = tr_root_find(t, gb);
  return ra < rb ? -1 : 1;
}

// visited id runs of the sibling scan (crdt_rope.rs:161,177-181)
struct Vis { uint32_t n; uint32_t lo, hi; bool open; };
LM_DEV bool vis_contains(const Tr& t, const Vis& v, uint32_t pid) {
  lmw::wave_sync();  // lane-0 stores of the previous step must land before any lane reloads (emulation rendezvous)
  int lane = lmw::lane();
  if (pid == NONE) return false;
  if (v.open && pid >= v.lo && pid <= v.hi) return true;
  for (uint32_t c = 0; c < v.n; c += 64) {
    uint32_t i = c + (uint32_t)lane;
    bool hit = false;
    if (i < v.n) { unsigned long long e = t.vis[i]; uint32_t lo = (uint32_t)(e >> 32), hi = (uint32_t)e; hit = pid >= lo && pid <= hi; }
    if (lmw::any(hit)) return true;
  }
  return false;
}
LM_DEV void vis_add(Tr& t, Vis& v, uint32_t pid) {
  if (v.open && pid == v.hi + 1 && pid_peer(pid) == pid_peer(v.hi)) { v.hi = pid; return; }
  if (v.open) {
    if (v.n >= VIS_CAP) { LM_SETERR(t.err, ST_UNSUPPORTED); return; }
    if (lmw::lane() == 0) t.vis[v.n] = ((unsigned long long)v.lo << 32) | v.hi;
    v.n++;
  }
  v.open = true; v.lo = pid; v.hi = pid;
}

// write `cnt` consecutive items of the logical sequence Q into leaf `dst` starting at Q index q0.
// Q = old[0,ins) ++ new run[0,len) ++ old[ins,n).  `old` lives in registers (R), new items are synthesised.
LM_DEV uint32_t tr_write_items(Tr& t, uint32_t dst, uint32_t q0, uint32_t cnt, const LeafRegs& R, uint32_t ins, uint32_t len,
                               uint32_t pid0, uint32_t run_off, uint32_t ol0, uint32_t orr, bool update_loc_old) {
  // run_off: index of the first element of this chunk of the run relative to pid0 (long runs are fed in pieces)
  int lane = lmw::lane();
  uint32_t q = q0 + (uint32_t)lane;
  bool in = (uint32_t)lane < cnt;
  bool is_new = q >= ins && q < ins + len;
  uint32_t src = q < ins ? q : (q >= ins + len ? q - len : 0);
  uint32_t vid = lmw::shfl(R.id, (int)(src & 63)), vol = lmw::shfl(R.ol, (int)(src & 63));
  uint32_t vor = lmw::shfl(R.orr, (int)(src & 63)), vst = lmw::shfl(R.st, (int)(src & 63));
  if (is_new) {
    uint32_t k = run_off + (q - ins);
    vid = pid0 + k;
    vol = k == 0 ? ol0 : pid0 + k - 1;
    vor = orr;
    vst = 0;
  }
  if (in) {
    t.it_id[dst * 64 + lane] = vid; t.it_ol[dst * 64 + lane] = vol; t.it_or[dst * 64 + lane] = vor; t.it_st[dst * 64 + lane] = vst;
    if (is_new || update_loc_old) t.loc[tr_g(t, vid)] = dst;
  }
  return (uint32_t)lmw::popc64(lmw::ballot(in && st_active(vst)));
}

// Insert run [pid0, pid0+len) at (leaf L, index ins) with origins (ol0, orr).  Long runs are placed in pieces.
LM_DEV void tr_place_run(Tr& t, uint32_t L, uint32_t ins, uint32_t pid0, uint32_t len, uint32_t ol0, uint32_t orr) {
  int lane = lmw::lane();
  uint32_t done = 0;
  while (done < len && !t.err) {
    // feed at most 64 new elements per step; the piece goes right after the previous piece
    uint32_t piece = len - done > 64 ? 64 : len - done;
    LeafRegs R = tr_leaf_load(t, L);
    uint32_t n = R.n;
    uint32_t old_act = (uint32_t)lmw::popc64(lmw::ballot((uint32_t)lane < n && st_active(R.st)));
    uint32_t total = n + piece;
    uint32_t p_ol = done == 0 ? ol0 : pid0 + done - 1;
    // origin_left of the first element of a later piece is the previous element of the run (same as in-run rule)
    if (total <= 64) {
      uint32_t na = tr_write_items(t, L, 0, total, R, ins, piece, pid0 + done, 0, p_ol, orr, false);
      if (lane == 0) t.lf_n[L] = total;
      tr_add_active(t, L, (int32_t)na - (int32_t)old_act);
      ins += piece;
    } else {
      // even split into two leaves (total <= 128)
      if (t.n_leaf >= t.leaf_cap) { LM_SETERR(t.err, ST_INTERNAL); return; }
      uint32_t NL = t.n_leaf++;
      uint32_t left = (total + 1) / 2, right = total - left;
      uint32_t na_l = tr_write_items(t, L, 0, left, R, ins, piece, pid0 + done, 0, p_ol, orr, false);
      uint32_t na_r = tr_write_items(t, NL, left, right, R, ins, piece, pid0 + done, 0, p_ol, orr, true);
      if (lane == 0) { t.lf_n[L] = left; t.lf_n[NL] = right; t.lf_next[NL] = t.lf_next[L]; t.lf_next[L] = NL; }
      tr_add_active(t, L, (int32_t)na_l - (int32_t)old_act);
      t.tot_active += na_r;
      tr_group_insert(t, L, NL, na_r);
      // continue after the piece: locate where the piece's last element landed
      uint32_t endq = ins + piece;  // Q index right after the piece
      if (endq <= left) { ins = endq; }
      else { L = NL; ins = endq - left; }
    }
    done += piece;
  }
}

// Fugue integrate of one insert run at active position `pos` (crdt_rope.rs:63-247)
LM_DEV void tr_insert(Tr& t, uint32_t pos, uint32_t pid0, uint32_t len) {
  int lane = lmw::lane();
  uint32_t L, ins, origin_left = NONE;
  if (pos > t.tot_active) pos = t.tot_active;  // beyond the end: clamp (query "missing" case)
  if (pos == 0) { L = t.first_leaf; ins = 0; }
  else {
    uint32_t slot;
    tr_find_kth(t, pos, L, slot);
    if (L == NONE) {
#ifdef LM_EMU_TRACE
      if (lane == 0) {
        uint32_t rs = 0; for (uint32_t q = 0; q < t.n_root; q++) rs += t.rt_act[q];
        fprintf(stderr, "find_kth pos=%u tot=%u rootsum=%u n_root=%u n_grp=%u n_leaf=%u pid0=%x len=%u\n", pos, t.tot_active, rs, t.n_root, t.n_grp, t.n_leaf, pid0, len);
        for (uint32_t q = 0; q < t.n_root; q++) { uint32_t G = t.rt_grp[q], gs = 0; for (uint32_t z = 0; z < t.gp_n[G]; z++) gs += t.gp_act[G * 64 + z]; fprintf(stderr, " root[%u] G=%u act=%u gsum=%u gn=%u\n", q, G, t.rt_act[q], gs, t.gp_n[G]); }
      }
#endif
      LM_SETERR(t.err, ST_INTERNAL); return; }
    ins = slot + 1;
  }
  LeafRegs R = tr_leaf_load(t, L);
  if (pos != 0) origin_left = lmw::bcast(R.id, (int)(ins - 1));
  // origin_right = first non-future element at/after the cursor; everything before it is "in between"
  uint32_t origin_right = NONE, r_ol = NONE, r_leaf = NONE, r_slot = 0;
  bool between = false;
  {
    uint32_t cl = L, from = ins;
    LeafRegs C = R;
    for (uint32_t guard = 0; guard <= t.n_leaf; guard++) {
      uint64_t nf = lmw::ballot((uint32_t)lane >= from && (uint32_t)lane < C.n && !(C.st & ST_FUT));
      if (nf) {
        int s = lmw::ffs64(nf);
        origin_right = lmw::bcast(C.id, s);
        r_ol = lmw::bcast(C.ol, s);
        r_leaf = cl; r_slot = (uint32_t)s;
        if ((uint32_t)s > from) between = true;
        break;
      }
      if (C.n > from) between = true;
      uint32_t nx = t.lf_next[cl];
      if (nx == NONE) break;
      cl = nx; from = 0;
      C = tr_leaf_load(t, cl);
    }
  }
  uint32_t ins_leaf = L, ins_idx = ins;
  if (between) {
    bool parent_right = origin_right != NONE && r_ol == origin_left;
    bool scanning = false;
    Vis v; v.n = 0; v.open = false; v.lo = v.hi = 0;
    uint32_t cl = L, ci = ins;
    LeafRegs C = R;
    uint32_t my_peer = pid_peer(pid0);
    for (uint32_t guard = 0; guard < (1u << 26) && !t.err; guard++) {
      if (ci >= C.n) {
        uint32_t nx = t.lf_next[cl];
        if (nx == NONE) break;
        cl = nx; ci = 0;
        C = tr_leaf_load(t, cl);
        continue;
      }
      if (origin_right != NONE && cl == r_leaf && ci == r_slot) break;
      uint32_t o_id = lmw::bcast(C.id, (int)ci), o_ol = lmw::bcast(C.ol, (int)ci), o_or = lmw::bcast(C.orr, (int)ci);
      if (o_ol != origin_left && !vis_contains(t, v, o_ol)) break;
      vis_add(t, v, o_id);
      if (o_ol == origin_left) {
        if (o_or == origin_right) {
          if (pid_peer(o_id) > my_peer) break;
          scanning = false;
        } else {
          uint32_t opr = NONE;
          if (o_or != NONE) {
            uint32_t xl = t.loc[tr_g(t, o_or)];
            if (xl >= t.n_leaf) { LM_SETERR(t.err, ST_INTERNAL); break; }
            uint32_t xid = (uint32_t)lane < t.lf_n[xl] ? t.it_id[xl * 64 + lane] : NONE;
            uint32_t xol = (uint32_t)lane < t.lf_n[xl] ? t.it_ol[xl * 64 + lane] : NONE;
            uint64_t xm = lmw::ballot(xid == o_or);
            if (!xm) { LM_SETERR(t.err, ST_INTERNAL); break; }
            if (lmw::bcast(xol, lmw::ffs64(xm)) == origin_left) opr = o_or;
          }
          int c;
          if (opr != NONE && parent_right) c = tr_cmp_pos(t, opr, origin_right);
          else if (opr != NONE) c = -1;
          else if (parent_right) c = 1;
          else c = 0;
          if (c < 0) scanning = true;
          else if (c == 0 && pid_peer(o_id) > my_peer) break;
          else scanning = false;
        }
      }
      if (!scanning) { ins_leaf = cl; ins_idx = ci + 1; }
      ci++;
    }
  }
  tr_place_run(t, ins_leaf, ins_idx, pid0, len, origin_left, origin_right);
}

// status update of the elements with ids [c0,c1) of `peer` (crdt_rope.rs:345-381 by id instead of by cursor)
enum { UPD_SET_FUT = 0, UPD_CLR_FUT = 1, UPD_DEL_INC = 2, UPD_DEL_DEC = 3 };
LM_DEV void tr_update_range(Tr& t, uint32_t peer, uint32_t c0, uint32_t c1, int mode) {
  int lane = lmw::lane();
  uint32_t eb = t.ebase[peer];
  for (uint32_t cb = c0; cb < c1 && !t.err; cb += 64) {
    lmw::wave_sync();
    uint32_t c = cb + (uint32_t)lane;
    bool valid = c < c1;
    uint32_t lf = valid ? t.loc[eb + c] : NONE;
    if (valid && lf >= t.n_leaf) lf = NONE;  // not an element of this container (malformed target): ignored
    uint64_t pend = lmw::ballot(lf != NONE);
    uint32_t chi = cb + 64 < c1 ? cb + 64 : c1;
    uint32_t lo_pid = pid_make(peer, cb), hi_pid = pid_make(peer, chi - 1);
    while (pend) {
      int l0 = lmw::ffs64(pend);
      uint32_t Lf = lmw::bcast(lf, l0);
      pend &= ~lmw::ballot(lf == Lf);
      uint32_t n = t.lf_n[Lf];
      bool in = (uint32_t)lane < n;
      uint32_t id = in ? t.it_id[Lf * 64 + lane] : NONE;
      uint32_t st = in ? t.it_st[Lf * 64 + lane] : ST_FUT;
      uint32_t old_act = (uint32_t)lmw::popc64(lmw::ballot(in && st_active(st)));
      bool hit = in && id >= lo_pid && id <= hi_pid;
      if (hit) {
        if (mode == UPD_SET_FUT) st |= ST_FUT;
        else if (mode == UPD_CLR_FUT) st &= ~ST_FUT;
        else if (mode == UPD_DEL_INC) st = (st + ST_DEL1) | ST_EVER;
        else if (st & ST_DELMASK) st -= ST_DEL1;
        t.it_st[Lf * 64 + lane] = st;
      }
      uint32_t new_act = (uint32_t)lmw::popc64(lmw::ballot(in && st_active(st)));
      tr_add_active(t, Lf, (int32_t)new_act - (int32_t)old_act);
    }
  }
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
    // first row whose end is past c0 (rows are counter-ordered inside a change)
    uint32_t lo = ch.op0, hr = ch.op0 + ch.n_op;
    while (lo < hr) { uint32_t mid = (lo + hr) >> 1; if (d.op[mid].ctr + d.op[mid].len <= c0) lo = mid + 1; else hr = mid; }
    for (uint32_t row = lo; row < ch.op0 + ch.n_op && !t.err; row++) {
      const OpRow r = d.op[row];
      if (r.ctr >= c1) break;
      if ((r.cidx_kind & 0xffff) != cidx) continue;
      uint32_t kind = (r.cidx_kind >> 16) & 0xff;
      uint32_t a = (c0 > r.ctr ? c0 : r.ctr) - r.ctr, b = (c1 < r.ctr + r.len ? c1 : r.ctr + r.len) - r.ctr;
      if (a >= b) continue;
      if (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END) {
        tr_update_range(t, peer, r.ctr + a, r.ctr + b, dir < 0 ? UPD_SET_FUT : UPD_CLR_FUT);
      } else if (kind == OK_DEL) {
        uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
        uint32_t t0, t1;
        if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + b; }
        else { t0 = r.a1 + (Ln - b); t1 = r.a1 + (Ln - a); }  // op offset j deletes target + (L-1-j)
        tr_update_range(t, r.a0, t0, t1, dir < 0 ? UPD_DEL_DEC : UPD_DEL_INC);
      }
    }
  }
}

#ifdef LM_EMU_CHECK
// debug-only (emulation): verify cached active counts against the leaves
inline bool tr_check(Tr& t, const char* what, uint32_t row) {
  bool ok = true;
  if (lmw::lane() == 0) {
    uint32_t tot = 0;
    for (uint32_t q = 0; q < t.n_root && ok; q++) {
      uint32_t G = t.rt_grp[q], gs = 0;
      for (uint32_t z = 0; z < t.gp_n[G]; z++) {
        uint32_t L = t.gp_leaf[G * 64 + z], a = 0;
        for (uint32_t i = 0; i < t.lf_n[L]; i++) a += st_active(t.it_st[L * 64 + i]) ? 1 : 0;
        if (a != t.gp_act[G * 64 + z] || t.lf_grp[L] != G) { fprintf(stderr, "CHECK %s row=%u: leaf %u (grp %u slot %u) act=%u cached=%u lf_grp=%u n=%u\n", what, row, L, G, z, a, t.gp_act[G * 64 + z], t.lf_grp[L], t.lf_n[L]); ok = false; break; }
        gs += a;
      }
      if (ok && gs != t.rt_act[q]) { fprintf(stderr, "CHECK %s row=%u: grp %u sum=%u cached=%u\n", what, row, G, gs, t.rt_act[q]); ok = false; }
      tot += gs;
    }
    if (ok && tot != t.tot_active) { fprintf(stderr, "CHECK %s row=%u: tot=%u cached=%u\n", what, row, tot, t.tot_active); ok = false; }
  }
  return lmw::any(!ok) ? false : true;
}
#define TR_CHECK(what, row) do { if (!t.err && !tr_check(t, what, row)) t.err = ST_INTERNAL; } while (0)
#else
#define TR_CHECK(what, row) do {} while (0)
#endif

// K9: one wave per document — replay every sequence container from the empty version.
LM_KERNEL void k_integrate(Dev d, DevDag g) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  LM_SHARED(uint32_t, s_ebase, MAX_PEERS);
  LM_SHARED(uint32_t, s_cur, MAX_PEERS);
  DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  uint32_t P = m.n_peers;
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_ebase[p] = d.elem_base[m.praw0 + p];
  lmw::block_sync();
  uint64_t elem0 = ((uint64_t)m.elem0_hi << 32) | m.elem0_lo;
  uint64_t vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
  Tr t;
  t.it_id = d.it_id + (uint64_t)m.leaf0 * 64; t.it_ol = d.it_ol + (uint64_t)m.leaf0 * 64;
  t.it_or = d.it_or + (uint64_t)m.leaf0 * 64; t.it_st = d.it_st + (uint64_t)m.leaf0 * 64;
  t.lf_n = d.lf_n + m.leaf0; t.lf_next = d.lf_next + m.leaf0; t.lf_grp = d.lf_grp + m.leaf0;
  t.gp_leaf = d.gp_leaf + (uint64_t)m.grp0 * 64; t.gp_act = d.gp_act + (uint64_t)m.grp0 * 64; t.gp_n = d.gp_n + m.grp0;
  t.loc = d.loc + elem0;
  t.ebase = s_ebase;
  t.vis = d.vis + (uint64_t)doc * VIS_CAP;
  t.leaf_cap = m.leaf_cap; t.grp_cap = m.grp_cap;
  t.n_leaf = 0; t.n_grp = 0;
  t.err = 0;
  uint32_t root_used = 0;
  for (uint32_t cidx = 0; cidx < m.n_cont && !t.err; cidx++) {
    uint32_t kr = d.cont[m.cid0 + cidx].kind_root;
    uint32_t ckind = kr & 0xff;
    if (ckind != CK_TEXT && ckind != CK_LIST) continue;
    // fresh tree: one empty leaf in one group
    if (t.n_leaf >= t.leaf_cap || t.n_grp >= t.grp_cap || root_used >= m.grp_cap) { LM_SETERR(t.err, ST_INTERNAL); break; }
    t.rt_grp = d.rt_grp + m.grp0 + root_used;
    t.rt_act = d.rt_act + m.grp0 + root_used;
    t.root_cap = m.grp_cap - root_used;
    uint32_t L0 = t.n_leaf++, G0 = t.n_grp++;
    if (lane == 0) {
      t.lf_n[L0] = 0; t.lf_next[L0] = NONE; t.lf_grp[L0] = G0;
      t.gp_leaf[G0 * 64] = L0; t.gp_act[G0 * 64] = 0; t.gp_n[G0] = 1;
      t.rt_grp[0] = G0; t.rt_act[0] = 0;
    }
    t.n_root = 1; t.tot_active = 0; t.first_leaf = L0;
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_cur[p] = 0;
    lmw::block_sync();
    bool touched = false;
    for (uint32_t oi = 0; oi < m.n_nodes && !t.err; oi++) {
      uint32_t n = d.node_order[m.chg0 + oi];
      uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
      const uint32_t* vv = d.vvh + vvh0 + (uint64_t)n * P;
      uint32_t node_peer = d.chg[d.chg_sorted[m.chg0 + first]].peer;
      bool checked_out = false;
      for (uint32_t ci = first; ci <= last && !t.err; ci++) {
        uint32_t crow = d.chg_sorted[m.chg0 + ci];
        const ChangeRow ch = d.chg[crow];
        uint32_t skip_to = ch.ctr + d.chg_skip[crow];
        for (uint32_t row = ch.op0; row < ch.op0 + ch.n_op && !t.err; row++) {
          const OpRow r = d.op[row];
          if ((r.cidx_kind & 0xffff) != cidx) continue;
          if (r.ctr + r.len <= skip_to) continue;
          uint32_t kind = (r.cidx_kind >> 16) & 0xff;
          uint32_t a = skip_to > r.ctr ? skip_to - r.ctr : 0;  // already-known prefix of a sliced change
          touched = true;
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
          }
          if (kind == OK_TEXT_INS || kind == OK_LIST_INS) {
            tr_insert(t, (uint32_t)r.prop + a, pid_make(node_peer, r.ctr + a), r.len - a);
            TR_CHECK("insert", row);
          } else if (kind == OK_DEL) {
            uint32_t Ln = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2);
            uint32_t t0, t1;
            if (r.a2 > 0) { t0 = r.a1 + a; t1 = r.a1 + Ln; }
            else { t0 = r.a1; t1 = r.a1 + (Ln - a); }
            tr_update_range(t, r.a0, t0, t1, UPD_DEL_INC);
            TR_CHECK("delete", row);
          } else if (kind == OK_STYLE_START) {
            tr_insert(t, (uint32_t)r.prop, pid_make(node_peer, r.ctr), 1);
          } else if (kind == OK_STYLE_END) {
            // diff_calc.rs:1105-1119: the matching StyleStart is the op right before (same peer, counter-1)
            uint32_t end_pos = NONE;
            if (row > ch.op0) {
              const OpRow pr = d.op[row - 1];
              if (((pr.cidx_kind >> 16) & 0xff) == OK_STYLE_START && pr.ctr + 1 == r.ctr && (pr.cidx_kind & 0xffff) == cidx)
                end_pos = (uint32_t)pr.prop + pr.a0;
            }
            if (end_pos == NONE) { LM_SETERR(t.err, ST_UNSUPPORTED); break; }
            uint32_t pos = end_pos + 1 < t.tot_active ? end_pos + 1 : t.tot_active;
            tr_insert(t, pos, pid_make(node_peer, r.ctr), 1);
          }
        }
        // the node's own ops advance the tracker version
        if (checked_out && lane == 0) s_cur[node_peer] = ch.ctr + ch.len;
      }
      lmw::block_sync();
    }
    if (lane == 0) {
      d.cont_root0[m.cid0 + cidx] = root_used;
      d.cont_nroot[m.cid0 + cidx] = t.n_root;
      if (touched) d.cont[m.cid0 + cidx].touched = 1;
    }
    root_used += t.n_root;
  }
  if (t.err && lane == 0) d.doc[doc].status = t.err;
}

}  // namespace lm
