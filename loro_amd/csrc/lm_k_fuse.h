// Row fusion for one-change-per-keystroke histories (plain documents only).
//
// The reference fuses a peer's consecutive self-dependent commits into one change before it exports them (change.rs:268-282,
// change_store.rs:1007-1017) and RLE-merges the ops inside a change (op.rs:143-152, list_op.rs:516-589) — but a document whose
// peers exported after every keystroke, or whose blobs come from a writer that does not fuse, arrives as tens of thousands of
// one-row changes: typing "hello" is five insert rows, a held backspace key one delete row per character.  The integrate kernel
// pays its full per-row price for each (≈200 instructions; the heterogeneous configs[1] batch spent most of its 141 ms on the
// 10k- and 40k-change documents).  Rows of ONE node of the causal graph — a peer's self-dependent run of changes, which the
// replay applies back to back with nothing in between — are chained here into the runs a fusing writer would have produced:
//   insert after insert:   next.pos == pos + len, next.counter == counter + len            (list_op.rs:516-540 is_mergable / merge)
//   forward delete chain:  same position, target ids ascending and contiguous             (list_op.rs:541-589 DeleteSpan merge)
//   backward delete chain: positions and target ids descending and contiguous (backspace)
// A chain never leaves its block, its 64-row window, its container, or its node (the next change must depend on nothing but its
// peer's previous op, and no other peer's change may depend on the change in front of it — k_dag_a's node cut).  The head row
// gets OPF_HEAD and its run's extent in Dev::fuse, the others OPF_CONT; k_integrate_span_plain_fuse replays heads as one row
// and skips the rest.  Everything else (payload fill, retreat / forward by rows, the element-granular kernel) keeps reading
// the rows as decoded: their fields are untouched.  One lane per op row.
#pragma once

namespace lm {

LM_KERNEL void k_fuse_rows(Dev d, uint32_t n_ops) {
  uint32_t t = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  int lane = lmw::lane();
  bool valid = t < n_ops;
  OpRow r;
  r.cidx_kind = 0; r.prop = 0; r.len = 0; r.ctr = 0; r.a0 = r.a1 = 0; r.a2 = 0; r.chg = 0;
  uint32_t blk = NONE;
  bool cand = false, first_of_chg = false, joinable_chg = false;
  if (valid) {
    r = d.op[t];
    blk = d.op_blk[t];
    const BlockDesc& bd = d.blk[blk];
    if (bd.status == ST_OK) {
      const DocMeta& m = d.doc[bd.doc];
      cand = m.status == ST_OK && (m.flags & (DF_PLAIN | DF_FUSED)) == (DF_PLAIN | DF_FUSED);
    }
    if (cand) {
      cand = (d.chg_flag[r.chg] & 1u) && d.chg_skip[r.chg] == 0;
      const ChangeRow& ch = d.chg[r.chg];
      first_of_chg = t == ch.op0;
      // the change continues its peer's previous change inside one node: its only dependency is the op right before it, and
      // no other peer's change depends on the change in front of it (chg_flag bit 1, k_dag_a)
      joinable_chg = cand && r.chg > 0 && ch.n_dep == 1 && d.dep_peer[ch.dep0] == ch.peer && ch.ctr > 0 && d.dep_ctr[ch.dep0] == ch.ctr - 1 &&
                     d.chg[r.chg - 1].blk == ch.blk && d.chg[r.chg - 1].ctr + d.chg[r.chg - 1].len == ch.ctr && !(d.chg_flag[r.chg - 1] & 2u);
    }
  }
  uint32_t kind = (r.cidx_kind >> 16) & 0xff;
  // the row in front of this one (lane 0 heads its window's first run)
  uint32_t p_ck = lmw::shfl_up(r.cidx_kind, 1), p_prop = lmw::shfl_up((uint32_t)r.prop, 1), p_len = lmw::shfl_up(r.len, 1), p_ctr = lmw::shfl_up(r.ctr, 1);
  uint32_t p_a0 = lmw::shfl_up(r.a0, 1), p_a1 = lmw::shfl_up(r.a1, 1), p_a2 = lmw::shfl_up((uint32_t)r.a2, 1), p_chg = lmw::shfl_up(r.chg, 1);
  uint32_t p_blk = lmw::shfl_up(blk, 1), p_cand = lmw::shfl_up(cand ? 1u : 0u, 1);
  uint32_t type = 0;   // 1 insert run, 2 forward delete chain, 3 backward delete chain
  if (cand && p_cand && lane > 0 && p_blk == blk && p_ck == r.cidx_kind && r.ctr == p_ctr + p_len &&
      (r.chg == p_chg || (r.chg == p_chg + 1 && first_of_chg && joinable_chg))) {
    if (kind == OK_TEXT_INS || kind == OK_LIST_INS) {
      if ((uint32_t)r.prop == p_prop + p_len) type = 1;
    } else if (kind == OK_DEL && r.a0 == p_a0 && r.len && p_len) {
      int32_t pa2 = (int32_t)p_a2;
      bool r_geom = (uint32_t)(r.a2 < 0 ? -r.a2 : r.a2) == r.len && r.prop >= 0, p_geom = (uint32_t)(pa2 < 0 ? -pa2 : pa2) == p_len && (int32_t)p_prop >= 0;
      if (r_geom && p_geom) {
        if (r.a2 > 0 && pa2 > 0 && (uint32_t)r.prop == p_prop && r.a1 == p_a1 + p_len) type = 2;
        else if ((r.a2 < 0 || r.len == 1) && (pa2 < 0 || p_len == 1) && r.a1 + r.len == p_a1) {
          // the previous row's leftmost target sits at p_left; this row's rightmost target must sit right in front of it
          uint32_t p_left = pa2 < 0 ? p_prop + 1 - p_len : p_prop;
          if (p_prop + 1 >= p_len && (uint32_t)r.prop + 1 == p_left && (uint32_t)r.prop + 1 >= r.len) type = 3;
        }
      }
    }
  }
  uint32_t p_type = lmw::shfl_up(type, 1);
  bool cont = type != 0 && (lane == 0 ? false : (p_type == 0 || p_type == type));
  uint64_t cm = lmw::ballot(cont);
  // a head: not a continuation itself, the next lane is one
  bool head = valid && !cont && lane < 63 && ((cm >> (lane + 1)) & 1);
  uint32_t inc = lmw::scan_incl_add(valid ? r.len : 0u);
  uint64_t stop = ~cm & ~((2ull << lane) - 1);                 // lanes behind this one that are not continuations
  int e = stop ? lmw::ffs64(stop) : 64;                        // first lane behind the run
  uint32_t tot = lmw::shfl(inc, e - 1) - inc + r.len;          // (every lane takes part in the permutes)
  uint32_t last_a1 = lmw::shfl(r.a1, e - 1);
  uint32_t run_type = lmw::shfl(type, (lane + 1) & 63);
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_FUSE") && (head || cont)) fprintf(stderr, "FUSE %s type %u\n", head ? "head" : "cont", head ? run_type : type);
#endif
  if (head) {
    d.op[t].cidx_kind = r.cidx_kind | OPF_HEAD;
    uint32_t a1 = run_type == 3 ? last_a1 : r.a1;
    int32_t sl = run_type == 3 ? -(int32_t)tot : (int32_t)tot;
    d.fuse[2 * (uint64_t)t] = a1;
    d.fuse[2 * (uint64_t)t + 1] = (uint32_t)sl;
  } else if (cont) d.op[t].cidx_kind = r.cidx_kind | OPF_CONT;
}

}  // namespace lm
