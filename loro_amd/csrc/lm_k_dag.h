// Document tables and causal graph: peer/container registries, applied-change selection (pending),
// DAG nodes, causal (topological) replay order, lamports, version vectors at node heads.
// Reference semantics (paths relative to /root/reference/crates/loro-internal/src):
//   import filtering / trim known prefix   oplog/change_store.rs:329-352, oplog.rs:367-382
//   pending changes                         encoding/outdated_encode_reordered.rs:48-75, oplog/pending_changes.rs:12-36
//   lamport from deps                       oplog/loro_dag.rs:1179-1187
//   DAG node = self-dependent run           oplog/loro_dag.rs:302-367,995-1019
//   vv of a node's deps                     oplog/loro_dag.rs:1083-1154,1192-1207
//   causal iteration                        dag/iter.rs:199-386, oplog.rs:591-669
#pragma once
#include "lm_k_decode.h"

namespace lm {

#ifndef LM_FUSE_ROWS
#define LM_FUSE_ROWS 1     // 0: no row fusion (rounds 1-3; A/B builds)
#endif
#ifndef LM_NO_NODE_CUT
#define LM_NO_NODE_CUT 0   // 1: nodes are whole self-dependent runs and ready nodes replay in ascending peer order (rounds 1-3; A/B builds)
#endif

struct DevDag {            // extra scratch of the DAG stage
  uint32_t* blk_sorted;    // [blk0 + i] blocks of a doc ordered by (peer, counter_start)
  uint32_t* chg_node;      // [chg0 + i] node id of the i-th sorted applied change
  uint32_t* node_done;     // [chg0 + n]
  uint32_t* node_lam;      // [chg0 + n] lamport of the node's first op
};

// K5a: one lane per doc — row ranges from the scanned block counters.
LM_KERNEL void k_doc_ranges(Dev d) {
  uint32_t doc = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  if (doc >= d.n_docs) return;
  DocMeta m;
  uint32_t* mw = (uint32_t*)&m;
  for (uint32_t i = 0; i < sizeof(DocMeta) / 4; i++) mw[i] = 0;
  uint32_t b0 = d.doc_blob[doc], b1 = d.doc_blob[doc + 1];
  uint32_t k0 = d.blob_blk0[b0], k1 = d.blob_blk0[b1];
  m.blk0 = k0;
  m.n_blk = k1 - k0;
  const uint32_t* o0 = d.boff + (uint64_t)k0 * BCN;
  const uint32_t* o1 = d.boff + (uint64_t)k1 * BCN;
  m.chg0 = o0[BC_CHG]; m.n_chg = o1[BC_CHG] - o0[BC_CHG];
  m.dep0 = o0[BC_DEP]; m.n_dep = o1[BC_DEP] - o0[BC_DEP];
  m.op0 = o0[BC_OP]; m.n_op = o1[BC_OP] - o0[BC_OP];
  if (d.doc_fused && d.doc_fused[doc]) { m.op0 = d.n_op_rows + o0[BC_MAPOP]; m.n_op = o1[BC_MAPOP] - o0[BC_MAPOP]; }   // (its record table: one slot per row at most, lm_k_map_fused.h)
  m.key0 = o0[BC_KEY]; m.n_key = o1[BC_KEY] - o0[BC_KEY];
  m.cid0 = o0[BC_CID]; m.n_cid = o1[BC_CID] - o0[BC_CID];
  m.praw0 = o0[BC_PEER]; m.n_praw = o1[BC_PEER] - o0[BC_PEER];
  m.atoms = o1[BC_ATOMS] - o0[BC_ATOMS];
  int32_t st = ST_OK;
  for (uint32_t b = b0; b < b1; b++) { int32_t s = d.blob_status[b]; if (s > st) st = s; }
  m.status = st;
  d.doc[doc] = m;
}

LM_DEV bool status_fatal(int32_t st) { return st != ST_OK; }

// K5: one wave per doc — unique sorted peers, peer map, container registry.
LM_KERNEL void k_doc_tables(Dev d) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  LM_SHARED(uint64_t, s_peers, MAX_PEERS);
  LM_SHARED(uint32_t, s_n, 4);
  DocMeta m = d.doc[doc];
  // aggregate block statuses
  {
    uint32_t st = 0;
    for (uint32_t i = (uint32_t)lane; i < m.n_blk; i += 64) { uint32_t s = (uint32_t)d.blk[m.blk0 + i].status; st = s > st ? s : st; }
    st = lmw::reduce_max(st);
    if ((int32_t)st > m.status) m.status = (int32_t)st;
  }
  if (m.status == ST_MF_BAIL) { if (lane == 0) { d.doc[doc].status = ST_DATA_CORRUPTION; d.doc[doc].flags = DF_REDO; } return; }   // (k_block_head: replayed through the row tables, lm_capi_impl.h redo)
  if (status_fatal(m.status)) { if (lane == 0) d.doc[doc].status = m.status; return; }
  // ---- peers: insertion into an LDS set, then rank sort
  uint32_t P = 0;
  bool too_many = false;
  for (uint32_t i = 0; i < m.n_praw; i++) {
    uint64_t v = d.peer_raw[m.praw0 + i];
    bool hit = false;
    for (uint32_t k = (uint32_t)lane; k < P; k += 64) hit |= s_peers[k] == v;
    if (!lmw::any(hit)) {
      if (P >= MAX_PEERS - 1) { too_many = true; break; }
      if (lane == 0) s_peers[P] = v;
      P++;
      lmw::block_sync();
    }
  }
  if (too_many) { if (lane == 0) LM_SETERR(d.doc[doc].status, ST_UNSUPPORTED); return; }
  lmw::block_sync();
  for (uint32_t e = (uint32_t)lane; e < P; e += 64) {
    uint64_t v = s_peers[e];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < P; j++) rank += s_peers[j] < v ? 1u : 0u;
    d.peer_uniq[m.praw0 + rank] = v;
  }
  lmw::block_sync();
  for (uint32_t i = (uint32_t)lane; i < m.n_praw; i += 64) {
    uint64_t v = d.peer_raw[m.praw0 + i];
    uint32_t lo = 0, hi = P;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < v) lo = mid + 1; else hi = mid; }
    d.peer_map[m.praw0 + i] = lo;
  }
  lmw::block_sync();   // peer_map (written lane-parallel above) is read below by every lane for child container ids
  // ---- containers
  uint32_t C = 0;
  bool cont_overflow = false;
  for (uint32_t i = 0; i < m.n_cid; i++) {
    const uint32_t* w = d.cid_raw + (uint64_t)(m.cid0 + i) * 4;
    uint32_t kr = w[0], blk = w[3];
    const uint32_t* bo = d.boff + (uint64_t)blk * BCN;
    bool is_root = (kr & 0x100) != 0;
    uint64_t noff = 0;
    uint32_t nlen = 0, cpeer = 0, cctr = 0;
    if (is_root) { noff = d.key_off[bo[BC_KEY] + w[2]]; nlen = d.key_len[bo[BC_KEY] + w[2]]; }
    else { cpeer = d.peer_map[bo[BC_PEER] + w[1]]; cctr = w[2]; }
    uint32_t idx = NONE;
    for (uint32_t c0 = 0; c0 < C && idx == NONE; c0 += 64) {
      bool match = false;
      if (c0 + (uint32_t)lane < C) {
        ContRow c = d.cont[m.cid0 + c0 + lane];
        if (c.kind_root == kr) {
          if (is_root) {
            if (c.name_len == nlen) {
              match = true;
              const uint8_t* a = d.data + c.name_off;
              const uint8_t* b = d.data + noff;
              for (uint32_t k = 0; k < nlen; k++) if (a[k] != b[k]) { match = false; break; }
            }
          } else match = c.peer == cpeer && c.counter == cctr;
        }
      }
      uint64_t mm = lmw::ballot(match);
      if (mm) idx = c0 + (uint32_t)lmw::ffs64(mm);
    }
    if (idx != NONE) {}
    else {
      if (C >= MAX_CONTAINERS) { cont_overflow = true; break; }
      idx = C;
      if (lane == 0) {
        ContRow c;
        c.name_off = noff; c.name_len = nlen; c.kind_root = kr; c.peer = cpeer; c.counter = cctr; c.touched = 0; c.pad = 0;
        d.cont[m.cid0 + C] = c;
      }
      C++;
      lmw::block_sync();
    }
    if (lane == 0) d.cid_map[m.cid0 + i] = idx;
  }
  if (lane == 0) {
    d.doc[doc].status = cont_overflow ? ST_UNSUPPORTED : m.status;
    d.doc[doc].n_peers = P;
    d.doc[doc].n_cont = C;
  }
  (void)s_n;
}

// K6: grid-stride — translate block-local peer/container indices to document-level ones.
LM_KERNEL void k_remap(Dev d, uint32_t n_ops, uint32_t n_chg) {
  uint32_t t = (uint32_t)(lmw::bid() * lmw::bdim() + lmw::tid());
  uint32_t m_chg = NONE, m_bit = 0;   // (change, container bit) this row contributes to the per-change container mask
  if (t < n_ops) {
    uint32_t blk = d.op_blk[t];
    const BlockDesc& bd = d.blk[blk];
    if (bd.status == ST_OK && d.doc[bd.doc].status == ST_OK) {
      const uint32_t* bo = d.boff + (uint64_t)blk * BCN;
      OpRow r = d.op[t];
      uint32_t local = r.cidx_kind & 0xffff, kind = (r.cidx_kind >> 16) & 0xff;
      uint32_t ci = d.cid_map[bo[BC_CID] + local];
      if (r.cidx_kind & OPF_NESTED) {
        // the reference decodes every value in full with the block: a nested map whose key index lies beyond the block's key table is
        // DataCorruption (value.rs read_value: `keys.get(key_idx).ok_or(DataCorruption)`). The decoders' walkers skip nested values
        // without the key table in hand (it costs them registers on every row); the few rows that hold a list / map come back here.
        const uint8_t* vend = d.data + bd.base + bd.sec_rel[SEC_VALUES] + bd.sec_len[SEC_VALUES];
        Rd v{d.data + d.op_val[t], vend, false};
        uint32_t n_keys = d.bcnt[(uint64_t)blk * BCN + BC_KEY];
        if (kind == OK_STYLE_START) { (void)rd_u8(v); (void)rd_uleb(v); (void)rd_uleb(v); }   // (mark: info, len, key idx — checked by the decoder)
        uint32_t vf = 0;
        skip_loro_value_keys(v, vf, n_keys);
        if ((vf & VF_CORRUPT) && !v.bad) LM_SETERR(d.doc[bd.doc].status, ST_DATA_CORRUPTION);
      }
      r.cidx_kind = ci | (kind << 16);
      if (kind == OK_DEL || kind == OK_LIST_MOVE || kind == OK_LIST_SET) r.a0 = d.peer_map[bo[BC_PEER] + r.a0];
      else if (kind == OK_MAP_SET || kind == OK_MAP_DEL) r.a0 = bo[BC_KEY] + (uint32_t)r.prop;   // the row's key row (the LWW kernels: no trip through op_blk → boff)
      d.op[t] = r;
      m_chg = r.chg; m_bit = ci & 63;
    }
  }
  // rows of a change are consecutive and mostly hit one container: only the first lane of a run issues the atomic
  {
    uint32_t p_chg = lmw::shfl_up(m_chg, 1), p_bit = lmw::shfl_up(m_bit, 1);
    bool lead = m_chg != NONE && (lmw::lane() == 0 || p_chg != m_chg || p_bit != m_bit);
    if (lead) lmw::atomic_or(&d.chg_mask[2 * (uint64_t)m_chg + (m_bit >> 5)], 1u << (m_bit & 31));
  }
  if (t < n_chg) {
    ChangeRow c = d.chg[t];
    const BlockDesc& bd = d.blk[c.blk];
    if (bd.status == ST_OK && d.doc[bd.doc].status == ST_OK) {
      const uint32_t* bo = d.boff + (uint64_t)c.blk * BCN;
      c.peer = d.peer_map[bo[BC_PEER] + 0];
      d.chg[t] = c;
      for (uint32_t k = c.dep0; k < c.dep0 + c.n_dep; k++) d.dep_peer[k] = d.peer_map[bo[BC_PEER] + d.dep_peer[k]];
    }
  }
}

// find the sorted applied change of `peer` that contains counter c; returns doc-relative sorted index or NONE
LM_DEV uint32_t find_change(const Dev& d, const DocMeta& m, uint32_t peer, uint32_t c) {
  uint32_t lo = d.peer_chg0[m.praw0 + peer], hi = d.peer_chg1[m.praw0 + peer];
  uint32_t end = hi;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    const ChangeRow& ch = d.chg[d.chg_sorted[m.chg0 + mid]];
    if (ch.ctr + ch.len <= c) lo = mid + 1; else hi = mid;
  }
  if (lo >= end) return NONE;
  const ChangeRow& ch = d.chg[d.chg_sorted[m.chg0 + lo]];
  return ch.ctr <= c ? lo : NONE;
}

// K7a: one wave per doc — order blocks, select applied changes, build DAG nodes.
#ifdef LM_PROF_DAG   // experiment build: ticks per pass of k_dag_a, summed over the batch's documents into d.prof[0..7] (tests/tools/gpu_prof_dag.py)
#define DAG_PH(i) do { uint64_t n_ = lmw::clock(); dpacc[i] += n_ - dptp; dptp = n_; } while (0)
#else
#define DAG_PH(i) do {} while (0)
#endif
// A document staged on a snapshot's STATE (lm_snapshot.h / lm_snapshot_base.h) carries its BASE version in d.vvo: 8 bytes = the
// synthetic peer that writes the state, then the snapshot's version vector (postcard map, ascending peers).  History below it is
// known without being staged: a peer's changes may start at its base end, and the version the document reports includes the base.
LM_DEV bool vvo_has(const Dev& d, uint32_t doc) { return d.vvo && d.vvo_off[doc + 1] - d.vvo_off[doc] > 8; }
LM_DEV uint64_t vvo_synth_peer(const Dev& d, uint32_t doc) { uint64_t v = 0; const uint8_t* p = d.vvo + d.vvo_off[doc]; for (int k = 0; k < 8; k++) v |= (uint64_t)p[k] << (8 * k); return v; }
LM_DEV uint32_t vvo_base_end(const Dev& d, uint32_t doc, uint64_t peer) {
  if (!vvo_has(d, doc)) return 0;
  Rd r = rd_make(d.vvo + d.vvo_off[doc] + 8, d.vvo_off[doc + 1] - d.vvo_off[doc] - 8);
  const uint64_t n = rd_uleb(r);
  for (uint64_t i = 0; i < n && !r.bad; i++) {
    const uint64_t q = rd_uleb(r);
    const uint64_t z = rd_uleb(r);
    if (q == peer) return (uint32_t)(z >> 1);
    if (q > peer) break;
  }
  return 0;
}
LM_KERNEL void k_dag_a(Dev d, DevDag g) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
#ifdef LM_PROF_DAG
  uint64_t dpacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dptp = lmw::clock();
#endif
  LM_SHARED(uint32_t, s_valid, MAX_PEERS);   // valid (applied) exclusive end per peer
  LM_SHARED(uint32_t, s_ext, MAX_PEERS);     // contiguous covered end per peer
  DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  uint32_t P = m.n_peers;
  // ---- 1. rank-sort the doc's blocks by (peer, counter_start, block index)
  for (uint32_t i = (uint32_t)lane; i < m.n_blk; i += 64) {
    uint32_t bi = m.blk0 + i;
    const BlockDesc& b = d.blk[bi];
    uint64_t key = ((uint64_t)d.chg[d.boff[(uint64_t)bi * BCN + BC_CHG]].peer << 32) | b.counter_start;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < m.n_blk; j++) {
      uint32_t bj = m.blk0 + j;
      uint64_t kj = ((uint64_t)d.chg[d.boff[(uint64_t)bj * BCN + BC_CHG]].peer << 32) | d.blk[bj].counter_start;
      rank += (kj < key || (kj == key && j < i)) ? 1u : 0u;
    }
    g.blk_sorted[m.blk0 + rank] = bi;
  }
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_ext[p] = vvo_base_end(d, doc, d.peer_uniq[m.praw0 + p]); d.peer_base[m.praw0 + p] = s_ext[p]; s_valid[p] = 0; d.peer_chg0[m.praw0 + p] = 0; d.peer_chg1[m.praw0 + p] = 0; }   // (the base version: 0 unless the document is staged on a snapshot's state)
  lmw::block_sync();
  DAG_PH(0);
  // ---- 2. coverage walk: drop known changes, slice straddling ones, park blocks behind a counter gap
  uint32_t n_sorted = 0;
  bool any_skip = false;   // a kept change whose prefix is already known (sliced): DF_PLAIN stays clear
  uint32_t pending = 0;
  uint32_t cur_peer = NONE, covered = 0;
  bool gap = false;
  for (uint32_t i = 0; i < m.n_blk; i++) {
    uint32_t bi = g.blk_sorted[m.blk0 + i];
    const BlockDesc& b = d.blk[bi];
    uint32_t c0 = d.boff[(uint64_t)bi * BCN + BC_CHG];
    uint32_t peer = d.chg[c0].peer;
    if (peer != cur_peer) {
      if (cur_peer != NONE && lane == 0) { s_ext[cur_peer] = covered; d.peer_chg1[m.praw0 + cur_peer] = n_sorted; }
      cur_peer = peer; covered = s_ext[peer]; gap = false;
      if (lane == 0) d.peer_chg0[m.praw0 + peer] = n_sorted;
    }
    if (b.counter_start > covered) gap = true;
    uint32_t bend = b.counter_start + b.counter_len;
    for (uint32_t k0 = 0; k0 < b.n_changes; k0 += 64) {
      uint32_t k = k0 + (uint32_t)lane;
      bool in = k < b.n_changes;
      uint32_t row = c0 + (in ? k : 0);
      ChangeRow ch = d.chg[row];
      bool keep = false;
      uint32_t skip = 0;
      if (in) {
        if (gap) { pending += ch.len; d.chg_flag[row] = 0; }
        else if (ch.ctr + ch.len <= covered) { d.chg_flag[row] = 0; }  // already known: dropped
        else { keep = true; skip = ch.ctr < covered ? covered - ch.ctr : 0; d.chg_flag[row] = 1; }
        d.chg_skip[row] = skip;
        any_skip |= keep && skip != 0;
      }
      uint64_t km = lmw::ballot(keep);
      if (keep) d.chg_sorted[m.chg0 + n_sorted + (uint32_t)lmw::popc64(km & ((1ull << lane) - 1))] = row;
      n_sorted += (uint32_t)lmw::popc64(km);
    }
    if (!gap && bend > covered) covered = bend;
  }
  if (cur_peer != NONE && lane == 0) { s_ext[cur_peer] = covered; d.peer_chg1[m.praw0 + cur_peer] = n_sorted; }
  pending = lmw::reduce_add(pending);
  lmw::block_sync();
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) s_valid[p] = s_ext[p];
  lmw::block_sync();
  DAG_PH(1);
  // ---- 3. dependency fixpoint: a change applies iff every dep is applied (pending otherwise)
  for (uint32_t iter = 0; iter < 4096; iter++) {
    bool changed = false;
    for (uint32_t i = (uint32_t)lane; i < n_sorted; i += 64) {
      uint32_t row = d.chg_sorted[m.chg0 + i];
      const ChangeRow& ch = d.chg[row];
      if (ch.ctr >= s_valid[ch.peer]) continue;
      if (d.chg_skip[row] != 0) continue;  // sliced change: its only dependency is the known prefix
      bool ok = true;
      for (uint32_t k = ch.dep0; k < ch.dep0 + ch.n_dep; k++) {
        uint32_t q = d.dep_peer[k], c = d.dep_ctr[k];
        if (c >= s_valid[q]) ok = false;
      }
      if (!ok) { lmw::atomic_min(&s_valid[ch.peer], ch.ctr); changed = true; }
    }
    lmw::block_sync();
    if (!lmw::any(changed)) break;
  }
  DAG_PH(2);
  // ---- 4. compact to applied changes, recompute per-peer ranges, count pending atoms
  uint32_t n_valid = 0;
  for (uint32_t i0 = 0; i0 < n_sorted; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    bool in = i < n_sorted;
    uint32_t row = in ? d.chg_sorted[m.chg0 + i] : 0;
    bool ok = false;
    if (in) {
      const ChangeRow& ch = d.chg[row];
      ok = ch.ctr < s_valid[ch.peer];
      if (!ok) { d.chg_flag[row] = 0; }
    }
    uint32_t pend_here = 0;
    if (in && !ok) pend_here = d.chg[row].len - d.chg_skip[row];
    pending += lmw::reduce_add(pend_here);
    uint64_t km = lmw::ballot(ok);
    lmw::block_sync();
    if (ok) d.chg_sorted[m.chg0 + n_valid + (uint32_t)lmw::popc64(km & ((1ull << lane) - 1))] = row;
    n_valid += (uint32_t)lmw::popc64(km);
    lmw::block_sync();
  }
  DAG_PH(3);
  // per-peer ranges in the compacted order (changes of a peer stay contiguous and ordered)
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { d.peer_chg0[m.praw0 + p] = NONE; d.peer_chg1[m.praw0 + p] = 0; }
  lmw::block_sync();
  for (uint32_t i = (uint32_t)lane; i < n_valid; i += 64) {
    uint32_t peer = d.chg[d.chg_sorted[m.chg0 + i]].peer;
    uint32_t prev = i ? d.chg[d.chg_sorted[m.chg0 + i - 1]].peer : NONE;
    uint32_t next = i + 1 < n_valid ? d.chg[d.chg_sorted[m.chg0 + i + 1]].peer : NONE;
    if (prev != peer) d.peer_chg0[m.praw0 + peer] = i;
    if (next != peer) d.peer_chg1[m.praw0 + peer] = i + 1;
  }
  lmw::block_sync();
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) {
    if (d.peer_chg0[m.praw0 + p] == NONE) { d.peer_chg0[m.praw0 + p] = 0; d.peer_chg1[m.praw0 + p] = 0; }
    d.peer_end[m.praw0 + p] = s_valid[p];
    d.peer_ext[m.praw0 + p] = s_ext[p];
  }
  // element bases: exclusive prefix of the covered extents
  {
    uint32_t run = 0;
    for (uint32_t p0 = 0; p0 < P; p0 += 64) {
      uint32_t p = p0 + (uint32_t)lane;
      uint32_t e = p < P ? s_ext[p] : 0;
      uint32_t inc = lmw::scan_incl_add(e);
      if (p < P) d.elem_base[m.praw0 + p] = run + inc - e;
      run += lmw::bcast(inc, 63);
    }
    // (a document staged on a snapshot's state: its peers' element slots begin at counter 0 like everyone's, so the base version's
    // extent is part of the layout although none of its ops is staged — ids below it that delete rows name find empty slots there)
    if (vvo_has(d, doc)) { if (run > m.atoms && lane == 0) d.doc[doc].atoms = run; }
    else if (run > m.atoms && lane == 0) LM_SETERR(d.doc[doc].status, ST_INTERNAL);
  }
  // ---- 5. DAG nodes: maximal runs linked only by a dependency on the peer's previous op — cut behind every change another
  // peer's change depends on (AppDagNode::has_succ and the lazy node split of the reference, loro_dag.rs:302-367,995-1019):
  // what follows such a change is concurrent with the dependent branch, and k_dag_b is then free to replay that branch FIRST
  // (a node is its scheduling unit: uncut, a peer's whole run — configs[1]: the base AND its own branch — had to precede
  // everything that depends on any part of it).
  lmw::mem_fence();
  lmw::block_sync();   // peer_chg0/1 (find_change) are complete
  DAG_PH(4);
  // (every dependency is resolved to its change HERE, once — dep_ci — so that k_dag_b's passes, one per node, test readiness and
  // merge version vectors with two loads per dependency instead of a binary search each.  A dependency on the peer's own previous
  // op — nearly all of them — is the change right in front, no search.)
  for (uint32_t i = (uint32_t)lane; i < n_valid; i += 64) {
    uint32_t row = d.chg_sorted[m.chg0 + i];
    const ChangeRow& ch = d.chg[row];
    for (uint32_t k = ch.dep0; k < ch.dep0 + ch.n_dep; k++) {
      uint32_t q = d.dep_peer[k], c = d.dep_ctr[k];
      uint32_t ci = NONE;
      if (q == ch.peer && i > 0 && c + 1 == ch.ctr) {
        const ChangeRow& pv = d.chg[d.chg_sorted[m.chg0 + i - 1]];
        if (pv.peer == q && pv.ctr <= c && c < pv.ctr + pv.len) ci = i - 1;
      }
      if (ci == NONE && q < P) ci = find_change(d, m, q, c);
      d.dep_ci[k] = ci;
      if (q != ch.peer && ci != NONE && d.chg_skip[row] == 0)   // (a sliced change's only dependency is the known prefix)
        lmw::atomic_or(&d.chg_flag[d.chg_sorted[m.chg0 + ci]], 2u);   // (chg_flag: bit 0 applied, bit 1 has a successor of another peer)
    }
  }
  lmw::mem_fence();
  lmw::block_sync();
  // The cut and the one-node-per-pass, largest-peer-first order of k_dag_b pay where concurrent branches are LONG (configs[1]: 27 %
  // of the integrate stage); a small document of many peers that sync every few ops only gets more nodes out of it — more passes
  // here, more tracker moves in the replay (measured, tests/tools/gpu_other.py: MovableList batch integrate 15.1 -> 20.6 ms, k_dag_b
  // 2.9 -> 6.8 ms; configs[3] k_dag_b 5.1 -> 13.7 ms).  So: documents of at least LM_CUT_MIN_ROWS op rows (an environment knob, 2,048 by default; the parity suites also run with 0).
  DAG_PH(5);
  const bool cut_doc = !LM_NO_NODE_CUT && m.n_op >= d.cut_min_rows;
  uint32_t n_nodes = 0;
  for (uint32_t i0 = 0; i0 < n_valid; i0 += 64) {
    uint32_t i = i0 + (uint32_t)lane;
    bool in = i < n_valid;
    bool head = false;
    if (in) {
      uint32_t row = d.chg_sorted[m.chg0 + i];
      const ChangeRow& ch = d.chg[row];
      bool cont = false;
      if (d.chg_skip[row] != 0) cont = true;
      else if (ch.n_dep == 1 && d.dep_peer[ch.dep0] == ch.peer && ch.ctr > 0 && d.dep_ctr[ch.dep0] == ch.ctr - 1) cont = true;
      // a continuation needs a predecessor of the same peer right before it in the sorted order
      if (cont && (i == 0 || d.chg[d.chg_sorted[m.chg0 + i - 1]].peer != ch.peer)) cont = false;
      if (cont && cut_doc && d.chg_skip[row] == 0 && (d.chg_flag[d.chg_sorted[m.chg0 + i - 1]] & 2u)) cont = false;
      head = !cont;
    }
    uint64_t hm = lmw::ballot(head);
    uint32_t nid = n_nodes + (uint32_t)lmw::popc64(hm & ((2ull << lane) - 1)) - 1;
    if (in) {
      g.chg_node[m.chg0 + i] = nid;
      if (head) d.node_first[m.chg0 + nid] = i;
    }
    n_nodes += (uint32_t)lmw::popc64(hm);
  }
  lmw::block_sync();
  for (uint32_t n = (uint32_t)lane; n < n_nodes; n += 64) {
    d.node_last[m.chg0 + n] = (n + 1 < n_nodes ? d.node_first[m.chg0 + n + 1] : n_valid) - 1;
    g.node_done[m.chg0 + n] = 0;
  }
  DAG_PH(6);
  // number of Map op rows (sizes the doc's LWW hash table)
  // (from the blocks' descriptors: the decoders count their rows by kind as they write them — lm_k_decode.h kc_add.  MovableList
  // move / set rows compete per element in the same LWW table, a move also places a new list item; rows of containers outside the
  // device scope count too: the LWW stage is what marks their containers, and a batch without any such row skips it)
  uint32_t n_map = 0, n_el = 0, n_style = 0;
  for (uint32_t i = (uint32_t)lane; i < m.n_blk; i += 64) {
    const BlockDesc& kb = d.blk[m.blk0 + i];
    if (kb.status != ST_OK) continue;
    n_map += kb.flags & 0x7fffffffu; n_el += kb.pad; n_style += kb.flags >> 31;
  }
  bool has_ml = false;
  for (uint32_t c = (uint32_t)lane; c < m.n_cont; c += 64) has_ml |= (d.cont[m.cid0 + c].kind_root & 0xff) == CK_MOVABLE;
  has_ml = lmw::any(has_ml);
  bool any_skip_doc = lmw::any(any_skip);
  n_map = lmw::reduce_add(n_map);
  n_el = lmw::reduce_add(n_el);
  n_style = lmw::reduce_add(n_style);
  if (lane == 0) {
    d.doc[doc].n_valid_chg = n_valid;
    d.doc[doc].n_nodes = n_nodes;
    d.doc[doc].pending_lo = pending;
    d.doc[doc].pending_hi = 0;
    d.doc[doc].n_mapop = n_map;
    d.doc[doc].n_elems = n_el;
    if (cut_doc) d.doc[doc].flags |= DF_CUT;
    if (has_ml) d.doc[doc].flags |= DF_MOVABLE;
    // (DF_FUSED: few rows per change — one change per keystroke — k_fuse_rows chains the rows into runs, lm_k_fuse.h)
    else if (!any_skip_doc && n_style == 0) d.doc[doc].flags |= DF_PLAIN | ((LM_FUSE_ROWS && (uint64_t)n_valid * 8 > m.n_op) ? DF_FUSED : 0u);
  }
#ifdef LM_PROF_DAG
  DAG_PH(7);
  if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], (unsigned long long)dpacc[i]);
#endif
}

// K7b: one wave per doc — Kahn passes over nodes: replay order, lamports, vv at node heads.
// res_mode (resident documents, lm_k_integrate_span.h DevRes): 0 = a batch: the DAG pass, then the optional checkout;
// 1 = the DAG pass only, and peer_end_all := the applied end of EVERY document (the resident trackers replay every applied
// op, whatever version is rendered); 2 = the checkout only, from peer_end_all — the part a run repeats when the same tables
// are rendered at another version.
LM_KERNEL void k_dag_b(Dev d, DevDag g, uint32_t res_mode) {
  uint32_t doc = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  uint32_t P = m.n_peers, N = m.n_nodes;
  uint64_t vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
  uint32_t n_done = 0;
  if (res_mode == 2) {
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) d.peer_end[m.praw0 + p] = d.peer_end_all[m.praw0 + p];
    lmw::mem_fence();
    lmw::block_sync();
    N = 0;   // (skips the DAG pass below)
  }
  // A peer's nodes become ready in counter order (the causal iterator enforces it, dag/iter.rs:242-266; every node of a
  // well-formed history has its peer's previous op in its causal past), and nodes are numbered by (peer, counter): the only
  // candidates of a pass are the first unfinished node of every peer — P checks per pass instead of N (a 1M-op document whose
  // two peers hand over every 1k ops has 1,000 nodes and needs 1,000 passes: 358 of the 900 ms of a configs[4] batch went here).
  LM_SHARED(uint32_t, s_next, MAX_PEERS);   // first unfinished node of the peer
  LM_SHARED(uint32_t, s_pend, MAX_PEERS);   // one past its last node
  for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_next[p] = 0; s_pend[p] = 0; }
  lmw::block_sync();
  if (res_mode != 2)
    for (uint32_t n = (uint32_t)lane; n < N; n += 64) {
      uint32_t pr = d.chg[d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n]]].peer;
      uint32_t prev = n ? d.chg[d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n - 1]]].peer : NONE;
      uint32_t next = n + 1 < N ? d.chg[d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n + 1]]].peer : NONE;
      if (pr < P && prev != pr) s_next[pr] = n;
      if (pr < P && next != pr) s_pend[pr] = n + 1;
    }
  lmw::block_sync();
  for (uint32_t pass = 0; pass <= N && n_done < N; pass++) {
    uint32_t batch0 = n_done;
    if (m.flags & DF_CUT) {
    // ONE node per pass: of the peers' first unfinished nodes whose dependencies are all done, the one of the LARGEST peer.
    // Any causal order gives the same sequence (Fugue converges; the reference's own order depends on a hash map,
    // dag/iter.rs:268-274) — but not at the same price: integrating a run next to concurrent runs of other peers walks over
    // every sibling of a SMALLER peer and stops at the first one of a larger peer (crdt_rope.rs:187,217: `other.peer >
    // new.peer → break`).  Concurrent branches replayed in descending peer order meet only larger peers' items: the scan ends
    // at its first sibling.  Staying with one peer as long as its next node is ready also keeps the tracker from moving
    // between branches more often than the graph demands.
    {
      uint32_t pick = NONE;
      for (uint32_t p0 = ((P - 1) >> 6) << 6;; p0 -= 64) {
        uint32_t pr = p0 + (uint32_t)lane;
        uint32_t n = pr < P ? s_next[pr] : NONE;
        bool ready = false;
        if (pr < P && n < s_pend[pr] && !g.node_done[m.chg0 + n]) {
          ready = true;
          const ChangeRow& ch = d.chg[d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n]]];
          for (uint32_t k = ch.dep0; k < ch.dep0 + ch.n_dep; k++) {
            uint32_t ci = d.dep_ci[k];
            if (ci == NONE || !g.node_done[m.chg0 + g.chg_node[m.chg0 + ci]]) ready = false;
          }
        }
        uint64_t rm = lmw::ballot(ready);
        if (rm) { pick = p0 + (uint32_t)(63 - __builtin_clzll((unsigned long long)rm)); break; }
        if (p0 == 0) break;
      }
      if (pick != NONE) {
        lmw::block_sync();
        uint32_t n = s_next[pick];
        lmw::block_sync();
        if (lane == 0) { d.node_order[m.chg0 + n_done] = n; s_next[pick] = n + 1; }
        n_done++;
      }
    }
    } else {
    // small documents: every peer's first unfinished node whose dependencies are all done, in one pass (ascending peer = ascending node index)
    for (uint32_t p0 = 0; p0 < P; p0 += 64) {
      uint32_t pr = p0 + (uint32_t)lane;
      uint32_t n = pr < P ? s_next[pr] : NONE;
      bool ready = false;
      if (pr < P && n < s_pend[pr] && !g.node_done[m.chg0 + n]) {
        ready = true;
        const ChangeRow& ch = d.chg[d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n]]];
        for (uint32_t k = ch.dep0; k < ch.dep0 + ch.n_dep; k++) {
          uint32_t ci = d.dep_ci[k];
          if (ci == NONE || !g.node_done[m.chg0 + g.chg_node[m.chg0 + ci]]) ready = false;
        }
      }
      uint64_t rm = lmw::ballot(ready);
      if (ready) { d.node_order[m.chg0 + n_done + (uint32_t)lmw::popc64(rm & ((1ull << lane) - 1))] = n; s_next[pr] = n + 1; }
      n_done += (uint32_t)lmw::popc64(rm);
    }
    }
    lmw::block_sync();   // (node_order is written and read by lanes of this one wave: program order is enough, as in rounds 1-3)
    if (n_done == batch0) { if (lane == 0) LM_SETERR(d.doc[doc].status, ST_INTERNAL); return; }  // cycle: malformed deps
    // finish the batch: lamport + vv at the head, lamports of every change of the node
    for (uint32_t bi = batch0; bi < n_done; bi++) {
      uint32_t n = d.node_order[m.chg0 + bi];
      uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
      const ChangeRow& hc = d.chg[d.chg_sorted[m.chg0 + first]];
      uint32_t* vv = d.vvh + vvh0 + (uint64_t)n * P;
      for (uint32_t p = (uint32_t)lane; p < P; p += 64) vv[p] = d.peer_base[m.praw0 + p];   // (the base version — zeros, ordinarily — is part of every version)
      uint32_t lam = 0;
      for (uint32_t k = hc.dep0; k < hc.dep0 + hc.n_dep; k++) {
        uint32_t q = d.dep_peer[k], c = d.dep_ctr[k];
        uint32_t ci = d.dep_ci[k];
        uint32_t drow = d.chg_sorted[m.chg0 + ci];
        uint32_t dl = d.chg_lamport[drow] + (c - d.chg[drow].ctr) + 1;
        lam = dl > lam ? dl : lam;
        const uint32_t* dv = d.vvh + vvh0 + (uint64_t)g.chg_node[m.chg0 + ci] * P;
        for (uint32_t p = (uint32_t)lane; p < P; p += 64) {
          uint32_t x = dv[p];
          if (p == q && c + 1 > x) x = c + 1;
          if (x > vv[p]) vv[p] = x;
        }
      }
      for (uint32_t i = first + (uint32_t)lane; i <= last; i += 64) {
        uint32_t row = d.chg_sorted[m.chg0 + i];
        d.chg_lamport[row] = lam + (d.chg[row].ctr - hc.ctr);
      }
      if (lane == 0) { g.node_done[m.chg0 + n] = 1; g.node_lam[m.chg0 + n] = lam; }
      lmw::block_sync();
    }
  }
  // ---- critical versions (integrate stage: ts_goto / ts_convert_base, lm_k_integrate_span.h).  The version in front of the
  // i-th node of the replay order is CRITICAL when every node from i on depends on all of it: vv_head(node j) >= everything the
  // nodes in front of i hold, for all j >= i (Eg-walker's critical versions; what lets the reference start a tracker at the
  // common ancestors with the history before them collapsed, tracker.rs:40-60).  Walked backwards with a running minimum of the
  // vv_heads and the version "everything in front of node i" (a peer's nodes are replayed in counter order, so taking node i
  // away sets its peer back to the node's first counter).  Flag (node_done bit 1): critical in front of node i but not in front
  // of node i+1 — a concurrent section begins with node i, the tracker gets its base there.  A history that is one chain
  // (critical everywhere) gets no flag at all, and neither does its last node.
  if (res_mode != 2 && N && (m.flags & DF_CUT)) {
    uint32_t* s_min = s_next;
    uint32_t* s_tot = s_pend;
    lmw::block_sync();
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) { s_min[p] = NONE; s_tot[p] = d.peer_end[m.praw0 + p]; }
    lmw::block_sync();
    bool crit_next = true;
    for (uint32_t i = N; i-- > 0;) {
      uint32_t n = d.node_order[m.chg0 + i];
      uint32_t row = d.chg_sorted[m.chg0 + d.node_first[m.chg0 + n]];
      const ChangeRow& hc = d.chg[row];
      const uint32_t* vv = d.vvh + vvh0 + (uint64_t)n * P;
      if (lane == 0) s_tot[hc.peer] = hc.ctr + d.chg_skip[row];
      for (uint32_t p = (uint32_t)lane; p < P; p += 64) { uint32_t v = vv[p]; if (v < s_min[p]) s_min[p] = v; }
      lmw::block_sync();
      bool below = false;
      for (uint32_t p = (uint32_t)lane; p < P; p += 64) below |= s_min[p] < s_tot[p];
      bool crit = !lmw::any(below);
      if (lane == 0) g.node_done[m.chg0 + n] = 1u | ((crit && !crit_next) ? 2u : 0u);
      crit_next = crit;
      lmw::block_sync();
    }
  }
  // ---- optional checkout (loro.rs:1625-1760): frontiers → version vector (AppDag::frontiers_to_vv,
  // loro_dag.rs:1190-1207) = merge over the ids of (vv at the head of the id's node) ∪ {peer: counter+1}.
  // The version replaces peer_end: integrate / LWW / emit only see ops below it.
  if (res_mode == 1) {
    for (uint32_t p = (uint32_t)lane; p < P; p += 64) d.peer_end_all[m.praw0 + p] = d.peer_end[m.praw0 + p];
    return;
  }
  uint64_t f0 = d.front_off[doc], f1 = d.front_off[doc + 1];
  if (f1 == f0) return;
  int32_t ferr = ST_OK;
  uint64_t cnt = 0;
  {
    // pass 1 (every lane parses the same bytes): well-formed, and every id inside the applied history
    Rd r = rd_make(d.front + f0, f1 - f0);
    cnt = rd_uleb(r);
    if (r.bad || cnt > f1 - f0) ferr = ST_DECODE_ERROR;
    for (uint64_t i = 0; i < cnt && !ferr; i++) {
      uint64_t peer = rd_uleb(r);
      int64_t ctr = rd_zigzag(r);
      if (r.bad) { ferr = ST_DECODE_ERROR; break; }
      uint32_t lo = 0, hi = P;
      while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < peer) lo = mid + 1; else hi = mid; }
      if (lo >= P || d.peer_uniq[m.praw0 + lo] != peer || ctr < 0 || (uint64_t)ctr >= d.peer_end[m.praw0 + lo] ||
          find_change(d, m, lo, (uint32_t)ctr) == NONE)
        ferr = ST_FRONTIERS_NOT_FOUND;
    }
    if (!ferr && rd_left(r) != 0) ferr = ST_DECODE_ERROR;
  }
  lmw::block_sync();  // every lane has read peer_end before it is rewritten
  if (ferr && res_mode == 2) { if (lane == 0) { d.doc[doc].front_err = ferr; d.doc[doc].flags |= DF_FRONT_ERR; } return; }   // (peer_end stays at the latest version)
  if (ferr) { if (lane == 0) LM_SETERR(d.doc[doc].status, ferr); return; }
  if (res_mode == 0) for (uint32_t p = (uint32_t)lane; p < P; p += 64) d.peer_end_all[m.praw0 + p] = d.peer_end[m.praw0 + p];
  lmw::block_sync();
  for (uint32_t p0 = 0; p0 < P; p0 += 64) {
    uint32_t p = p0 + (uint32_t)lane, acc = 0;
    Rd r = rd_make(d.front + f0, f1 - f0);
    (void)rd_uleb(r);
    for (uint64_t i = 0; i < cnt; i++) {
      uint64_t peer = rd_uleb(r);
      uint32_t ctr = (uint32_t)rd_zigzag(r);
      uint32_t lo = 0, hi = P;
      while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < peer) lo = mid + 1; else hi = mid; }
      uint32_t ci = find_change(d, m, lo, ctr);
      if (p < P) {
        uint32_t x = d.vvh[vvh0 + (uint64_t)g.chg_node[m.chg0 + ci] * P + p];
        if (p == lo && ctr + 1 > x) x = ctr + 1;
        acc = x > acc ? x : acc;
      }
    }
    if (p < P) d.peer_end[m.praw0 + p] = acc;
  }
}

// cooperative UTF-8 → scalars of ONE long string with coalesced loads: 64 bytes per step while the text is ASCII (byte =
// scalar, no lane traffic), otherwise 61 — a scalar spans at most 4 bytes, so a lead byte below lane 61 has its tail loaded;
// scalar boundaries come from a ballot over the lead bytes and each lead lane assembles its scalar from the following lanes.
// (32-bit offsets: a value is shorter than its blob, and a blob shorter than 4 GiB.)
// BYTES: the span-granular layout — tb[] takes one byte per scalar, cp[] only the scalars beyond ASCII.
template <bool BYTES>
LM_DEV bool fill_text_coop(const Dev& d, const uint8_t* s, uint32_t nbytes, uint64_t e0, uint32_t len) {
  uint32_t lane = (uint32_t)lmw::lane();
  bool bad = false;
  uint32_t n = 0;  // scalars emitted so far
  for (uint32_t c = 0; c < nbytes;) {
    uint32_t i = c + lane;
    bool inb = i < nbytes;
    uint32_t b = inb ? s[i] : 0u;
    if (!lmw::ballot(b >= 0x80)) {
      if (inb & (n + lane < len)) { if (BYTES) d.tb[e0 + n + lane] = (uint8_t)b; else d.cp[e0 + n + lane] = b; }
      uint32_t took = nbytes - c < 64 ? nbytes - c : 64u;
      n += took; c += 64;
      continue;
    }
    bool last_chunk = c + 61 >= nbytes;
    bool mine = inb & (last_chunk | (lane < 61));   // this lane's byte belongs to this step
    bool lead = mine & ((b & 0xC0) != 0x80);
    uint32_t b1 = lmw::shfl(b, (int)((lane + 1) & 63)), b2 = lmw::shfl(b, (int)((lane + 2) & 63)), b3 = lmw::shfl(b, (int)((lane + 3) & 63));
    uint64_t lm_ = lmw::ballot(lead);
    uint32_t rank = (uint32_t)lmw::popc64(lm_ & ((1ull << lane) - 1));
    if (lead) {
      uint32_t cpv, extra;
      if (b < 0x80) { cpv = b; extra = 0; }
      else if ((b & 0xE0) == 0xC0) { cpv = ((b & 0x1F) << 6) | (b1 & 0x3F); extra = 1; }
      else if ((b & 0xF0) == 0xE0) { cpv = ((b & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F); extra = 2; }
      else if ((b & 0xF8) == 0xF0) { cpv = ((b & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F); extra = 3; }
      else { cpv = 0; extra = 0; bad = true; }
      if (i + extra >= nbytes) bad = true;
      if (extra >= 1 && (b1 & 0xC0) != 0x80) bad = true;
      if (extra >= 2 && (b2 & 0xC0) != 0x80) bad = true;
      if (extra >= 3 && (b3 & 0xC0) != 0x80) bad = true;
      if (lane + extra > 63) bad = true;  // cannot happen: leads above lane 60 only exist in the last chunk
      if (n + rank < len) {
        if (BYTES) { d.tb[e0 + n + rank] = (uint8_t)(extra ? TB_WIDE : cpv); if (extra) d.cp[e0 + n + rank] = cpv; }
        else d.cp[e0 + n + rank] = cpv;
      }
    }
    n += (uint32_t)lmw::popc64(lm_);
    c += 61;
  }
  return lmw::any(bad) || n != len;
}

// K8: one wave per change block — element payload tables.
//  * span-granular leaves (the default): tb[] holds one BYTE per Text element — the scalar itself when it is ASCII, TB_WIDE
//    when the scalar needs more (then cp[] holds it), TB_ANCHOR for a style anchor.  A string with as many bytes as the row
//    has elements is copied four bytes per load / store (it is ASCII, or it is not UTF-8 of that many scalars: the reference
//    decodes the string and counts its chars, outdated_encode_reordered.rs:215-476); the emit stage gathers one byte per
//    element.  (Rounds 1-2 kept a 4-byte scalar per element: 4x the traffic on both sides and a byte-by-byte decode here.
//    Rendering straight from blob byte ranges through a row index was built and measured, tests/tools/wip/
//    text_from_blob_ranges.patch: fine for long rows, 1.8x slower on keystroke-per-change histories, where a run spans
//    hundreds of one-character rows — profiles/r03_bench_text_from_blob_ranges.json.)
//  * List / MovableList inserts: cp[] keeps each item's byte offset (values have no fixed size).
//  * element-granular leaves (LM_SPAN=0, the second implementation): cp[] holds every Text scalar, as in rounds 1-2.
// Rows are taken 64 at a time, ONE ROW PER LANE: typing produces short runs, so each lane handles its own string
// (neighbouring rows' payloads and element slots are adjacent, so the per-lane accesses of a wave fall into the same few
// cache lines).  Strings longer than 64 bytes are left to the cooperative routine above.
LM_KERNEL void k_elem_fill(Dev d) {
  uint32_t bi = (uint32_t)lmw::bid();
  int lane = lmw::lane();
  const BlockDesc& bd = d.blk[bi];
  if (bd.status != ST_OK) return;
  uint32_t doc = bd.doc;
  const DocMeta& m = d.doc[doc];
  if (status_fatal(m.status)) return;
  // resident document whose element layout did not move: the payload slots of the blocks earlier runs filled are still right
  if (d.res_old_blobs && (m.flags & DF_FILL_KEPT) && bd.blob - d.doc_blob[doc] < d.res_old_blobs[doc]) return;
  const bool span = d.span != 0;
  const uint32_t* off = d.boff + (uint64_t)bi * BCN;
  uint32_t op0 = off[BC_OP], n_op = d.bcnt[(uint64_t)bi * BCN + BC_OP];
  uint32_t chg0 = off[BC_CHG];
  uint32_t peer = d.chg[chg0].peer;
  uint64_t ebase = (((uint64_t)m.elem0_hi << 32) | m.elem0_lo) + d.elem_base[m.praw0 + peer];
  uint32_t ext = d.peer_ext[m.praw0 + peer];
  uint64_t doc_data0 = d.blob_off[d.doc_blob[doc]];
  const uint8_t* vsec = d.data + bd.base + bd.sec_rel[SEC_VALUES];
  const uint8_t* lim = vsec + bd.sec_len[SEC_VALUES];
  LM_SHARED(uint32_t, s_inc, 64);
  LM_SHARED(uint32_t, s_nb, 64);
  LM_SHARED(uint32_t, s_src, 64);
  LM_SHARED(uint32_t, s_dst, 64);
  bool bad = false;
  for (uint32_t g0 = 0; g0 < n_op; g0 += 64) {
    uint32_t row = op0 + g0 + (uint32_t)lane;
    bool valid = g0 + (uint32_t)lane < n_op;
    OpRow r;
    r.cidx_kind = 0; r.prop = 0; r.len = 0; r.ctr = 0; r.a0 = r.a1 = 0; r.a2 = 0; r.chg = 0;
    if (valid) r = d.op[row];
    uint32_t kind = (r.cidx_kind >> 16) & 0xff;
    bool want = valid && (kind == OK_TEXT_INS || kind == OK_LIST_INS || kind == OK_STYLE_START || kind == OK_STYLE_END);
    const bool cand_text = want && kind == OK_TEXT_INS;
    if (want && !d.chg_flag[r.chg]) want = false;
    if (want && r.ctr + r.len > ext) want = false;
    // a Text insert of a change that is NOT applied (pending: a dependency is missing) fills no slot, but its string is decoded by the
    // reference all the same when the block is read (String::from_utf8 in the value reader, value.rs:608-859): not UTF-8, or not as
    // many scalars as the row's length, fails the import there whether or not the change is ever applied.  (Found on damaged rich-text
    // documents whose flipped dependency made the damaged change pending: rendered here, DecodeDataCorruptionError there.)
    if (cand_text && !want) {
      const uint8_t* q = d.data + d.op_val[row];
      Rd vq = rd_make(q, q < lim ? (uint64_t)(lim - q) : 0ull);
      uint64_t nb = rd_uleb(vq);
      if (vq.bad || nb > rd_left(vq)) bad = true;
      else {
        uint32_t n = 0;
        for (uint64_t i = 0; i < nb && !bad;) {
          uint32_t b0 = vq.p[i], extra = b0 < 0x80 ? 0u : (b0 & 0xE0) == 0xC0 ? 1u : (b0 & 0xF0) == 0xE0 ? 2u : (b0 & 0xF8) == 0xF0 ? 3u : 4u;
          if (extra == 4 || i + extra >= nb + (extra ? 0 : 1)) { bad = true; break; }
          for (uint32_t k = 1; k <= extra; k++) if ((vq.p[i + k] & 0xC0) != 0x80) bad = true;
          n++; i += extra + 1;
        }
        if (n != r.len) bad = true;
      }
    }
    uint64_t e0 = ebase + r.ctr;
    // a style anchor: cp[] names its op row (CP_ANCHOR | row inside the document) — what k_richtext resolves the StyleOp from; the
    // renderers only look at the marker (tb[] == TB_ANCHOR, cp[] >= CP_ANCHOR)
    if (want && (kind == OK_STYLE_START || kind == OK_STYLE_END)) { if (span) d.tb[e0] = (uint8_t)TB_ANCHOR; d.cp[e0] = CP_ANCHOR | (row - m.op0); want = false; }
    const uint8_t* p = want ? d.data + d.op_val[row] : lim;
    Rd v = rd_make(p, (uint64_t)(lim - p));
    uint64_t nbytes = 0;
    bool is_long = false, flat = false;
    if (want && kind == OK_TEXT_INS) {
      nbytes = rd_uleb(v);
      if (v.bad || nbytes > rd_left(v)) { bad = true; want = false; }
      else if (span && nbytes == r.len) flat = true;   // as many bytes as elements: ASCII (copied below) — or not UTF-8 of that many scalars
      else if (nbytes > 64) { is_long = true; }
      else {
        const uint8_t* s = v.p;
        uint32_t n = 0;
        for (uint32_t i = 0; i < (uint32_t)nbytes;) {
          if (span && i + 4 <= (uint32_t)nbytes && n + 3 < r.len) {
            uint32_t w = ld32u(s + i);
            if (!(w & 0x80808080u)) { st32u(d.tb + e0 + n, w); n += 4; i += 4; continue; }
          }
          if (!span && i + 4 <= (uint32_t)nbytes) {
            // four ASCII bytes at once: the loads go out together and four scalars are stored (the pipeline is issue bound —
            // one byte per trip cost ≈25 instructions per character)
            uint32_t q0 = s[i], q1 = s[i + 1], q2 = s[i + 2], q3 = s[i + 3];
            if (((q0 | q1 | q2 | q3) & 0x80) == 0) {
              if (n + 3 < r.len) { d.cp[e0 + n] = q0; d.cp[e0 + n + 1] = q1; d.cp[e0 + n + 2] = q2; d.cp[e0 + n + 3] = q3; }
              else { if (n < r.len) d.cp[e0 + n] = q0; if (n + 1 < r.len) d.cp[e0 + n + 1] = q1; if (n + 2 < r.len) d.cp[e0 + n + 2] = q2; }
              n += 4; i += 4;
              continue;
            }
          }
          uint32_t b = s[i], cpv, extra;
          if (b < 0x80) { cpv = b; extra = 0; }
          else if ((b & 0xE0) == 0xC0) { cpv = b & 0x1F; extra = 1; }
          else if ((b & 0xF0) == 0xE0) { cpv = b & 0x0F; extra = 2; }
          else if ((b & 0xF8) == 0xF0) { cpv = b & 0x07; extra = 3; }
          else { bad = true; break; }
          if (i + extra >= (uint32_t)nbytes) { bad = true; break; }
          for (uint32_t k = 1; k <= extra; k++) {
            uint32_t t = s[i + k];
            if ((t & 0xC0) != 0x80) bad = true;
            cpv = (cpv << 6) | (t & 0x3F);
          }
          if (n < r.len) {
            if (span) { d.tb[e0 + n] = (uint8_t)(extra ? TB_WIDE : cpv); if (extra) d.cp[e0 + n] = cpv; }
            else d.cp[e0 + n] = cpv;
          }
          n++;
          i += extra + 1;
        }
        if (n != r.len) bad = true;
      }
    } else if (want) {  // list insert: tag 7, count, then the items; the element table keeps each item's byte offset
      (void)rd_u8(v);
      uint64_t cnt = rd_uleb(v);
      if (cnt != r.len) bad = true;
      for (uint32_t k = 0; k < r.len && !bad; k++) {
        d.cp[e0 + k] = (uint32_t)((uint64_t)(v.p - d.data) - doc_data0);
        bool u = false;
        skip_loro_value(v, u);
        if (v.bad) bad = true;
      }
    }
    if (span) {
      // the ASCII strings of these 64 rows as ONE flat copy, four bytes per lane and step: lane → (row, dword of the row) by a
      // search over the running dword counts kept in LDS.  Typing comes in runs: 70 % of a configs[1] document's text sits in
      // rows longer than 64 bytes, and a row-at-a-time copy is a chain of dependent round trips per row; here every load of a
      // step is independent of the others
      uint32_t nd = flat ? ((uint32_t)nbytes + 3) / 4 : 0u;
      uint32_t inc = lmw::scan_incl_add(nd);
      uint32_t total = lmw::bcast(inc, 63);
      if (total) {
        lmw::block_sync();
        s_inc[lane] = inc; s_nb[lane] = (uint32_t)nbytes; s_src[lane] = (uint32_t)(v.p - vsec); s_dst[lane] = r.ctr;
        lmw::block_sync();
        uint32_t hi = 0;
        for (uint32_t f0 = 0; f0 < total; f0 += 64) {
          uint32_t f = f0 + (uint32_t)lane;
          if (f < total) {
            uint32_t lo = 0, hh = 63;                    // first row whose running count exceeds f
            while (lo < hh) { uint32_t mid = (lo + hh) >> 1; if (s_inc[mid] > f) hh = mid; else lo = mid + 1; }
            uint32_t nbr = s_nb[lo];
            uint32_t k4 = 4u * (f - (s_inc[lo] - (nbr + 3) / 4));
            const uint8_t* src = vsec + s_src[lo] + k4;
            uint8_t* dst = d.tb + ebase + s_dst[lo] + k4;
            if (k4 + 4 <= nbr) { uint32_t w = ld32u(src); hi |= w; st32u(dst, w); }
            else for (uint32_t i = 0; k4 + i < nbr; i++) { uint32_t b = src[i]; hi |= b; dst[i] = (uint8_t)b; }
          }
        }
        if (hi & 0x80808080u) bad = true;
      }
    }
    // long strings: the whole wave works on one at a time
    uint64_t lm_ = lmw::ballot(is_long);
    while (lm_) {
      int l = lmw::ffs64(lm_);
      lm_ &= lm_ - 1;
      uint64_t sp = (uint64_t)(v.p - d.data);
      uint32_t sp_lo = lmw::bcast((uint32_t)sp, l), sp_hi = lmw::bcast((uint32_t)(sp >> 32), l);
      uint32_t nb = lmw::bcast((uint32_t)nbytes, l);
      uint32_t e_lo = lmw::bcast((uint32_t)e0, l), e_hi = lmw::bcast((uint32_t)(e0 >> 32), l);
      uint32_t ln = lmw::bcast(r.len, l);
      const uint8_t* str = d.data + (((uint64_t)sp_hi << 32) | sp_lo);
      uint64_t e0l = ((uint64_t)e_hi << 32) | e_lo;
      if (span ? fill_text_coop<true>(d, str, nb, e0l, ln) : fill_text_coop<false>(d, str, nb, e0l, ln)) bad = true;
    }
  }
  if (lmw::any(bad) && lane == 0) LM_SETERR(d.doc[doc].status, ST_DATA_CORRUPTION);
}

}  // namespace lm
