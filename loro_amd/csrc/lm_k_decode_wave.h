// K4 (default): change-block decode, one wavefront per GROUP of DEC_G = 8 consecutive blocks.
//
//  * every block's bytes up to its value payloads (header, meta, cids, keys, op and delete-start columns — ≈1 KB of a
//    ≈3.4 KB text block) are read from HBM once, as coalesced 16-byte vectors, into a per-block LDS slot; the parsers
//    below then read LDS (the one-lane-per-block kernel k_block_decode re-fetched each block ≈16x because 64 lanes x 9
//    live cursors thrash the vector L1).  The payload bytes themselves are only skipped over: the walker reads each
//    value's length prefix from HBM through one monotonic cursor per block;
//  * lane = (block b = lane / 8, role r = lane % 8).  Roles 0-3 own the four EncodedOp columns (container_index
//    DeltaRle, prop DeltaRle, value_type Rle<u8>, len Rle<u32>; block_encode.rs:417-428), roles 4-6 the three
//    EncodedDeleteStartId columns (outdated_encode_reordered.rs:480-489), role 7 walks the value payloads
//    (encoding/value.rs) and assembles the 32-byte OpRow.  All column lanes run the SAME AnyRle cursor code on their own
//    byte range, so one pass of the row loop advances 7 columns x 8 blocks; the row's fields meet in role 7 through
//    lane permutes.  Header, change meta, keys and container ids (small, sequential by format) are parsed by role 0 of
//    every block before the row loop.
//  * nothing lives in scratch: the nested-value frame stack of role 7 is in LDS.
// A block whose head exceeds the slot (thousands of changes or keys) is decoded by the same code straight from HBM.
// Reference: decode_block block_encode.rs:535-706, decode_changes_header block_meta_encode.rs:90-242,
// decode_op outdated_encode_reordered.rs:215-476.  Output tables are identical to k_block_decode's.
#pragma once

namespace lm {

static constexpr uint32_t DEC_G = 8;            // blocks per wave
static constexpr uint32_t DEC_KINDS = 32;       // container kinds of a block cached in LDS (more: read back from cid_raw)
// error bookkeeping of the row loop: the earliest row wins, then the role order of the sequential decoder
LM_DEV void dec_err(uint32_t& key, uint32_t row, uint32_t prio, int32_t code) {
  uint32_t k = (row << 8) | (prio << 4) | (uint32_t)code;
  if (k < key) key = k;
}

LM_KERNEL void k_block_decode_wave(Dev d, uint32_t slot_cap) {
  int lane = lmw::lane();
  uint32_t g0 = (uint32_t)lmw::bid() * DEC_G;
  uint32_t b = (uint32_t)lane >> 3, r = (uint32_t)lane & 7;
  uint32_t bi = g0 + b;
  LM_DYN_SHARED(uint32_t, s_mem);   // DEC_G slots of slot_cap staged bytes | DEC_G x 16 frame-stack words | DEC_G x DEC_KINDS kind bytes
  uint32_t* s_fs = s_mem + DEC_G * (slot_cap / 4);
  uint8_t* s_kinds = (uint8_t*)(s_fs + DEC_G * 16);
  bool have = bi < d.n_blocks;
  // only the scalar fields of the descriptor stay in registers; section extents are read where a section is opened
  struct { uint64_t base; uint32_t counter_start, counter_len, n_changes; } bd = {0, 0, 0, 0};
  const BlockDesc* bdp = d.blk + (have ? bi : 0);
  bool ok = have && bdp->status == ST_OK;
  if (ok) { bd.base = bdp->base; bd.counter_start = bdp->counter_start; bd.counter_len = bdp->counter_len; bd.n_changes = bdp->n_changes; }
  // ---- stage the block up to its value payloads: header | change_meta | cids | keys | positions | ops | delete_start_ids.
  // The payload bytes (mostly text) are not needed here — the walker only reads each value's length prefix, straight from
  // HBM (one monotonic cursor per block: 8 hot lines per wave) — so a slot of `slot_cap` bytes per block is enough.
  uint64_t org = bd.base & ~(uint64_t)15;
  uint32_t head_len = ok ? bdp->sec_rel[SEC_VALUES] : 0u;          // bytes of the block before the values section
  uint32_t span = ok ? (uint32_t)(bd.base - org) + head_len : 0u;  // staged bytes, from the 16-byte boundary below the block
  bool staged = ok && span <= slot_cap;
  uint8_t* slot = (uint8_t*)s_mem + (size_t)b * slot_cap;
  if (staged) {
    struct V16 { uint32_t x, y, z, w; };
    const V16* gsrc = (const V16*)(d.data + org);
    V16* ldst = (V16*)slot;
    uint32_t nv = (span + 15) / 16;
    for (uint32_t i = r; i < nv; i += 8) ldst[i] = gsrc[i];   // `data` carries 64 bytes of slack behind the last blob
  }
  lmw::block_sync();
  if (!lmw::any(ok)) return;   // no decodable block in the group
  const uint8_t* blk_p = staged ? (const uint8_t*)slot + (bd.base - org) : d.data + bd.base;   // first byte of the block
  auto sec = [&](int s_) { return rd_make(blk_p + bdp->sec_rel[s_], bdp->sec_len[s_]); };
  auto abs_of = [&](const uint8_t* p) { return (uint64_t)(p - blk_p) + bd.base; };

  const uint32_t* off = d.boff + (uint64_t)(have ? bi : 0) * BCN;
  const uint32_t* cnt = d.bcnt + (uint64_t)(have ? bi : 0) * BCN;
  uint32_t N = bd.n_changes;
  uint32_t chg0 = 0, dep0 = 0, op0 = 0, key0 = 0, cid0 = 0, peer0 = 0, n_peers = 0, n_ops = 0, n_keys = 0, n_cids = 0;
  if (ok) {
    chg0 = off[BC_CHG]; dep0 = off[BC_DEP]; op0 = off[BC_OP]; key0 = off[BC_KEY]; cid0 = off[BC_CID]; peer0 = off[BC_PEER];
    n_peers = cnt[BC_PEER]; n_ops = cnt[BC_OP]; n_keys = cnt[BC_KEY]; n_cids = cnt[BC_CID];
  }
  int32_t st = ST_OK;          // errors of the sequential sections (role 0) — they precede every row error
  // ---- role 0: header, change meta, keys, container ids
  if (ok && r == 0) {
    Rd h = sec(SEC_HEADER);
    (void)rd_uleb(h);
    for (uint32_t i = 0; i < n_peers; i++) {
      uint64_t v = 0;
      for (int k = 0; k < 8; k++) v |= (uint64_t)rd_u8(h) << (8 * k);
      d.peer_raw[peer0 + i] = v;
    }
    {
      // change lens → counters
      uint64_t known = 0;
      uint32_t ctr = bd.counter_start;
      for (uint32_t i = 0; i < N; i++) {
        uint64_t l;
        if (i + 1 < N) { l = rd_uleb(h); known += l; if (known > bd.counter_len) { st = ST_DECODE_ERROR; l = 0; } }
        else l = bd.counter_len - (known > bd.counter_len ? bd.counter_len : known);
        ChangeRow c;
        c.peer = 0; c.ctr = ctr; c.len = (uint32_t)l; c.dep0 = 0; c.n_dep = 0; c.op0 = 0; c.n_op = 0; c.blk = bi;
        d.chg[chg0 + i] = c;
        ctr += (uint32_t)l;
      }
      // dep_on_self BoolRle[N] and other-dep counts AnyRle<usize>[N] advance together: one pass over the changes
      BoolCur bc = bool_make(h);
      // the dep-count column starts where the BoolRle ends: run the bool cursor to its end first (N values)
      Rd after_bool = h;
      {
        BoolCur t = bc;
        for (uint32_t i = 0; i < N; i++) (void)bool_next(t);
        if (t.rem != 0) t.r.bad = true;
        after_bool = t.r;
      }
      RleCur dc = rle_make(after_bool);
      uint32_t dcur = dep0;
      for (uint32_t i = 0; i < N; i++) {
        uint32_t ds = bool_next(bc) ? 1u : 0u;
        uint64_t others = rle_next_uvar(dc);
        ChangeRow c = d.chg[chg0 + i];
        if (dcur + ds + others > dep0 + cnt[BC_DEP]) { st = ST_DECODE_ERROR; others = 0; ds = 0; }
        c.dep0 = dcur;
        c.n_dep = ds + (uint32_t)others;
        c.op0 = ds;   // kept until the dep columns are read
        if (ds) {
          if (c.ctr == 0) st = ST_DECODE_ERROR;
          d.dep_peer[dcur] = 0;
          d.dep_ctr[dcur] = c.ctr ? c.ctr - 1 : 0;
        }
        dcur += c.n_dep;
        d.chg[chg0 + i] = c;
      }
      if (dc.rem != 0) dc.r.bad = true;
      if (dcur - dep0 != cnt[BC_DEP]) st = ST_DECODE_ERROR;
      // dep peer idx AnyRle<u32>[D]
      RleCur pc = rle_make(dc.r);
      uint64_t D = 0;
      for (uint32_t i = 0; i < N; i++) {
        ChangeRow c = d.chg[chg0 + i];
        for (uint32_t k = c.dep0 + c.op0; k < c.dep0 + c.n_dep; k++) {
          uint64_t pi = rle_next_uvar(pc);
          if (pi >= n_peers) { st = ST_DECODE_ERROR; pi = 0; }
          d.dep_peer[k] = (uint32_t)pi;
          D++;
        }
      }
      if (pc.rem != 0) pc.r.bad = true;
      // dep counters DeltaOfDelta[D]
      Rd hr = pc.r;
      DodCur dd = dod_make(hr);
      for (uint32_t i = 0; i < N && D; i++) {
        ChangeRow c = d.chg[chg0 + i];
        for (uint32_t k = c.dep0 + c.op0; k < c.dep0 + c.n_dep; k++) {
          int64_t v = dod_next(dd);
          if (v < 0 || v >= (int64_t)MAX_COUNTER) { st = st ? st : ST_DECODE_ERROR; v = 0; }
          d.dep_ctr[k] = (uint32_t)v;
        }
      }
      dod_finish(dd, hr, D);
      // wire lamports (DeltaOfDelta[N-1]): shape only — lamports are recomputed from deps on import
      // (outdated_encode_reordered.rs:61-62, loro_dag.rs:1179-1187)
      DodCur ld = dod_make(hr);
      for (uint32_t i = 0; i + 1 < N; i++) (void)dod_next(ld);
      dod_finish(ld, hr, N - 1);
      if (h.bad || bc.r.bad || after_bool.bad || dc.r.bad || pc.r.bad || hr.bad) st = st ? st : ST_DECODE_ERROR;
      for (uint32_t i = 0; i < N; i++) d.chg[chg0 + i].op0 = 0;
    }
    {  // change_meta: timestamps + message lengths, shape only (block_encode.rs:563-571)
      Rd m = sec(SEC_META);
      DodCur td = dod_make(m);
      for (uint32_t i = 0; i < N; i++) (void)dod_next(td);
      dod_finish(td, m, N);
      RleCur mc = rle_make(m);
      uint64_t tot = 0;
      for (uint32_t i = 0; i < N; i++) tot += rle_next_uvar(mc);
      if (mc.r.bad || tot > rd_left(mc.r)) st = st ? st : ST_DATA_CORRUPTION;
    }
    {  // keys
      Rd k = sec(SEC_KEYS);
      for (uint32_t i = 0; i < n_keys; i++) {
        uint64_t l = rd_uleb(k);
        d.key_off[key0 + i] = abs_of(k.p);
        d.key_len[key0 + i] = (uint32_t)l;
        rd_skip(k, l);
      }
      if (k.bad) st = st ? st : ST_DECODE_ERROR;
    }
    {  // container ids (arena.rs:39-105)
      Rd k = sec(SEC_CIDS);
      if (n_cids) (void)rd_uleb(k);
      for (uint32_t i = 0; i < n_cids; i++) {
        uint64_t fields = rd_uleb(k);
        uint32_t is_root = rd_u8(k), kind = rd_u8(k);
        uint64_t pidx = rd_uleb(k);
        int64_t koc = rd_zigzag(k);
        if (fields != 4) st = st ? st : ST_DECODE_ERROR;
        uint32_t* w = d.cid_raw + (uint64_t)(cid0 + i) * 4;
        if (is_root) { if (koc < 0 || (uint64_t)koc >= n_keys) { st = st ? st : ST_DATA_CORRUPTION; koc = 0; } }
        else { if (pidx >= n_peers) { st = st ? st : ST_DATA_CORRUPTION; pidx = 0; } if (koc < 0 || koc >= (int64_t)MAX_COUNTER) { st = st ? st : ST_UNSUPPORTED; koc = 0; } }
        w[0] = kind | (is_root ? 0x100u : 0u);
        w[1] = (uint32_t)pidx;
        w[2] = (uint32_t)koc;
        w[3] = bi;
        if (kind > CK_COUNTER) st = st ? st : ST_DECODE_ERROR;   // ContainerType::try_from_u8 fails (loro-common/src/lib.rs:748-793)
        if (i < DEC_KINDS) s_kinds[b * DEC_KINDS + i] = (uint8_t)kind;
      }
      if (k.bad) st = st ? st : ST_DECODE_ERROR;
    }
  }
  lmw::block_sync();   // kinds (LDS) and the change rows (HBM, read back by role 7 of the same block) are in place
  // ---- column cursors
  RleCur col = rle_make(rd_make(blk_p, 0));
  bool has_del = false;
  uint32_t errk = 0xffffffffu;   // earliest row error of this lane
  bool shape_bad = false;        // ops / delete section framing
  if (ok) {
    Rd o = sec(SEC_OPS);
    uint64_t outer = rd_uleb(o), ncols = rd_uleb(o);
    Rd mine = rd_make(blk_p, 0);
    for (uint32_t c = 0; c < 4; c++) { Rd cc = rd_bytes(o); if (r == c) mine = cc; }   // (kept in registers: no indexed array of readers)
    if (outer != 1 || ncols != 4 || o.bad) shape_bad = true;
    Rd ds = sec(SEC_DEL);
    has_del = ds.p < ds.end;
    if (has_del) {
      uint64_t douter = rd_uleb(ds), dcols = rd_uleb(ds);
      if (douter != 1 || dcols != 3) shape_bad = true;
      for (uint32_t c = 4; c < 7; c++) { Rd cc = rd_bytes(ds); if (r == c) mine = cc; }
    }
    col = rle_make(mine);
  }
  // ---- role 7: value walker + row→change bookkeeping
  Rd v = rd_make(blk_p, 0);
  uint64_t counter = bd.counter_start;
  uint32_t change_index = 0, rows_in_change = 0;
  uint64_t next_boundary = 0;
  bool unsupported = false;
  if (ok && r == 7) {
    v = rd_make(d.data + bd.base + bdp->sec_rel[SEC_VALUES], bdp->sec_len[SEC_VALUES]);
    next_boundary = N > 1 ? d.chg[chg0 + 1].ctr : (uint64_t)bd.counter_start + bd.counter_len;
    d.chg[chg0].op0 = op0;
  }
  uint32_t max_rows = lmw::reduce_max(ok ? n_ops : 0u);
  int base_lane = lane & ~7;
  for (uint32_t row = 0; row < max_rows; row++) {
    bool act = ok && row < n_ops;
    // 1. the four op columns advance
    uint32_t x = 0;
    if (act && r < 4) {
      if (r == 2) x = rle_next_u8(col) & 0x7f;
      else if (r == 3) { uint64_t l = rle_next_uvar(col); if (l > MAX_COUNTER) { dec_err(errk, row, 5, ST_UNSUPPORTED); l = MAX_COUNTER; } x = (uint32_t)l; }
      else {
        int64_t w = rle_next_delta(col);
        if (r == 0) { if (w < 0 || (uint64_t)w >= n_cids) { dec_err(errk, row, 0, ST_DATA_CORRUPTION); w = 0; } }
        else if (w < INT32_MIN || w > INT32_MAX) { dec_err(errk, row, 1, ST_DECODE_ERROR); w = 0; }
        x = (uint32_t)w;
      }
    }
    uint32_t ci = lmw::shfl(x, base_lane), prop = lmw::shfl(x, base_lane + 1), vt = lmw::shfl(x, base_lane + 2), len = lmw::shfl(x, base_lane + 3);
    uint32_t ckind = 0xff;
    if (act && n_cids) ckind = ci < DEC_KINDS ? s_kinds[b * DEC_KINDS + ci] : (d.cid_raw[(uint64_t)(cid0 + ci) * 4] & 0xff);
    // 2. delete-start columns advance on DeleteSeq rows of sequence containers
    bool take_del = act && vt == 9 && (ckind == CK_TEXT || ckind == CK_LIST || ckind == CK_MOVABLE);
    uint32_t y = 0;
    if (take_del && has_del && r >= 4 && r < 7) {
      int64_t w = rle_next_delta(col);
      if (r == 4) { if (w < 0 || (uint64_t)w >= n_peers) { dec_err(errk, row, 6, ST_DATA_CORRUPTION); w = 0; } }
      else if (r == 5) { if (w < 0 || w >= (int64_t)MAX_COUNTER) { dec_err(errk, row, 8, ST_DATA_CORRUPTION); w = 0; } }
      else if (w == 0 || w > (int64_t)MAX_COUNTER || w < -(int64_t)MAX_COUNTER) { dec_err(errk, row, 7, ST_DATA_CORRUPTION); w = 1; }
      if (col.r.bad) dec_err(errk, row, 9, ST_DATA_CORRUPTION);
      y = (uint32_t)w;
    }
    uint32_t dpeer = lmw::shfl(y, base_lane + 4), dctr = lmw::shfl(y, base_lane + 5), dlen = lmw::shfl(y, base_lane + 6);
    // 3. role 7: value payload (docs/encoding.md §10), decode_op mapping, the 32-byte row
    if (act && r == 7) {
      OpRow orow;
      orow.cidx_kind = ci;  // block-local until k_remap
      orow.prop = (int32_t)prop;
      orow.len = len;
      orow.ctr = (uint32_t)counter;
      orow.a0 = 0; orow.a1 = 0; orow.a2 = 0;
      orow.chg = chg0 + change_index;
      uint64_t val_at = (uint64_t)(v.p - d.data);
      uint32_t kind = OK_OTHER, mark_len = 0;
      bool is_list_value = false;
      uint32_t* fs = s_fs + b * 16;
      switch (vt) {
        case 0: case 1: case 2: case 8: case 9: break;
        case 3: (void)rd_sleb(v); break;
        case 4: rd_skip(v, 8); break;
        case 5: case 6: { uint64_t l = rd_uleb(v); rd_skip(v, l); break; }
        case 7: (void)rd_uleb(v); break;
        case 10: (void)rd_sleb(v); break;
        case 11: {
          is_list_value = v.p < v.end && *v.p == 7;
          if (is_list_value) { Rd t = v; (void)rd_u8(t); orow.a0 = (uint32_t)rd_uleb(t); }
          // (values of containers outside the device scope are never rendered: any shape is accepted)
          skip_loro_value_fs(v, unsupported, ckind == CK_MAP ? 0 : (is_list_value && ckind == CK_LIST ? 1 : (ckind > CK_TEXT ? 16 : -1)), fs);
          break;
        }
        case 12: {
          (void)rd_u8(v);
          mark_len = (uint32_t)rd_uleb(v);
          uint64_t key_idx = rd_uleb(v);
          if (key_idx >= n_keys) dec_err(errk, row, 2, ST_DATA_CORRUPTION);
          bool u = false;
          skip_loro_value_fs(v, u, -1, fs);
          break;
        }
        case 13: { (void)rd_uleb(v); uint32_t isn = rd_u8(v); (void)rd_uleb(v); if (!isn) (void)rd_uleb(v); break; }
        case 14: (void)rd_uleb(v); (void)rd_uleb(v); (void)rd_uleb(v); break;
        case 15: { (void)rd_uleb(v); (void)rd_uleb(v); bool u = false; skip_loro_value_fs(v, u, -1, fs); break; }
        case 16: {
          (void)rd_uleb(v); (void)rd_uleb(v); (void)rd_uleb(v);
          uint32_t isn = rd_u8(v);
          if (!isn) { (void)rd_uleb(v); (void)rd_uleb(v); }
          break;
        }
        default: { uint64_t l = rd_uleb(v); rd_skip(v, l); break; }
      }
      // decode_op mapping (outdated_encode_reordered.rs:215-476)
      if (ckind == CK_TEXT) {
        if (vt == 5) kind = OK_TEXT_INS;
        else if (vt == 9) kind = OK_DEL;
        else if (vt == 12) { kind = OK_STYLE_START; orow.a0 = mark_len; }
        else if (vt == 0) kind = OK_STYLE_END;
        else dec_err(errk, row, 3, ST_DATA_CORRUPTION);
      } else if (ckind == CK_MAP) {
        if ((int32_t)prop < 0 || prop >= n_keys) dec_err(errk, row, 3, ST_DATA_CORRUPTION);
        if (vt == 8) kind = OK_MAP_DEL;
        else if (vt == 11) kind = OK_MAP_SET;
        else dec_err(errk, row, 3, ST_DATA_CORRUPTION);
      } else if (ckind == CK_LIST) {
        if (vt == 11) { if (is_list_value) kind = OK_LIST_INS; else dec_err(errk, row, 3, ST_DATA_CORRUPTION); }
        else if (vt == 9) kind = OK_DEL;
        else dec_err(errk, row, 3, ST_DATA_CORRUPTION);
      }
      if (take_del) {
        if (!has_del) dec_err(errk, row, 4, ST_DATA_CORRUPTION);
        else { orow.a0 = dpeer; orow.a1 = dctr; orow.a2 = (int32_t)dlen; }
      }
      orow.cidx_kind |= kind << 16;
      d.op[op0 + row] = orow;
      d.op_val[op0 + row] = val_at;
      d.op_blk[op0 + row] = bi;
      counter += len;
      if (counter > MAX_COUNTER) { dec_err(errk, row, 10, ST_UNSUPPORTED); counter = MAX_COUNTER; }
      if (change_index >= N) { dec_err(errk, row, 11, ST_DATA_CORRUPTION); change_index = N - 1; }
      rows_in_change++;
      if (counter >= next_boundary && change_index + 1 < N) {
        d.chg[chg0 + change_index].n_op = rows_in_change;
        rows_in_change = 0;
        change_index++;
        d.chg[chg0 + change_index].op0 = op0 + row + 1;
        next_boundary = change_index + 1 < N ? d.chg[chg0 + change_index + 1].ctr : (uint64_t)bd.counter_start + bd.counter_len;
      }
    }
  }
  // ---- close the block: remaining change rows, the reader flags, the block status
  uint32_t tail = 0;   // low-priority findings (bit 0: a reader ran off its column, bit 1: counter does not add up, bit 2: unsupported shape)
  if (ok && r == 7) {
    d.chg[chg0 + change_index].n_op = rows_in_change;
    // changes that received no rows still need a valid op0
    for (uint32_t i = change_index + 1; i < N; i++) { d.chg[chg0 + i].op0 = op0 + n_ops; d.chg[chg0 + i].n_op = 0; }
    if (v.bad) tail |= 1;
    if (counter != (uint64_t)bd.counter_start + bd.counter_len) tail |= 2;
    if (unsupported) tail |= 4;
  }
  if (ok && r < 4 && col.r.bad) tail |= 1;
  if (ok && shape_bad) tail |= 8;
  // combine over the 8 lanes of the block
  for (int m = 1; m < 8; m <<= 1) {
    uint32_t oe = lmw::shfl_xor(errk, m), ot = lmw::shfl_xor(tail, m);
    errk = oe < errk ? oe : errk;
    tail |= ot;
  }
  if (ok && r == 0) {
    if (st == ST_OK && (tail & 8)) st = ST_DECODE_ERROR;
    if (st == ST_OK && errk != 0xffffffffu) st = (int32_t)(errk & 0xf);
    if (st == ST_OK && (tail & 1)) st = ST_DECODE_ERROR;
    if (st == ST_OK && (tail & 2)) st = ST_DATA_CORRUPTION;
    if (st == ST_OK && (tail & 4)) st = ST_UNSUPPORTED;
    d.blk[bi].status = st;
  }
}

}  // namespace lm
