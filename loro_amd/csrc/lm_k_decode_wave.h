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
//    (encoding/value.rs).  Rows are decoded eight at a time in phases that keep the active lanes on ONE code path:
//    every column lane runs the same mode-selected AnyRle cursor (rle_next_any) and the same table-driven range check
//    on its own byte range — first the op columns for the 8 rows, then the delete-start columns once per DeleteSeq row
//    of the chunk; the walker then passes over the 8 payloads; finally lane = (block, row) maps each row (decode_op)
//    and stores the 32-byte OpRow, 64 rows per store.  Fields meet through two small LDS tables (s_x, s_w), not lane
//    permutes.  (The first version ran all seven roles' code for every single row: 0.66 M instructions per configs[1]
//    document against 0.47 M now.)
//    Header, change meta, keys and container ids (small, sequential by format) are parsed by role 0 of every block
//    before the row loop.
//  * nothing lives in scratch: the nested-value frame stack of role 7 is in LDS.
// A block whose head exceeds the slot (thousands of changes or keys) has its op / delete-start columns staged alone; one whose
// columns do not fit either is decoded by the same code straight from HBM.
// Reference: decode_block block_encode.rs:535-706, decode_changes_header block_meta_encode.rs:90-242,
// decode_op outdated_encode_reordered.rs:215-476.  Output tables are identical to k_block_decode's.
#pragma once

namespace lm {

#ifndef LM_DEC_G
#define LM_DEC_G 8
#endif
#ifndef LM_NO_PARTIAL_STAGE
#define LM_NO_PARTIAL_STAGE 0   // 1: a block whose head exceeds the LDS slot is not staged at all (rounds 2-3; A/B builds)
#endif
static constexpr uint32_t DEC_G = LM_DEC_G;     // blocks per wave (8 lanes each; -DLM_DEC_G=4: half the lanes idle, half the LDS per wave — twice the waves per CU)
static constexpr uint32_t DEC_R = 8;            // rows per chunk (one lane per row in the assembly phase)
static constexpr uint32_t DEC_WW = 6;           // words the walker hands over per row: value offset lo/hi, aux, flags, counter, change
static constexpr uint32_t DEC_VW2 = 128;        // window of the integer-values fast path (8 values of up to 11 bytes, rounded to 16-byte loads)
static constexpr uint32_t DEC_LDS_FIXED = DEC_G * 16 * 4 + DEC_G * DEC_R * 8 * 4 + DEC_G * DEC_R * DEC_WW * 4 + DEC_G * DEC_VW + DEC_G * DEC_VW2;   // frame stacks + s_x + s_w + value windows
static constexpr uint32_t DEC_KINDS = 16;       // container kinds of a block cached in LDS (more: read back from cid_raw)
// error bookkeeping of the row loop: the earliest row wins, then the role order of the sequential decoder
LM_DEV void dec_err(uint32_t& key, uint32_t row, uint32_t prio, int32_t code) {
  uint32_t k = (row << 8) | (prio << 4) | (uint32_t)code;
  if (k < key) key = k;
}

#ifdef LM_PROF_DEC
#define DEC_PH(i) do { uint64_t n_ = lmw::clock(); pacc[i] += n_ - ptp; ptp = n_; } while (0)
#else
#define DEC_PH(i) do {} while (0)
#endif
// (the LDS slots bound the occupancy at ≈2.5 waves per SIMD: a register budget for three — 168 VGPRs — costs nothing and keeps everything out of scratch)
// (head_lo, head_hi]: the launch takes the groups whose largest head span lies in that range — the host launches the kernel once
// for every batch and a second time, with larger slots, when k_block_count met heads beyond the default slot (a batch of Map
// documents: every block carries a key table of a few KB; parsed from HBM by one lane, key after key, those tables were half of
// configs[2]'s decode).  Each group is decoded by exactly one of the launches; the other's wave leaves at once.
// I64 = true: the instantiation with the integer-values fast path of the walker (k_block_decode_wave_map: the second launch — batches
// made of Map blocks); text blocks, which never take it, are decoded by the one without.
template <bool I64>
LM_DEV void block_decode_wave_body(Dev d, uint32_t slot_cap, uint32_t head_lo, uint32_t head_hi) {
  int lane = lmw::lane();
#ifdef LM_PROF_DEC
  uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptp = lmw::clock();   // 0 stage, 1 head (role 0), 2 cursors, 3 A1, 4 T + A2, 5 W, 6 B, 7 close
#endif
  uint32_t g0 = (uint32_t)lmw::bid() * DEC_G;
  uint32_t b = (uint32_t)lane >> 3, r = (uint32_t)lane & 7;
  uint32_t bi = g0 + b;
  LM_DYN_SHARED(uint32_t, s_mem);   // DEC_G slots of slot_cap staged bytes | DEC_G x 16 frame-stack words | s_x | s_w | DEC_G value windows | DEC_G x DEC_KINDS kind bytes
  uint32_t* s_fs = s_mem + DEC_G * (slot_cap / 4);
  uint32_t* s_x = s_fs + DEC_G * 16;                  // DEC_G x DEC_R rows x 8 column words
  uint32_t* s_w = s_x + DEC_G * DEC_R * 8;            // DEC_G x DEC_R rows x DEC_WW walker words
  uint8_t* s_vw = (uint8_t*)(s_w + DEC_G * DEC_R * DEC_WW);   // DEC_G value windows of DEC_VW bytes (16-byte aligned: slot_cap and every table in front are multiples of 16)
  uint8_t* s_vw2 = s_vw + DEC_G * DEC_VW;             // DEC_G windows of DEC_VW2 bytes (the integer-values fast path)
  uint8_t* s_kinds = s_vw2 + DEC_G * DEC_VW2;
  bool have = b < DEC_G && bi < d.n_blocks;
  // only the scalar fields of the descriptor stay in registers; section extents are read where a section is opened
  struct { uint64_t base; uint32_t counter_start, counter_len, n_changes; } bd = {0, 0, 0, 0};
  const BlockDesc* bdp = d.blk + (have ? bi : 0);
  bool ok = have && bdp->status == ST_OK && !(d.doc_fused && d.doc_fused[bdp->doc]);   // (a fused Map document's blocks: k_block_head + k_map_fused)
  if (ok) { bd.base = bdp->base; bd.counter_start = bdp->counter_start; bd.counter_len = bdp->counter_len; bd.n_changes = bdp->n_changes; }
  // ---- stage the block up to its value payloads: header | change_meta | cids | keys | positions | ops | delete_start_ids.
  // The payload bytes (mostly text) are not needed here — the walker only reads each value's length prefix, straight from
  // HBM (one monotonic cursor per block: 8 hot lines per wave) — so a slot of `slot_cap` bytes per block is enough.
  // A head that does not fit the slot — a block of Map writes carries its own key table, hundreds of keys — still has its op and
  // delete-start COLUMNS staged when those fit (they are the last two sections of the head): header / meta / cids / keys are then
  // parsed from HBM by role 0, once, and the row loop's cursors read LDS as for any other block.  (Unstaged, every cursor of the
  // block chases bytes through HBM — and slows the trips of all eight blocks of its wave: configs[2] spent 57 of its 118 ms here.)
  // Every LDS address is formed as slot + (non-negative offset): a generic pointer rebased to the block's first byte would lie
  // BELOW the LDS aperture for a block staged from its ops section on, and a flat access whose base register is outside the
  // aperture goes to global memory whatever its immediate offset adds (the round-3 attempt died on the GPU this way).
  uint32_t head_len = ok ? bdp->sec_rel[SEC_VALUES] : 0u;          // bytes of the block before the values section
  {
    uint32_t gmax = lmw::reduce_max(ok ? (uint32_t)(bd.base & 15) + head_len : 0u);
    if (!(gmax > head_lo && gmax <= head_hi) && !(gmax == 0 && head_lo == 0)) return;   // the other launch's group
  }
  const bool head_fits = ok && (uint32_t)(bd.base & 15) + head_len <= slot_cap;
  const uint32_t from = (!ok || head_fits || LM_NO_PARTIAL_STAGE) ? 0u : bdp->sec_rel[SEC_OPS];   // first staged byte of the block
  const uint64_t org = (bd.base + from) & ~(uint64_t)15;           // staged bytes start at the 16-byte boundary below it
  uint32_t span = ok ? (uint32_t)(bd.base + head_len - org) : 0u;
  bool staged = ok && span <= slot_cap;                            // the op / delete-start columns sit in LDS
  const bool head_lds = staged && from == 0;                       // … and so does the rest of the head
  uint8_t* slot = (uint8_t*)s_mem + (size_t)b * slot_cap;
#ifdef LM_EMU_TRACE
  if (getenv("LM_EMU_STAGE") && ok && r == 0) fprintf(stderr, "STAGE %s head %u cols %u slot %u\n", head_lds ? "whole" : staged ? "columns" : "none", head_len, head_len - bdp->sec_rel[SEC_OPS], slot_cap);
#endif
  if (staged) {
    struct V16 { uint32_t x, y, z, w; };
    const V16* gsrc = (const V16*)(d.data + org);
    V16* ldst = (V16*)slot;
    uint32_t nv = (span + 15) / 16;
    for (uint32_t i = r; i < nv; i += 8) ldst[i] = gsrc[i];   // `data` carries 64 bytes of slack behind the last blob
  }
  lmw::block_sync();
  DEC_PH(0);
  if (!lmw::any(ok)) return;   // no decodable block in the group
  auto sec = [&](int s_) {
    const uint64_t at = bd.base + bdp->sec_rel[s_];   // absolute offset of the section
    const bool lds = head_lds || (staged && (s_ == (int)SEC_OPS || s_ == (int)SEC_DEL));
    return rd_make(lds ? (const uint8_t*)slot + (uint32_t)(at - org) : d.data + at, bdp->sec_len[s_]);
  };
  auto abs_of = [&](const uint8_t* p) { return head_lds ? (uint64_t)(p - (const uint8_t*)slot) + org : (uint64_t)(p - d.data); };   // (head sections only)
  const uint8_t* const blk_p = d.data;   // (start value of readers that are set later)

  const uint32_t* off = d.boff + (uint64_t)(have ? bi : 0) * BCN;
  const uint32_t* cnt = d.bcnt + (uint64_t)(have ? bi : 0) * BCN;
  uint32_t N = bd.n_changes;
  uint32_t chg0 = 0, dep0 = 0, op0 = 0, key0 = 0, cid0 = 0, peer0 = 0, n_peers = 0, n_ops = 0, n_keys = 0, n_cids = 0;
  if (ok) {
    chg0 = off[BC_CHG]; dep0 = off[BC_DEP]; op0 = off[BC_OP]; key0 = off[BC_KEY]; cid0 = off[BC_CID]; peer0 = off[BC_PEER];
    n_peers = cnt[BC_PEER]; n_ops = cnt[BC_OP]; n_keys = cnt[BC_KEY]; n_cids = cnt[BC_CID];
  }
  int32_t st = ST_OK;          // errors of the sequential sections (role 0) — they precede every row error
  // ---- a block of many changes (one change per keystroke: ≈200 per block): the change rows and the own-peer dependency rows are
  // written by ALL EIGHT lanes of the block, an eighth of the changes each.  The three columns involved — change lengths (N-1
  // varints), dep_on_self (BoolRle), other-dependency counts (AnyRle) — follow one another without length prefixes, so every lane
  // walks them whole, but cheaply (sums and positions only: runs are taken in one step), and spends the full price — rows, stores —
  // on its own changes only.  No lane talks to another: all eight compute the same totals and reach the same verdict.  ANY anomaly
  // (a reader that runs out, lengths beyond the block's counters, dependency rows that do not add up, a first change that depends on
  // "its peer's counter -1") leaves everything to role 0's sequential walk below, which owns the error codes.
  // (Role 0 alone spent 57 % of the decoder's time on such blocks — profiles/r04_decoder_phases.log — with seven lanes idle.)
  bool par = false;
  Rd par_h = rd_make(blk_p, 0);           // role 0 continues from these when the shared walk stands
  Rd par_after_bool = par_h, par_dc_end = par_h;
  uint64_t par_known = 0, par_others = 0;
  if (ok && N >= 16) {
    Rd h = sec(SEC_HEADER);
    (void)rd_uleb(h);
    rd_skip(h, (uint64_t)n_peers * 8);
    const uint32_t C = (N + 7) / 8;
    const uint32_t i0 = r * C < N ? r * C : N, i1 = i0 + C < N ? i0 + C : N;
    // change lengths: the counter my chunk starts at, the sum of all, where the column ends
    Rd h_mine = h;
    uint64_t pre = 0, known = 0;
    bool empty_change = false;
    for (uint32_t i = 0; i + 1 < N; i++) {
      if (i == i0) { h_mine = h; pre = known; }
      const uint64_t l = rd_uleb(h);
      empty_change |= l == 0;
      known += l;
    }
    if (i0 + 1 >= N) { h_mine = h; pre = known; }
    bool fine = !h.bad && !empty_change && known < bd.counter_len;   // (every change holds an id: no counter is met twice, none is 0 behind the first)
    // dep_on_self: values in front of my chunk, in the whole column; the column must end with its N-th value
    BoolCur bcur = bool_make(h);
    { BoolCur f = bcur; if (bool_next(f) && bd.counter_start == 0) fine = false; }
    const uint64_t t_pre = bool_count(bcur, i0);
    BoolCur b_mine = bcur;
    const uint64_t t_all = t_pre + bool_count(bcur, N - i0);
    if (bcur.rem != 0 || bcur.r.bad) fine = false;
    // other-dependency counts: likewise
    RleCur dcur_c = rle_make(bcur.r);
    const uint64_t o_pre = rle_sum_uvar(dcur_c, i0);
    RleCur d_mine = dcur_c;
    const uint64_t o_all = o_pre + rle_sum_uvar(dcur_c, N - i0);
    if (dcur_c.rem != 0 || dcur_c.r.bad || t_all + o_all != (uint64_t)cnt[BC_DEP] || o_all > (uint64_t)cnt[BC_DEP]) fine = false;
    if (fine) {
      par = true;
      par_h = h; par_after_bool = bcur.r; par_dc_end = dcur_c.r; par_known = known; par_others = o_all;
      uint32_t ctr = bd.counter_start + (uint32_t)pre;
      uint32_t dc_at = dep0 + (uint32_t)(t_pre + o_pre);
      uint64_t kn = pre;
      for (uint32_t i = i0; i < i1; i++) {
        uint64_t l;
        if (i + 1 < N) { l = rd_uleb(h_mine); kn += l; } else l = bd.counter_len - known;
        const uint32_t ds = bool_next(b_mine) ? 1u : 0u;
        const uint32_t others = (uint32_t)rle_next_uvar(d_mine);
        ChangeRow c;
        c.peer = 0; c.ctr = ctr; c.len = (uint32_t)l; c.dep0 = dc_at; c.n_dep = ds + others; c.op0 = 0; c.n_op = 0; c.blk = bi;
        d.chg[chg0 + i] = c;
        if (ds && dc_at < dep0 + cnt[BC_DEP]) { d.dep_peer[dc_at] = 0; d.dep_ctr[dc_at] = ctr - 1; }   // (the bound: a sum that wrapped around must not become an address)
        dc_at += ds + others;
        ctr += (uint32_t)l;
      }
    }
  }
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
      // change lens → counters.  The passes below never read a ChangeRow back: what a later pass needs of an earlier one (a
      // change's counter, its dep_on_self bit, its dependency count) comes from walking the same columns again — LDS reads for a
      // staged head.  (Each pass used to load every ChangeRow it had stored: three dependent global round trips per change for
      // ONE lane per block — with one change per keystroke, ≈200 changes per block, that was most of the decoder's time on such
      // blocks: 52 ms of the heterogeneous batch's decode.)
      const Rd h_lens = h;              // the N-1 change lengths start here
      uint64_t known = par ? par_known : 0;
      uint32_t ctr = bd.counter_start;
      if (par) h = par_h;
      for (uint32_t i = 0; i < N && !par; i++) {
        uint64_t l;
        if (i + 1 < N) { l = rd_uleb(h); known += l; if (known > bd.counter_len) { st = ST_DECODE_ERROR; l = 0; } }
        else l = bd.counter_len - (known > bd.counter_len ? bd.counter_len : known);
        ChangeRow c;
        c.peer = 0; c.ctr = ctr; c.len = (uint32_t)l; c.dep0 = 0; c.n_dep = 0; c.op0 = 0; c.n_op = 0; c.blk = bi;
        d.chg[chg0 + i] = c;
        ctr += (uint32_t)l;
      }
      // dep_on_self BoolRle[N] and other-dep counts AnyRle<usize>[N] advance together: one pass over the changes
      const BoolCur bc0 = bool_make(h);
      // the dep-count column starts where the BoolRle ends: run the bool cursor to its end first (N values)
      Rd after_bool = h;
      if (par) after_bool = par_after_bool;
      else {
        BoolCur t = bc0;
        bool_skip(t, N);
        if (t.rem != 0) t.r.bad = true;
        after_bool = t.r;
      }
      const RleCur dc0 = rle_make(after_bool);
      // one change's (dep_on_self, other dependencies) and where its dependency rows begin — the same walk in every pass, the
      // clamp of a damaged count included
      const uint32_t dep_lim = dep0 + cnt[BC_DEP];
      auto next_deps = [&](BoolCur& bc, RleCur& dc, uint32_t& dcur, uint32_t& ds, uint32_t& others, bool& clamp) {
        ds = bool_next(bc) ? 1u : 0u;
        uint64_t o = rle_next_uvar(dc);
        clamp = (uint64_t)dcur + ds + o > dep_lim;
        if (clamp) { o = 0; ds = 0; }
        others = (uint32_t)o;
      };
      BoolCur bc = bc0;
      RleCur dc = dc0;
      uint32_t dcur = dep0;
      uint64_t others_total = 0;      // dependencies on other peers in the whole block (0: the two columns that describe them are empty — no pass over the changes)
      if (par) { others_total = par_others; dc.r = par_dc_end; dc.rem = 0; dcur = dep0 + cnt[BC_DEP]; }
      else {
        Rd hl = h_lens;               // the changes' counters again, from their lengths
        uint64_t kn = 0;
        uint32_t cctr = bd.counter_start;
        for (uint32_t i = 0; i < N; i++) {
          uint32_t ds, others;
          bool clamp;
          next_deps(bc, dc, dcur, ds, others, clamp);
          if (clamp) st = ST_DECODE_ERROR;
          d.chg[chg0 + i].dep0 = dcur;
          d.chg[chg0 + i].n_dep = ds + others;
          if (ds) {
            if (cctr == 0) st = ST_DECODE_ERROR;
            d.dep_peer[dcur] = 0;
            d.dep_ctr[dcur] = cctr ? cctr - 1 : 0;
          }
          dcur += ds + others;
          others_total += others;
          uint64_t l = 0;
          if (i + 1 < N) { l = rd_uleb(hl); kn += l; if (kn > bd.counter_len) l = 0; }
          cctr += (uint32_t)l;
        }
      }
      if (dc.rem != 0) dc.r.bad = true;
      if (dcur - dep0 != cnt[BC_DEP]) st = ST_DECODE_ERROR;
      // dep peer idx AnyRle<u32>[D]
      RleCur pc = rle_make(dc.r);
      uint64_t D = 0;
      if (others_total) {
        BoolCur b2 = bc0;
        RleCur d2 = dc0;
        uint32_t cur2 = dep0;
        for (uint32_t i = 0; i < N; i++) {
          uint32_t ds, others;
          bool clamp;
          next_deps(b2, d2, cur2, ds, others, clamp);
          for (uint32_t k = cur2 + ds; k < cur2 + ds + others; k++) {
            uint64_t pi = rle_next_uvar(pc);
            if (pi >= n_peers) { st = ST_DECODE_ERROR; pi = 0; }
            d.dep_peer[k] = (uint32_t)pi;
            D++;
          }
          cur2 += ds + others;
        }
      }
      if (pc.rem != 0) pc.r.bad = true;
      // dep counters DeltaOfDelta[D]
      Rd hr = pc.r;
      DodCur dd = dod_make(hr);
      if (D) {
        BoolCur b3 = bc0;
        RleCur d3 = dc0;
        uint32_t cur3 = dep0;
        for (uint32_t i = 0; i < N; i++) {
          uint32_t ds, others;
          bool clamp;
          next_deps(b3, d3, cur3, ds, others, clamp);
          for (uint32_t k = cur3 + ds; k < cur3 + ds + others; k++) {
            int64_t v = dod_next(dd);
            if (v < 0 || v >= (int64_t)MAX_COUNTER) { st = st ? st : ST_DECODE_ERROR; v = 0; }
            d.dep_ctr[k] = (uint32_t)v;
          }
          cur3 += ds + others;
        }
      }
      dod_finish(dd, hr, D);
      // wire lamports (DeltaOfDelta[N-1]): shape only — lamports are recomputed from deps on import
      // (outdated_encode_reordered.rs:61-62, loro_dag.rs:1179-1187)
      DodCur ld = dod_make(hr);
      dod_skip(ld, N ? N - 1 : 0);
      dod_finish(ld, hr, N - 1);
      {   // the last change's lamport = lamport_start + lamport_len - its length, in u32 with checked arithmetic (block_meta_encode.rs:215-221):
          // the wire lamports are not used (recomputed from the dependencies on import), this verdict is
        const uint64_t kn_ = known > bd.counter_len ? bd.counter_len : known;
        const uint64_t lend = (uint64_t)bdp->lamport_start + (uint64_t)bdp->lamport_len, last_len = (uint64_t)bd.counter_len - kn_;
        if (lend > 0xFFFFFFFFull || lend < last_len) st = st ? st : ST_DECODE_ERROR;
      }
      if (h.bad || bc.r.bad || after_bool.bad || dc.r.bad || pc.r.bad || hr.bad) st = st ? st : ST_DECODE_ERROR;
    }
    {  // change_meta: timestamps + message lengths, shape only (block_encode.rs:563-571)
      Rd m = sec(SEC_META);
      DodCur td = dod_make(m);
      dod_skip(td, N);
      dod_finish(td, m, N);
      RleCur mc = rle_make(m);
      const uint64_t tot = rle_sum_uvar(mc, N);
      if (mc.rem != 0) mc.r.bad = true;   // (a run that announces more than N values does not decode either)
      if (mc.r.bad || tot > rd_left(mc.r)) st = st ? st : ST_DATA_CORRUPTION;   // (both change_meta columns: DecodeDataCorruptionError, block_encode.rs:563-571 — lm_k_decode.h)
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
        // (a kind beyond Counter is ContainerType::Unknown(kind), loro-common/src/lib.rs:793-804 — try_from_u8 never fails: the container is outside the device scope like Tree / Counter)
        if (i < DEC_KINDS) s_kinds[b * DEC_KINDS + i] = (uint8_t)kind;
      }
      if (k.bad) st = st ? st : ST_DECODE_ERROR;
    }
  }
  lmw::block_sync();   // kinds (LDS) and the change rows (HBM, read back by role 7 of the same block) are in place
  DEC_PH(1);
  // ---- column cursors (roles 0-6) and the per-role parameters that let ONE code path decode all seven columns
  RleCur col = rle_make(rd_make(blk_p, 0));
  ColCur fcol = col_make((lm_lds_bytes)slot, 0, 0, false);   // the same cursor for a staged block (LDS offsets, lean varints)
  bool has_del = false;
  uint32_t errk = 0xffffffffu;   // earliest row error of this lane
  uint32_t kc_map = 0, kc_el = 0, kc_style = 0;   // this lane's rows (phase B: lane = (block, row of the chunk)), lm_k_decode.h kc_add
  bool colbad = false;           // this lane's column does not decode (lm_dev_util.h rle_drain)
  uint32_t n_del_read = 0;       // delete-start columns: values read for the block's DeleteSeq rows
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
    if (staged) fcol = col_make((lm_lds_bytes)slot, (uint32_t)(mine.p - (const uint8_t*)slot), (uint32_t)(mine.end - (const uint8_t*)slot), mine.bad);
  }
  // value domain of each column: [v_lo, v_lo + v_span) (v_nz: zero is invalid too); a value outside it records the
  // row error (v_prio orders the findings of one row like the sequential decoder) and is replaced by v_repl
  uint32_t mode = r == 2 ? 0u : (r == 3 ? 1u : 2u);        // Rle<u8> | Rle<u32> | DeltaRle
  int64_t v_lo = 0, v_repl = 0;
  uint64_t v_span = 256;
  uint32_t v_prio = 0, v_mask = 0xffffffffu;
  int32_t v_code = ST_DATA_CORRUPTION;
  bool v_nz = false;
  if (r == 0) { v_span = n_cids; v_prio = 0; }                                                       // container_index
  else if (r == 1) { v_lo = INT32_MIN; v_span = 1ull << 32; v_prio = 1; v_code = ST_DECODE_ERROR; }    // prop
  else if (r == 2) { v_mask = 0x7f; }                                                                 // value_type
  else if (r == 3) { v_span = (uint64_t)MAX_COUNTER + 1; v_repl = MAX_COUNTER; v_prio = 5; v_code = ST_UNSUPPORTED; }   // len
  else if (r == 4) { v_span = n_peers; v_prio = 6; }                                                  // delete start: peer_idx
  else if (r == 5) { v_span = MAX_COUNTER; v_prio = 8; }                                              //   counter
  else if (r == 6) { v_lo = -(int64_t)MAX_COUNTER; v_span = 2ull * MAX_COUNTER + 1; v_nz = true; v_repl = 1; v_prio = 7; }   //   signed len
  // ---- role 7: value walker + row→change bookkeeping
  Rd v = rd_make(blk_p, 0);
  const uint64_t v_end_abs = ok ? bd.base + bdp->sec_rel[SEC_VALUES] + bdp->sec_len[SEC_VALUES] : 0ull;   // end of the block's values section
  uint64_t counter = bd.counter_start;
  uint32_t change_index = 0, rows_in_change = 0;
  uint64_t next_boundary = 0;
  Rd hl7 = rd_make(d.data, 0);   // role 7: the header's change lengths (N - 1 of them)
  uint64_t kn7 = 0;
  auto next_len7 = [&]() -> uint64_t { uint64_t l = rd_uleb(hl7); kn7 += l; return kn7 > bd.counter_len ? 0ull : l; };
  bool unsupported = false;
  if (ok && r == 7) {
    v = rd_make(d.data + bd.base + bdp->sec_rel[SEC_VALUES], bdp->sec_len[SEC_VALUES]);
    // (the changes' first counters come from the header's length column — LDS for a staged head — not from the ChangeRows role 0
    // stored: one dependent global load per change boundary was a third of the walker's time on one-change-per-keystroke blocks)
    hl7 = sec(SEC_HEADER);
    (void)rd_uleb(hl7);
    rd_skip(hl7, (uint64_t)n_peers * 8);
    next_boundary = N > 1 ? (uint64_t)bd.counter_start + next_len7() : (uint64_t)bd.counter_start + bd.counter_len;
    d.chg[chg0].op0 = op0;
  }
  uint32_t max_rows = lmw::reduce_max(ok ? n_ops : 0u);
  // Rows are decoded DEC_R = 8 at a time in four phases that each keep the lanes on one code path:
  //   A  lane = (block, column): the four op columns advance 8 rows, then the three delete-start columns advance once
  //      per DeleteSeq row of the chunk; values land in s_x[block][row][column]
  //   W  lane = (block, role 7): the value walker passes over the 8 payloads (sequential by format) → s_w[block][row]
  //   B  lane = (block, row): decode_op mapping, the 32-byte OpRow and its side tables, 64 rows per store
  uint32_t* sx = s_x + (size_t)b * (DEC_R * 8);       // this block's rows: 8 words each (columns 0-6, word 7 = container kind)
  uint32_t* sw = s_w + (size_t)b * (DEC_R * DEC_WW);
  DEC_PH(2);
  // (the chunk's delete-start columns — one value per DeleteSeq row — advance in the SAME trips as the next chunk's op columns:
  // roles 0-3 and roles 4-6 run one cursor code path on different lanes.  The order inside a chunk is therefore T (which rows
  // are DeleteSeq rows; the op columns go to registers), W (reads value type / length / kind), then the column trips — the next
  // chunk's op columns overwrite words 0-3 of s_x, which nobody reads any more, while this chunk's delete-start values land in
  // words 4-6 — then B.  As a phase of their own the delete columns were 25 % of the decoder's time on text blocks.)
  for (uint32_t k = 0; k < DEC_R; k++) {   // op columns of the first chunk
    if (ok && k < n_ops && r < 4) {
      int64_t w = staged ? col_next_any(fcol, mode) : rle_next_any(col, mode);
      if ((uint64_t)(w - v_lo) >= v_span) { dec_err(errk, k, v_prio, v_code); w = v_repl; }
      sx[k * 8 + r] = (uint32_t)w & v_mask;
    }
  }
  lmw::wave_sync();
  for (uint32_t c0 = 0; c0 < max_rows; c0 += DEC_R) {
    DEC_PH(3);
    // T. lane = (block, row r): which rows are DeleteSeq ops of a sequence container
    uint32_t row = c0 + r;
    bool act = ok && row < n_ops;
    uint32_t ci = 0, prop = 0, vt = 0, len = 0;
    if (act) { ci = sx[r * 8]; prop = sx[r * 8 + 1]; vt = sx[r * 8 + 2]; len = sx[r * 8 + 3]; }
    uint32_t ckind = 0xff;
    if (act && n_cids) ckind = ci < DEC_KINDS ? s_kinds[b * DEC_KINDS + ci] : (d.cid_raw[(uint64_t)(cid0 + ci) * 4] & 0xff);
    bool take_del = act && vt == 9 && (ckind == CK_TEXT || ckind == CK_LIST || ckind == CK_MOVABLE);
    if (act) sx[r * 8 + 7] = ckind;
    uint64_t tdm = lmw::ballot(take_del && has_del);
    uint32_t todo = (r >= 4 && r < 7) ? (uint32_t)(tdm >> (b * 8)) & 0xffu : 0u;   // delete-start columns: one value per DeleteSeq row, in row order
    DEC_PH(4);
    // W. role 7: value payloads (docs/encoding.md §10) and the row → change bookkeeping
    {
      uint32_t wn = (ok && r == 7 && c0 < n_ops) ? (n_ops - c0 < DEC_R ? n_ops - c0 : DEC_R) : 0u;
      uint32_t* fs = s_fs + b * 16;
      // S. A chunk of plain typing — every row a string insert or a row without payload — needs no walk through memory: if the
      // strings are ASCII, string k begins where the lengths of the rows in front of it say (length prefix + as many bytes as
      // the row has elements).  The block's eight lanes (lane = row here, as in T) compute those positions with one scan and check
      // all eight prefixes with one load each; only if every prefix holds exactly its row's length does the walker advance by
      // arithmetic.  (The prefix of string k+1 sits behind string k: read one after the other they were a chain of dependent
      // HBM / L2 round trips of one lane — 40 % of the decoder's time on text blocks.)
      bool spec = false;
      {
        const bool str_r = act && vt == 5;
        const uint32_t hdr_r = len < 128u ? 1u : (len < 16384u ? 2u : (len < 2097152u ? 3u : 4u));
        const uint32_t pay_r = str_r ? hdr_r + len : 0u;
        bool ok_r = !act || str_r || vt <= 2 || vt == 8 || vt == 9;
        if (lmw::any(str_r)) {
          uint32_t inc = lmw::scan_incl_add(pay_r);
          uint32_t base = lmw::shfl(inc, (int)(b ? b * 8 - 1 : 0));   // (every lane takes part in a lane permute: a lane that sat it out would read as 0)
          if (!b) base = 0;
          int wl_ = (int)(b * 8 + 7);
          uint64_t vp = (uint64_t)(v.p - d.data);
          uint64_t wpos = ((uint64_t)lmw::shfl((uint32_t)(vp >> 32), wl_) << 32) | lmw::shfl((uint32_t)vp, wl_);
          if (str_r) {
            uint64_t pos = wpos + (inc - pay_r - base);
            ok_r = pos + pay_r <= v_end_abs;
            if (ok_r) {
              uint32_t w4 = ld32u(d.data + pos);   // (the prefix; `data` carries 64 bytes of slack behind the last blob)
              uint32_t b0 = w4 & 0xff, b1 = (w4 >> 8) & 0xff, b2 = (w4 >> 16) & 0xff, b3 = w4 >> 24;
              uint32_t val, used;
              if (!(b0 & 0x80)) { val = b0; used = 1; }
              else if (!(b1 & 0x80)) { val = (b0 & 0x7f) | (b1 << 7); used = 2; }
              else if (!(b2 & 0x80)) { val = (b0 & 0x7f) | ((b1 & 0x7f) << 7) | (b2 << 14); used = 3; }
              else { val = (b0 & 0x7f) | ((b1 & 0x7f) << 7) | ((b2 & 0x7f) << 14) | ((b3 & 0x7f) << 21); used = (b3 & 0x80) ? 0u : 4u; }
              ok_r = val == len && used == hdr_r;
            }
          }
          uint64_t okm = lmw::ballot(ok_r);
          spec = ((okm >> (b * 8)) & 0xffull) == 0xffull;
          // … and when no change of the block ends inside the chunk, the rows' bookkeeping words (value offset, counter, change) are
          // a scan away too: every row lane writes its own s_w entry and the walker only moves its state on — eight rows one after
          // the other on ONE lane were what was left of this phase
          uint32_t cinc = lmw::scan_incl_add(act ? len : 0u);
          uint32_t cbase = lmw::shfl(cinc, (int)(b ? b * 8 - 1 : 0));
          if (!b) cbase = 0;
          uint32_t tot_len = lmw::shfl(cinc, wl_) - cbase, tot_pay = lmw::shfl(inc, wl_) - base;
          uint32_t ctr7 = lmw::shfl((uint32_t)counter, wl_), ci7 = lmw::shfl(change_index, wl_);
          uint32_t nb7 = lmw::shfl((uint32_t)(next_boundary > 0xffffffffull ? 0xffffffffull : next_boundary), wl_);
          const bool whole = spec && ci7 < N && (uint64_t)ctr7 + tot_len <= MAX_COUNTER && ((uint64_t)ctr7 + tot_len < nb7 || ci7 + 1 >= N);
          if (whole) {
            if (act) {
              uint64_t va = wpos + (inc - pay_r - base);
              uint32_t* o = sw + r * DEC_WW;
              o[0] = (uint32_t)va; o[1] = (uint32_t)(va >> 32); o[2] = 0; o[3] = 0; o[4] = ctr7 + (cinc - len - cbase); o[5] = chg0 + ci7;
            }
            if (wn) { v.p += tot_pay; counter += tot_len; rows_in_change += wn; wn = 0; }
          }
        }
      }
#ifndef LM_NO_I64_FAST
      if (I64)
      // I. A chunk of Map writes whose values are integers — `set(key, i64)`: counters, timestamps, ids — needs no walk either.  Each
      // value is a tag byte (3) and a signed LEB128: bytes with a clear top bit ALTERNATE tag, last LEB byte, tag, last LEB byte …
      // The block's eight lanes fetch the 128 bytes behind the walker's cursor with one 16-byte load each; every row lane then reads
      // the window, finds its value between the (2k)-th and the (2k+1)-th such byte, and checks its tag; when all eight hold and no
      // change ends inside the chunk the rows' bookkeeping words come from a scan, as for plain typing above.  (One lane stepped
      // through the eight values byte by byte before: 55 % of the decoder's time on configs[2], profiles/r04_decoder_phases.log.)
      {
        const bool int_r = act && vt == 11 && ckind == CK_MAP;
        const uint64_t cand_m = lmw::ballot(int_r || !act);
        const bool blk_cand = ((cand_m >> (b * 8)) & 0xffull) == 0xffull;
        const uint32_t wn_blk = lmw::shfl(wn, (int)(b * 8 + 7));
        if (lmw::any(blk_cand && wn_blk != 0)) {
          int wl_ = (int)(b * 8 + 7);
          uint64_t vp = (uint64_t)(v.p - d.data);
          uint64_t wpos = ((uint64_t)lmw::shfl((uint32_t)(vp >> 32), wl_) << 32) | lmw::shfl((uint32_t)vp, wl_);
          const bool go = blk_cand && wn_blk != 0;
          uint8_t* win = s_vw2 + (size_t)b * DEC_VW2;
          if (go && wpos + 16ull * r < v_end_abs) {   // (a 16-byte piece that starts inside the section: `data` carries 64 bytes of slack)
            struct V16b { uint32_t x, y, z, w; } pc;
            __builtin_memcpy(&pc, d.data + wpos + 16ull * r, 16);
            *(V16b*)(win + 16u * r) = pc;
          }
          lmw::wave_sync();
          bool ok_r = true;
          uint32_t st_k = 0, en_k = 0;
          if (go && act) {
            // bytes of the window with a clear top bit, as two 64-bit masks (only bytes inside the values section count)
            uint64_t left = v_end_abs - wpos;
            uint64_t t0 = 0, t1 = 0;
            const uint32_t* w32 = (const uint32_t*)win;
            for (uint32_t q = 0; q < 16; q++) {
              uint32_t x = ~w32[q] & 0x80808080u;                                       // bit 7 of each byte: set where the byte terminates
              uint64_t nib = ((x >> 7) & 1u) | ((x >> 14) & 2u) | ((x >> 21) & 4u) | ((x >> 28) & 8u);
              t0 |= nib << (4 * q);
            }
            for (uint32_t q = 16; q < 32; q++) {
              uint32_t x = ~w32[q] & 0x80808080u;
              uint64_t nib = ((x >> 7) & 1u) | ((x >> 14) & 2u) | ((x >> 21) & 4u) | ((x >> 28) & 8u);
              t1 |= nib << (4 * (q - 16));
            }
            if (left < 64) { t0 &= (1ull << left) - 1; t1 = 0; }
            else if (left < 128) t1 &= (1ull << (left - 64)) - 1;
            // the (2r)-th and (2r+1)-th set bits (r = this lane's row of the chunk)
            auto nth = [&](uint32_t n, uint32_t& pos) -> bool {
              uint64_t a = t0, bq = t1;
              uint32_t c0 = (uint32_t)lmw::popc64(a);
              if (n < c0) { for (uint32_t i = 0; i < n; i++) a &= a - 1; pos = (uint32_t)lmw::ffs64(a); return true; }
              n -= c0;
              if (n >= (uint32_t)lmw::popc64(bq)) return false;
              for (uint32_t i = 0; i < n; i++) bq &= bq - 1;
              pos = 64u + (uint32_t)lmw::ffs64(bq);
              return true;
            };
            ok_r = nth(2 * r, st_k) && nth(2 * r + 1, en_k);
            // the tag is an I64's, the LEB128 is at most ten bytes (the bytes between two set bits all carry the continuation bit by
            // construction), and the value begins right behind the one in front of it
            if (ok_r) ok_r = win[st_k] == 3 && en_k - st_k <= 10;
            if (ok_r) {
              uint32_t pen = 0;
              if (r == 0) ok_r = st_k == 0;
              else ok_r = nth(2 * r - 1, pen) && st_k == pen + 1;
            }
          }
          uint64_t okm = lmw::ballot(!go || !act || ok_r);
          const bool all_ok = go && ((okm >> (b * 8)) & 0xffull) == 0xffull;
          // end of the chunk's last value = where the walker continues; the rows' counters / change as in S
          uint32_t n_act = (uint32_t)lmw::popc64((lmw::ballot(act) >> (b * 8)) & 0xffull);
          uint32_t last_en = lmw::shfl(en_k, (int)(b * 8 + (n_act ? n_act - 1 : 0)));
          uint32_t cinc = lmw::scan_incl_add(act ? len : 0u);
          uint32_t cbase = lmw::shfl(cinc, (int)(b ? b * 8 - 1 : 0));
          if (!b) cbase = 0;
          uint32_t tot_len = lmw::shfl(cinc, wl_) - cbase;
          uint32_t ctr7 = lmw::shfl((uint32_t)counter, wl_), ci7 = lmw::shfl(change_index, wl_);
          uint32_t nb7 = lmw::shfl((uint32_t)(next_boundary > 0xffffffffull ? 0xffffffffull : next_boundary), wl_);
          const bool whole = all_ok && n_act == wn_blk && ci7 < N && (uint64_t)ctr7 + tot_len <= MAX_COUNTER && ((uint64_t)ctr7 + tot_len < nb7 || ci7 + 1 >= N);
#ifdef LM_EMU_TRACE
          if (getenv("LM_EMU_I64") && go && r == 7) fprintf(stderr, "I64 %s\n", whole ? "fast" : (all_ok ? "boundary" : "declined"));
#endif
          if (whole) {
            if (act) {
              uint64_t va = wpos + st_k;
              uint32_t* o = sw + r * DEC_WW;
              o[0] = (uint32_t)va; o[1] = (uint32_t)(va >> 32); o[2] = 0; o[3] = 0; o[4] = ctr7 + (cinc - len - cbase); o[5] = chg0 + ci7;
            }
            if (wn) { v.p += last_en + 1; counter += tot_len; rows_in_change += wn; wn = 0; }
          }
        }
      }
#endif
      // one value (row k of the chunk), through either reader
      auto walk = [&](auto& v, const uint32_t k) {
        uint32_t wvt = sx[k * 8 + 2], wlen = sx[k * 8 + 3], wkind = sx[k * 8 + 7];
        uint64_t val_at = (uint64_t)(v.p - d.data);
        uint32_t aux = 0, flags = 0;   // aux: element count of a list value | mark length; flags bit 0: the value is a list
        uint32_t vfl = 0;              // (skip_loro_value: VF_UNSUPPORTED | VF_CORRUPT — a nested key index beyond the key table, an undefined value tag)
        switch (wvt) {
          case 0: case 1: case 2: case 8: case 9: break;
          case 3: (void)rd_sleb(v); break;
          case 4: rd_skip(v, 8); break;
          case 5: if (spec) { v.p += (wlen < 128u ? 1u : (wlen < 16384u ? 2u : (wlen < 2097152u ? 3u : 4u))) + wlen; break; }   // (checked above: S)
            [[fallthrough]];
          case 6: { uint64_t l = rd_uleb(v); rd_skip(v, l); break; }
          case 7: (void)rd_uleb(v); break;
          case 10: (void)rd_sleb(v); break;
          case 11: {
            uint32_t tag0 = v.p < v.end ? rd_peek(v) : 0xffu;   // (an empty reader goes through the general routine, which latches `bad`)
            bool is_list_value = tag0 == 7;
            if (is_list_value) { auto t = v; (void)rd_u8(t); aux = (uint32_t)rd_uleb(t); flags = 1; }
            if (tag0 == 7 || tag0 == 8) flags |= 4;   // a list / map value: its nested key indices are checked by k_remap (OPF_NESTED)
            // (values of containers outside the device scope are never rendered: any shape is accepted)
            skip_loro_value_top(v, vfl, wkind == CK_MAP ? 0 : (is_list_value && (wkind == CK_LIST || wkind == CK_MOVABLE) ? 1 : (wkind > CK_TEXT && wkind != CK_MOVABLE ? 16 : -1)), fs, tag0);
            break;
          }
          case 12: {
            (void)rd_u8(v);
            aux = (uint32_t)rd_uleb(v);
            uint64_t key_idx = rd_uleb(v);
            if (key_idx >= n_keys) dec_err(errk, c0 + k, 2, ST_DATA_CORRUPTION);
            uint32_t u = 0;
            skip_loro_value_fs(v, u, -1, fs);
            vfl |= u & VF_CORRUPT;
            flags |= 4;
            break;
          }
          case 13: { (void)rd_uleb(v); uint32_t isn = rd_u8(v); (void)rd_uleb(v); if (!isn) (void)rd_uleb(v); break; }
          // ListMove / ListSet (MovableList): the element id travels in the row's delete-start words, which such a row never uses
          case 14: {
            uint64_t f = rd_uleb(v), pi = rd_uleb(v), lm_ = rd_uleb(v);
            if (pi >= n_peers || lm_ > 0xFFFFFFFFull || f > 0x7FFFFFFFull) flags |= 2;
            sx[k * 8 + 4] = (uint32_t)pi; sx[k * 8 + 5] = (uint32_t)lm_; sx[k * 8 + 6] = (uint32_t)f;
            break;
          }
          case 15: {
            uint64_t pi = rd_uleb(v), lm_ = rd_uleb(v);
            if (pi >= n_peers || lm_ > 0xFFFFFFFFull) flags |= 2;
            sx[k * 8 + 4] = (uint32_t)pi; sx[k * 8 + 5] = (uint32_t)lm_; sx[k * 8 + 6] = 0;
            val_at = (uint64_t)(v.p - d.data);   // op_val of a set row points at the nested value
            if (wkind == CK_MOVABLE) skip_loro_value_fs(v, vfl, 0, fs);
            else { uint32_t u = 0; skip_loro_value_fs(v, u, -1, fs); vfl |= u & VF_CORRUPT; }
            flags |= 4;
            break;
          }
          case 16: {
            (void)rd_uleb(v); (void)rd_uleb(v); (void)rd_uleb(v);
            uint32_t isn = rd_u8(v);
            if (!isn) { (void)rd_uleb(v); (void)rd_uleb(v); }
            break;
          }
          default: { uint64_t l = rd_uleb(v); rd_skip(v, l); break; }
        }
        if (vfl & VF_UNSUPPORTED) unsupported = true;
        if (vfl & VF_CORRUPT) dec_err(errk, c0 + k, 2, ST_DATA_CORRUPTION);
        uint32_t* o = sw + k * DEC_WW;
        o[0] = (uint32_t)val_at; o[1] = (uint32_t)(val_at >> 32); o[2] = aux; o[3] = flags;
        o[4] = (uint32_t)counter; o[5] = chg0 + change_index;
        counter += wlen;
        if (counter > MAX_COUNTER) { dec_err(errk, c0 + k, 10, ST_UNSUPPORTED); counter = MAX_COUNTER; }
        if (change_index >= N) { dec_err(errk, c0 + k, 11, ST_DATA_CORRUPTION); change_index = N - 1; }
        rows_in_change++;
        // (an op that crosses a change boundary — or any op of a change whose header length is zero: docs/encoding.md §10.6, "independent
        // validators should reject these inputs"; the oracle does the same)
        if (counter > next_boundary && change_index + 1 < N) dec_err(errk, c0 + k, 12, ST_DATA_CORRUPTION);
        if (counter >= next_boundary && change_index + 1 < N) {
          d.chg[chg0 + change_index].n_op = rows_in_change;
          rows_in_change = 0;
          change_index++;
          d.chg[chg0 + change_index].op0 = op0 + c0 + k + 1;
          next_boundary = change_index + 1 < N ? next_boundary + next_len7() : (uint64_t)bd.counter_start + bd.counter_len;
        }
      };
      // A chunk whose payloads are scalars or nested values — blocks of Map sets, list items — is walked through a WINDOW: the
      // block's eight lanes fetch the DEC_VW bytes behind the walker's cursor into LDS (one coalesced load for all such blocks of
      // the wave) and the walker passes over the values that begin inside it, trip by trip; every byte was a dependent HBM / L2
      // round trip of one lane before (62 % of the decoder's time on configs[2], profiles/r03_decoder_phases.log).  A chunk with
      // a string or binary payload — a length prefix, then a jump — reads straight from HBM: with windows a text block's walk took
      // 17 % longer.  A wave without a dense chunk runs the plain loop.
      bool dense = false, jumps = false;   // (a chunk of delete rows has no payload at all: nothing to fetch)
      for (uint32_t k = 0; k < wn; k++) { uint32_t t_ = sx[k * 8 + 2]; jumps |= t_ == 5 || t_ == 6 || t_ > 16; dense |= t_ == 3 || t_ == 4 || (t_ >= 10 && t_ <= 16); }
      dense &= !jumps;
      if (!lmw::any(dense)) {
        for (uint32_t k = 0; k < wn; k++) walk(v, k);
      } else {
        RdW w;
        w.p = v.p; w.end = v.end; w.bad = v.bad; w.wl = (lm_lds_bytes)(s_vw + (size_t)b * DEC_VW);
        uint32_t wk = 0;
        while (lmw::any(wk < wn)) {
          uint64_t vp = (uint64_t)(w.p - d.data);
          int wl_ = (int)(b * 8 + 7);
          uint64_t wpos = ((uint64_t)lmw::shfl((uint32_t)(vp >> 32), wl_) << 32) | lmw::shfl((uint32_t)vp, wl_);
          bool more = lmw::shfl((dense && wk < wn) ? 1u : 0u, wl_) != 0;
          if (more && wpos + 8ull * r < v_end_abs) {   // (an 8-byte piece that starts inside the section: `data` carries 64 bytes of slack)
            struct V8 { uint32_t x, y; } pc;
            __builtin_memcpy(&pc, d.data + wpos + 8ull * r, 8);
            *(V8*)(s_vw + (size_t)b * DEC_VW + 8u * r) = pc;
          }
          lmw::wave_sync();
          if (wk < wn) {
            w.wp = dense ? w.p : w.p - DEC_VW;   // (not dense: nothing lies inside the window)
            for (bool first = true; wk < wn && (first || !dense || (uint64_t)(w.p - w.wp) + 12 <= DEC_VW); wk++, first = false) walk(w, wk);
          }
          lmw::wave_sync();
        }
        v.p = w.p; v.bad = w.bad;
      }
    }
    lmw::wave_sync();
    DEC_PH(5);
    // A. the column trips: this chunk's delete-start values (roles 4-6 → words 4-6) beside the next chunk's op columns (roles 0-3 → words 0-3)
    for (uint32_t k = 0; k < DEC_R; k++) {
      const bool a1 = ok && r < 4 && c0 + DEC_R + k < n_ops, a2 = todo != 0;
      if (!lmw::any(a1 | a2)) break;
      if (a1 | a2) {
        uint32_t kk = a2 ? (uint32_t)__builtin_ctz(todo) : k;
        uint32_t rowx = a2 ? c0 + kk : c0 + DEC_R + k;
        todo &= todo - 1;   // (0 stays 0)
        // (a delete-start column that is used up has no id for this row: the row's finding; one that does not decode is the column's)
        const bool used_up = a2 && (staged ? col_exhausted(fcol) : rle_exhausted(col));
        int64_t w = used_up ? v_repl : (staged ? col_next_any(fcol, mode) : rle_next_any(col, mode));
        if (a2 && !used_up) { if (staged ? fcol.bad : col.r.bad) colbad = true; else n_del_read++; }
        if (!used_up && ((uint64_t)(w - v_lo) >= v_span || (v_nz && w == 0))) { dec_err(errk, rowx, v_prio, v_code); w = v_repl; }
        if (a2 && (used_up || (staged ? fcol.bad : col.r.bad))) dec_err(errk, rowx, 9, ST_DATA_CORRUPTION);
        sx[kk * 8 + r] = (uint32_t)w & v_mask;
      }
    }
    lmw::wave_sync();
    // B. lane = (block, row): decode_op mapping (outdated_encode_reordered.rs:215-476) and the row itself
    if (act) {
      const uint32_t* o = sw + r * DEC_WW;
      uint32_t aux = o[2];
      bool is_list_value = (o[3] & 1) != 0;
      OpRow orow;
      orow.cidx_kind = ci;  // block-local until k_remap
      orow.prop = (int32_t)prop;
      orow.len = len;
      orow.ctr = o[4];
      orow.a0 = (vt == 11 && is_list_value) ? aux : 0u; orow.a1 = 0; orow.a2 = 0;
      orow.chg = o[5];
      uint32_t kind = OK_OTHER;
      if (ckind == CK_TEXT) {
        if (vt == 5) kind = OK_TEXT_INS;
        else if (vt == 9) kind = OK_DEL;
        else if (vt == 12) { kind = OK_STYLE_START; orow.a0 = aux; }
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
      } else if (ckind == CK_MOVABLE) {   // outdated_encode_reordered.rs:388-459
        if (vt == 11) { if (is_list_value) kind = OK_LIST_INS; else dec_err(errk, row, 3, ST_DATA_CORRUPTION); }
        else if (vt == 9) kind = OK_DEL;
        else if (vt == 14 || vt == 15) {
          kind = vt == 14 ? OK_LIST_MOVE : OK_LIST_SET;
          orow.a0 = sx[r * 8 + 4]; orow.a1 = sx[r * 8 + 5]; orow.a2 = (int32_t)sx[r * 8 + 6];
          if ((o[3] & 2) || (int32_t)prop < 0 || len != 1) { dec_err(errk, row, 3, ST_DATA_CORRUPTION); orow.a0 = 0; }
        } else dec_err(errk, row, 3, ST_DATA_CORRUPTION);
      }
      if (take_del) {
        if (!has_del) dec_err(errk, row, 4, ST_DATA_CORRUPTION);
        else { orow.a0 = sx[r * 8 + 4]; orow.a1 = sx[r * 8 + 5]; orow.a2 = (int32_t)sx[r * 8 + 6]; }
      }
      orow.cidx_kind |= (kind << 16) | (((o[3] & 4) && (vt != 12 || kind == OK_STYLE_START)) ? OPF_NESTED : 0u);
      kc_add(kc_map, kc_el, kc_style, kind, len);
      d.op[op0 + row] = orow;
      d.op_val[op0 + row] = (uint64_t)o[0] | ((uint64_t)o[1] << 32);
      d.op_blk[op0 + row] = bi;
    }
    lmw::wave_sync();   // the next chunk overwrites s_x / s_w
    DEC_PH(6);
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
  // every column decodes in full: the op columns to exactly one value per row, the delete-start columns to equally many values each
  // (what is left of them is counted here) — the reference decodes the columns before it looks at a row, so this finding precedes
  // every row's (bit 4)
  if (ok && r < 4 && ((staged ? fcol.bad : col.r.bad) || !(staged ? col_exhausted(fcol) : rle_exhausted(col)))) tail |= 16;
  {
    uint32_t tot = n_del_read;
    if (ok && has_del && r >= 4 && r < 7) {
      tot += staged ? col_drain(fcol, mode) : rle_drain(col, mode);
      if (colbad || (staged ? fcol.bad : col.r.bad)) tail |= 16;
    }
    uint32_t t4 = lmw::shfl(tot, (int)(b * 8 + 4)), t5 = lmw::shfl(tot, (int)(b * 8 + 5)), t6 = lmw::shfl(tot, (int)(b * 8 + 6));
    if (ok && has_del && (t4 != t5 || t4 != t6)) tail |= 16;
  }
  if (ok && shape_bad) tail |= 8;
  // combine over the 8 lanes of the block
  for (int m = 1; m < 8; m <<= 1) {
    uint32_t oe = lmw::shfl_xor(errk, m), ot = lmw::shfl_xor(tail, m);
    errk = oe < errk ? oe : errk;
    tail |= ot;
    kc_map += lmw::shfl_xor(kc_map, m); kc_el += lmw::shfl_xor(kc_el, m); kc_style += lmw::shfl_xor(kc_style, m);
  }
  if (ok && r == 0) {
#ifdef LM_EMU_TRACE
    if (getenv("LM_EMU_DECERR") && (st != ST_OK || errk != 0xffffffffu || tail)) fprintf(stderr, "DECERR blk %u st %d errk %x tail %x\n", bi, st, errk, tail);
#endif
    if (st == ST_OK && (tail & 8)) st = ST_DECODE_ERROR;
    if (st == ST_OK && (tail & 16)) st = ST_DECODE_ERROR;
    if (st == ST_OK && errk != 0xffffffffu) st = (int32_t)(errk & 0xf);
    if (st == ST_OK && (tail & 1)) st = ST_DECODE_ERROR;
    if (st == ST_OK && (tail & 2)) st = ST_DATA_CORRUPTION;
    if (st == ST_OK && (tail & 4)) st = ST_UNSUPPORTED;
    d.blk[bi].status = st;
    d.blk[bi].flags = kc_pack(kc_map, kc_style); d.blk[bi].pad = (st == ST_DECODE_ERROR || st == ST_DATA_CORRUPTION) ? DEC_RECLASS : kc_el;   // (k_block_reclassify looks at such a block once more)
  }
#ifdef LM_PROF_DEC
  DEC_PH(7);
  if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&d.prof[i], (unsigned long long)pacc[i]);
#endif
}

LM_KERNEL LM_WAVES_PER_SIMD(3) LM_ONE_WAVE_GROUPS void k_block_decode_wave(Dev d, uint32_t slot_cap, uint32_t head_lo, uint32_t head_hi) {
  block_decode_wave_body<false>(d, slot_cap, head_lo, head_hi);
}
LM_KERNEL LM_WAVES_PER_SIMD(3) LM_ONE_WAVE_GROUPS void k_block_decode_wave_map(Dev d, uint32_t slot_cap, uint32_t head_lo, uint32_t head_hi) {
  block_decode_wave_body<true>(d, slot_cap, head_lo, head_hi);
}

}  // namespace lm
