// K4m + K10m (round 6): LWW Map documents WITHOUT op rows — decode → LWW fused, one workgroup per document.
//
// configs[2] (10,000 documents x 16 peers x 10,000 writes on 1,024 keys) spent 217 of its 236 ms materialising 160,000 op rows per
// document (44 bytes each: k_block_decode 137 ms), rewriting them (k_remap 24 ms), walking every block's key table twice with one
// lane (k_block_count 23 ms) and reading the rows back (k_map_lww_doc 33 ms): 5.4 x the algorithmic bytes crossed HBM.  An LWW
// history needs none of that: the winner per key is max (lamport, peer) — diff_calc.rs:515-538, delta/map_delta.rs:20-46 — and a
// row's lamport is its change's lamport + its offset (block_encode.rs:654-704: a Map op has one id), so the op columns of a block can
// be folded straight into the document's table.
//
// A workgroup of MF_WG lanes owns one document whose blocks hold Map ops with scalar values only (k_block_kind / k_doc_kind); its
// (container, key) -> best-write table lives in LDS exactly as k_map_lww_doc's (lm_k_lww_doc.h); every wave takes a block at a time,
// last block first (the latest writes enter the table first, so most earlier writes lose on a plain LDS read and are dropped):
//   * key table: `uleb len, bytes` per key.  The starts are found WITHOUT walking the chain: a key byte is >= 0x20 and a length
//     below 0x20 is not, so the candidates (bytes < 0x20) are found 256 bytes per step by all lanes, and accepted only when every
//     candidate's successor is candidate + 1 + length — which makes the candidates the chain (induction from offset 0).  Keys of 32
//     bytes or more, or with control characters, fail that test: the document leaves the kernel (below);
//   * op columns (block_encode.rs:417-428: container_index DeltaRle, prop DeltaRle, value_type Rle<u8>, len Rle<u32>): 64 rows per
//     step — a run fills its lanes at once, a literal segment is cut at its varint terminators (one ballot per 64 bytes), the delta
//     columns finish with one wave scan;
//   * values: `tag, payload`.  A step of 64 integer values (tag 3 + sleb128) is cut by the parity of the bytes without a
//     continuation bit — tags and terminators alternate — and verified (every tag is 3 and follows a terminator); any other mix of
//     scalars is walked by one lane;
//   * every row: container / key index range checks, applied-change and version filters as in k_map_lww_doc, hash of the key,
//     LDS probe, atomic maximum.  Only a row that RAISES its key's maximum leaves a record — an OpRow in the document's candidate
//     table (dense: the record number is the row word of the LWW value, as the op row index is for k_map_lww_doc) — so the emit
//     stage finds the winner's row, value offset and block exactly where it finds them for every other document.
// Nothing else is written: no op rows, no key rows (a claimed slot's key row is slot-numbered and written at the end), no remap.
//
// Anything this kernel is not built for — a nested list / map value, a key table that fails the candidate test, a block beyond the
// LDS tables, a table that fills up, any column / value that does not decode — does not get a verdict here: the document is flagged
// DF_REDO and replayed through the row tables by the side engine (lm_capi_impl.h redo), whose decoders own every error code.
#pragma once

namespace lm {

static constexpr uint32_t MF_WG = 1024;          // lanes per document (16 waves)
static constexpr uint32_t MF_WAVES = MF_WG / 64;
static constexpr uint32_t MF_KMAX = 1020;        // keys of one block (kpos[] in LDS)
static constexpr uint32_t MF_WAVE_LDS = (MF_KMAX + 4) * 2 + 5 * 64 * 4;   // per wave: kpos u16[], five 64-word exchange rows
static constexpr uint32_t MF_LDS = LWW_LDS_CAP * 24 + (MAX_PEERS + MAX_CONTAINERS / 32 + 8) * 4 + MF_WAVES * MF_WAVE_LDS;

struct DevMf {
  const uint32_t* docs;      // the fused documents of the batch (workgroup -> document)
  const uint8_t* doc_fused;  // per document: 1 = its blocks are decoded by this kernel
  const uint32_t* key0;      // per document: first of its slot-numbered key rows (d.key_off / d.key_len, behind the decoders' rows)
};

// one column of the block's EncodedOp table, read 64 values at a time by the whole wave; every field is wave-uniform
struct MfCol { uint64_t p, end; int64_t rem; int64_t runv; int32_t acc; uint32_t mode; bool run; };

LM_DEV void mf_wave_lds_sync() {
#ifndef LM_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  lmw::wave_sync();
#endif
}

// the next `want` (<= 64) values of the column into lane order; false: the column does not decode the way this kernel reads it
LM_DEV bool mf_fetch(const uint8_t* data, MfCol& c, uint32_t want, uint32_t* tmp, int lane, int32_t& out) {
  uint32_t filled = 0;
  bool ok = true;
  while (filled < want && ok) {
    if (c.rem == 0) {
      Rd r = rd_make(data + c.p, c.end - c.p);
      if (r.p >= r.end) { ok = false; break; }
      int64_t k = rd_zigzag(r);
      if (k == 0 || r.bad || k > (1 << 28) || k < -(1 << 28)) { ok = false; break; }
      c.run = k > 0;
      c.rem = k > 0 ? k : -k;
      if (c.run) { c.runv = rd_any(r, c.mode); if (r.bad || c.runv > (1 << 24) || c.runv < -(1 << 24)) { ok = false; break; } }
      c.p = (uint64_t)(r.p - data);
    }
    uint32_t take = c.rem < (int64_t)(want - filled) ? (uint32_t)c.rem : want - filled;
    if (c.run) {
      if ((uint32_t)lane >= filled && (uint32_t)lane < filled + take) tmp[lane] = (uint32_t)(int32_t)c.runv;
    } else {
      uint32_t got = 0;
      while (got < take && ok) {
        const uint64_t pos = c.p + (uint32_t)lane;
        const bool inb = pos < c.end;
        const uint32_t b = inb ? data[pos] : 0x80u;
        if (c.mode == 0) {   // Rle<u8>: every byte is a value
          uint64_t left = c.end - c.p;
          uint32_t n = take - got;
          if (n > 64) n = 64;
          if (left < n) { ok = false; break; }
          if ((uint32_t)lane < n) tmp[filled + got + lane] = b;
          c.p += n; got += n;
          continue;
        }
        const bool term = inb && !(b & 0x80u);
        const uint64_t mask = lmw::ballot(term);
        if (!mask) { ok = false; break; }
        const uint64_t below = mask & ((1ull << lane) - 1ull);
        const uint32_t rank = (uint32_t)lmw::popc64(below);
        const uint32_t navail = (uint32_t)lmw::popc64(mask);
        const uint32_t n = take - got < navail ? take - got : navail;
        const uint32_t start = below ? 64u - (uint32_t)__builtin_clzll(below) : 0u;
        const uint32_t len = (uint32_t)lane - start + 1;
        const bool mine = term && rank < n;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          uint32_t src = start + j;
          uint32_t bj = lmw::shfl(b, (int)(src & 63));
          if (src <= (uint32_t)lane) v |= (bj & 0x7fu) << (7 * j);
        }
        if (lmw::any(mine && len > 4)) { ok = false; break; }   // (a value beyond 2^28: not a key index / container index / length of a block this kernel takes)
        if (mine) tmp[filled + got + rank] = c.mode == 2 ? (uint32_t)((int32_t)(v >> 1) ^ -(int32_t)(v & 1)) : v;
        // bytes consumed: up to and including the n-th terminator
        const uint64_t last = lmw::ballot(term && rank == n - 1);
        c.p += (uint32_t)lmw::ffs64(last) + 1;
        got += n;
      }
    }
    c.rem -= take;
    filled += take;
  }
  if (!ok) return false;
  mf_wave_lds_sync();
  int32_t v = (uint32_t)lane < want ? (int32_t)tmp[lane] : 0;
  mf_wave_lds_sync();
  if (c.mode == 2) {
    if (lmw::any(v > (1 << 24) || v < -(1 << 24))) return false;
    int32_t s = (int32_t)lmw::scan_incl_add((uint32_t)v);
    v = c.acc + s;
    c.acc = (int32_t)lmw::bcast((uint32_t)v, (int)want - 1);
  }
  out = v;
  return true;
}
LM_DEV bool mf_col_done(const MfCol& c) { return c.rem == 0 && c.p == c.end; }

// `want` integer values (tag 3 + sleb128) from vp: value offsets (absolute in `data`) into tmpv[0..want); false: the bytes are not that
LM_DEV bool mf_values_i64(const uint8_t* data, uint64_t& vp, uint64_t vend, uint32_t want, uint32_t* tmpv, uint64_t base, int lane) {
  const uint32_t need = 2 * want;
  uint32_t cnt = 0;
  uint32_t carry_clear = 1;          // the byte in front of the window has no continuation bit (a terminator, or the start of the values)
  uint32_t carry_run = 0;            // continuation bytes at the end of the previous window (length check of a sleb that straddles it)
  bool bad = false;
  uint64_t p = vp;
  while (cnt < need) {
    const uint64_t pos = p + (uint32_t)lane;
    const bool inb = pos < vend;
    const uint32_t b = inb ? data[pos] : 0x80u;
    const bool clear = inb && !(b & 0x80u);
    const uint64_t mask = lmw::ballot(clear);
    if (!mask) { bad = true; break; }
    const uint64_t below = mask & ((1ull << lane) - 1ull);
    const uint32_t rank = cnt + (uint32_t)lmw::popc64(below);
    const bool valid = clear && rank < need;
    const bool is_tag = valid && !(rank & 1u);
    const bool prev_clear = lane == 0 ? carry_clear != 0 : ((mask >> (lane - 1)) & 1ull) != 0;
    // a terminator: the sleb's bytes = this one + the continuation bytes behind the tag
    const uint32_t run_before = below ? (uint32_t)lane - (64u - (uint32_t)__builtin_clzll(below)) : (uint32_t)lane + carry_run;
    bool lane_bad = (is_tag && (b != 3u || !prev_clear)) || (valid && (rank & 1u) && run_before + 1 > 10);
    if (is_tag) tmpv[(rank >> 1)] = (uint32_t)(pos - base);
    if (lmw::any(lane_bad)) { bad = true; break; }
    const uint32_t navail = (uint32_t)lmw::popc64(mask);
    if (need - cnt >= navail) {
      // the whole window (its tail of continuation bytes belongs to a value that ends in the next window)
      uint64_t left = vend - p;
      uint32_t adv = left < 64 ? (uint32_t)left : 64u;
      const uint32_t hi = 64u - (uint32_t)__builtin_clzll(mask);   // index of the last clear byte + 1
      carry_clear = hi == adv ? 1u : 0u;
      carry_run = adv - hi;
      p += adv; cnt += navail;
      if (adv == 0) { bad = true; break; }
    } else {
      const uint64_t last = lmw::ballot(clear && rank == need - 1);
      p += (uint32_t)lmw::ffs64(last) + 1;
      cnt = need;
    }
  }
  if (bad) return false;
  vp = p;
  return true;
}

// any mix of scalar values (and map deletes, which carry none), one lane: offsets into tmpv; false: a nested value / an undefined tag / overrun
LM_DEV bool mf_values_slow(const uint8_t* data, uint64_t& vp, uint64_t vend, uint32_t want, const uint32_t* vts, uint32_t* tmpv, uint64_t base, int lane) {
  uint32_t okw = 1;
  uint64_t np = vp;
  if (lane == 0) {
    Rd r = rd_make(data + vp, vend - vp);
    for (uint32_t i = 0; i < want; i++) {
      tmpv[i] = (uint32_t)((uint64_t)(r.p - data) - base);
      if (vts[i] != 11u) continue;   // (8: a map delete — no payload)
      uint32_t tag = rd_u8(r);
      switch (tag) {
        case 0: case 1: case 2: break;
        case 3: (void)rd_sleb(r); break;
        case 4: rd_skip(r, 8); break;
        case 5: case 6: { uint64_t l = rd_uleb(r); rd_skip(r, l); break; }
        case 9: (void)rd_u8(r); break;   // a child container (any kind byte)
        default: okw = 0; break;         // 7 / 8: a nested value — the row tables' walkers; anything else: their verdict
      }
      if (r.bad || !okw) { okw = 0; break; }
    }
    np = (uint64_t)(r.p - data);
  }
  okw = lmw::bcast(okw, 0);
  uint32_t lo = lmw::bcast((uint32_t)np, 0), hi = lmw::bcast((uint32_t)(np >> 32), 0);
  vp = ((uint64_t)hi << 32) | lo;
  return okw != 0;
}

LM_KERNEL LM_WAVES_PER_SIMD(4) void k_map_fused(Dev d, DevMf f, uint32_t* retry_count) {
  const uint32_t doc = f.docs[(uint32_t)lmw::bid()];
  const uint32_t tid = (uint32_t)lmw::tid();
  const int lane = lmw::lane();
  const uint32_t wv = (uint32_t)lmw::wave_in_block();
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  const uint32_t cap = d.ht_cap[doc];
  LM_DYN_SHARED(unsigned long long, s_mem64);
  unsigned long long* s_key = s_mem64;                 // [cap] cidx (8) | key length (16) | absolute offset of the claimer's key bytes (40); ~0 = empty
  unsigned long long* s_pfx = s_key + LWW_LDS_CAP;     // [cap] first eight key bytes, big endian, zero padded
  unsigned long long* s_best = s_pfx + LWW_LDS_CAP;    // [cap] (lamport, peer, record) + 1 of the best write so far, 0 = none
  uint32_t* s_end = (uint32_t*)(s_best + LWW_LDS_CAP); // [MAX_PEERS] version being rendered per peer
  uint32_t* s_touch = s_end + MAX_PEERS;               // [MAX_CONTAINERS / 32]
  uint32_t* s_misc = s_touch + MAX_CONTAINERS / 32;    // [0] claimed slots, [1] bail, [2] soft-unsupported, [3] flush cursor, [4] error, [5] records
  uint8_t* s_wave = (uint8_t*)(s_misc + 8) + (size_t)wv * MF_WAVE_LDS;
  uint16_t* kpos = (uint16_t*)s_wave;                  // [nk + 1] start of every key's length byte, relative to the key section; [nk] = its end
  uint32_t* tmp = (uint32_t*)(s_wave + (MF_KMAX + 4) * 2);   // 5 rows of 64 words: cidx, prop, value type, len, value offset
  if (cap == 0 || cap > LWW_LDS_CAP) {   // (more Map rows than the LDS table is sized for cannot happen: ht_opt caps it; 0 = no Map row at all)
    if (cap != 0 && tid == 0) { lmw::atomic_or(&d.doc[doc].flags, DF_REDO); LM_SETERR(d.doc[doc].status, ST_DATA_CORRUPTION); }
    return;
  }
  for (uint32_t i = tid; i < cap; i += MF_WG) { s_key[i] = HT_EMPTY; s_pfx[i] = LWW_PFX_UNSET; s_best[i] = 0; }
  for (uint32_t i = tid; i < m.n_peers && i < MAX_PEERS; i += MF_WG) s_end[i] = d.peer_end[m.praw0 + i];
  for (uint32_t i = tid; i < MAX_CONTAINERS / 32 + 8; i += MF_WG) s_touch[i] = 0;   // (+ s_misc)
  lmw::block_sync();
  const uint64_t seed = 0xcbf29ce484222325ull;
  const uint8_t* data = d.data;
  // ---- every wave: a block at a time, last block first
  for (uint32_t bk = wv; bk < m.n_blk; bk += MF_WAVES) {
    const uint32_t bi = m.blk0 + (m.n_blk - 1 - bk);
    const BlockDesc* bdp = d.blk + bi;
    const uint64_t base = bdp->base;
    const uint32_t N = bdp->n_changes, cs = bdp->counter_start, cl = bdp->counter_len;
    const uint32_t n_ops = bdp->flags & 0x7fffffffu;          // (k_block_head: the rows of the value_type column)
    const uint32_t* bo = d.boff + (uint64_t)bi * BCN;
    const uint32_t chg0 = bo[BC_CHG], cidr0 = bo[BC_CID];
    const uint32_t n_cids = d.bcnt[(uint64_t)bi * BCN + BC_CID];
    bool bail = lmw::bcast(s_misc[1] != 0 ? 1u : 0u, 0) != 0;   // (another wave gave up on the document: nothing left to do; lane 0's read, so that the wave agrees)
    // ---- key starts: candidates = bytes below 0x20, four bytes per lane and step
    const uint64_t k0 = base + bdp->sec_rel[SEC_KEYS];
    const uint32_t klen_sec = bdp->sec_len[SEC_KEYS];
    uint32_t nk = 0;
    if (!bail) {
      if (klen_sec > 0xfff0u) bail = true;
      uint32_t expect = 0;            // where the next candidate has to be
      for (uint32_t w0 = 0; w0 < klen_sec && !bail; w0 += 256) {
        const uint32_t o = w0 + 4u * (uint32_t)lane;
        uint32_t by[4];
#pragma unroll
        for (int q = 0; q < 4; q++) by[q] = o + q < klen_sec ? data[k0 + o + q] : 0xffu;
        // ranks in byte order: lane-major, then q
        uint32_t cand[4], ncl = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { cand[q] = by[q] < 0x20u ? 1u : 0u; ncl += cand[q]; }
        const uint32_t incl = lmw::scan_incl_add(ncl);
        uint32_t rk = nk + incl - ncl;
        const uint32_t tot = lmw::bcast(incl, 63);
        if (nk + tot > MF_KMAX) { bail = true; break; }
#pragma unroll
        for (int q = 0; q < 4; q++) if (cand[q]) { kpos[rk] = (uint16_t)(o + q); rk++; }
        nk += tot;
        (void)expect;
      }
      if (!bail) {
        if ((uint32_t)lane == 0) kpos[nk] = (uint16_t)klen_sec;
        mf_wave_lds_sync();
        // the candidates are the chain iff the first is 0 and every one's successor is itself + 1 + its length
        bool okc = nk == 0 ? klen_sec == 0 : kpos[0] == 0;
        for (uint32_t k = (uint32_t)lane; k < nk; k += 64) {
          const uint32_t a = kpos[k], nx = kpos[k + 1];
          okc &= a + 1u + (uint32_t)data[k0 + a] == nx;
        }
        if (lmw::ballot(!okc)) bail = true;
      }
    }
    // ---- the op columns
    MfCol col[4];
    uint64_t vp = base + bdp->sec_rel[SEC_VALUES];
    const uint64_t vsec = vp, vend = vp + bdp->sec_len[SEC_VALUES];
    if (!bail) {
      Rd o = rd_make(data + base + bdp->sec_rel[SEC_OPS], bdp->sec_len[SEC_OPS]);
      uint64_t outer = rd_uleb(o), ncols = rd_uleb(o);
      if (outer != 1 || ncols != 4) bail = true;
      for (int q = 0; q < 4; q++) {
        Rd cq = rd_bytes(o);
        col[q].p = (uint64_t)(cq.p - data); col[q].end = (uint64_t)(cq.end - data);
        col[q].rem = 0; col[q].runv = 0; col[q].acc = 0; col[q].run = false;
        col[q].mode = q < 2 ? 2u : (q == 2 ? 0u : 1u);
      }
      if (o.bad || o.p != o.end) bail = true;
      if (bdp->sec_len[SEC_DEL] != 0 || n_ops != cl || N == 0 || N > n_ops) bail = true;   // (a Map op has one id: rows = ids; delete-start ids belong to sequences)
    }
    // ---- rows, 64 at a time
    uint32_t cj = 0;                                   // change of the chunk's first row (rows and changes advance together)
    ChangeRow ch;
    ch.peer = 0; ch.ctr = 0; ch.len = 0; ch.dep0 = ch.n_dep = ch.op0 = ch.n_op = ch.blk = 0;
    uint32_t ch_flag = 0, ch_skip = 0, ch_lam = 0;
    bool have_ch = false;
    for (uint32_t r0 = 0; r0 < n_ops && !bail; r0 += 64) {
      const uint32_t want = n_ops - r0 < 64 ? n_ops - r0 : 64u;
      int32_t v_ci = 0, v_prop = 0, v_vt = 0, v_len = 0;
      if (!mf_fetch(data, col[0], want, tmp, lane, v_ci) || !mf_fetch(data, col[1], want, tmp + 64, lane, v_prop) ||
          !mf_fetch(data, col[2], want, tmp + 128, lane, v_vt) || !mf_fetch(data, col[3], want, tmp + 192, lane, v_len)) { bail = true; break; }
      const bool act = (uint32_t)lane < want;
      v_vt &= 0x7f;
      if (lmw::any(act && (v_len != 1 || (v_vt != 11 && v_vt != 8) || v_ci < 0 || (uint32_t)v_ci >= n_cids || v_prop < 0 || (uint32_t)v_prop >= nk))) { bail = true; break; }
      // values
      uint32_t* tmpv = tmp + 256;
      bool fast = !lmw::any(act && v_vt != 11);
      if (fast) { uint64_t vq = vp; fast = mf_values_i64(data, vq, vend, want, tmpv, 0, lane); if (fast) vp = vq; }
      if (!fast) {
        if (act) tmp[128 + lane] = (uint32_t)v_vt;
        mf_wave_lds_sync();
        if (!mf_values_slow(data, vp, vend, want, tmp + 128, tmpv, 0, lane)) { bail = true; break; }
      }
      mf_wave_lds_sync();
      const uint64_t val_at = act ? (uint64_t)tmpv[lane] | (vsec & ~0xffffffffull) : 0;   // (offsets are kept as 32-bit words: the high half is the section's)
      mf_wave_lds_sync();
      // rows of this chunk, change by change (wave-uniform walk over the block's changes; a chunk meets one or two)
      const uint32_t ctr = cs + r0 + (uint32_t)lane;
      uint32_t done_to = r0;   // rows below are assigned
      while (done_to < r0 + want && !bail) {
        if (!have_ch) {
          if (cj >= N) { bail = true; break; }
          ch = d.chg[chg0 + cj];
          ch_flag = d.chg_flag[chg0 + cj]; ch_skip = d.chg_skip[chg0 + cj]; ch_lam = d.chg_lamport[chg0 + cj];
          have_ch = true;
          if (ch.len == 0 || ch.ctr != cs + done_to) { bail = true; break; }   // (a zero-length change / rows that do not start where the change does)
        }
        const uint32_t c_end = ch.ctr + ch.len;                       // first id behind the change
        const uint32_t upto = c_end - cs < r0 + want ? c_end - cs : r0 + want;
        const bool in = act && r0 + (uint32_t)lane >= done_to && r0 + (uint32_t)lane < upto;
        if (ch_flag && ch.peer < MAX_PEERS) {
          // (k_map_lww_doc's filters: the known prefix of a sliced change, a write beyond the rendered version)
          const uint32_t cidx = in ? d.cid_map[cidr0 + (uint32_t)v_ci] : 0u;
          bool go = in && ctr >= ch.ctr + ch_skip;
          if (lmw::any(go && cidx >= MAX_CONTAINERS)) { s_misc[4] = 1; go = false; }
          if (go) lmw::atomic_or(&s_touch[cidx >> 5], 1u << (cidx & 31));
          go = go && ctr < s_end[ch.peer];
          if (go) {
            const uint32_t ka = kpos[v_prop];
            const uint64_t kat = k0 + ka + 1;
            const uint32_t kl = data[k0 + ka];
            unsigned long long pf = 0;
            for (uint32_t q = 0; q < 8; q++) pf = (pf << 8) | (q < kl ? data[kat + q] : 0u);
            uint64_t h = (seed ^ cidx ^ ((uint64_t)kl << 32)) * 0x100000001b3ull;
            h = (h ^ pf) * 0x9E3779B97F4A7C15ull;
            if (kl > 8) h = fnv1a(data + kat + 8, kl - 8, h);
            h ^= h >> 29;
            const unsigned long long lp = ((unsigned long long)(ch_lam + (ctr - ch.ctr)) << 8) | (unsigned long long)ch.peer;   // (lamport, peer): what competes
            const unsigned long long head = ((unsigned long long)cidx << 56) | ((unsigned long long)kl << 40);
            const unsigned long long mine = head | (kat & 0xffffffffffull);
            uint32_t slot = (uint32_t)h & (cap - 1);
            bool placed = false;
            for (uint32_t probe = 0; probe < cap; probe++, slot = (slot + 1) & (cap - 1)) {
              unsigned long long cur = s_key[slot];
              if (cur == HT_EMPTY) {
                cur = lmw::atomic_cas64(&s_key[slot], HT_EMPTY, mine);
                if (cur == HT_EMPTY) {
                  s_pfx[slot] = pf;
                  uint32_t at = lmw::atomic_add(&s_misc[0], 1u);
                  if (at >= cap / 2) s_misc[1] = 1;
                  cur = mine;
                }
              }
              bool same = cur == mine;
              if (!same && (cur >> 40) == (mine >> 40)) {   // same container, same length: the bytes decide
                unsigned long long op = s_pfx[slot];
                if (op != LWW_PFX_UNSET && op != pf) same = false;
                else if (op != LWW_PFX_UNSET && kl <= 8) same = true;
                else same = bytes_eq(data + (cur & 0xffffffffffull), data + kat, kl);
              }
              if (same) {
                const unsigned long long cb = s_best[slot];
                if (cb == 0 || ((cb - 1) >> 24) < lp) {
                  // this row raises the key's maximum (as far as a plain read can tell): it gets a record
                  const uint32_t x = lmw::atomic_add(&s_misc[5], 1u);
                  if (x >= (1u << 24) || x >= m.n_op) { s_misc[1] = 1; }
                  else {
                    lmw::atomic_max64(&s_best[slot], ((lp << 24) | x) + 1);
                    OpRow r;
                    r.cidx_kind = cidx | ((v_vt == 11 ? OK_MAP_SET : OK_MAP_DEL) << 16);
                    r.prop = v_prop; r.len = 1; r.ctr = ctr; r.a0 = f.key0[doc] + slot; r.a1 = 0; r.a2 = 0; r.chg = chg0 + cj;
                    d.op[m.op0 + x] = r;
                    d.op_val[m.op0 + x] = val_at;
                    d.op_blk[m.op0 + x] = bi;
                  }
                }
                placed = true;
                break;
              }
            }
            if (!placed) s_misc[1] = 1;
          }
        } else if (ch_flag) { s_misc[4] = 1; }
        done_to = upto;
        if (upto == c_end - cs) { cj++; have_ch = false; }
      }
    }
    if (!bail) {
      // everything is used up exactly: the columns, the values, the changes
      if (!mf_col_done(col[0]) || !mf_col_done(col[1]) || !mf_col_done(col[2]) || !mf_col_done(col[3]) || vp != vend || cj != N || have_ch) bail = true;
    }
    if (bail && lane == 0) s_misc[1] = 1;
  }
  lmw::block_sync();
  // ---- results (k_map_lww_doc's): containers, flags, the claimed slots slot for slot into the document's global table
  if (s_misc[1]) {
    if (tid == 0) { lmw::atomic_or(&d.doc[doc].flags, DF_REDO); LM_SETERR(d.doc[doc].status, ST_DATA_CORRUPTION); d.ht_cnt[doc] = 0; }
    return;
  }
  for (uint32_t c = tid; c < m.n_cont && c < MAX_CONTAINERS; c += MF_WG)
    if ((s_touch[c >> 5] >> (c & 31)) & 1) d.cont[m.cid0 + c].touched = 1;
  if (tid == 0 && s_misc[4]) LM_SETERR(d.doc[doc].status, ST_INTERNAL);
  unsigned long long* keys = d.ht_key + d.ht0[doc];
  unsigned long long* best = d.ht_best + d.ht0[doc];
  uint32_t* list = d.ht_list + 2 * d.ht0[doc];
  const uint32_t kf0 = f.key0[doc];
  for (uint32_t s = tid; s < cap; s += MF_WG) {
    unsigned long long k = s_key[s];
    if (k == HT_EMPTY) continue;
    keys[s] = ((unsigned long long)(uint32_t)(k >> 56) << 32) | (kf0 + s);
    best[s] = s_best[s];
    d.key_off[kf0 + s] = k & 0xffffffffffull;
    d.key_len[kf0 + s] = (uint32_t)(k >> 40) & 0xffffu;
    list[lmw::atomic_add(&s_misc[3], 1u)] = s;
  }
  lmw::block_sync();
  if (tid == 0) d.ht_cnt[doc] = s_misc[3];
  (void)retry_count;
}

}  // namespace lm
