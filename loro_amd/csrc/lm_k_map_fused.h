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
// Every pass reads the block straight from HBM with loads whose ADDRESSES do not depend on what was read before (256 bytes per wave
// and step, four per lane), so they pipeline; what is sequential in the format is resolved in registers:
//   * key table: `uleb len, bytes` per key.  The starts are found WITHOUT walking the chain: a key byte is >= 0x20 and a length
//     below 0x20 is not, so the candidates (bytes < 0x20) are ranked by one wave scan per step, and accepted only when every
//     candidate's successor is candidate + 1 + length — which makes the candidates the chain (induction from offset 0).  Keys of 32
//     bytes or more, or with control characters, take the general path: every lane works out where the key that would start at ITS
//     byte ends, and the chain hops from start to start through v_readlane (no memory round trip per key);
//   * the key-index column (prop, DeltaRle): a literal segment is cut at its varint terminators — rank by the same scan — and
//     decoded by the lane that holds the terminator; runs are filled by arithmetic; one more pass turns deltas into indices.  The
//     other three columns (container_index, value_type, len) are runs in practice: read once when they are, 64 rows at a time otherwise;
//   * values: `tag, payload`.  Integer values (tag 3 + sleb128) are cut by the PARITY of the bytes without a continuation bit — tags
//     and terminators alternate — and verified (every tag is 3 and directly follows a terminator); any other mix of scalars takes
//     the general path of the key table (every lane: "a value that starts here ends there", then the register chain);
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

static constexpr uint32_t MF_WG = 1024;          // lanes per document (16 waves: one workgroup per CU by LDS)
static constexpr uint32_t MF_WAVES = MF_WG / 64;
static constexpr uint32_t MF_KMAX = 1020;        // keys of one block (kpos[] in LDS)
static constexpr uint32_t MF_RMAX = 1024;        // rows of one block (a_prop[], a_voff[] in LDS)
static constexpr uint32_t MF_WAVE_LDS = (MF_KMAX + 4) * 2 + MF_RMAX * 2 * 2 + 3 * 64 * 4 + 128;   // per wave: kpos, a_prop, a_voff (u16), three 64-word exchange rows, the delete bitmap
static constexpr uint32_t MF_LDS = LWW_LDS_CAP * 24 + (MAX_PEERS + MAX_CONTAINERS / 32 + 8) * 4 + MF_WAVES * MF_WAVE_LDS;

struct DevMf {
  const uint32_t* docs;      // the fused documents of the batch (workgroup -> document)
  const uint8_t* doc_fused;  // per document: 1 = its blocks are decoded by this kernel
  const uint32_t* key0;      // per document: first of its slot-numbered key rows (d.key_off / d.key_len, behind the decoders' rows)
  uint32_t stop_after;       // timing experiments only (LM_MF_STOP): 1 = after the key tables, 2 = + key index column, 3 = + values, 4 = + row columns / changes (no table work); 0 = the kernel
};

// one of the three small columns of the block's EncodedOp table, read 64 values at a time by the whole wave (or once, when the column
// is a single run); every field is wave-uniform
struct MfCol { uint64_t p, end; int64_t rem; int64_t runv; int32_t acc; uint32_t mode; bool run, uniform; int32_t uval; };

#ifdef LM_PROF_MF   // experiment build: ticks per part of k_map_fused, summed over the batch into d.prof[doc][0..15] (tests/tools/gpu_mf_phases.py)
#define MF_PH(i) do { uint64_t n_ = lmw::clock(); mfp[i] += n_ - mft; mft = n_; } while (0)
#define MF_PH_ARGS , uint64_t* mfp, uint64_t& mft
#define MF_PH_PASS , mfp, mft
#else
#define MF_PH(i) do {} while (0)
#define MF_PH_ARGS
#define MF_PH_PASS
#endif
LM_DEV void mf_wave_lds_sync() {
#ifndef LM_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  lmw::wave_sync();
#endif
}
// four bytes of `data` at `pos` (any alignment), bytes at or beyond `end` read as `fill`
// (`data` carries 64 bytes of slack behind the last blob — lm_pipeline.h stage / import_more — so a word that STARTS in front of `end`
// may be read whole: one predicated load and a mask instead of a byte loop for the words that straddle the end)
LM_DEV uint32_t mf_ld4(const uint8_t* data, uint64_t pos, uint64_t end, uint32_t fill) {
  const uint32_t f4 = fill * 0x01010101u;
  uint32_t x = f4;
  if (pos < end) {
    x = ld32u(data + pos);
    const uint64_t left = end - pos;
    if (left < 4) { const uint32_t keep = (1u << (8 * (uint32_t)left)) - 1u; x = (x & keep) | (f4 & ~keep); }
  }
  return x;
}

// A 256-byte WINDOW of `data` held in registers, four bytes per lane: the headers of the op columns (segment counts, run values)
// are parsed out of it with v_readlane — every position is wave-uniform — instead of one dependent HBM round trip per byte.
struct MfWin { uint64_t w0; uint32_t x; };
LM_DEV MfWin mf_win(const uint8_t* data, uint64_t at, uint64_t end, int lane) { MfWin w; w.w0 = at; w.x = mf_ld4(data, at + 4u * (uint32_t)lane, end, 0u); return w; }
struct MfRd { uint64_t p, end; bool bad; };
LM_DEV uint32_t mw_u8(const uint8_t* data, const MfWin& w, MfRd& r) {
  if (r.p >= r.end) { r.bad = true; return 0; }
  const uint64_t o = r.p - w.w0;
  uint32_t b;
  if (r.p >= w.w0 && o < 256) b = (lmw::bcast(w.x, (int)(o >> 2)) >> (8 * (uint32_t)(o & 3))) & 0xffu;
  else b = data[r.p];
  r.p++;
  return b;
}
LM_DEV uint64_t mw_uleb(const uint8_t* data, const MfWin& w, MfRd& r) {
  uint64_t v = 0;
  for (int i = 0; i < 10; i++) {
    uint32_t b = mw_u8(data, w, r);
    v |= (uint64_t)(b & 0x7f) << (7 * i);
    if (!(b & 0x80)) return v;
  }
  r.bad = true;
  return v;
}
LM_DEV int64_t mw_zigzag(const uint8_t* data, const MfWin& w, MfRd& r) { uint64_t v = mw_uleb(data, w, r); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
LM_DEV int64_t mw_any(const uint8_t* data, const MfWin& w, MfRd& r, uint32_t mode) {   // == rd_any for the values this kernel takes (a value of ten bytes and more: not its business)
  if (mode == 0) return (int64_t)mw_u8(data, w, r);
  if (mode == 1) return (int64_t)mw_uleb(data, w, r);
  return mw_zigzag(data, w, r);
}

// the next `want` (<= 64) values of the column into lane order; false: the column does not decode the way this kernel reads it
LM_DEV bool mf_fetch(const uint8_t* data, MfCol& c, uint32_t want, uint32_t* tmp, int lane, int32_t& out) {
  if (c.uniform) { out = c.uval; return true; }
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
        if (lmw::any(mine && len > 4)) { ok = false; break; }   // (a value beyond 2^28: not a container index / length of a block this kernel takes)
        if (mine) tmp[filled + got + rank] = c.mode == 2 ? (uint32_t)((int32_t)(v >> 1) ^ -(int32_t)(v & 1)) : v;
        const uint64_t last = lmw::ballot(term && rank == n - 1);   // bytes consumed: up to and including the n-th terminator
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
// a column that is ONE run over all the block's rows is read once (the three small columns nearly always are)
LM_DEV bool mf_col_open(const uint8_t* data, const MfWin& w, MfCol& c, uint32_t n_rows) {
  c.uniform = false; c.uval = 0;
  MfRd r{c.p, c.end, false};
  if (r.p >= r.end) return n_rows == 0;
  int64_t k = mw_zigzag(data, w, r);
  if (r.bad || k != (int64_t)n_rows) return true;          // (anything else: the chunked reader, which owns the checks)
  int64_t v = mw_any(data, w, r, c.mode);
  if (r.bad || r.p != r.end || v > (1 << 24) || v < -(1 << 24)) return true;
  if (c.mode == 2 && v != 0 && n_rows > 1) return true;    // (a delta run: the values differ from row to row)
  c.uniform = true; c.uval = (int32_t)v; c.p = c.end; c.rem = 0;
  return true;
}
LM_DEV bool mf_col_done(const MfCol& c) { return c.rem == 0 && c.p == c.end; }

// ---- the key-index column: DeltaRle over all rows -> a_prop[row] (absolute index, u16).  false: not what this kernel reads.
// A block numbers its keys in the order it meets them, so the column is MANY short segments — runs of +1 while new keys turn up,
// two or three literal deltas where an earlier key is written again (95 segments in a 500-row block of configs[2]): the column is
// copied into LDS once (`cb`: the value-offset table's space, which is filled afterwards) and walked segment by segment from there —
// a head is a couple of LDS reads, a run one store per 64 rows, a literal one ballot over its bytes; from HBM every segment was a
// dependent round trip of its own (125 us per block).  A column beyond `cb_cap` bytes is walked in HBM by the same code.
LM_DEV bool mf_prop_all(const uint8_t* data, uint64_t p, uint64_t end, uint32_t n_rows, uint32_t nk, uint16_t* a_prop, uint8_t* cb, uint32_t cb_cap, uint8_t* mk, uint32_t mk_cap, int lane MF_PH_ARGS) {
  const uint32_t L = (uint32_t)(end - p);
  const uint8_t* bp = data + p;
  if (L <= cb_cap) {
    for (uint32_t o = 4u * (uint32_t)lane; o < L; o += 256) st32u(cb + o, mf_ld4(data, p + o, end, 0u));   // (cb is 4-byte aligned; the last store may pad up to three bytes)
    mf_wave_lds_sync();
    bp = cb;
  }
  MF_PH(3);
  uint32_t q = 0;                      // position in the column
  uint32_t row = 0;
  bool bad = false;
  const bool in_lds = bp == cb;
  auto slow_segment = [&]() -> bool {
    // ---- one segment, step by step (a column that is walked in HBM; a head the window could not take)
    Rd r = rd_make(bp + q, L - q);
    if (r.p >= r.end) return false;
    int64_t k = rd_zigzag(r);
    if (k == 0 || r.bad || k > (int64_t)(n_rows - row) || -k > (int64_t)(n_rows - row)) return false;
    if (k > 0) {
      int64_t dv = rd_zigzag128(r);
      if (r.bad || dv > 32767 || dv < -32768) return false;
      q = (uint32_t)(r.p - bp);
      for (uint32_t i = (uint32_t)lane; i < (uint32_t)k; i += 64) a_prop[row + i] = (uint16_t)(int16_t)dv;
      row += (uint32_t)k;
      MF_PH(5);
      return true;
    }
    q = (uint32_t)(r.p - bp);
    const uint32_t lit0 = q;             // no varint of the segment reaches in front of this byte
    uint32_t need = (uint32_t)(-k), got = 0;
    while (got < need) {
      const uint32_t pos = q + (uint32_t)lane;
      const bool inb = pos < L;
      const uint32_t b = inb ? bp[pos] : 0x80u;
      const bool term = inb && !(b & 0x80u);
      const uint64_t mask = lmw::ballot(term);
      if (!mask) return false;           // 64 bytes without a terminator
      const uint32_t rank = (uint32_t)lmw::popc64(mask & ((1ull << lane) - 1ull));
      const uint32_t navail = (uint32_t)lmw::popc64(mask);
      const uint32_t n = need - got < navail ? need - got : navail;
      // (the three bytes in front are requested with the lane's own: one round trip)
      const uint32_t b1 = inb && pos >= lit0 + 1 ? bp[pos - 1] : 0u, b2 = inb && pos >= lit0 + 2 ? bp[pos - 2] : 0u, b3 = inb && pos >= lit0 + 3 ? bp[pos - 3] : 0u;
      if (term && rank < n) {
        // the bytes in front of the terminator that carry a continuation bit belong to it (least significant group first)
        const bool c1 = (b1 & 0x80u) != 0, c2 = c1 && (b2 & 0x80u), c3 = c2 && (b3 & 0x80u);
        const uint32_t u = c2 ? ((b2 & 0x7fu) | ((b1 & 0x7fu) << 7) | (b << 14)) : c1 ? ((b1 & 0x7fu) | (b << 7)) : b;
        const int32_t dv = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);
        if (c3 || dv > 32767 || dv < -32768) bad = true;   // four bytes and more: not a key index
        a_prop[row + got + rank] = (uint16_t)(int16_t)dv;
      }
      const uint64_t last = lmw::ballot(term && rank == n - 1);   // bytes consumed: up to and including the n-th terminator
      q += (uint32_t)lmw::ffs64(last) + 1;
      got += n;
    }
    row += need;
    MF_PH(6);
      return true;
  };
  // Staged column: every lane works out the segment that WOULD start at its byte of the 64-byte window behind q — head (a count of
  // at most 63 in one byte), a run's delta, or a literal of up to four values that end inside the lane's eight bytes — and the walk
  // hops from head to head through v_readlane: three readlanes and one LDS store per segment instead of a dependent parse of each
  // head (25 % of the kernel before).  Anything else at a head (a long literal, a two-byte count, a three-byte delta) is taken by
  // the step-by-step code below, after which the walk goes on.
  while (row < n_rows) {
    if (q >= L) return false;
    bool slow = !in_lds;
    if (in_lds) {
      const uint32_t w0 = q;
      const uint32_t at = w0 + (uint32_t)lane;
      // T: the window's bytes without a continuation bit (bit l = byte w0 + l ends a varint) — one ballot; a segment that would start
      // at a lane's byte is then a matter of bit counting: a run's delta ends at the first such byte behind the head, a literal of n
      // values at the n-th
      uint32_t info = 0;               // adv (8) | rows (8) | kind (1 run, 2 literal) << 16; 0 = not a head these lanes can take
      uint32_t rdelta = 0;             // a run's delta (i16)
      const uint32_t myb = at < L ? cb[at] : 0x80u;
      const uint64_t T = lmw::ballot(at < L && !(myb & 0x80u));
      if (at < L && !(myb & 0x80u) && myb) {
        const int32_t k = (int32_t)(myb >> 1) ^ -(int32_t)(myb & 1u);
        const uint64_t Tr = lane < 63 ? T >> (lane + 1) : 0ull;          // terminators behind the head byte
        if (k > 0) {
          const uint32_t e = Tr ? (uint32_t)__builtin_ctzll(Tr) : 64u;    // the delta's last byte, relative to the byte behind the head
          if (e < 2) {
            const uint32_t b0 = cb[at + 1], b1 = cb[at + 2];
            const uint32_t u = e == 0 ? b0 : ((b0 & 0x7fu) | (b1 << 7));
            rdelta = (uint32_t)(uint16_t)(int16_t)((int32_t)(u >> 1) ^ -(int32_t)(u & 1u));
            info = (2u + e) | ((uint32_t)k << 8) | (1u << 16);
          }
        } else if (k >= -24) {
          const uint32_t n = (uint32_t)(-k);
          if ((uint32_t)lmw::popc64(Tr) >= n) {
            uint64_t m = Tr;
            for (uint32_t j = 1; j < n; j++) m &= m - 1;                  // (the n-th terminator: n - 1 lowest bits cleared)
            const uint32_t e = (uint32_t)__builtin_ctzll(m);
            info = (2u + e) | (n << 8) | (2u << 16);
          }
        }
        if ((info & 0xffu) + at > L) info = 0;                          // (the segment would run past the column)
      }
      // the walk through this window only MARKS the heads (a readlane and a handful of scalar operations per segment) …
      uint32_t off = 0;
      for (;;) {
      uint64_t heads = 0;
      uint32_t rows_w = 0;
      slow = false;
      while (off < 64) {
        const uint32_t inf = lmw::bcast(info, (int)off);
        if (!inf) { slow = true; break; }
        const uint32_t n = (inf >> 8) & 0xffu;
        if (n > n_rows - row - rows_w) return false;
        if (rows_w + n > mk_cap) break;                                  // (the marks' space: the next window starts at this head)
        heads |= 1ull << off;
        rows_w += n;
        off += inf & 0xffu;
        if (row + rows_w >= n_rows) break;
      }
      // … the rows then PULL their deltas: every head lane knows its first row (one scan of the row counts) and leaves its lane number
      // there; a running maximum over the marks tells every row its head, whose delta(s) it fetches with a lane permute
      if (rows_w) {
        const bool is_head = (heads >> lane) & 1ull;
        const uint32_t my_n = is_head ? (info >> 8) & 0xffu : 0u;
        const uint32_t r_h = lmw::scan_incl_add(my_n) - my_n;
        for (uint32_t i = (uint32_t)lane; i < (rows_w + 3) / 4; i += 64) ((uint32_t*)mk)[i] = 0;
        mf_wave_lds_sync();
        if (is_head) mk[r_h] = (uint8_t)(lane + 1);
        mf_wave_lds_sync();
        uint32_t carry = 0;
        for (uint32_t c0 = 0; c0 < rows_w; c0 += 64) {
          const uint32_t rho = c0 + (uint32_t)lane;
          uint32_t h = rho < rows_w ? mk[rho] : 0u;
          h = lmw::scan_incl_max(h);
          h = h > carry ? h : carry;
          carry = lmw::bcast(h, 63);
          const int hl = (int)((h - 1u) & 63u);
          const uint32_t inf = lmw::shfl(info, hl), rd = lmw::shfl(rdelta, hl), rh = lmw::shfl(r_h, hl);
          if (rho < rows_w) {
            uint32_t dv = rd;
            if ((inf >> 16) != 1u) {
              // the (rho - rh)-th varint behind the head byte: between the terminator in front of it and its own
              const uint32_t j = rho - rh;
              uint64_t m = hl < 63 ? T >> (hl + 1) : 0ull;
              uint32_t st = 0;
              for (uint32_t i = 0; i < j; i++) { st = (uint32_t)__builtin_ctzll(m) + 1; m &= m - 1; }
              const uint32_t en = (uint32_t)__builtin_ctzll(m);
              const uint32_t p0 = w0 + (uint32_t)hl + 1 + st;
              const uint32_t nb = en - st + 1;
              const uint32_t c0 = cb[p0], c1 = cb[p0 + 1], c2 = cb[p0 + 2];
              const uint32_t u = nb == 1 ? c0 : nb == 2 ? ((c0 & 0x7fu) | (c1 << 7)) : ((c0 & 0x7fu) | ((c1 & 0x7fu) << 7) | (c2 << 14));
              const int32_t d = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);
              if (nb > 3 || d > 32767 || d < -32768) bad = true;
              dv = (uint32_t)(uint16_t)(int16_t)d;
            }
            a_prop[row + rho] = (uint16_t)dv;
          }
        }
        mf_wave_lds_sync();
        row += rows_w;
      }
      // a head the lanes could not take: that one segment step by step, then on with the SAME window (its lanes' work is not redone)
      if (!slow || row >= n_rows) break;
      q = w0 + off;
      if (q >= L) return false;
      MF_PH(4);
      if (!slow_segment()) return false;
      off = q - w0;
      if (off >= 64 || row >= n_rows) { slow = false; break; }
      }
      q = w0 + off;
      MF_PH(4);
      continue;
    }
    if (!slow_segment()) return false;
  }
  if (q != L || lmw::any(bad)) return false;                      // (surplus values / bytes: the row decoders' verdict)
  mf_wave_lds_sync();
  // deltas -> indices
  int32_t acc = 0;
  for (uint32_t r0 = 0; r0 < n_rows; r0 += 64) {
    const uint32_t i = r0 + (uint32_t)lane;
    int32_t dv = i < n_rows ? (int32_t)(int16_t)a_prop[i] : 0;
    int32_t s = acc + (int32_t)lmw::scan_incl_add((uint32_t)dv);
    acc = (int32_t)lmw::bcast((uint32_t)s, 63);
    if (i < n_rows) { if (s < 0 || (uint32_t)s >= nk) bad = true; a_prop[i] = (uint16_t)s; }
  }
  mf_wave_lds_sync();
  MF_PH(7);
  return !lmw::any(bad);
}

// ---- values, every one an integer (tag 3 + sleb128): offsets of the tags -> a_voff[row] (relative to the section).  false: they are not
LM_DEV bool mf_values_int_all(const uint8_t* data, uint64_t vsec, uint64_t vend, uint32_t n_rows, uint16_t* a_voff, int lane) {
  const uint32_t need = 2 * n_rows;
  uint32_t got = 0;
  uint32_t prev_x = 0;
  bool first = true, bad = false;
  uint32_t xq[4] = {0, 0, 0, 0};   // four steps of 256 bytes are requested together
  uint32_t qi = 4;
  for (uint64_t p = vsec; p < vend; p += 256) {
    const uint64_t pos = p + 4u * (uint32_t)lane;
    if (qi == 4) {
#pragma unroll
      for (uint32_t u = 0; u < 4; u++) xq[u] = mf_ld4(data, pos + 256u * u, vend, 0x80u);
      qi = 0;
    }
    const uint32_t x = qi == 0 ? xq[0] : qi == 1 ? xq[1] : qi == 2 ? xq[2] : xq[3];
    qi++;
    const uint32_t cl = ~x & 0x80808080u;                          // bytes without a continuation bit: tags and terminators, alternating
    const uint32_t cnt = (uint32_t)__builtin_popcount(cl);
    const uint32_t incl = lmw::scan_incl_add(cnt);
    const uint32_t tot = lmw::bcast(incl, 63);
    uint32_t px = lmw::shift_up(x, 1);
    if (lane == 0) px = first ? 0u : prev_x;                        // (in front of the section: "a terminator")
    uint32_t rk = got + incl - cnt;
    // a sleb128 holds at most ten bytes: eight continuation bytes in a row (this lane's dword and the one in front) are taken as
    // "beyond this path" — values near +-2^63 are the row decoders'
    if ((x & 0x80808080u) == 0x80808080u && (px & 0x80808080u) == 0x80808080u && pos < vend && !(first && lane == 0)) bad = true;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
      if (!((cl >> (8 * q + 7)) & 1u)) continue;
      if (!(rk & 1u)) {   // a tag
        const uint32_t b = (x >> (8 * q)) & 0xffu;
        const uint32_t pb = q ? (x >> (8 * (q - 1))) & 0x80u : (px >> 24) & 0x80u;   // continuation bit of the byte in front
        if (b != 3u || pb || rk >= need) bad = true;
        else a_voff[rk >> 1] = (uint16_t)(pos + q - vsec);
      }
      rk++;
    }
    got += tot;
    prev_x = lmw::bcast(x, 63);
    first = false;
  }
  // every row has its tag and its terminator, and the section ends with a terminator
  if (got != need) bad = true;
  if (vend > vsec && (data[vend - 1] & 0x80u)) bad = true;
  mf_wave_lds_sync();
  return !lmw::any(bad);
}

// ---- the general chain: every lane says where the record that would start at ITS byte ends, the chain hops through v_readlane.
// KEYS: `uleb len, bytes`; values: `tag, payload` of a scalar (a map delete — bit set in `dels` — has none).  Starts -> out[] (u16,
// relative to the section).  Returns the number of records, or NONE when the section is not a chain of such records.
template <bool KEYS>
LM_DEV uint32_t mf_chain(const uint8_t* data, uint64_t sec, uint64_t send, uint32_t max_n, uint32_t want_n, const uint32_t* dels, uint16_t* out, int lane) {
  uint32_t n = 0;
  uint64_t start = sec;             // where the next record starts (wave-uniform)
  bool bad = false;
  if (send - sec > 0xfff0u) return NONE;
  while (KEYS ? start < send : n < want_n) {
    if (!KEYS) {
      // rows without a payload in front of the next value
      bool d = (dels[n >> 5] >> (n & 31)) & 1u;
      if (d) { if (lane == 0) out[n] = (uint16_t)(start - sec); n++; continue; }
      if (start >= send) { bad = true; break; }
    }
    if (n >= max_n) { bad = true; break; }
    // the 64 bytes from `start`: lane l = the record that would start at start + l
    const uint64_t w0 = start;
    const uint64_t at = w0 + (uint32_t)lane;
    uint32_t len = 0;               // bytes of that record; 0 = none (beyond the section / not a record this kernel reads)
    if (at < send) {
      Rd r = rd_make(data + at, send - at);
      if (KEYS) { uint64_t l = rd_uleb(r); rd_skip(r, l); }
      else {
        uint32_t tag = rd_u8(r);
        switch (tag) {
          case 0: case 1: case 2: break;
          case 3: (void)rd_sleb(r); break;
          case 4: rd_skip(r, 8); break;
          case 5: case 6: { uint64_t l = rd_uleb(r); rd_skip(r, l); break; }
          case 9: (void)rd_u8(r); break;   // a child container (any kind byte)
          default: r.bad = true; break;    // 7 / 8: a nested value — the row tables' walkers; anything else: their verdict
        }
      }
      if (!r.bad) len = (uint32_t)(r.p - (data + at));
    }
    // hop from record to record while the starts stay inside this window
    uint32_t off = 0;
    while (off < 64 && (KEYS ? w0 + off < send : n < want_n)) {
      if (!KEYS) {
        bool d = (dels[n >> 5] >> (n & 31)) & 1u;
        if (d) { if (lane == 0) out[n] = (uint16_t)(w0 + off - sec); n++; continue; }
        if (w0 + off >= send) { bad = true; break; }
      }
      if (n >= max_n) { bad = true; break; }
      const uint32_t l = lmw::bcast(len, (int)off);
      if (l == 0) { bad = true; break; }
      if (lane == 0) out[n] = (uint16_t)(w0 + off - sec);
      n++;
      off += l;
    }
    if (bad) break;
    start = w0 + off;
  }
  if (bad || start > send || (KEYS && start != send)) return NONE;
  if (!KEYS && start != send) return NONE;
  if (KEYS && lane == 0) out[n] = (uint16_t)(send - sec);
  mf_wave_lds_sync();
  return n;
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
  uint16_t* kpos = (uint16_t*)s_wave;                  // [nk + 1] start of every key's length prefix, relative to the key section; [nk] = its end
  uint16_t* a_prop = kpos + (MF_KMAX + 4);             // [rows] key index of every row
  uint16_t* a_voff = a_prop + MF_RMAX;                 // [rows] its value, relative to the value section
  uint32_t* tmp = (uint32_t*)(a_voff + MF_RMAX);       // 3 rows of 64 words: container index, value type, len (the chunked column reader)
  uint32_t* dels = tmp + 3 * 64;                       // [MF_RMAX / 32] rows that are map deletes (the general value chain)
  if (cap == 0 || cap > LWW_LDS_CAP) {   // (0 = no Map row at all)
    if (cap != 0 && tid == 0) { lmw::atomic_or(&d.doc[doc].flags, DF_REDO); LM_SETERR(d.doc[doc].status, ST_DATA_CORRUPTION); }
    return;
  }
  for (uint32_t i = tid; i < cap; i += MF_WG) { s_key[i] = HT_EMPTY; s_pfx[i] = LWW_PFX_UNSET; s_best[i] = 0; }
  for (uint32_t i = tid; i < m.n_peers && i < MAX_PEERS; i += MF_WG) s_end[i] = d.peer_end[m.praw0 + i];
  for (uint32_t i = tid; i < MAX_CONTAINERS / 32 + 8; i += MF_WG) s_touch[i] = 0;   // (+ s_misc)
  lmw::block_sync();
  const uint64_t seed = 0xcbf29ce484222325ull;
  const uint8_t* data = d.data;
#ifdef LM_PROF_MF
  uint64_t mfp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mft = lmw::clock();   // 0 block set-up | 1 keys | 2 op header + small columns | 3 prop: staging | 4 heads | 5 runs | 6 literals | 7 prefix | 8 values | 9 row columns / changes | 10 table
#endif
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
    if (n_ops > MF_RMAX || n_ops != cl || N == 0 || N > n_ops || bdp->sec_len[SEC_DEL] != 0) bail = true;   // (a Map op has one id: rows = ids; delete-start ids belong to sequences)
    MF_PH(0);
    // ---- key starts
    const uint64_t k0 = base + bdp->sec_rel[SEC_KEYS];
    const uint32_t klen_sec = bdp->sec_len[SEC_KEYS];
    uint32_t nk = 0;
    if (!bail) {
      if (klen_sec > 0xfff0u) bail = true;
      bool fastk = !bail;
      for (uint32_t w0 = 0; w0 < klen_sec && fastk; w0 += 256) {   // candidates = bytes below 0x20, four bytes per lane and step
        const uint32_t o = w0 + 4u * (uint32_t)lane;
        const uint32_t x = mf_ld4(data, k0 + o, k0 + klen_sec, 0xffu);
        // bytes below 0x20: none of their bits 5-7 is set (folded onto bit 5 of each byte: the shifts stay inside the byte)
        const uint32_t c5 = ~(x | (x >> 1) | (x >> 2)) & 0x20202020u;
        const uint32_t cand = ((c5 >> 5) & 1u) | ((c5 >> 12) & 2u) | ((c5 >> 19) & 4u) | ((c5 >> 26) & 8u);
        const uint32_t ncl = (uint32_t)__builtin_popcount(c5);
        const uint32_t incl = lmw::scan_incl_add(ncl);
        uint32_t rk = nk + incl - ncl;
        const uint32_t tot = lmw::bcast(incl, 63);
        if (nk + tot > MF_KMAX) { fastk = false; bail = true; break; }
#pragma unroll
        for (int q = 0; q < 4; q++) if ((cand >> q) & 1u) { kpos[rk] = (uint16_t)(o + q); rk++; }
        nk += tot;
      }
      if (fastk) {
        if ((uint32_t)lane == 0) kpos[nk] = (uint16_t)klen_sec;
        mf_wave_lds_sync();
        // the candidates are the chain iff the first is 0 and every one's successor is itself + 1 + its length
        bool okc = nk == 0 ? klen_sec == 0 : kpos[0] == 0;
        for (uint32_t k = (uint32_t)lane; k < nk; k += 64) {
          const uint32_t a = kpos[k], nx = kpos[k + 1];
          okc &= a + 1u + (uint32_t)data[k0 + a] == nx;
        }
        if (lmw::ballot(!okc)) fastk = false;
        mf_wave_lds_sync();
      }
      if (!fastk && !bail) {   // long keys / control characters: the general chain
        nk = mf_chain<true>(data, k0, k0 + klen_sec, MF_KMAX, 0, nullptr, kpos, lane);
        if (nk == NONE) { nk = 0; bail = true; }
      }
    }
    MF_PH(1);
    if (f.stop_after == 1) continue;
    // ---- the op columns
    MfCol c_ci, c_vt, c_len;   // container index, value type, len (the key index column is read as a whole, below)
    c_ci.p = c_ci.end = c_vt.p = c_vt.end = c_len.p = c_len.end = 0; c_ci.rem = c_vt.rem = c_len.rem = 0; c_ci.uniform = c_vt.uniform = c_len.uniform = false;
    c_ci.runv = c_vt.runv = c_len.runv = 0; c_ci.acc = c_vt.acc = c_len.acc = 0; c_ci.mode = 2; c_vt.mode = 0; c_len.mode = 1; c_ci.run = c_vt.run = c_len.run = false; c_ci.uval = c_vt.uval = c_len.uval = 0;
    const uint64_t vsec = base + bdp->sec_rel[SEC_VALUES], vend = vsec + bdp->sec_len[SEC_VALUES];
    if (!bail) {
      // the section is `1, 4, (uleb len, bytes) x 4`: the lengths sit in front of each column, so the header is walked column by column —
      // through a register window (the first one covers the container index column and the head of the key index column; the value
      // type / len columns behind the key indices get a window of their own)
      const uint64_t os = base + bdp->sec_rel[SEC_OPS], oe = os + bdp->sec_len[SEC_OPS];
      MfWin w = mf_win(data, os, oe, lane);
      MfRd o{os, oe, false};
      uint64_t outer = mw_uleb(data, w, o), ncols = mw_uleb(data, w, o);
      if (outer != 1 || ncols != 4) bail = true;
      uint64_t pp = 0, pe = 0;
      MfWin wprop = w;
      // (three named columns, the loop unrolled by hand: an indexed private array lives in scratch memory — a round trip per field)
      auto open_col = [&](MfCol& c, uint32_t mode) {
        if (bail) return;
        if (o.p + 12 > w.w0 + 256) w = mf_win(data, o.p, oe, lane);
        uint64_t n = mw_uleb(data, w, o);
        if (o.bad || n > o.end - o.p) { bail = true; return; }
        c.p = o.p; c.end = o.p + n;
        c.rem = 0; c.runv = 0; c.acc = 0; c.run = false; c.uniform = false; c.uval = 0; c.mode = mode;
        if (c.p + 24 > w.w0 + 256 && n) w = mf_win(data, c.p, oe, lane);
        if (!mf_col_open(data, w, c, n_ops)) bail = true;
        o.p += n;
      };
      open_col(c_ci, 2u);
      if (!bail) {
        if (o.p + 12 > w.w0 + 256) w = mf_win(data, o.p, oe, lane);
        uint64_t n = mw_uleb(data, w, o);
        if (o.bad || n > o.end - o.p) bail = true;
        else { pp = o.p; pe = o.p + n; wprop = w; o.p += n; }
      }
      open_col(c_vt, 0u);
      open_col(c_len, 1u);
      if (o.bad || o.p != o.end || bdp->sec_len[SEC_VALUES] > 0xfff0u) bail = true;
      (void)wprop;
      MF_PH(2);
      if (!bail && !mf_prop_all(data, pp, pe, n_ops, nk, a_prop, (uint8_t*)a_voff, MF_RMAX * 2 - 8, (uint8_t*)tmp, 3 * 64 * 4, lane MF_PH_PASS)) bail = true;
    }
    if (f.stop_after == 2) continue;
    // ---- values: integers by parity when every row is a set; the general chain otherwise
    if (!bail) {
      const bool all_sets = c_vt.uniform && (c_vt.uval & 0x7f) == 11;
      bool done = all_sets && mf_values_int_all(data, vsec, vend, n_ops, a_voff, lane);
      if (!done) {
        // which rows are deletes (no payload)
        for (uint32_t i = (uint32_t)lane; i < MF_RMAX / 32; i += 64) dels[i] = 0;
        mf_wave_lds_sync();
        if (!c_vt.uniform) {
          MfCol c = c_vt;
          for (uint32_t r0 = 0; r0 < n_ops && !bail; r0 += 64) {
            const uint32_t want = n_ops - r0 < 64 ? n_ops - r0 : 64u;
            int32_t vt = 0;
            if (!mf_fetch(data, c, want, tmp + 64, lane, vt)) { bail = true; break; }
            const uint64_t dm = lmw::ballot((uint32_t)lane < want && (vt & 0x7f) == 8);
            if (lane == 0) { dels[(r0 >> 5)] = (uint32_t)dm; dels[(r0 >> 5) + 1] = (uint32_t)(dm >> 32); }
          }
        } else if ((c_vt.uval & 0x7f) == 8) {
          for (uint32_t i = (uint32_t)lane; i < MF_RMAX / 32; i += 64) dels[i] = 0xffffffffu;
        }
        mf_wave_lds_sync();
        if (!bail && mf_chain<false>(data, vsec, vend, n_ops, n_ops, dels, a_voff, lane) == NONE) bail = true;
      }
    }
    MF_PH(8);
    if (f.stop_after == 3) continue;
    // the block's changes, one per lane (a block of more than 64 changes reads them where it meets them)
    uint32_t L_ctr = 0, L_len = 0, L_peer = 0, L_flag = 0, L_skip = 0, L_lam = 0;
    if (!bail && (uint32_t)lane < N && (uint32_t)lane < 64u) {
      const ChangeRow c = d.chg[chg0 + lane];
      L_ctr = c.ctr; L_len = c.len; L_peer = c.peer;
      L_flag = d.chg_flag[chg0 + lane]; L_skip = d.chg_skip[chg0 + lane]; L_lam = d.chg_lamport[chg0 + lane];
    }
    const uint32_t L_cid = !bail && (uint32_t)lane < n_cids ? d.cid_map[cidr0 + lane] : 0u;   // the block's containers, one per lane (k_block_kind: at most 32)
    // ---- rows, 64 at a time
    uint32_t cj = 0;                                   // change of the chunk's first row (rows and changes advance together)
    ChangeRow ch;
    ch.peer = 0; ch.ctr = 0; ch.len = 0; ch.dep0 = ch.n_dep = ch.op0 = ch.n_op = ch.blk = 0;
    uint32_t ch_flag = 0, ch_skip = 0, ch_lam = 0;
    bool have_ch = false;
    for (uint32_t r0 = 0; r0 < n_ops && !bail; r0 += 64) {
      const uint32_t want = n_ops - r0 < 64 ? n_ops - r0 : 64u;
      int32_t v_ci = 0, v_vt = 0, v_len = 0;
      if (!mf_fetch(data, c_ci, want, tmp, lane, v_ci) || !mf_fetch(data, c_vt, want, tmp + 64, lane, v_vt) || !mf_fetch(data, c_len, want, tmp + 128, lane, v_len)) { bail = true; break; }
      const bool act = (uint32_t)lane < want;
      v_vt &= 0x7f;
      if (lmw::any(act && (v_len != 1 || (v_vt != 11 && v_vt != 8) || v_ci < 0 || (uint32_t)v_ci >= n_cids))) { bail = true; break; }
      const uint32_t v_prop = act ? a_prop[r0 + lane] : 0u;
      const uint64_t val_at = vsec + (act ? a_voff[r0 + lane] : 0u);
      const uint32_t cidx_row = lmw::shfl(L_cid, v_ci & 63);
      // the row's key: length prefix + first eight bytes in ONE round trip (three words from its start), requested here so that the
      // change walk below runs beside it
      const uint32_t ka = act ? kpos[v_prop] : 0u;
      const uint64_t kend = k0 + klen_sec;
      const uint32_t kw0 = act ? mf_ld4(data, k0 + ka, kend, 0u) : 0u, kw1 = act ? mf_ld4(data, k0 + ka + 4, kend, 0u) : 0u, kw2 = act ? mf_ld4(data, k0 + ka + 8, kend, 0u) : 0u;
      // rows of this chunk, change by change (wave-uniform walk over the block's changes; a chunk meets one or two)
      const uint32_t ctr = cs + r0 + (uint32_t)lane;
      uint32_t done_to = r0;   // rows below are assigned
      while (done_to < r0 + want && !bail) {
        if (!have_ch) {
          if (cj >= N) { bail = true; break; }
          if (cj < 64) {
            ch.ctr = lmw::bcast(L_ctr, (int)cj); ch.len = lmw::bcast(L_len, (int)cj); ch.peer = lmw::bcast(L_peer, (int)cj);
            ch_flag = lmw::bcast(L_flag, (int)cj); ch_skip = lmw::bcast(L_skip, (int)cj); ch_lam = lmw::bcast(L_lam, (int)cj);
          } else {
            ch = d.chg[chg0 + cj];
            ch_flag = d.chg_flag[chg0 + cj]; ch_skip = d.chg_skip[chg0 + cj]; ch_lam = d.chg_lamport[chg0 + cj];
          }
          have_ch = true;
          if (ch.len == 0 || ch.ctr != cs + done_to) { bail = true; break; }   // (a zero-length change / rows that do not start where the change does)
        }
        const uint32_t c_end = ch.ctr + ch.len;                       // first id behind the change
        const uint32_t upto = c_end - cs < r0 + want ? c_end - cs : r0 + want;
        const bool in = act && r0 + (uint32_t)lane >= done_to && r0 + (uint32_t)lane < upto;
        if (ch_flag && ch.peer < MAX_PEERS) {
          // (k_map_lww_doc's filters: the known prefix of a sliced change, a write beyond the rendered version)
          const uint32_t cidx = in ? cidx_row : 0u;
          bool go = in && ctr >= ch.ctr + ch_skip;
          if (lmw::any(go && cidx >= MAX_CONTAINERS)) { s_misc[4] = 1; go = false; }
          if (go) lmw::atomic_or(&s_touch[cidx >> 5], 1u << (cidx & 31));
          go = go && ctr < s_end[ch.peer] && f.stop_after != 4;
          MF_PH(9);
          if (go) {
            const uint32_t b0 = kw0 & 0xffu;
            const uint32_t hdr = (b0 & 0x80u) ? 2u : 1u;
            const uint32_t kl = (b0 & 0x80u) ? ((b0 & 0x7fu) | (((kw0 >> 8) & 0xffu) << 7)) : b0;   // (a key section is below 64 KB: two length bytes at most)
            const uint64_t kat = k0 + ka + hdr;
            unsigned long long pf;
            {
              // the first eight key bytes, big endian, zero padded
              const unsigned long long lo = ((unsigned long long)kw1 << 32) | kw0;
              unsigned long long le = hdr == 1 ? (lo >> 8) | ((unsigned long long)kw2 << 56) : (lo >> 16) | ((unsigned long long)kw2 << 48);
              if (kl < 8) le &= (1ull << (8 * kl)) - 1ull;
              pf = __builtin_bswap64(le);
            }
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
                    r.prop = (int32_t)v_prop; r.len = 1; r.ctr = ctr; r.a0 = f.key0[doc] + slot; r.a1 = 0; r.a2 = 0; r.chg = chg0 + cj;
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
          MF_PH(10);
        } else if (ch_flag) { s_misc[4] = 1; }
        done_to = upto;
        if (upto == c_end - cs) { cj++; have_ch = false; }
      }
    }
    if (!bail) {
      // everything is used up exactly: the columns and the changes (the key index column and the values checked their ends themselves)
      if (!mf_col_done(c_ci) || !mf_col_done(c_vt) || !mf_col_done(c_len) || cj != N || have_ch) bail = true;
    }
    if (bail && lane == 0) s_misc[1] = 1;
    mf_wave_lds_sync();
  }
#ifdef LM_PROF_MF
  if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd((unsigned long long*)&d.prof[(uint64_t)doc * 16 + i], (unsigned long long)mfp[i]);
#endif
  lmw::block_sync();
  // ---- results (k_map_lww_doc's): containers, flags, the claimed slots slot for slot into the document's global table
  if (s_misc[1] || s_misc[4]) {   // (an index beyond the tables' limits too: whatever is behind it, the row tables' verdict)
    if (tid == 0) { lmw::atomic_or(&d.doc[doc].flags, DF_REDO); LM_SETERR(d.doc[doc].status, ST_DATA_CORRUPTION); d.ht_cnt[doc] = 0; }
    return;
  }
  for (uint32_t c = tid; c < m.n_cont && c < MAX_CONTAINERS; c += MF_WG)
    if ((s_touch[c >> 5] >> (c & 31)) & 1) d.cont[m.cid0 + c].touched = 1;
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
