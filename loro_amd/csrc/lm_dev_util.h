// Device helpers: bounded byte readers, varints, serde_columnar cursors, xxHash32.
// Wire grammar: /root/reference/docs/encoding.md §1, §8.1; docs/encoding-xxhash32.md.
#pragma once
#include "lm_wave.h"
#include "lm_types.h"

namespace lm {

struct Rd {                 // bounded reader over global memory; `bad` latches on any overrun
  const uint8_t* p;
  const uint8_t* end;
  bool bad;
};
LM_DEV Rd rd_make(const uint8_t* p, uint64_t n) { Rd r; r.p = p; r.end = p + n; r.bad = false; return r; }
template <class R> LM_DEV uint64_t rd_left(const R& r) { return (uint64_t)(r.end - r.p); }
// byte loads: consecutive reads of a lane stay inside one cache line, so they are L1 hits; buffering eight
// bytes in registers was measured slower (more VALU per byte than the load it saves).
LM_DEV uint32_t rd_u8(Rd& r) {
  if (r.p >= r.end) { r.bad = true; return 0; }
  return *r.p++;
}
// The value walker of k_block_decode_wave reads through a WINDOW: the next DEC_VW bytes behind its cursor, fetched by the
// block's eight lanes with one coalesced load and kept in LDS — a byte inside it is an LDS read, anything else the plain load.
// (Every byte used to be a dependent HBM / L2 round trip of one lane: 31 % of the decoder's time on text blocks, 62 % on blocks
// of Map values — profiles/r03_decoder_phases.log.)  Same fields and the same `bad` latch as Rd, so the templated readers
// below serve both.
static constexpr uint32_t DEC_VW = 64;    // (LDS decides the decoder's occupancy: 15,616 bytes per wave are 10 waves per CU, 14,848 are 11 — and 10 % of its time)
struct RdW {
  const uint8_t* p;
  const uint8_t* end;
  bool bad;
  const uint8_t* wp;      // global address of the window's first byte
  lm_lds_bytes wl;        // the window in LDS
};
LM_DEV uint32_t rd_u8(RdW& r) {
  if (r.p >= r.end) { r.bad = true; return 0; }
  uint64_t o = (uint64_t)(r.p - r.wp);
  uint32_t b = o < DEC_VW ? r.wl[o] : *r.p;
  r.p++;
  return b;
}
LM_DEV uint32_t rd_peek(const Rd& r) { return *r.p; }   // r.p < r.end
LM_DEV uint32_t rd_peek(const RdW& r) { uint64_t o = (uint64_t)(r.p - r.wp); return o < DEC_VW ? r.wl[o] : *r.p; }   // r.p < r.end
template <class R> LM_DEV uint64_t rd_uleb(R& r) {
  uint64_t v = 0;
  for (int i = 0; i < 10; i++) {
    uint32_t b = rd_u8(r);
    v |= (uint64_t)(b & 0x7f) << (7 * i);
    if (!(b & 0x80)) return v;
  }
  r.bad = true;
  return v;
}
LM_DEV int64_t rd_zigzag(Rd& r) {
  uint64_t v = rd_uleb(r);
  return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
}
// i128 zigzag varint (DeltaRle deltas); values beyond i64 are flagged bad
LM_DEV int64_t rd_zigzag128(Rd& r) {
  uint64_t lo = 0, hi = 0;
  for (int i = 0; i < 19; i++) {
    uint32_t b = rd_u8(r);
    uint64_t part = b & 0x7f;
    int sh = 7 * i;
    if (sh < 64) { lo |= part << sh; if (sh > 57) hi |= part >> (64 - sh); }
    else hi |= part << (sh - 64);
    if (!(b & 0x80)) {
      // value = (hi:lo); zigzag decode needs it to fit in 65 bits
      if (hi > 1) { r.bad = true; return 0; }
      uint64_t mag = (lo >> 1) | (hi << 63);
      return (int64_t)mag ^ -(int64_t)(lo & 1);
    }
  }
  r.bad = true;
  return 0;
}
template <class R> LM_DEV int64_t rd_sleb(R& r) {
  int64_t result = 0;
  int shift = 0;
  uint32_t b;
  do {
    b = rd_u8(r);
    if (shift < 64) result |= (int64_t)((uint64_t)(b & 0x7f) << shift);
    shift += 7;
    if (shift > 70) { r.bad = true; return 0; }
  } while ((b & 0x80) && !r.bad);
  if (shift < 64 && (b & 0x40)) result |= -((int64_t)1 << shift);
  return result;
}
template <class R> LM_DEV void rd_skip(R& r, uint64_t n) {
  if (n > rd_left(r)) { r.bad = true; r.p = r.end; return; }
  r.p += n;
}
// uleb length + bytes → sub-reader
LM_DEV Rd rd_bytes(Rd& r) {
  uint64_t n = rd_uleb(r);
  if (n > rd_left(r)) { r.bad = true; n = rd_left(r); }
  Rd s = rd_make(r.p, n);
  s.bad = r.bad;
  r.p += n;
  return s;
}

// AnyRle cursor: segments {zigzag k; k>0: one value ×k | k<0: |k| literals}
struct RleCur {
  Rd r;
  int64_t rem;
  bool run;
  int64_t val;   // run value, or running sum for delta columns
  int64_t runv;  // run delta for delta columns
};
LM_DEV RleCur rle_make(Rd r) { RleCur c; c.r = r; c.rem = 0; c.run = false; c.val = 0; c.runv = 0; return c; }
LM_DEV bool rle_head(RleCur& c) {  // start next segment; false at end / error
  if (c.r.p >= c.r.end) { c.r.bad = true; return false; }
  int64_t k = rd_zigzag(c.r);
  if (k == 0) { c.r.bad = true; return false; }
  c.run = k > 0;
  c.rem = k > 0 ? k : -k;
  return true;
}
LM_DEV uint64_t rle_next_uvar(RleCur& c) {
  if (c.rem == 0) { if (!rle_head(c)) return 0; if (c.run) c.val = (int64_t)rd_uleb(c.r); }
  c.rem--;
  return c.run ? (uint64_t)c.val : rd_uleb(c.r);
}
// the (wrapping) sum of the next n values of a uvar column: a run is k x its value
LM_DEV uint64_t rle_sum_uvar(RleCur& c, uint64_t n) {
  uint64_t tot = 0;
  while (n) {
    if (c.rem > 0 && c.run) {
      const uint64_t k = (uint64_t)c.rem < n ? (uint64_t)c.rem : n;
      tot += k * (uint64_t)c.val; c.rem -= (int64_t)k; n -= k;
    } else { tot += rle_next_uvar(c); n--; }
  }
  return tot;
}
LM_DEV uint32_t rle_next_u8(RleCur& c) {
  if (c.rem == 0) { if (!rle_head(c)) return 0; if (c.run) c.val = (int64_t)rd_u8(c.r); }
  c.rem--;
  return c.run ? (uint32_t)c.val : rd_u8(c.r);
}
LM_DEV int64_t rle_next_delta(RleCur& c) {  // DeltaRle: running sum of i128 deltas
  if (c.rem == 0) { if (!rle_head(c)) return c.val; if (c.run) c.runv = rd_zigzag128(c.r); }
  c.rem--;
  if (__builtin_add_overflow(c.val, c.run ? c.runv : rd_zigzag128(c.r), &c.val)) c.r.bad = true;   // the reference sums in i128 and fails the narrowing
  return c.val;
}
// The three cursors above as ONE code path selected by a per-lane `mode` (0 = Rle<u8>, 1 = Rle<uvar>, 2 = DeltaRle), so lanes
// that own different columns do not diverge (k_block_decode_wave).  Same results and the same `bad` latching as
// rle_next_u8 / rle_next_uvar / rle_next_delta.
LM_DEV int64_t rd_any(Rd& r, uint32_t mode) {
  uint64_t lo = 0, hi = 0;
  uint32_t lim = mode == 0 ? 1u : (mode == 1 ? 10u : 19u);
  bool term = false;
  for (uint32_t i = 0; i < lim; i++) {
    uint32_t b = rd_u8(r);
    uint64_t part = mode == 0 ? b : (b & 0x7f);
    uint32_t sh = 7 * i;
    if (sh < 64) { lo |= part << sh; if (sh > 57) hi |= part >> (64 - sh); }
    else hi |= part << (sh - 64);
    if (mode == 0 || !(b & 0x80)) { term = true; break; }
  }
  if (!term) { r.bad = true; return mode == 1 ? (int64_t)lo : 0; }
  if (mode != 2) return (int64_t)lo;
  if (hi > 1) { r.bad = true; return 0; }   // a delta beyond i64
  uint64_t mag = (lo >> 1) | (hi << 63);
  return (int64_t)mag ^ -(int64_t)(lo & 1);
}
LM_DEV int64_t rle_next_any(RleCur& c, uint32_t mode) {   // the run value lives in runv for every mode; val is the delta sum
  bool need = !c.run;
  if (c.rem == 0) { if (!rle_head(c)) return mode == 2 ? c.val : 0; need = true; }
  c.rem--;
  if (need) c.runv = rd_any(c.r, mode);
  if (mode != 2) return c.runv;
  if (__builtin_add_overflow(c.val, c.runv, &c.val)) c.r.bad = true;   // the reference sums in i128 and fails the narrowing
  return c.val;
}
// The reference decodes every column of a block IN FULL before it looks at a row (serde_columnar: each column into its own vector,
// block_encode.rs:587-595, outdated_encode_reordered.rs:480-489), so a column that does not decode, or that holds more or fewer
// values than the rows need, fails the block with DecodeError whatever the rows say.  The decoders read columns row by row; these
// two give them the same verdict afterwards: is the column used up exactly, and how many values does the rest hold.
LM_DEV bool rle_exhausted(const RleCur& c) { return c.rem == 0 && c.r.p >= c.r.end; }
// A RUN is counted, not stepped through (a crafted column of a few bytes with a run count near 2^28 made one lane spin
// 2^28 times per column): the rest of a run adds `rem` values at once — for a delta column the sum after the run must still fit
// i64, and the sums inside a run are monotonic, so the end of the run decides — literals are decoded one by one (each consumes
// at least a byte of the column), and once the count passes `cap` the verdict "more values than the rows need" stands.
LM_DEV bool drain_run(int64_t& rem, int64_t& val, int64_t runv, uint32_t mode, uint32_t& n, uint32_t cap) {   // false: the sum overflows
  bool ok = true;
  if (mode == 2) { int64_t d; ok = !__builtin_mul_overflow(runv, rem, &d) && !__builtin_add_overflow(val, d, &val); }
  uint64_t tot = (uint64_t)n + (uint64_t)rem;
  n = tot > (uint64_t)cap + 1 ? cap + 1 : (uint32_t)tot;
  rem = 0;
  return ok;
}
LM_DEV uint32_t rle_drain(RleCur& c, uint32_t mode, uint32_t cap = (1u << 28)) {   // values left in the column, saturating at cap + 1 (c.r.bad: the rest does not decode)
  uint32_t n = 0;
  while (!c.r.bad && !rle_exhausted(c) && n <= cap) {
    if (c.rem > 0 && c.run) {   // inside a run: its value was read with the run's first value (rle_next_any), the rest are copies
      if (!drain_run(c.rem, c.val, c.runv, mode, n, cap)) c.r.bad = true;
      continue;
    }
    (void)rle_next_any(c, mode);
    if (!c.r.bad) n++;
  }
  return n;
}
// The same cursor over a column of a block STAGED IN LDS (k_block_decode_wave): byte offsets into the slot instead of 64-bit
// pointers, LDS loads instead of flat ones, and a varint of up to three bytes — nearly all of them — is read with three
// independent loads and no per-byte bounds test, whenever three bytes are left in the column (the generic reader above spends
// ≈12 instructions per byte on the bounds test, the latch and the pointer; 80 % of the decoder's instructions were inside it).
// Anything else — the last bytes of a column, a longer varint — goes through the generic reader on the same bytes, so the
// values, the `bad` latch and the position after every call are those of rle_next_any.
struct ColCur { lm_lds_bytes base; uint32_t p, end; bool bad; int64_t rem; bool run; int64_t val, runv; };
LM_DEV ColCur col_make(lm_lds_bytes base, uint32_t p, uint32_t end, bool bad) {
  ColCur c; c.base = base; c.p = p; c.end = end; c.bad = bad; c.rem = 0; c.run = false; c.val = 0; c.runv = 0; return c;
}
LM_DEV bool col_var3(ColCur& c, uint32_t& v) {   // a uleb128 that ends within three bytes, all inside the column; false: nothing consumed
  if (c.end - c.p < 3) return false;
  uint32_t b0 = c.base[c.p], b1 = c.base[c.p + 1], b2 = c.base[c.p + 2];
  if (b0 & b1 & b2 & 0x80) return false;
  uint32_t c0 = b0 >> 7, c1 = c0 & (b1 >> 7);
  v = (b0 & 0x7f) | (c0 ? (b1 & 0x7f) << 7 : 0u) | (c1 ? (b2 & 0x7f) << 14 : 0u);
  c.p += 1 + c0 + c1;
  return true;
}
LM_DEV int64_t col_any(ColCur& c, uint32_t mode) {   // == rd_any on the column's reader
  uint32_t v;
  if (mode == 0) { if (c.p < c.end) return c.base[c.p++]; }
  else if (col_var3(c, v)) return mode == 1 ? (int64_t)v : (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  Rd r = rd_make((const uint8_t*)c.base + c.p, c.end - c.p);
  int64_t w = rd_any(r, mode);
  c.p = (uint32_t)(r.p - (const uint8_t*)c.base);
  if (r.bad) c.bad = true;
  return w;
}
LM_DEV int64_t col_next_any(ColCur& c, uint32_t mode) {   // == rle_next_any
  bool need = !c.run;
  if (c.rem == 0) {
    // == rle_head
    if (c.p >= c.end) { c.bad = true; return mode == 2 ? c.val : 0; }
    int64_t k;
    uint32_t v;
    if (col_var3(c, v)) k = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
    else {
      Rd r = rd_make((const uint8_t*)c.base + c.p, c.end - c.p);
      k = rd_zigzag(r);
      c.p = (uint32_t)(r.p - (const uint8_t*)c.base);
      if (r.bad) c.bad = true;
    }
    if (k == 0) { c.bad = true; return mode == 2 ? c.val : 0; }
    c.run = k > 0;
    c.rem = k > 0 ? k : -k;
    need = true;
  }
  c.rem--;
  if (need) c.runv = col_any(c, mode);
  if (mode != 2) return c.runv;
  if (__builtin_add_overflow(c.val, c.runv, &c.val)) c.bad = true;   // the reference sums in i128 and fails the narrowing
  return c.val;
}
LM_DEV bool col_exhausted(const ColCur& c) { return c.rem == 0 && c.p >= c.end; }
LM_DEV uint32_t col_drain(ColCur& c, uint32_t mode, uint32_t cap = (1u << 28)) {   // == rle_drain
  uint32_t n = 0;
  while (!c.bad && !col_exhausted(c) && n <= cap) {
    if (c.rem > 0 && c.run) {
      if (!drain_run(c.rem, c.val, c.runv, mode, n, cap)) c.bad = true;
      continue;
    }
    (void)col_next_any(c, mode);
    if (!c.bad) n++;
  }
  return n;
}
// number of values in a whole AnyRle payload whose literals are single bytes (Rle<u8>)
LM_DEV uint64_t rle_count_u8(Rd r) {
  uint64_t n = 0;
  while (r.p < r.end && !r.bad) {
    int64_t k = rd_zigzag(r);
    if (k == 0) { r.bad = true; break; }
    if (k > 0) { (void)rd_u8(r); n += (uint64_t)k; }
    else { rd_skip(r, (uint64_t)(-k)); n += (uint64_t)(-k); }
    if (n > (1u << 30)) { r.bad = true; break; }
  }
  return r.bad ? ~0ull : n;
}
// number of values in an AnyRle payload with varint values (uleb or zigzag128 share the skip rule)
LM_DEV uint64_t rle_count_var(Rd r) {
  uint64_t n = 0;
  while (r.p < r.end && !r.bad) {
    int64_t k = rd_zigzag(r);
    if (k == 0) { r.bad = true; break; }
    uint64_t m = k > 0 ? 1 : (uint64_t)(-k);
    for (uint64_t i = 0; i < m && !r.bad; i++) {
      for (int j = 0; j < 19; j++) { uint32_t b = rd_u8(r); if (!(b & 0x80)) break; }
    }
    n += k > 0 ? (uint64_t)k : (uint64_t)(-k);
    if (n > (1u << 30)) { r.bad = true; break; }
  }
  return r.bad ? ~0ull : n;
}

// BoolRle cursor: alternating run lengths, first run is FALSE
struct BoolCur { Rd r; uint64_t rem; bool cur; bool started; };
LM_DEV BoolCur bool_make(Rd r) { BoolCur c; c.r = r; c.rem = 0; c.cur = true; c.started = false; return c; }
LM_DEV bool bool_next(BoolCur& c) {
  int guard = 0;
  while (c.rem == 0) {
    c.cur = !c.cur;
    c.rem = rd_uleb(c.r);
    if (c.r.bad || ++guard > 4) { c.r.bad = true; return false; }
  }
  c.rem--;
  return c.cur;
}

// n x bool_next with the values dropped: whole runs at a time
LM_DEV void bool_skip(BoolCur& c, uint64_t n) {
  while (n) {
    if (c.rem == 0) { (void)bool_next(c); n--; continue; }
    const uint64_t k = c.rem < n ? c.rem : n;
    c.rem -= k; n -= k;
  }
}

// n x bool_next: how many of the values are true
LM_DEV uint64_t bool_count(BoolCur& c, uint64_t n) {
  uint64_t t = 0;
  while (n) {
    if (c.rem == 0) { t += bool_next(c) ? 1u : 0u; n--; continue; }
    const uint64_t k = c.rem < n ? c.rem : n;
    if (c.cur) t += k;
    c.rem -= k; n -= k;
  }
  return t;
}

// DeltaOfDelta cursor (docs/encoding.md:713-726)
struct DodCur { const uint8_t* p; uint64_t nbits, pos; int64_t prev, delta; bool has_first, first_taken, bad; uint32_t last_used; };
LM_DEV uint64_t dod_bits(DodCur& c, int n) {
  if (c.pos + (uint64_t)n > c.nbits) { c.bad = true; return 0; }
  uint64_t v = 0;
  for (int i = 0; i < n; i++) {
    v = (v << 1) | ((c.p[c.pos >> 3] >> (7 - (c.pos & 7))) & 1);
    c.pos++;
  }
  return v;
}
LM_DEV DodCur dod_make(Rd& r) {  // consumes the head (Option<i64> + last-used byte); bitstream follows at r.p
  DodCur c;
  c.bad = false; c.pos = 0; c.delta = 0; c.prev = 0; c.first_taken = false;
  uint32_t tag = rd_u8(r);
  c.has_first = tag == 1;
  if (tag > 1) c.bad = true;
  if (c.has_first) c.prev = rd_zigzag(r);
  c.last_used = rd_u8(r);
  c.p = r.p;
  c.nbits = rd_left(r) * 8;
  if (r.bad) c.bad = true;
  return c;
}
LM_DEV int64_t dod_next(DodCur& c) {
  if (!c.has_first) { c.bad = true; return 0; }
  if (!c.first_taken) { c.first_taken = true; return c.prev; }
  int64_t dd;
  if (dod_bits(c, 1) == 0) dd = 0;
  else if (dod_bits(c, 1) == 0) dd = (int64_t)dod_bits(c, 7) - 63;
  else if (dod_bits(c, 1) == 0) dd = (int64_t)dod_bits(c, 9) - 255;
  else if (dod_bits(c, 1) == 0) dd = (int64_t)dod_bits(c, 12) - 2047;
  else if (dod_bits(c, 1) == 0) dd = (int64_t)dod_bits(c, 21) - 1048575;
  else dd = (int64_t)dod_bits(c, 64);
  c.delta = (int64_t)((uint64_t)c.delta + (uint64_t)dd);      // wrapping, like the release-mode Rust reader (damaged input only)
  c.prev = (int64_t)((uint64_t)c.prev + (uint64_t)c.delta);
  return c.prev;
}
// n x dod_next with the values dropped (columns read for their shape only): a byte of zero bits is eight values with a
// delta-of-delta of 0 — constant strides, the common case (lamports of one-op changes, equal timestamps) — taken in one step
LM_DEV void dod_skip(DodCur& c, uint64_t n) {
  if (n == 0) return;
  if (!c.has_first) { c.bad = true; return; }
  if (!c.first_taken) { c.first_taken = true; n--; }
  while (n) {
    if (n >= 8 && (c.pos & 7) == 0 && c.pos + 8 <= c.nbits && c.p[c.pos >> 3] == 0) {
      c.pos += 8; n -= 8;
      c.prev = (int64_t)((uint64_t)c.prev + 8ull * (uint64_t)c.delta);
      continue;
    }
    (void)dod_next(c); n--;
  }
}
// after taking `n` values: validate the used-bits byte and advance the byte reader past the stream
LM_DEV void dod_finish(DodCur& c, Rd& r, uint64_t n) {
  if (c.bad) r.bad = true;   // (an option tag other than 0 / 1 in front of an EMPTY stream too: the early return below used to skip it — a
                             // block with no foreign dependency and a damaged tag byte was accepted where the reference fails, found on damaged resident sessions)
  if (!c.has_first) { if (n != 0 || c.last_used != 0) r.bad = true; return; }
  if (n == 0) { r.bad = true; return; }
  if (n == 1) { if (c.last_used != 0) r.bad = true; }
  else { uint32_t e = (uint32_t)(c.pos % 8 ? c.pos % 8 : 8); if (c.last_used != e) r.bad = true; }
  if (c.bad) r.bad = true;
  rd_skip(r, (c.pos + 7) / 8);
}

// unaligned 32-bit access (gfx950 global memory takes it in one instruction)
LM_DEV uint32_t ld32u(const uint8_t* p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }
LM_DEV void st32u(uint8_t* p, uint32_t w) { __builtin_memcpy(p, &w, 4); }

LM_DEV uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
LM_DEV uint32_t ld32le(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
// xxHash32, one lane (docs/encoding-xxhash32.md).  `p` must be 4-byte aligned.
LM_DEV uint32_t xxh32_lane(const uint8_t* p, uint64_t len, uint32_t seed) {
  const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
  const uint8_t* end = p + len;
  uint32_t h;
  if (len >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* limit = end - 16;
    const uint32_t* w = (const uint32_t*)p;
    // 8 stripes (128 bytes) per step: all loads are issued before the first multiply, so one memory latency is paid
    // per 128 bytes instead of per 16 (the accumulator chain itself is serial by construction of the hash)
    while ((const uint8_t*)(w + 32) <= end) {
      uint32_t x[32];
#pragma unroll
      for (int i = 0; i < 32; i++) x[i] = w[i];
#pragma unroll
      for (int i = 0; i < 32; i += 4) {
        v1 = rotl32(v1 + x[i] * P2, 13) * P1;
        v2 = rotl32(v2 + x[i + 1] * P2, 13) * P1;
        v3 = rotl32(v3 + x[i + 2] * P2, 13) * P1;
        v4 = rotl32(v4 + x[i + 3] * P2, 13) * P1;
      }
      w += 32;
    }
    if ((const uint8_t*)w <= limit) do {
      uint32_t a = w[0], b = w[1], c = w[2], d = w[3];
      v1 = rotl32(v1 + a * P2, 13) * P1;
      v2 = rotl32(v2 + b * P2, 13) * P1;
      v3 = rotl32(v3 + c * P2, 13) * P1;
      v4 = rotl32(v4 + d * P2, 13) * P1;
      w += 4;
    } while ((const uint8_t*)w <= limit);
    p = (const uint8_t*)w;
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  while (p + 4 <= end) { h = rotl32(h + ld32le(p) * P3, 17) * P4; p += 4; }
  while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}

// xxHash32 of one LARGE buffer by a whole wave: 1 KB (64 stripes) per step is loaded with coalesced dword loads, then lanes
// 0..3 — the four accumulators — consume their words stripe by stripe through lane permutes; the rest (< 1 KB) and the
// finalisation are done redundantly by every lane.  (One lane per blob streams a 2.4 MB blob at one 16-byte load per
// instruction per lane: 12 ms for a configs[2] document; this takes ≈2.)  `p` must be 4-byte aligned; wave-uniform arguments.
LM_DEV uint32_t xxh32_wave(const uint8_t* p, uint64_t len, uint32_t seed) {
  const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
  if (len < 1024) return xxh32_lane(p, len, seed);
  int lane = lmw::lane();
  const uint8_t* end = p + len;
  const uint32_t* w = (const uint32_t*)p;
  uint32_t v = lane == 0 ? seed + P1 + P2 : lane == 1 ? seed + P2 : lane == 2 ? seed : seed - P1;   // lanes >= 4 compute along, unused
  int src0 = lane & 3;
  while ((const uint8_t*)(w + 256) <= end) {
    uint32_t x0 = w[lane], x1 = w[64 + lane], x2 = w[128 + lane], x3 = w[192 + lane];
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) v = rotl32(v + lmw::shfl(x0, (s_ * 4 + src0) & 63) * P2, 13) * P1;
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) v = rotl32(v + lmw::shfl(x1, (s_ * 4 + src0) & 63) * P2, 13) * P1;
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) v = rotl32(v + lmw::shfl(x2, (s_ * 4 + src0) & 63) * P2, 13) * P1;
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) v = rotl32(v + lmw::shfl(x3, (s_ * 4 + src0) & 63) * P2, 13) * P1;
    w += 256;
  }
  uint32_t v1 = lmw::bcast(v, 0), v2 = lmw::bcast(v, 1), v3 = lmw::bcast(v, 2), v4 = lmw::bcast(v, 3);
  const uint8_t* q = (const uint8_t*)w;
  while (q + 16 <= end) {
    const uint32_t* u = (const uint32_t*)q;
    v1 = rotl32(v1 + u[0] * P2, 13) * P1; v2 = rotl32(v2 + u[1] * P2, 13) * P1;
    v3 = rotl32(v3 + u[2] * P2, 13) * P1; v4 = rotl32(v4 + u[3] * P2, 13) * P1;
    q += 16;
  }
  uint32_t h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  h += (uint32_t)len;
  while (q + 4 <= end) { h = rotl32(h + ld32le(q) * P3, 17) * P4; q += 4; }
  while (q < end) { h = rotl32(h + (*q) * P5, 11) * P1; q++; }
  h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
  return h;
}

// packed element id helpers
LM_DEV uint32_t pid_make(uint32_t peer, uint32_t ctr) { return (peer << 24) | ctr; }
LM_DEV uint32_t pid_peer(uint32_t pid) { return pid >> 24; }
LM_DEV uint32_t pid_ctr(uint32_t pid) { return pid & 0xFFFFFFu; }

}  // namespace lm
