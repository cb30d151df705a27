// f64 → JSON number text, shortest digits that round-trip, laid out like serde_json / ryu's "pretty" printer
// (what `LoroValue::to_json_value` produces for LoroValue::Double; reference: crates/loro-common/src/value.rs:719-738).
// Digit generation is the free-format algorithm of Burger & Dybvig ("Printing Floating-Point Numbers Quickly and
// Accurately", PLDI'96, figure 3: scale by an estimate of log10, fix up, generate) on a small fixed-width bignum:
// integer arithmetic only, so the host build and the gfx950 build agree bit for bit.  f64 values are rare in the
// documents this engine merges; the routine favours being obviously exact over being fast.
#pragma once
#include <cstdint>
#include "lm_wave.h"

namespace lm {

struct Big {                      // little-endian magnitude, 40 × 32 bits (≥ 2^55 · 2^1077 · small slack)
  static constexpr int N = 40;
  uint32_t w[N];
};
// every routine works on the low `nl` limbs only: nl is chosen per value from its binary and decimal exponents, so an
// everyday double costs 4-5 limbs per operation instead of 40
LM_DEV void big_set(Big& a, uint64_t v, int nl) { for (int i = 0; i < nl; i++) a.w[i] = 0; a.w[0] = (uint32_t)v; a.w[1] = (uint32_t)(v >> 32); }
LM_DEV void big_mul_small(Big& a, uint32_t m, int nl) {
  uint64_t c = 0;
  for (int i = 0; i < nl; i++) { uint64_t t = (uint64_t)a.w[i] * m + c; a.w[i] = (uint32_t)t; c = t >> 32; }
}
LM_DEV void big_shl(Big& a, uint32_t bits, int nl) {
  uint32_t ws = bits >> 5, bs = bits & 31;
  for (int i = nl - 1; i >= 0; i--) {
    uint32_t lo = (i >= (int)ws) ? a.w[i - (int)ws] : 0u;
    uint32_t lo2 = (i >= (int)ws + 1) ? a.w[i - (int)ws - 1] : 0u;
    a.w[i] = bs ? ((lo << bs) | (lo2 >> (32 - bs))) : lo;
  }
}
LM_DEV int big_cmp(const Big& a, const Big& b, int nl) {
  for (int i = nl - 1; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
LM_DEV void big_add(Big& a, const Big& b, int nl) {
  uint64_t c = 0;
  for (int i = 0; i < nl; i++) { uint64_t t = (uint64_t)a.w[i] + b.w[i] + c; a.w[i] = (uint32_t)t; c = t >> 32; }
}
LM_DEV void big_sub(Big& a, const Big& b, int nl) {   // a >= b
  uint64_t br = 0;
  for (int i = 0; i < nl; i++) { uint64_t t = (uint64_t)a.w[i] - b.w[i] - br; a.w[i] = (uint32_t)t; br = (t >> 63) & 1; }
}
LM_DEV void big_copy(Big& a, const Big& b, int nl) { for (int i = 0; i < nl; i++) a.w[i] = b.w[i]; }
LM_DEV void big_mul_pow10(Big& a, uint32_t e, int nl) {
  while (e >= 9) { big_mul_small(a, 1000000000u, nl); e -= 9; }
  static const uint32_t P[9] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u};
  if (e) big_mul_small(a, P[e], nl);
}

// writes at most 32 bytes to `out`, returns the length.  `ws` = workspace of 6 bignums (the device keeps it in LDS and
// lets one lane run the routine: the data is wave-uniform, and 1 KiB of per-lane scratch would cost every wave of the
// emit kernel occupancy for a value that almost never occurs).
LM_DEV int f64_json(uint64_t bits, char* out, Big* ws) {
  int n = 0;
  uint32_t be = (uint32_t)((bits >> 52) & 0x7ff);
  uint64_t frac = bits & 0xFFFFFFFFFFFFFull;
  bool neg = (bits >> 63) != 0;
  if (be == 0x7ff) { out[0] = 'n'; out[1] = 'u'; out[2] = 'l'; out[3] = 'l'; return 4; }   // NaN / inf: serde_json prints null
  if (be == 0 && frac == 0) { if (neg) out[n++] = '-'; out[n++] = '0'; out[n++] = '.'; out[n++] = '0'; return n; }
  uint64_t f = be ? (frac | (1ull << 52)) : frac;
  int e2 = be ? (int)be - 1075 : -1074;
  // integers below 10^16 print as "<n>.0" (the layout rule for k >= #digits); no bignum needed
  if (e2 <= 0 && e2 > -53 && (f & ((1ull << (-e2)) - 1)) == 0) {
    uint64_t iv = f >> (-e2);
    if (iv < 10000000000000000ull) {
      if (neg) out[n++] = '-';
      // (digits least significant first, one per byte of two registers — an indexed private array is scratch memory on the device:
      // a round trip per digit)
      uint64_t t0 = 0, t1 = 0;
      int tn = 0;
      // (two chunks in 32-bit arithmetic: a 64-bit division per digit is a software routine on this target, see sink_i64)
      uint64_t q1 = iv < 1000000000ull ? 0ull : iv / 1000000000ull;
      uint32_t c0 = (uint32_t)(iv - q1 * 1000000000ull), c1 = (uint32_t)q1;
      if (c1) { for (int i = 0; i < 9; i++) { uint64_t c = '0' + c0 % 10u; c0 /= 10u; if (tn < 8) t0 |= c << (8 * tn); else t1 |= c << (8 * (tn - 8)); tn++; } c0 = c1; }
      do { uint64_t c = '0' + c0 % 10u; c0 /= 10u; if (tn < 8) t0 |= c << (8 * tn); else t1 |= c << (8 * (tn - 8)); tn++; } while (c0);
      while (tn) { --tn; out[n++] = (char)(tn < 8 ? t0 >> (8 * tn) : t1 >> (8 * (tn - 8))); }
      out[n++] = '.'; out[n++] = '0';
      return n;
    }
  }
  bool even = (f & 1) == 0;                    // round-to-even: the interval ends belong to v
  bool lower_closer = be > 1 && frac == 0;      // v is a power of two: the gap below is half the gap above
  int bitlen = 64 - __builtin_clzll(f);
  double t = (double)(e2 + bitlen - 1) * 0.30102999566398120;   // log10(2); a lower bound of log10(v)
  int est = (int)t;
  if ((double)est < t - 1e-10) est++;           // ceil(t - 1e-10)
  if (t < 0 && (double)est > t + 1.0) est--;    // (int) truncates toward zero for negative t
  // the digits, most significant first, one per byte of three registers (no indexed private array: that is scratch memory)
  uint64_t dw0 = 0, dw1 = 0, dw2 = 0;
  int nd = 0;
  auto dput = [&](int dv) { uint64_t c = (uint64_t)('0' + dv); if (nd < 8) dw0 |= c << (8 * nd); else if (nd < 16) dw1 |= c << (8 * (nd - 8)); else dw2 |= c << (8 * (nd - 16)); nd++; };
  auto dget = [&](int i) -> char { return (char)(i < 8 ? dw0 >> (8 * i) : (i < 16 ? dw1 >> (8 * (i - 8)) : dw2 >> (8 * (i - 16)))); };
  int k = est;
  if (e2 >= -121 && e2 <= 60) {
    // Everyday magnitudes (≈1.7e-21 … 1e34): the same algorithm, the same numbers — in two 64-bit registers each instead of limb
    // arrays in LDS.  Every quantity stays below 2^127: for e2 >= 0 the largest is 10·r < 2^(53 + e2 + 2 + 4); for e2 < 0, v >= 1 it
    // is 10·s <= 2^60; for v < 1 it is 10·s = 10·2^(2 - e2).  (A List of random doubles spent ≈60k cycles per value in the limb
    // routine — one lane, every limb an LDS round trip; profiles/r03_renderer_phases.log: 97 % of configs[3]'s rendering.)
    typedef unsigned __int128 u128;
    u128 r, s, mp, mm;
    if (e2 >= 0) {
      r = (u128)f << ((uint32_t)e2 + (lower_closer ? 2u : 1u));
      s = lower_closer ? 4 : 2;
      mp = (u128)1 << ((uint32_t)e2 + (lower_closer ? 1u : 0u));
      mm = (u128)1 << (uint32_t)e2;
    } else {
      r = (u128)f << (lower_closer ? 2u : 1u);
      s = (u128)1 << ((uint32_t)(-e2) + (lower_closer ? 2u : 1u));
      mp = lower_closer ? 2 : 1;
      mm = 1;
    }
    {
      uint32_t e = (uint32_t)(est < 0 ? -est : est);
      u128 p10 = 1;
      while (e >= 9) { p10 *= 1000000000u; e -= 9; }
      while (e) { p10 *= 10u; e--; }
      if (est >= 0) s *= p10; else { r *= p10; mp *= p10; mm *= p10; }
    }
    {
      u128 hi = r + mp;
      if (even ? hi >= s : hi > s) k++;          // estimate was one too low
      else { r *= 10u; mp *= 10u; mm *= 10u; }
    }
    for (int guard = 0; guard < 19; guard++) {
      int d = 0;
      // (no 128-bit division on the device: the digit by compare-and-subtract of 8s, 4s, 2s, s)
      u128 s2 = s << 1, s4 = s << 2, s8 = s << 3;
      if (r >= s8) { r -= s8; d += 8; }
      if (r >= s4) { r -= s4; d += 4; }
      if (r >= s2) { r -= s2; d += 2; }
      if (r >= s) { r -= s; d += 1; }
      bool tc1 = even ? r <= mm : r < mm;
      u128 hi = r + mp;
      bool tc2 = even ? hi >= s : hi > s;
      if (!tc1 && !tc2) { dput(d); r *= 10u; mp *= 10u; mm *= 10u; continue; }
      if (tc1 && tc2) { u128 r2 = r << 1; if (r2 > s || (r2 == s && (d & 1))) d++; }   // nearer digit; an exact tie goes to the even digit
      else if (tc2) d++;
      dput(d);
      break;
    }
  } else {
  Big &r = ws[0], &s = ws[1], &mp = ws[2], &mm = ws[3], &hi = ws[4], &r2 = ws[5];
  // limbs: the largest value formed is below 2^(57 + |e2| + 3.33·|est| + 8)
  int aest = est < 0 ? -est : est, ae2 = e2 < 0 ? -e2 : e2;
  int nl = (57 + ae2 + (aest * 3322 + 999) / 1000 + 8 + 31) / 32 + 1;
  if (nl > Big::N) nl = Big::N;
  if (e2 >= 0) {
    big_set(r, f, nl); big_shl(r, (uint32_t)e2 + (lower_closer ? 2u : 1u), nl);
    big_set(s, lower_closer ? 4 : 2, nl);
    big_set(mp, 1, nl); big_shl(mp, (uint32_t)e2 + (lower_closer ? 1u : 0u), nl);
    big_set(mm, 1, nl); big_shl(mm, (uint32_t)e2, nl);
  } else {
    big_set(r, f, nl); big_shl(r, lower_closer ? 2u : 1u, nl);
    big_set(s, 1, nl); big_shl(s, (uint32_t)(-e2) + (lower_closer ? 2u : 1u), nl);
    big_set(mp, lower_closer ? 2 : 1, nl);
    big_set(mm, 1, nl);
  }
  if (est >= 0) big_mul_pow10(s, (uint32_t)est, nl);
  else { big_mul_pow10(r, (uint32_t)(-est), nl); big_mul_pow10(mp, (uint32_t)(-est), nl); big_mul_pow10(mm, (uint32_t)(-est), nl); }
  {
    big_copy(hi, r, nl);
    big_add(hi, mp, nl);
    int c = big_cmp(hi, s, nl);
    if (even ? c >= 0 : c > 0) k++;              // estimate was one too low
    else { big_mul_small(r, 10, nl); big_mul_small(mp, 10, nl); big_mul_small(mm, 10, nl); }
  }
  for (int guard = 0; guard < 19; guard++) {
    int d = 0;
    while (big_cmp(r, s, nl) >= 0) { big_sub(r, s, nl); d++; }
    int c1 = big_cmp(r, mm, nl);
    bool tc1 = even ? c1 <= 0 : c1 < 0;
    big_copy(hi, r, nl);
    big_add(hi, mp, nl);
    int c2 = big_cmp(hi, s, nl);
    bool tc2 = even ? c2 >= 0 : c2 > 0;
    if (!tc1 && !tc2) { dput(d); big_mul_small(r, 10, nl); big_mul_small(mp, 10, nl); big_mul_small(mm, 10, nl); continue; }
    if (tc1 && tc2) { big_copy(r2, r, nl); big_shl(r2, 1, nl); int c3 = big_cmp(r2, s, nl); if (c3 > 0 || (c3 == 0 && (d & 1))) d++; }   // nearer digit; an exact tie goes to the even digit (as ryu / std::to_chars)
    else if (tc2) d++;
    dput(d);
    break;
  }
  }
  // v = 0.d1d2…dn × 10^k ; layout rules of ryu's pretty printer (as serde_json prints f64)
  if (neg) out[n++] = '-';
  int kk = k;                                   // position of the decimal point relative to the first digit
  if (nd <= kk && kk <= 16) {                   // integer value: digits, zeros, ".0"
    for (int i = 0; i < nd; i++) out[n++] = dget(i);
    for (int i = nd; i < kk; i++) out[n++] = '0';
    out[n++] = '.'; out[n++] = '0';
  } else if (0 < kk && kk <= 16) {              // point inside the digits
    for (int i = 0; i < kk; i++) out[n++] = dget(i);
    out[n++] = '.';
    for (int i = kk; i < nd; i++) out[n++] = dget(i);
  } else if (-5 < kk && kk <= 0) {              // 0.000ddd
    out[n++] = '0'; out[n++] = '.';
    for (int i = 0; i < -kk; i++) out[n++] = '0';
    for (int i = 0; i < nd; i++) out[n++] = dget(i);
  } else {                                      // d[.ddd]e[-]xx
    out[n++] = dget(0);
    if (nd > 1) { out[n++] = '.'; for (int i = 1; i < nd; i++) out[n++] = dget(i); }
    out[n++] = 'e';
    int ex = kk - 1;
    if (ex < 0) { out[n++] = '-'; ex = -ex; }
    if (ex >= 100) { out[n++] = (char)('0' + ex / 100); ex %= 100; out[n++] = (char)('0' + ex / 10); out[n++] = (char)('0' + ex % 10); }
    else if (ex >= 10) { out[n++] = (char)('0' + ex / 10); out[n++] = (char)('0' + ex % 10); }
    else out[n++] = (char)('0' + ex);
  }
  return n;
}

}  // namespace lm
