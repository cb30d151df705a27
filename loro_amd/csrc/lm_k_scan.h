// Exclusive prefix sums over interleaved u32 counters (rows × ncomp) — table offsets of the decode stage.
#pragma once
#include "lm_wave.h"

namespace lm {

static constexpr int SCAN_TILE = 256;

// phase 1: per-tile exclusive scan; out[row] = exclusive-in-tile, tile_sum[tile] = tile total
LM_KERNEL void k_scan_tile(const uint32_t* in, uint32_t* out, uint32_t* tile_sum, uint32_t n, uint32_t ncomp) {
  LM_SHARED(uint32_t, s_w, 4);
  uint32_t tile = (uint32_t)lmw::bid();
  uint32_t row = tile * SCAN_TILE + (uint32_t)lmw::tid();
  int lane = lmw::lane(), w = lmw::wave_in_block();
  for (uint32_t c = 0; c < ncomp; c++) {
    uint32_t v = row < n ? in[(uint64_t)row * ncomp + c] : 0;
    uint32_t inc = lmw::scan_incl_add(v);
    if (lane == 63) s_w[w] = inc;
    lmw::block_sync();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += s_w[k];
    if (row < n) out[(uint64_t)row * ncomp + c] = base + inc - v;
    if (lmw::tid() == SCAN_TILE - 1) tile_sum[(uint64_t)tile * ncomp + c] = base + inc;
    lmw::block_sync();
  }
}
// phase 2: one wave scans the tile sums in place (exclusive)
LM_KERNEL void k_scan_sums(uint32_t* tile_sum, uint32_t n_tiles, uint32_t ncomp, uint32_t* totals) {
  int lane = lmw::lane();
  for (uint32_t c = 0; c < ncomp; c++) {
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 64) {
      uint32_t t = t0 + (uint32_t)lane;
      uint32_t v = t < n_tiles ? tile_sum[(uint64_t)t * ncomp + c] : 0;
      uint32_t inc = lmw::scan_incl_add(v);
      if (t < n_tiles) tile_sum[(uint64_t)t * ncomp + c] = run + inc - v;
      run += lmw::bcast(inc, 63);
    }
    if (lane == 0) totals[c] = run;
  }
}
// phase 3: add tile bases; row n receives the totals
LM_KERNEL void k_scan_add(uint32_t* out, const uint32_t* tile_sum, const uint32_t* totals, uint32_t n, uint32_t ncomp) {
  uint32_t tile = (uint32_t)lmw::bid();
  uint32_t row = tile * SCAN_TILE + (uint32_t)lmw::tid();
  for (uint32_t c = 0; c < ncomp; c++) {
    if (row < n) out[(uint64_t)row * ncomp + c] += tile_sum[(uint64_t)tile * ncomp + c];
    if (row == n) out[(uint64_t)row * ncomp + c] = totals[c];
  }
}

// wave-primitive self test: DPP scan vs the bpermute formulation on pseudo-random lane values
LM_KERNEL void k_selftest(uint32_t* out, uint32_t rounds) {
  int lane = lmw::lane();
  uint32_t bad = 0;
  uint32_t x = 0x9E3779B9u * (uint32_t)(lane + 1) + (uint32_t)lmw::bid();
  for (uint32_t r = 0; r < rounds; r++) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    uint32_t v = (r & 1) ? (x & 0xffff) : (x % 65u);
    uint32_t a = lmw::scan_incl_add(v), b = lmw::scan_incl_add_shfl(v);
    bad += a != b ? 1u : 0u;
    uint64_t m = lmw::ballot((v & 1) != 0);
    uint32_t c = (uint32_t)lmw::popc64(m & ((2ull << lane) - 1));
    uint32_t dref = lmw::scan_incl_add_shfl(v & 1);
    bad += c != dref ? 1u : 0u;
    // whole-wave shifts on the DPP crossbar vs the LDS permute (lanes below the distance are don't-care)
    uint32_t s1 = lmw::shift_up(x, 1), s2 = lmw::shift_up(x, 2), r1 = lmw::shfl_up(x, 1), r2 = lmw::shfl_up(x, 2);
    bad += (lane >= 1 && s1 != r1) ? 1u : 0u;
    bad += (lane >= 2 && s2 != r2) ? 1u : 0u;
  }
  bad = lmw::reduce_add(bad);
  if (lane == 0) out[lmw::bid()] = bad;
}

// loc[] := NONE in front of the integrate stage (lm_pipeline.h): 16 bytes per lane and store, grid-stride.  hipMemsetAsync's fill
// kernel wrote these 2 GB per 5,000 configs[1] documents at ≈2 TB/s (profiles/r06_kernel_stats.md: __amd_rocclr_fillBufferAligned,
// ≈1 ms per launch of the pipeline); HBM takes stores faster than that.
LM_KERNEL void k_fill_words(uint32_t* p, uint64_t n4, uint32_t v, uint32_t n_threads) {
  struct alignas(16) U4 { uint32_t x, y, z, w; };
  U4* q = (U4*)p;
  const U4 w4 = {v, v, v, v};
  for (uint64_t i = (uint64_t)lmw::bid() * (uint64_t)lmw::bdim() + (uint64_t)lmw::tid(); i < n4; i += n_threads) q[i] = w4;
}

}  // namespace lm
