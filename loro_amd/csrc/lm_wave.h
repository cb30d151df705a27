// Wave-level programming layer for the merge kernels.
//
// Product build (hipcc, gfx950): thin inline wrappers over the CDNA4 wave64 builtins.
// Test build (g++, -DLM_EMU, used ONLY by tests/emu): every lane of a workgroup runs as a cooperative
// fiber and the cross-lane primitives rendezvous through a per-wave mailbox, so the very same kernel
// source can be exercised on a machine without a GPU.  The emulation is a debugging harness for the
// kernels' logic; it is never compiled into, nor reachable from, the shipped library.
#pragma once
#include <cstdint>
#include <cstddef>

#ifndef LM_EMU
#include <hip/hip_runtime.h>
#define LM_DEV __device__ __forceinline__
#define LM_DEV_NOINLINE __device__ __noinline__
#define LM_KERNEL extern "C" __global__
#define LM_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))   // register budget: 512 / n VGPRs
#define LM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // the instruction scheduler moves nothing across this point
#define LM_ONE_WAVE_GROUPS __launch_bounds__(64)   // launched with 64-thread workgroups only: lifts the 128-VGPR cap of a 1,024-thread group
#define LM_SHARED(type, name, n) __shared__ type name[n]
#define LM_DYN_SHARED(type, name) extern __shared__ type name[]
typedef const __attribute__((address_space(3))) uint8_t* lm_lds_bytes;   // bytes known to sit in LDS (ds_read_u8 instead of a flat load)

namespace lmw {
static constexpr int WAVE = 64;
LM_DEV int tid() { return (int)threadIdx.x; }
LM_DEV int bid() { return (int)blockIdx.x; }
LM_DEV int bdim() { return (int)blockDim.x; }
LM_DEV int lane() { return (int)(threadIdx.x & 63); }
LM_DEV int wave_in_block() { return (int)(threadIdx.x >> 6); }
LM_DEV void block_sync() { __syncthreads(); }
LM_DEV uint64_t ballot(bool p) { return __ballot(p); }
LM_DEV uint32_t shfl(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src, 64); }
LM_DEV uint64_t shfl64(uint64_t v, int src) {
  uint32_t lo = shfl((uint32_t)v, src), hi = shfl((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}
LM_DEV uint32_t shfl_up(uint32_t v, int d) { return (uint32_t)__shfl_up((int)v, d, 64); }
LM_DEV uint32_t shfl_xor(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }
// value of `v` in lane `src`; src must be wave-uniform
LM_DEV uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
LM_DEV uint32_t first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
LM_DEV uint32_t atomic_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
LM_DEV uint32_t atomic_min(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
LM_DEV uint32_t atomic_max32(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
// LDS word += v, no value returned (ds_add_u32: no round trip to wait for)
LM_DEV void lds_add(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
LM_DEV uint64_t atomic_max64(unsigned long long* p, uint64_t v) { return atomicMax(p, (unsigned long long)v); }
LM_DEV uint32_t atomic_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
LM_DEV uint64_t atomic_cas64(unsigned long long* p, uint64_t cmp, uint64_t v) {
  return atomicCAS(p, (unsigned long long)cmp, (unsigned long long)v);
}
LM_DEV int popc64(uint64_t m) { return __popcll(m); }
LM_DEV int ffs64(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }  // -1 if empty
// Separates a "every lane loads" phase from a "every lane stores" phase over the same addresses.  A wave
// executes each instruction for all lanes at once, so this is only a scheduling fence here; the fiber
// emulation needs a real rendezvous.
LM_DEV void wave_sync() { __builtin_amdgcn_wave_barrier(); }
LM_DEV void mem_fence() { __threadfence(); }   // global stores of this wave are visible to later loads of any lane
LM_DEV uint64_t clock() { return __builtin_readcyclecounter(); }
}  // namespace lmw

#else  // ------------------------------------------------------------------ LM_EMU (tests only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <ucontext.h>
#define LM_DEV inline
#define LM_DEV_NOINLINE inline
#define LM_KERNEL inline
#define LM_WAVES_PER_SIMD(n)
#define LM_ONE_WAVE_GROUPS
#define LM_SCHED_FENCE() do {} while (0)
#define LM_SHARED(type, name, n) static type name[n]
#define LM_DYN_SHARED(type, name) type* name = (type*)lmw::emu_dyn_shared()
typedef const uint8_t* lm_lds_bytes;

namespace lmw {
static constexpr int WAVE = 64;
struct EmuFiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  int tid = 0;
};
struct EmuWave {
  uint64_t box[2][64];
  int arrived = 0;
  int live = 0;
  uint64_t gen = 0;
};
struct EmuBlock {
  std::vector<EmuFiber> fibers;
  std::vector<EmuWave> waves;
  ucontext_t sched;
  int cur = 0, bid = 0, bdim = 0;
  int blk_arrived = 0, blk_live = 0;
  uint64_t blk_gen = 0;
  uint64_t progress = 0;
  std::function<void()> body;
};
inline const char*& emu_kname() {   // the kernel being run (diagnostics of the deadlock report below)
  static thread_local const char* k = "?";
  return k;
}
inline EmuBlock*& emu_cur() {
  static thread_local EmuBlock* b = nullptr;
  return b;
}
inline void emu_yield() {
  EmuBlock* b = emu_cur();
  swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}
inline void emu_trampoline() {
  EmuBlock* b = emu_cur();
  b->body();
  EmuFiber& f = b->fibers[b->cur];
  f.done = true;
  EmuWave& w = b->waves[f.tid >> 6];
  w.live--;
  b->blk_live--;
  b->progress++;
  if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
  if (b->blk_live > 0 && b->blk_arrived == b->blk_live) { b->blk_arrived = 0; b->blk_gen++; }
  swapcontext(&f.ctx, &b->sched);
}
// run one workgroup of `bdim` threads
inline void emu_run_block(int bid, int bdim, std::function<void()> body) {
  static const size_t STACK = 256 * 1024;
  EmuBlock blk;
  blk.bid = bid;
  blk.bdim = bdim;
  blk.body = body;
  blk.fibers.resize(bdim);
  blk.waves.resize((bdim + 63) / 64);
  blk.blk_live = bdim;
  EmuBlock* prev = emu_cur();
  emu_cur() = &blk;
  for (int t = 0; t < bdim; t++) {
    EmuFiber& f = blk.fibers[t];
    f.tid = t;
    f.stack = (char*)malloc(STACK);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STACK;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())emu_trampoline, 0);
    blk.waves[t >> 6].live++;
  }
  int remaining = bdim;
  uint64_t last_progress = ~0ull;
  while (remaining > 0) {
    if (blk.progress == last_progress) {
      fprintf(stderr, "lm emu: deadlock in %s, workgroup %d of %d threads — lanes blocked at different collectives (non-uniform control flow)\n", emu_kname(), bid, bdim);
      abort();
    }
    last_progress = blk.progress;
    remaining = 0;
    for (int t = 0; t < bdim; t++) {
      if (blk.fibers[t].done) continue;
      blk.cur = t;
      swapcontext(&blk.sched, &blk.fibers[t].ctx);
      if (!blk.fibers[t].done) remaining++;
    }
  }
  for (auto& f : blk.fibers) free(f.stack);
  emu_cur() = prev;
}
inline std::vector<uint64_t>& emu_dyn_buf() { static std::vector<uint64_t> v; return v; }
inline void* emu_dyn_shared() { return emu_dyn_buf().data(); }
template <class F>
inline void emu_launch(int grid, int block, size_t dyn_shared_bytes, F&& body) {
  emu_dyn_buf().assign(dyn_shared_bytes / 8 + 2, 0);
  for (int b = 0; b < grid; b++) emu_run_block(b, block, body);
}

inline int tid() { return emu_cur()->fibers[emu_cur()->cur].tid; }
inline int bid() { return emu_cur()->bid; }
inline int bdim() { return emu_cur()->bdim; }
inline int lane() { return tid() & 63; }
inline int wave_in_block() { return tid() >> 6; }
inline void block_sync() {
  EmuBlock* b = emu_cur();
  uint64_t g = b->blk_gen;
  b->blk_arrived++;
  b->progress++;
  if (b->blk_arrived == b->blk_live) { b->blk_arrived = 0; b->blk_gen++; return; }
  while (b->blk_gen == g) emu_yield();
}
// all live lanes of the wave publish `v`; returns pointer to the 64 published values (dead lanes = 0)
inline const uint64_t* emu_exchange(uint64_t v) {
  EmuBlock* b = emu_cur();
  int t = tid();
  EmuWave& w = b->waves[t >> 6];
  uint64_t g = w.gen;
  uint64_t* box = w.box[g & 1];
  if (w.arrived == 0) memset(box, 0, sizeof(uint64_t) * 64);
  box[t & 63] = v;
  w.arrived++;
  b->progress++;
  if (w.arrived == w.live) { w.arrived = 0; w.gen++; return box; }
  while (w.gen == g) emu_yield();
  return box;
}
inline uint64_t ballot(bool p) {
  const uint64_t* x = emu_exchange(p ? 1 : 0);
  uint64_t m = 0;
  for (int i = 0; i < 64; i++) if (x[i]) m |= 1ull << i;
  return m;
}
inline uint64_t shfl64(uint64_t v, int src) {
  // src may differ per lane: publish value, then read after the rendezvous
  const uint64_t* x = emu_exchange(v);
  return x[src & 63];
}
inline uint32_t shfl(uint32_t v, int src) { return (uint32_t)shfl64(v, src); }
inline uint32_t shfl_up(uint32_t v, int d) {
  int l = lane();
  const uint64_t* x = emu_exchange(v);
  return l >= d ? (uint32_t)x[l - d] : v;
}
inline uint32_t shfl_xor(uint32_t v, int m) {
  int l = lane();
  const uint64_t* x = emu_exchange(v);
  return (uint32_t)x[(l ^ m) & 63];
}
inline uint32_t bcast(uint32_t v, int src) { return shfl(v, src); }
inline uint32_t first(uint32_t v) {
  // value of the lowest live lane: publish (v | 1<<32) so dead lanes (0) are distinguishable
  const uint64_t* x = emu_exchange((uint64_t)v | (1ull << 32));
  for (int i = 0; i < 64; i++) if (x[i] >> 32) return (uint32_t)x[i];
  return v;
}
inline uint32_t atomic_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline uint32_t atomic_min(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
inline uint32_t atomic_max32(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline void lds_add(uint32_t* p, uint32_t v) { *p += v; }
inline uint64_t atomic_max64(unsigned long long* p, uint64_t v) { uint64_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomic_or(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
inline uint64_t atomic_cas64(unsigned long long* p, uint64_t cmp, uint64_t v) { uint64_t o = *p; if (o == cmp) *p = v; return o; }
inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
inline int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : -1; }
inline void wave_sync() { (void)emu_exchange(0); }
inline void mem_fence() {}
inline uint64_t clock() { return 0; }
}  // namespace lmw
#endif

#if defined(LM_EMU) && defined(LM_EMU_TRACE)
#define LM_SETERR(lhs, code) do { if (lmw::lane() == 0 || getenv("LM_EMU_ERR_ALL")) fprintf(stderr, "lm emu: " #lhs " = " #code " at %s:%d\n", __FILE__, __LINE__); (lhs) = (code); } while (0)
#else
#define LM_SETERR(lhs, code) do { (lhs) = (code); } while (0)
#endif

namespace lmw {
// ---- derived wave collectives (same source for both builds); all lanes must participate
// reference formulation (bpermute based); kept for the self-test and the emulation build
LM_DEV uint32_t scan_incl_add_shfl(uint32_t v) {
  int l = lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = shfl_up(v, d);
    if (l >= d) v += t;
  }
  return v;
}
#ifndef LM_EMU
// wave64 inclusive prefix sum on the DPP crossbar: row_shr 1/2/4/8 inside each 16-lane row, then
// row_bcast:15 and row_bcast:31 to carry the row totals (six VALU ops instead of six LDS round trips).
// A lane without a source (row_shr past the row start, a row masked out of a row_bcast) receives `old` = 0, so every step
// is an unconditional add — no lane predicates to keep in (or reload into) SGPR pairs.
LM_DEV uint32_t scan_incl_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
// the same crossbar walk with max instead of add (unsigned; a lane without a source sees 0)
LM_DEV uint32_t scan_incl_max(uint32_t v) {
  uint32_t t;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = t > v ? t : v;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = t > v ? t : v;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;
  t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;
  return v;
}
#else
LM_DEV uint32_t scan_incl_add(uint32_t v) { return scan_incl_add_shfl(v); }
LM_DEV uint32_t scan_incl_max(uint32_t v) {
  int l = lane();
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = shfl_up(v, d); if (l >= d && t > v) v = t; }
  return v;
}
#endif
#ifndef LM_EMU
// lane i receives lane i-d (d = 1 or 2) across the whole wave64 through the DPP crossbar (wave_shr:1, one or two VALU
// ops) instead of an LDS permute; lanes below d keep their own value
LM_DEV uint32_t shift_up(uint32_t v, int d) {
  uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);
  if (d == 2) t = (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x138, 0xf, 0xf, false);
  return t;
}
// same, but lanes below d receive 0: no tied `old` operand, so no register copy in front of the DPP move
LM_DEV uint32_t shift_up0(uint32_t v, int d) {
  uint32_t t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true);
  if (d == 2) t = (uint32_t)__builtin_amdgcn_mov_dpp((int)t, 0x138, 0xf, 0xf, true);
  return t;
}
#else
LM_DEV uint32_t shift_up(uint32_t v, int d) { return shfl_up(v, d); }
LM_DEV uint32_t shift_up0(uint32_t v, int d) { uint32_t t = shfl_up(v, d); return lane() < d ? 0u : t; }
#endif
// lane i receives lane i+N of its own 16-lane row (DPP row_shl:N, one VALU move; lanes whose source lies beyond the row receive 0)
#ifndef LM_EMU
template <int N> LM_DEV uint32_t row_down(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x100 | N, 0xf, 0xf, true); }
#else
template <int N> LM_DEV uint32_t row_down(uint32_t v) { int l = lane(); uint32_t t = shfl(v, (l & ~15) | ((l + N) & 15)); return (l & 15) + N > 15 ? 0u : t; }
#endif
// wave sum: the DPP prefix scan's last lane (six VALU ops + one readlane instead of six dependent LDS swizzles)
LM_DEV uint32_t reduce_add(uint32_t v) { return bcast(scan_incl_add(v), 63); }
LM_DEV uint32_t reduce_max(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { uint32_t t = shfl_xor(v, m); v = t > v ? t : v; }
  return v;
}
LM_DEV uint32_t reduce_min(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { uint32_t t = shfl_xor(v, m); v = t < v ? t : v; }
  return v;
}
LM_DEV bool any(bool p) { return ballot(p) != 0; }
}  // namespace lmw
