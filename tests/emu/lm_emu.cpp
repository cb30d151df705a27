// TEST HARNESS ONLY — compiles the device kernels for the host with every lane as a cooperative fiber
// (lm_wave.h, LM_EMU) so the kernels' logic can be checked against the oracle on a machine without a GPU.
// It is built and loaded exclusively by tests/ (never by loro_amd/, bench.py or the C-ABI product library),
// and its exports carry the lmemu_ prefix so it cannot be mistaken for libloromerge.so.
#define LM_EMU 1
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../loro_amd/csrc/lm_wave.h"

namespace lmbe {
static uint64_t g_alloc = 0;
static std::chrono::steady_clock::time_point g_t0;
inline bool init(int) { return true; }
struct StreamCtx { int unused = 0; };
inline StreamCtx* stream_create(int) { return new StreamCtx(); }
inline void stream_destroy(StreamCtx* c) { delete c; }
inline void bind(StreamCtx*) {}
inline void* dalloc(size_t n) { g_alloc += n; void* p = malloc(n ? n : 1); memset(p, 0xA5, n); return p; }  // poisoned: kernels must not rely on fresh memory
inline void dfree(void* p) { free(p); }
inline void dmemset(void* p, int v, size_t n) { memset(p, v, n); }
inline void h2d(void* d, const void* h, size_t n) { memcpy(d, h, n); }
inline void h2d_async(void* d, const void* h, size_t n) { memcpy(d, h, n); }
inline void d2h(void* h, const void* d, size_t n) { memcpy(h, d, n); }
inline void d2d(void* dst, const void* src, size_t n) { memmove(dst, src, n); }
inline void sync() {}
inline void* halloc(size_t n) { return malloc(n ? n : 1); }
inline void hfree(void* p) { free(p); }
inline uint64_t allocated_bytes() { return g_alloc; }
inline void tic(bool) { g_t0 = std::chrono::steady_clock::now(); }
template <class V> inline void flush_times(V&) {}
inline void reset_times() {}
template <class V>
inline void toc(const char* name, V& times, bool profiling) {
  if (!profiling) return;
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count();
  times.push_back({name, ms});
}
}  // namespace lmbe

#define LM_LAUNCH(kern, grid, block, ...) do { lmw::emu_kname() = #kern; lmw::emu_launch((int)(grid), (int)(block), 0, [&]() { lm::kern(__VA_ARGS__); }); } while (0)
#define LM_LAUNCH_DYN(kern, grid, block, shmem, ...) do { lmw::emu_kname() = #kern; lmw::emu_launch((int)(grid), (int)(block), (size_t)(shmem), [&]() { lm::kern(__VA_ARGS__); }); } while (0)
#define LM_API(name) lmemu_##name

#include "../../loro_amd/csrc/lm_capi_impl.h"

#include "../../loro_amd/csrc/lm_f64.h"
extern "C" int lmemu_f64_json(uint64_t bits, char* out) { static lm::Big ws[6]; return lm::f64_json(bits, out, ws); }
