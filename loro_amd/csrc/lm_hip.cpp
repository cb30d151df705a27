// Product translation unit: HIP backend + kernels + C ABI → libloromerge.so (gfx950).
// There is no CPU path in this library: lm_create() returns NULL unless a HIP device is present.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <map>
#include "lm_wave.h"

namespace lm { struct KernelTime; }

namespace lmbe {
// Every Engine owns one StreamCtx (a non-blocking HIP stream + two timing events).  The host thread that drives an
// engine binds it with bind(); all stream-ordered calls below go to the bound context, so several engines (the
// sub-batches of one lm_ctx, or several lm_ctx) can be driven from different host threads and overlap on the device.
struct StreamCtx {
  hipStream_t s = nullptr;
  int device = 0;                        // the HIP device this stream (and every buffer of its engine) lives on
  std::vector<hipEvent_t> ev;            // pairs (start, stop), one pair per timed stage of a run
  std::vector<const char*> names;        // stages recorded since the last flush
  bool open = false;
  // small host → device copies of a run (sizing tables, offsets): staged in a pinned ring and copied stream-ordered WITHOUT a host
  // wait — the ring is reused only after the next synchronisation of the stream (rounds 1-4a synchronised after every one of the
  // ≈10 table uploads of a run)
  uint8_t* ring = nullptr;
  size_t ring_cap = 0, ring_top = 0;
  std::vector<uint8_t*> ring_old;        // outgrown rings: still the source of copies in flight, freed at the next synchronisation
};
static thread_local StreamCtx* cur = nullptr;
static std::atomic<uint64_t> g_alloc{0};

#define LM_HIP_CHECK(x)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline bool init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return false;
  return hipSetDevice(device) == hipSuccess;
}
inline StreamCtx* stream_create(int device) {
  if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
  StreamCtx* c = new StreamCtx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking) != hipSuccess) { delete c; throw std::runtime_error("hipStreamCreate failed"); }
  return c;
}
inline void stream_destroy(StreamCtx* c) {
  if (!c) return;
  if (cur == c) cur = nullptr;
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  (void)hipStreamSynchronize(c->s);
  for (uint8_t* r : c->ring_old) (void)hipHostFree(r);
  if (c->ring) (void)hipHostFree(c->ring);
  (void)hipStreamDestroy(c->s);
  delete c;
}
// bind the calling host thread to a stream context (HIP's current device is per host thread)
inline void bind(StreamCtx* c) { (void)hipSetDevice(c->device); cur = c; }
inline void* dalloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n) != hipSuccess) return nullptr;
  g_alloc += n;
  return p;
}
inline void dfree(void* p) { (void)hipFree(p); }
inline void dmemset(void* p, int v, size_t n) { LM_HIP_CHECK(hipMemsetAsync(p, v, n, cur->s)); }
inline void synced() {   // the stream has just been synchronised: every staged copy has completed
  cur->ring_top = 0;
  for (uint8_t* r : cur->ring_old) (void)hipHostFree(r);
  cur->ring_old.clear();
}
inline void sync_stream() { LM_HIP_CHECK(hipStreamSynchronize(cur->s)); synced(); }
inline void h2d(void* d, const void* h, size_t n) {
  if (!n) return;
  if (n > ((size_t)32 << 20)) { LM_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, cur->s)); sync_stream(); return; }
  size_t at = (cur->ring_top + 63) & ~(size_t)63;
  if (at + n > cur->ring_cap) {
    if (n > cur->ring_cap / 2) {           // a larger ring; the old one may still feed copies in flight
      size_t cap = cur->ring_cap ? cur->ring_cap * 2 : ((size_t)4 << 20);
      while (cap < 2 * n) cap *= 2;
      void* nr = nullptr;
      if (hipHostMalloc(&nr, cap, hipHostMallocDefault) != hipSuccess) throw std::runtime_error("pinned staging ring allocation failed");
      if (cur->ring) cur->ring_old.push_back(cur->ring);
      cur->ring = (uint8_t*)nr; cur->ring_cap = cap;
    } else sync_stream();                  // full: wait once, start over
    at = 0;
  }
  memcpy(cur->ring + at, h, n);
  LM_HIP_CHECK(hipMemcpyAsync(d, cur->ring + at, n, hipMemcpyHostToDevice, cur->s));
  cur->ring_top = at + n;
}
inline void h2d_async(void* d, const void* h, size_t n) { LM_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, cur->s)); }   // h pinned; completed by the next sync()
inline void d2h(void* h, const void* d, size_t n) { LM_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, cur->s)); sync_stream(); }
inline void d2d(void* dst, const void* src, size_t n) { LM_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, cur->s)); }   // stream-ordered
inline void sync() { sync_stream(); }
inline void* halloc(size_t n) { void* p = nullptr; if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return nullptr; return p; }
inline void hfree(void* p) { (void)hipHostFree(p); }
inline uint64_t allocated_bytes() { return g_alloc.load(); }
// Stage timing: events are recorded on the engine's stream without any host synchronisation, so a profiled run
// overlaps its streams exactly like an unprofiled one; flush_times() waits once, at the end of the run.
inline void tic(bool profiling) {
  if (!profiling) return;
  size_t k = cur->names.size();
  while (cur->ev.size() < 2 * (k + 1)) { hipEvent_t e; LM_HIP_CHECK(hipEventCreate(&e)); cur->ev.push_back(e); }
  (void)hipEventRecord(cur->ev[2 * k], cur->s);
  cur->open = true;
}
template <class V>
inline void toc(const char* name, V&, bool profiling) {
  if (!profiling || !cur->open) return;
  (void)hipEventRecord(cur->ev[2 * cur->names.size() + 1], cur->s);
  cur->names.push_back(name);
  cur->open = false;
}
inline void reset_times() { cur->names.clear(); cur->open = false; }
template <class V>
inline void flush_times(V& times) {
  if (cur->names.empty()) return;
  sync_stream();
  for (size_t k = 0; k < cur->names.size(); k++) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, cur->ev[2 * k], cur->ev[2 * k + 1]);
    times.push_back({cur->names[k], (double)ms});
  }
  cur->names.clear();
}
inline void check_launch(const char* name) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("launch ") + name + ": " + hipGetErrorString(e));
}
}  // namespace lmbe

#define LM_LAUNCH(kern, grid, block, ...)                                                          \
  do {                                                                                             \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, lmbe::cur->s, __VA_ARGS__); \
    lmbe::check_launch(#kern);                                                                     \
  } while (0)
// (the dynamic-LDS ceiling of a kernel is raised when a launch needs more than any launch of that kernel before it — not on every
// launch: hipFuncSetAttribute is a driver round trip.  Per kernel function, whatever site launches it.)
namespace lmbe {
inline bool dyn_lds_needs_raise(const void* fn, size_t bytes) {
  // (keyed by (device, function): the attribute belongs to the function as loaded on ONE device — a context on a second device
  // of the same process must raise it again)
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> set;
  std::lock_guard<std::mutex> g(mu);
  size_t& cur_max = set[std::make_pair(cur ? cur->device : 0, fn)];
  if (bytes <= cur_max) return false;
  cur_max = bytes;
  return true;
}
}  // namespace lmbe
#define LM_LAUNCH_DYN(kern, grid, block, shmem, ...)                                               \
  do {                                                                                             \
    if (lmbe::dyn_lds_needs_raise((const void*)kern, (size_t)(shmem)))                             \
      LM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shmem))); \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (shmem), lmbe::cur->s, __VA_ARGS__); \
    lmbe::check_launch(#kern);                                                                     \
  } while (0)
#define LM_API(name) lm_##name
#define LM_PARALLEL_PARTS 1

#include "lm_capi_impl.h"
