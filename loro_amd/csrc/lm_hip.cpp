// Product translation unit: HIP backend + kernels + C ABI → libloromerge.so (gfx950).
// There is no CPU path in this library: lm_create() returns NULL unless a HIP device is present.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <atomic>
#include "lm_wave.h"

namespace lm { struct KernelTime; }

namespace lmbe {
static hipStream_t g_stream = nullptr;
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static std::atomic<uint64_t> g_alloc{0};
static int g_device = -1;

#define LM_HIP_CHECK(x)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline bool init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return false;
  if (hipSetDevice(device) != hipSuccess) return false;
  g_device = device;
  if (!g_stream) {
    if (hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return false;
    hipEventCreate(&g_ev0);
    hipEventCreate(&g_ev1);
  }
  return true;
}
inline void* dalloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n) != hipSuccess) return nullptr;
  g_alloc += n;
  return p;
}
inline void dfree(void* p) { (void)hipFree(p); }
inline void dmemset(void* p, int v, size_t n) { LM_HIP_CHECK(hipMemsetAsync(p, v, n, g_stream)); }
inline void h2d(void* d, const void* h, size_t n) { LM_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, g_stream)); LM_HIP_CHECK(hipStreamSynchronize(g_stream)); }
inline void d2h(void* h, const void* d, size_t n) { LM_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, g_stream)); LM_HIP_CHECK(hipStreamSynchronize(g_stream)); }
inline void sync() { LM_HIP_CHECK(hipStreamSynchronize(g_stream)); }
inline void* halloc(size_t n) { void* p = nullptr; if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return nullptr; return p; }
inline void hfree(void* p) { (void)hipHostFree(p); }
inline uint64_t allocated_bytes() { return g_alloc.load(); }
inline void tic() { (void)hipEventRecord(g_ev0, g_stream); }
template <class V>
inline void toc(const char* name, V& times, bool profiling) {
  if (!profiling) return;
  (void)hipEventRecord(g_ev1, g_stream);
  LM_HIP_CHECK(hipEventSynchronize(g_ev1));
  float ms = 0;
  (void)hipEventElapsedTime(&ms, g_ev0, g_ev1);
  times.push_back({name, (double)ms});
}
inline void check_launch(const char* name) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("launch ") + name + ": " + hipGetErrorString(e));
}
}  // namespace lmbe

#define LM_LAUNCH(kern, grid, block, ...)                                                          \
  do {                                                                                             \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, lmbe::g_stream, __VA_ARGS__); \
    lmbe::check_launch(#kern);                                                                     \
  } while (0)
#define LM_LAUNCH_DYN(kern, grid, block, shmem, ...)                                               \
  do {                                                                                             \
    LM_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(shmem))); \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3((unsigned)(block)), (shmem), lmbe::g_stream, __VA_ARGS__); \
    lmbe::check_launch(#kern);                                                                     \
  } while (0)
#define LM_API(name) lm_##name

#include "lm_capi_impl.h"
