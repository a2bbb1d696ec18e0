// Shared helpers for the gfx950 kernels of the OpenPVSG hot path.
// Everything here targets CDNA4 (MI355X) directly: 64-lane wavefronts, 8 XCDs,
// f32-input MFMA.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

// ---- status codes shared with include/openpvsg_hip.h ----------------------
#define PVSG_OK 0
#define PVSG_ERR_INVALID_ARG 1
#define PVSG_ERR_UNSUPPORTED 2
#define PVSG_ERR_HIP 3

namespace pvsg {

// thread-local last-error text, read through pvsg_last_error()
char* err_buf();
int set_err(int code, const char* fmt, ...);

#define PVSG_REQUIRE(cond, ...)                                      \
  do {                                                               \
    if (!(cond)) return ::pvsg::set_err(PVSG_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

#define PVSG_LAUNCH_CHECK(name)                                                    \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess)                                                         \
      return ::pvsg::set_err(PVSG_ERR_HIP, "%s: launch failed: %s", name,          \
                             hipGetErrorString(e__));                              \
  } while (0)

// Observed on gfx950: workgroup b runs on XCD (b % 8), each XCD has a private
// 4 MiB L2.  Remap the linear block id so that every XCD walks ONE contiguous
// range of logical work items (speed only; correctness never depends on it).
// Bijective for every nblk (cdna_hip_programming.md, 256^2 template note).
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}

// streaming (touched once) operands: non-temporal so they do not evict the gathered rows from L2
__device__ __forceinline__ float4 ld4_stream(const float* p) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 t = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<v4*>(p));
}

// Dynamic LDS above 64 KB needs an opt-in per kernel function; done once per (function, device) -- one `done` mask per
// call site, bit = device ordinal.  Racing threads at worst set the same value twice.
inline hipError_t ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// Zero a small device buffer with a KERNEL, never hipMemsetAsync: captured into a hipGraph, the memset node of these
// 16-byte .. few-KiB buffers did not reliably run before the kernel node that follows it (ROCm 7.2, MI355X) -- replays kept
// the previous call's 'has an unblocked key' flags and produced NaN rows (scripts/lab/graph_bisect.py found it).
static __global__ void zero_words_kernel(unsigned* __restrict__ p, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
inline hipError_t zero_words_async(void* p, size_t bytes, hipStream_t stream) {
  const long long n = (long long)(bytes / 4);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, static_cast<unsigned*>(p), n);
  return hipGetLastError();
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace pvsg
