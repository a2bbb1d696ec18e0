// Register-only 16-bit MFMA spin kernels (the co-runner of scripts/coresidency_repro.hip as a shared object, so that Python
// drivers can put it next to the backend's kernels and to RCCL):  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o
// /tmp/libspin_mfma.so scripts/lab/spin_mfma.hip
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void spin_kernel(float* sink, int iters) {
  const unsigned l = threadIdx.x;
  const u32x4 a = {0x3f803f80u + l, 0x3e803f00u + 3 * l, 0x3f203f40u + 5 * l, 0x3f603ec0u + 7 * l};   // plausible 16-bit pairs
  const u32x4 b = {0x3f003f80u + 11 * l, 0x3ea03f10u + l, 0x3f283f48u + 2 * l, 0x3f683ed0u + l};
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[j], 0, 0, 0);
    }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) sink[0] = acc[0][0];
}

extern "C" int spin_launch(int kind, int blocks, int iters, float* sink, void* stream) {
  if (kind == 0) hipLaunchKernelGGL(spin_kernel<0>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sink, iters);
  else hipLaunchKernelGGL(spin_kernel<1>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sink, iters);
  return (int)hipGetLastError();
}
