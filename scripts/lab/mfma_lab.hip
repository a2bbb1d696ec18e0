// Is the split-bf16 GEMM stalled or power-capped?  A register-only MFMA loop (the GEMM's 24-MFMA K-step pattern, no memory,
// no LDS, no barrier) at 1..4 waves per SIMD on three operand fills: zeros, the three limbs of N(0,1) data, random bits.
// Reports TF/s, shader cycles per MFMA and SIMD (s_memtime) and the shader clock (s_memtime vs the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_lab scripts/lab/mfma_lab.hip && ./mfma_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: v_mfma_f32_32x32x16_bf16, 2x2 accumulators of 16 registers (the GEMM's K-step: 24 MFMAs, 32 cycles each)
// SHAPE 1: v_mfma_f32_16x16x32_bf16, 4x4 accumulators of 4 registers (same flops per step: 96 MFMAs, 8 passes... 16 cycles each)
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop_kernel(const bf16x8* __restrict__ src, float* __restrict__ sink,
                                                           unsigned long long* __restrict__ ticks, int iters) {
  const int tid = threadIdx.x;
  constexpr int NB = SHAPE == 0 ? 2 : 4;               // distinct operand blocks per limb (64 x 64 wave tile either way)
  bf16x8 av[3][NB], wv[3][NB];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      av[l][b] = src[(size_t)(l * 8 + b) * 256 + tid];
      wv[l][b] = src[(size_t)(l * 8 + 4 + b) * 256 + tid];
    }
  constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float total = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) total += acc[i][j][r];
  } else {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) total += acc[i][j][r];
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (total == 123.456f) sink[tid] = total;
  if ((tid & 63) == 0) {
    ticks[(blockIdx.x * 4 + (tid >> 6)) * 2] = t1 - t0;
    ticks[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = r1 - r0;
  }
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// fill 0: zeros; 1: limb l of N(0,1)-like data (slot / 4 = limb); 2: random bits
__global__ void fill_src(unsigned* p, int fill) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one packed bf16 pair; 24 slots x 256 lanes x 4 pairs
  if (i >= 24 * 256 * 4) return;
  if (fill == 0) { p[i] = 0u; return; }
  const unsigned a = hash32(i * 2 + 1), b = hash32(i * 2 + 2);
  if (fill == 2) { p[i] = a ^ (b << 7); return; }
  const int limb = (i / (256 * 4)) / 8;
  float v[2];
  for (int k = 0; k < 2; ++k) {
    const unsigned x = k ? b : a, y = hash32(x + 77u);
    v[k] = ((x >> 8) * (1.f / 8388608.f) - 1.f) + ((y >> 8) * (1.f / 8388608.f) - 1.f);
  }
  unsigned out = 0;
  for (int k = 0; k < 2; ++k) {
    float r = v[k];
    unsigned bits = 0;
    for (int l = 0; l <= limb; ++l) {
      const bf16x2 h = __builtin_convertvector(f32x2{r, 0.f}, bf16x2);
      bits = __builtin_bit_cast(unsigned, h) & 0xffffu;
      r -= __builtin_bit_cast(float, bits << 16);
    }
    out |= bits << (16 * k);
  }
  p[i] = out;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned* src;
  float* sink;
  unsigned long long* ticks;
  CK(hipMalloc(&src, 24 * 256 * 16));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&ticks, (size_t)cus * 4 * 4 * 2 * 8));
  const int iters = 2048;
  const char* fills[3] = {"zeros", "limbs_of_normal", "random_bits"};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%-8s %-16s %6s %10s %9s %12s %9s\n", "mfma", "operands", "w/SIMD", "ms", "TF/s", "pipe cyc/MFMA", "clk_GHz");
  for (int shape = 0; shape < 2; ++shape)
    for (int fill = 0; fill < 3; ++fill) {
      fill_src<<<96, 256>>>(src, fill);
      for (int wps = 1; wps <= 4; ++wps) {
        const unsigned grid = (unsigned)(cus * wps);
        auto launch = [&]() {
          if (shape == 0) mfma_loop_kernel<0><<<grid, 256>>>(reinterpret_cast<const bf16x8*>(src), sink, ticks, iters);
          else mfma_loop_kernel<1><<<grid, 256>>>(reinterpret_cast<const bf16x8*>(src), sink, ticks, iters);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        std::vector<unsigned long long> h((size_t)grid * 4 * 2);
        CK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, real = 0;
        for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; real += (double)h[i + 1]; }
        cyc /= (double)(h.size() / 2);
        real /= (double)(h.size() / 2);
        const double mfma_per_wave = (double)iters * (shape == 0 ? 24 : 96);
        const double flops = (double)grid * 4 * iters * (shape == 0 ? 24.0 * 2.0 * 32 * 32 * 16 : 96.0 * 2.0 * 16 * 16 * 32);
        printf("%-8s %-16s %6d %10.3f %9.0f %12.2f %9.2f\n", shape == 0 ? "32x32x16" : "16x16x32", fills[fill], wps, ms,
               flops / ms / 1e9, cyc / mfma_per_wave / wps, cyc / (real * 10.0));
      }
    }
  return 0;
}
