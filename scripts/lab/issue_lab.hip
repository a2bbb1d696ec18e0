// Issue-slot lab: how do MFMA (bf16, separate matrix pipe) and VALU instructions of ONE wave share a SIMD?
// Each wave runs REPS x [ CHAIN dependent v_mfma_f32_16x16x32_bf16, each followed by NV independent VALU ops ].
// Reports cycles per MFMA (s_memtime is the constant 100 MHz clock, so wall time x the reported clock is used instead).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/lab/issue_lab scripts/lab/issue_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NCH, int KIND>   // NCH independent accumulator chains round-robin; KIND 0: 16x16x32, 1: 16x16x16 (legacy), 2: no MFMA
__global__ __launch_bounds__(512) void k(float* sink, int reps, float seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
  s16x4 a4 = {1, 2, 3, 4}, b4 = {5, 6, 7, (short)threadIdx.x};
  f32x4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      if (KIND == 0) acc[m % NCH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % NCH], 0, 0, 0);
      if (KIND == 1) acc[m % NCH] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[m % NCH], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NV; ++i) { v[i % 8] = v[i % 8] * 1.0001f + 0.5f; asm volatile("" : "+v"(v[i % 8])); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s;
}

template <int NV, int NCH, int KIND>
void run(int wg_threads, const char* name) {
  float* sink; hipMalloc(&sink, 4);
  const int reps = 4000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NCH, KIND>), dim3(blocks), dim3(wg_threads), 0, 0, sink, 100, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NCH, KIND>), dim3(blocks), dim3(wg_threads), 0, 0, sink, reps, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_chunk_ns = ms * 1e6 / (reps * 12.0);
  printf("%-10s NV=%d chains=%d waves/SIMD=%d  %.2f ns per [MFMA + %d VALU] per wave (%.1f cyc @2.1GHz)\n", name, NV, NCH,
         wg_threads / 256, per_chunk_ns, NV, per_chunk_ns * 2.1);
  hipFree(sink);
}

int main() {
  for (int wt : {256, 512}) {
    run<0, 1, 0>(wt, "x32"); run<2, 1, 0>(wt, "x32"); run<4, 1, 0>(wt, "x32"); run<6, 1, 0>(wt, "x32"); run<8, 1, 0>(wt, "x32");
    run<0, 2, 0>(wt, "x32"); run<4, 2, 0>(wt, "x32"); run<8, 2, 0>(wt, "x32");
    run<0, 6, 0>(wt, "x32"); run<4, 6, 0>(wt, "x32"); run<8, 6, 0>(wt, "x32");
    run<0, 1, 1>(wt, "x16"); run<4, 1, 1>(wt, "x16"); run<0, 2, 1>(wt, "x16"); run<4, 2, 1>(wt, "x16"); run<0, 6, 1>(wt, "x16"); run<4, 6, 1>(wt, "x16");
    run<4, 1, 2>(wt, "valu"); run<8, 1, 2>(wt, "valu");
  }
  return 0;
}
