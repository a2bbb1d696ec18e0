// VALU-port cost of the instructions the bf16-split attention kernel is made of (gfx950), 2 waves per SIMD, no MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/lab/valu_lab scripts/lab/valu_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(float* sink, int reps, float seed) {
  float v[8]; unsigned u[8];
  for (int i = 0; i < 8; ++i) { v[i] = seed + i + threadIdx.x * 0.01f; u[i] = threadIdx.x * 7 + i; }
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) v[i] = v[i] * 1.0001f + 0.5f;                                             // v_fma_f32
        if (OP == 1) { u[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[i], v[(i + 1) & 7]}, bf16x2)); }  // v_cvt_pk_bf16_f32
        if (OP == 2) { f32x2 a = {v[i], v[(i + 1) & 7]}, b = {1.5f, 2.5f}; a = a - b; asm volatile("" : "+v"(a)); v[i] = a[0]; }   // v_pk_add_f32 (+mov)
        if (OP == 3) v[i] = __builtin_amdgcn_exp2f(v[i]);                                      // v_exp_f32
        if (OP == 4) v[i] = (u[i] & 4u) ? -1.f : v[i];                                         // v_and + v_cmp + v_cndmask
        if (OP == 5) u[i] = (u[i] << 16) ^ 0x1234u;                                            // v_lshl + v_xor (or v_lshl_xor?)
        if (OP == 6) v[i] = fmaxf(v[i], v[(i + 1) & 7]);                                       // v_max_f32
        asm volatile("" : "+v"(v[i]), "+v"(u[i]));
      }
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + u[i];
  if (s == 12345.678f) sink[0] = s;
}
template <int OP> void run(const char* name) {
  float* sink; (void)hipMalloc(&sink, 4);
  const int reps = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, sink, 100, 1.f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(512), 0, 0, sink, reps, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.2f ns per op-group per wave pair -> %.2f ns per SIMD op-group\n", name, ms * 1e6 / (reps * 64.0), ms * 1e6 / (reps * 64.0) / 2);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_cvt_pk_bf16_f32"); run<2>("v_pk_add_f32(+mov?)"); run<3>("v_exp_f32"); run<4>("and+cmp+cndmask");
  run<5>("lshl+xor"); run<6>("v_max_f32"); run<0>("v_fma_f32 again");
  return 0;
}
