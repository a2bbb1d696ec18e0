// How do the GEMM's epilogue stores behave on their own?  Workgroups of 256 threads (48 KB of LDS each: three per CU, like
// the GEMM) write 128 x 128 f32 tiles of a row-major (M, N) matrix in the accumulator's register order; optionally after a
// register-only MFMA loop of `iters` K-steps (24 MFMAs each).  Store flavours: 0 plain dword, 1 plain dwordx4 (256 B rows),
// 2 nontemporal dwordx4, 3 dwordx4 with sc1 (write-through), 4 no stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_lab scripts/lab/store_lab.hip && ./store_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(1);                                                                             \
    }                                                                                      \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + slot;
}

__device__ unsigned g_cu_slots[4096];

// STAG: how the workgroups that are resident at launch get a start offset of slot * delay_us: 0 none, 1 slot = blockIdx >> 8,
// 2 slot = (blockIdx >> 3) % 3, 3 slot = arrival order on this CU (atomic counter keyed by XCC / SE / SH / CU id)
template <int FLAVOUR, bool XCD, int STAG = 0>
__global__ __launch_bounds__(256, 2) void tile_store_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int N,
                                                            int tiles_n, int iters, int delay_us = 0) {
  __shared__ float pad[12288];                           // 48 KB: three workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (STAG > 0 && blockIdx.x < 768u) {
    unsigned slot;
    if (STAG == 1) slot = blockIdx.x >> 8;
    else if (STAG == 2) slot = (blockIdx.x >> 3) % 3u;
    else {
      __shared__ unsigned s_slot;
      if (tid == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);
        s_slot = atomicAdd(&g_cu_slots[((xcc & 15u) << 8) | ((hw >> 8) & 0xffu)], 1u) % 3u;
      }
      __syncthreads();
      slot = s_slot;
    }
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)slot * delay_us * 100ull;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
  }
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = XCD ? xcd_contiguous_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  if (iters < 0) pad[tid] = 1.f;                         // keeps the array
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)(tid + r);
  if (iters > 0) {
    bf16x8 av[3][2], wv[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = src[(size_t)(l * 4 + b) * 256 + tid];
        wv[l][b] = src[(size_t)(l * 4 + 2 + b) * 256 + tid];
      }
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
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
  }
  const int kg = lane >> 5, li = lane & 31;
  if (FLAVOUR == 4) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (acc[i][j][3] == 123.456f) out[tid] = acc[i][j][5];
    return;
  }
  if (FLAVOUR == 0) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * kg) * N + n0 + wc * 64 + cb * 32 + li;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) op[(size_t)(rb * 32 + (r & 3) + 8 * (r >> 2)) * N] = acc[rb][cb][r];
    }
    return;
  }
  // 16-byte stores, 256 contiguous bytes per row and 16 lanes (values are not transposed here: timing only)
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + rr;
      const f32x4 v = {acc[rb][i >> 2][4 * (i & 3)], acc[rb][i >> 2][4 * (i & 3) + 1], acc[rb][i >> 2][4 * (i & 3) + 2],
                       acc[rb][i >> 2][4 * (i & 3) + 3]};
      float* p = out + (size_t)(m0 + wr * 64 + rb * 32 + row) * N + n0 + wc * 64 + c4;
      if (FLAVOUR == 1) *reinterpret_cast<f32x4*>(p) = v;
      else if (FLAVOUR == 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
      else asm volatile("global_store_dwordx4 %0, %1, off sc1\n" ::"v"(p), "v"(v) : "memory");
    }
}

template <int FLAVOUR, bool XCD>
float run(const bf16x8* src, float* out, int M, int N, int iters) {
  const int tiles_n = N / 128;
  const unsigned grid = (unsigned)(M / 128 * tiles_n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  tile_store_kernel<FLAVOUR, XCD><<<grid, 256>>>(src, out, N, tiles_n, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) tile_store_kernel<FLAVOUR, XCD><<<grid, 256>>>(src, out, N, tiles_n, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}


// (A) persistent workgroups, the stores of tile t issued one per K-step inside the MFMA loop of tile t+1 (accumulators of the
//     finished tile parked in 64 more registers);  (B) the plain kernel at LDSF floats of LDS per workgroup (more per CU).
template <int LDSF, bool DEFER>
__global__ __launch_bounds__(256, 2) void tile_store_persist_kernel(const bf16x8* __restrict__ src, float* __restrict__ out, int N,
                                                                    int tiles_n, int ntiles, int iters) {
  __shared__ float pad[LDSF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  if (iters < 0) pad[tid] = 1.f;
  bf16x8 av[3][2], wv[3][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      av[l][b] = src[(size_t)(l * 4 + b) * 256 + tid];
      wv[l][b] = src[(size_t)(l * 4 + 2 + b) * 256 + tid];
    }
  constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
  const int rr = lane >> 4, c4 = (lane & 15) * 4;
  f32x16 prev[2][2];
  float* pbase = nullptr;
  for (int tile = blockIdx.x; tile < ntiles + (DEFER ? (int)gridDim.x : 0); tile += gridDim.x) {
    const bool live = tile < ntiles;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)(tid + r);
#pragma unroll 1
    for (int it = 0; it < iters; it += 16) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (live) {
#pragma unroll
          for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
        }
        if (DEFER && it == 0 && pbase) {               // one 16-byte store of the previous tile per K-step
          const int rb = u >> 3, i = u & 7;
          const f32x4 v = {prev[rb][i >> 2][4 * (i & 3)], prev[rb][i >> 2][4 * (i & 3) + 1], prev[rb][i >> 2][4 * (i & 3) + 2],
                           prev[rb][i >> 2][4 * (i & 3) + 3]};
          *reinterpret_cast<f32x4*>(pbase + (size_t)(rb * 32 + 4 * i + rr) * N) = v;
        }
      }
    }
    float* base = out + (size_t)(tm * 128 + wr * 64) * N + tn * 128 + wc * 64 + c4;
    if (DEFER) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) prev[i][j] = acc[i][j];
      pbase = live ? base : nullptr;
    } else if (live) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 v = {acc[rb][i >> 2][4 * (i & 3)], acc[rb][i >> 2][4 * (i & 3) + 1], acc[rb][i >> 2][4 * (i & 3) + 2],
                           acc[rb][i >> 2][4 * (i & 3) + 3]};
          *reinterpret_cast<f32x4*>(base + (size_t)(rb * 32 + 4 * i + rr) * N) = v;
        }
    }
  }
}

template <int LDSF, bool DEFER>
float run_persist(const bf16x8* src, float* out, int M, int N, int iters, int wg_per_cu) {
  const int tiles_n = N / 128, ntiles = M / 128 * tiles_n;
  const unsigned grid = wg_per_cu > 0 ? (unsigned)(256 * wg_per_cu) : (unsigned)ntiles;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  tile_store_persist_kernel<LDSF, DEFER><<<grid, 256>>>(src, out, N, tiles_n, ntiles, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) tile_store_persist_kernel<LDSF, DEFER><<<grid, 256>>>(src, out, N, tiles_n, ntiles, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

template <int STAG>
float run_stag(const bf16x8* src, float* out, int M, int N, int iters, int delay_us) {
  const int tiles_n = N / 128;
  const unsigned grid = (unsigned)(M / 128 * tiles_n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  tile_store_kernel<1, true, STAG><<<grid, 256>>>(src, out, N, tiles_n, iters, delay_us);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) tile_store_kernel<1, true, STAG><<<grid, 256>>>(src, out, N, tiles_n, iters, delay_us);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

int main() {
  const int M = 618240;
  float* out;
  bf16x8* src;
  CK(hipMalloc(&out, (size_t)M * 1024 * 4));
  CK(hipMalloc(&src, 12 * 256 * 16));
  CK(hipMemset(src, 0x3c, 12 * 256 * 16));
  const char* names[5] = {"dword (shipped order)", "dwordx4, 256 B rows", "dwordx4 nontemporal", "dwordx4 sc1", "no stores"};
  for (int N : {1024, 256}) {
    for (int iters : {0, 16, 64}) {
      printf("N=%d, %d K-steps of MFMA before the stores (%.2f GB written)\n", N, iters, (double)M * N * 4 / 1e9);
      float t[5];
      t[0] = run<0, true>(src, out, M, N, iters);
      t[1] = run<1, true>(src, out, M, N, iters);
      t[2] = run<2, true>(src, out, M, N, iters);
      t[3] = run<3, true>(src, out, M, N, iters);
      t[4] = run<4, true>(src, out, M, N, iters);
      const float tn = run<1, false>(src, out, M, N, iters);
      for (int f = 0; f < 5; ++f)
        printf("   %-24s %.3f ms  %s%.2f TB/s\n", names[f], t[f], f == 4 ? "(" : "", f == 4 ? 0.0 : (double)M * N * 4 / t[f] / 1e9);
      printf("   %-24s %.3f ms  %.2f TB/s\n", "dwordx4, plain block order", tn, (double)M * N * 4 / tn / 1e9);
    }
  }
  printf("N = 1024, 16 K-steps, dwordx4 stores: one workgroup per tile at 3 / 4 / 5 per CU (LDS 48 / 36 / 30 KB): %.3f %.3f %.3f ms\n",
         run_persist<12288, false>(src, out, M, 1024, 16, 0), run_persist<9216, false>(src, out, M, 1024, 16, 0),
         run_persist<7680, false>(src, out, M, 1024, 16, 0));
  printf("   persistent workgroups, stores at the end of each tile, 2 / 3 / 4 per CU: %.3f %.3f %.3f ms\n",
         run_persist<12288, false>(src, out, M, 1024, 16, 2), run_persist<12288, false>(src, out, M, 1024, 16, 3),
         run_persist<9216, false>(src, out, M, 1024, 16, 4));
  printf("   persistent workgroups, previous tile's stores spread over the next tile's K-steps, 2 / 3 per CU: %.3f %.3f ms\n",
         run_persist<12288, true>(src, out, M, 1024, 16, 2), run_persist<12288, true>(src, out, M, 1024, 16, 3));
  printf("   same at N = 256 (2 / 3 per CU): %.3f %.3f ms; end-of-tile stores, 3 per CU: %.3f ms\n",
         run_persist<12288, true>(src, out, M, 256, 16, 2), run_persist<12288, true>(src, out, M, 256, 16, 3),
         run_persist<12288, false>(src, out, M, 256, 16, 3));
  printf("start stagger of the resident workgroups (dwordx4 stores, N = 1024, 16 K-steps): ms by delay per slot\n");
  for (int d : {4, 8, 12}) {
    printf("   delay %2d us: slot=blk>>8 %.3f   slot=(blk>>3)%%3 %.3f   slot=arrival order on the CU %.3f\n", d,
           run_stag<1>(src, out, M, 1024, 16, d), run_stag<2>(src, out, M, 1024, 16, d), run_stag<3>(src, out, M, 1024, 16, d));
  }
  return 0;
}
