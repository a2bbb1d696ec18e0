// Stand-alone laboratory for the split-bf16 GEMM (no torch): ablations of the shipped kernel's K-step and the
// pre-split / LDS-DMA-fed variants, timed with HIP events on random data, interleaved rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_lab scripts/lab/gemm_lab.hip && ./gemm_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e__ = (x);                                                                      \
    if (e__ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__));     \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned xcd = bid & 7u, slot = bid >> 3;
  const unsigned base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + slot;
}

static int g_cus = 256;
constexpr int GB_M = 128, GB_N = 128, GB_K = 16;
constexpr int GB_LIMB = 2 * GB_M * 8;
constexpr int GB_TILE = 3 * GB_LIMB;
constexpr int GB_STAGE = 2 * GB_TILE;

__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
  const bf16x2 hh = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
  const bf16x2 mm = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
  m = __builtin_bit_cast(unsigned, mm);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

// ---------------------------------------------------------------------------------------------------------------
// The shipped kernel with ablation switches.  ABL: 0 = as shipped, 1 = no VALU split (raw bits staged),
// 2 = no staging writes (loads kept alive), 3 = no global loads either, 4 = MFMAs only (no LDS reads, no barrier)
__device__ unsigned long long g_phase[8];      // [0] prologue, [1] loop, [2] epilogue cycles (wave 0 of every block), [3] blocks

// ABL 5 = 4 without the epilogue stores; ABL 10 = shipped + phase timing (s_memtime)
template <int ABL>
__global__ __launch_bounds__(256, 2)
void gemm_abl_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * GB_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned long long tp0 = __builtin_readcyclecounter();
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  const int ar = tid >> 1, akg = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 8 * akg) * 4) : 0x80000000u;
  const size_t a_base = (size_t)m0 * K * 4;
  const auto asrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + a_base), 0,
                                                        (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;

  f32x4 a_regs[2][2];
  u32x4 w_regs[2][3];
  auto fetch = [&](int slot, int kt) {
    if (ABL >= 3 && ABL < 10) return;
    const unsigned so = (unsigned)kt * (GB_K * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a_regs[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc_t, a_voff + 16 * q, so, 0));
    const __bf16* wk = wsrc + (size_t)kt * 3 * w_limb_stride;
#pragma unroll
    for (int l = 0; l < 3; ++l) w_regs[slot][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
  };
  auto stash = [&](int slot, __bf16* st) {
    if (ABL >= 3 && ABL < 10) return;
    if (ABL == 2) {
#pragma unroll
      for (int q = 0; q < 2; ++q) asm volatile("" ::"v"(a_regs[slot][q]));
#pragma unroll
      for (int i = 0; i < 3; ++i) asm volatile("" ::"v"(w_regs[slot][i]));
      return;
    }
    u32x4 h, m, l;
    if (ABL == 1) {
      h = __builtin_bit_cast(u32x4, a_regs[slot][0]);
      m = __builtin_bit_cast(u32x4, a_regs[slot][1]);
      l = h;
    } else {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        split2(a_regs[slot][q][0], a_regs[slot][q][1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(a_regs[slot][q][2], a_regs[slot][q][3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      h = u32x4{hh[0], hh[1], hh[2], hh[3]};
      m = u32x4{mm[0], mm[1], mm[2], mm[3]};
      l = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
    __bf16* pa = st + (akg * GB_M + ar) * 8;
    *reinterpret_cast<u32x4*>(pa) = h;
    *reinterpret_cast<u32x4*>(pa + GB_LIMB) = m;
    *reinterpret_cast<u32x4*>(pa + 2 * GB_LIMB) = l;
    __bf16* pw = st + GB_TILE + (wkg * GB_N + wcol) * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(pw + i * GB_LIMB) = w_regs[slot][i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = K / GB_K;
  const int kg = lane >> 5, li = lane & 31;
  const int a_off = (kg * GB_M + wr * 64 + li) * 8, w_off = GB_TILE + (kg * GB_N + wc * 64 + li) * 8;
  fetch(0, 0);
  stash(0, lds);
  fetch(1, KT > 1 ? 1 : 0);
  fetch(0, KT > 2 ? 2 : KT - 1);
  bf16x8 av[3][2], wv[3][2];
  f32x4 acc4[2][2][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc4[i >> 3][(i >> 2) & 1][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ABL == 4 || ABL == 5) {   // operands: whatever the (never written) LDS holds once -- random-ish bits from A instead
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = *reinterpret_cast<const bf16x8*>(A + (size_t)(tid * 8 + l * 2 + b) * 4);
        wv[l][b] = *reinterpret_cast<const bf16x8*>(Wp + (size_t)(tid * 8 + l * 2 + b) * 8);
      }
  }
  f32x4 acc16[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc16[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ABL 22: v_mfma_f32_16x16x32_bf16 on the K = 16 stages -- the 32-deep K of one instruction holds TWO limb products of the same
  // 16 k values: lanes 0-31 (k slots 0..15) read one limb, lanes 32-63 (slots 16..31) another:
  //   [a_h|a_h].[w_h|w_m] = hh + hm,   [a_m|a_l].[w_h|w_h] = mh + lh,   [a_h|a_m].[w_l|w_m] = hl + mm
  const int l15 = lane & 15, kgp = (lane >> 4) & 1, hsel = lane >> 5;
  const int a16 = (kgp * GB_M + wr * 64 + l15) * 8, w16 = GB_TILE + (kgp * GB_N + wc * 64 + l15) * 8;
  const int la[3] = {0, (hsel ? 2 : 1) * GB_LIMB, (hsel ? 1 : 0) * GB_LIMB};      // A: [h|h], [m|l], [h|m]
  const int lw[3] = {(hsel ? 1 : 0) * GB_LIMB, 0, (hsel ? 1 : 2) * GB_LIMB};      // W: [h|m], [h|h], [l|m]
  auto kstep = [&](int kt, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    if (ABL == 22) {
      __syncthreads();
      const __bf16* cur = lds + par * GB_STAGE;
      stash(par ^ 1, lds + (par ^ 1) * GB_STAGE);
      fetch(par ^ 1, kt + 3 < KT ? kt + 3 : KT - 1);
#pragma unroll
      for (int c = 2; c >= 0; --c) {             // small terms first
        bf16x8 ap[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) ap[rb] = *reinterpret_cast<const bf16x8*>(cur + a16 + la[c] + rb * 16 * 8);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const bf16x8 wp = *reinterpret_cast<const bf16x8*>(cur + w16 + lw[c] + cb * 16 * 8);
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
            acc16[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[rb], wp, acc16[rb][cb], 0, 0, 0);
        }
      }
      return;
    }
    if (ABL != 4 && ABL != 5) {
      __syncthreads();
      const __bf16* cur = lds + par * GB_STAGE;
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          av[l][b] = *reinterpret_cast<const bf16x8*>(cur + a_off + l * GB_LIMB + b * 32 * 8);
          wv[l][b] = *reinterpret_cast<const bf16x8*>(cur + w_off + l * GB_LIMB + b * 32 * 8);
        }
      stash(par ^ 1, lds + (par ^ 1) * GB_STAGE);
      fetch(par ^ 1, kt + 3 < KT ? kt + 3 : KT - 1);
    }
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
    if (ABL == 21) {
      // TIMING ONLY (numerically meaningless): every v_mfma_f32_32x32x16_bf16 replaced by two v_mfma_f32_16x16x32_bf16 on the
      // same operand registers -- same flops and matrix-pipe cycles, everything else of the kernel unchanged.  Does the
      // instruction's lower energy per flop (mfma_lab: 0.85 vs 0.73 of the roof on limb data) survive inside the real kernel?
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc4[rb][cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc4[rb][cb][0], 0, 0, 0);
            acc4[rb][cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc4[rb][cb][1], 0, 0, 0);
          }
    } else {
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  const unsigned long long tp1 = __builtin_readcyclecounter();
  int kt = 0;
  for (; kt + 2 <= KT; kt += 2) {
    kstep(kt, P0{});
    kstep(kt + 1, P1{});
  }
  if (kt < KT) kstep(kt, P0{});
  unsigned long long tp2 = 0;
  if (ABL == 10) {
    asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][1]));
    tp2 = __builtin_readcyclecounter();
  }

  unsigned long long tpe[2] = {0, 0};
  if (ABL == 22) {
    const int g4 = lane >> 4;
    const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
    if (full) {                                   // no guards, bias waited for once: the 64 stores issue back to back
      float bv[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc16[rb][cb][r] + bv[cb];
      }
      return;
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = n0 + wc * 64 + cb * 16 + l15;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
          if (row < M) out[(size_t)row * N + col] = acc16[rb][cb][r] + bv;
        }
    }
    return;
  }
  if (ABL == 21) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = acc4[rb][cb][r >> 2][r & 3] + acc4[rb][cb][(r >> 2) ^ 1][r & 3];
  }
  if (ABL == 11 || ABL == 12 || ABL == 13 || ABL == 21) {
    // full tiles: no per-element guards, bias fetched and waited for ONCE -> 64 stores issue back to back
    // (the guarded form makes the compiler put an s_waitcnt vmcnt(0) in front of every store: each waits for its predecessor)
    const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
    if (full) {
      float bv[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 32 + li] : 0.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long te0 = __builtin_readcyclecounter();
      if (ABL == 11 || ABL == 13 || ABL == 21) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          float* op = out + (size_t)(m0 + wr * 64 + 4 * kg) * N + n0 + wc * 64 + cb * 32 + li;
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[(size_t)(rb * 32 + (r & 3) + 8 * (r >> 2)) * N] = acc[rb][cb][r] + bv[cb];
        }
        if (ABL == 13) {
          asm volatile("" ::: "memory");
          const unsigned long long te1 = __builtin_readcyclecounter();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const unsigned long long te2 = __builtin_readcyclecounter();
          if (tid == 0) {
            atomicAdd(&g_phase[0], tp1 - tp0);
            atomicAdd(&g_phase[1], te0 - tp1);
            atomicAdd(&g_phase[2], te2 - te0);
            atomicAdd(&g_phase[3], 1ull);
            atomicAdd(&g_phase[4], te1 - te0);
            atomicAdd(&g_phase[5], 0ull);
            atomicAdd(&g_phase[6], te2 - te1);
          }
        }
      } else {
        // transposed through LDS: 16-byte stores, 256 contiguous bytes per row
        __syncthreads();
        float* reg = reinterpret_cast<float*>(lds) + wave * (32 * 64);
        const int rr = lane >> 4, c4 = (lane & 15) * 4;
        const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + n0 + wc * 64 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + 4 * kg) * 64 + cb * 32 + li] = acc[rb][cb][r];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + rr;
            f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 64 + c4);
            v += b4;
            *reinterpret_cast<f32x4*>(out + (size_t)(m0 + wr * 64 + rb * 32 + row) * N + n0 + wc * 64 + c4) = v;
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int col = n0 + wc * 64 + cb * 32 + li;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
    float* op = out + (size_t)(m0 + wr * 64 + 4 * kg) * N + col;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rrel = rb * 32 + (r & 3) + 8 * (r >> 2);
        if (ABL == 5) {
          if (acc[rb][cb][r] == 123.456f) op[(size_t)rrel * N] = acc[rb][cb][r] + bv;
        } else if (m0 + wr * 64 + 4 * kg + rrel < M) op[(size_t)rrel * N] = acc[rb][cb][r] + bv;
      }
    if (ABL == 10) {
      asm volatile("" ::: "memory");
      tpe[cb] = __builtin_readcyclecounter();
    }
  }
  if (ABL == 10) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tp3 = __builtin_readcyclecounter();
    if (tid == 0) {
      atomicAdd(&g_phase[0], tp1 - tp0);
      atomicAdd(&g_phase[1], tp2 - tp1);
      atomicAdd(&g_phase[2], tp3 - tp2);
      atomicAdd(&g_phase[3], 1ull);
      atomicAdd(&g_phase[4], tpe[0] - tp2);
      atomicAdd(&g_phase[5], tpe[1] - tpe[0]);
      atomicAdd(&g_phase[6], tp3 - tpe[1]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pre-split operands, both staged by LDS-DMA.  Ap: [row tile M/128][k-tile K/16][limb 3][kg 2][row 128][8] bf16 (12 KB
// per (row tile, k-tile)); Wp as packed today.  NS LDS stages of 24 KB, DMA NS-1 K-steps ahead, counted vmcnt, raw
// barrier.  Wave w brings 6 KB of a stage: waves 0,1 the A chunk, waves 2,3 the W pieces.
// PERSIST: grid = resident workgroups, each loops over tiles; the first stages of the next tile are requested before the
// epilogue stores.
template <int NS, bool PERSIST, int WPS /* min waves per SIMD for launch bounds */>
__global__ __launch_bounds__(256, WPS)
void gemm_ps_kernel(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                    float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int KT = K / GB_K;
  const int kg = lane >> 5, li = lane & 31;
  const int a_off = (kg * GB_M + wr * 64 + li) * 8, w_off = GB_TILE + (kg * GB_N + wc * 64 + li) * 8;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  constexpr int D = NS - 1;

  int tile = PERSIST ? (int)blockIdx.x : (int)xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tstep = PERSIST ? (int)gridDim.x : ntiles;

  // DMA of k-tile kt of tile (tm, tn) into stage st: 6 x 1 KB per wave
  auto issue = [&](int tm, int tn, int kt, int st) {
    __bf16* sbase = lds + st * GB_STAGE + wave * (6 * 512);
    if (wave < 2) {
      const __bf16* g = Ap + ((size_t)tm * KT + kt) * GB_TILE + wave * (6 * 512) + lane * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds(g + i * 512, (lds_ptr_t)(sbase + i * 512), 16, 0, 0);
    } else {
      const __bf16* gk = Wp + (size_t)kt * 3 * w_limb_stride + (size_t)(tn * GB_N + lane) * 8;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int p = (wave - 2) * 6 + i;                  // piece 0..11 = (limb p/4, kg (p/2)%2, half p%2)
        const __bf16* g = gk + (size_t)(p >> 2) * w_limb_stride + (size_t)((p >> 1) & 1) * Npad * 8 + (p & 1) * 512;
        __builtin_amdgcn_global_load_lds(g, (lds_ptr_t)(sbase + i * 512), 16, 0, 0);
      }
    }
  };

  int tm = tile / tiles_n, tn = tile % tiles_n;
  if (tile < ntiles) {
#pragma unroll
    for (int s = 0; s < D; ++s) issue(tm, tn, s < KT ? s : KT - 1, s);
  }
  int st_rd = 0;                                          // stage holding k-tile kt
  while (tile < ntiles) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int ntile_next = tile + tstep;
    const int tm2 = ntile_next / tiles_n, tn2 = ntile_next % tiles_n;
    const bool more = PERSIST && ntile_next < ntiles;
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
      if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (D == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // the stage read in the previous K-step is free now: request k-tile kt + D (or the next tile's first stages)
      int st_wr = st_rd + D;
      if (st_wr >= NS) st_wr -= NS;
      const int kn = kt + D;
      if (kn < KT) issue(tm, tn, kn, st_wr);
      else if (more) issue(tm2, tn2, kn - KT, st_wr);
      else issue(tm, tn, KT - 1, st_wr);                  // keeps the vmcnt arithmetic uniform; never read
      const __bf16* cur = lds + st_rd * GB_STAGE;
      bf16x8 av[3][2], wv[3][2];
#pragma unroll
      for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          av[l][b] = *reinterpret_cast<const bf16x8*>(cur + a_off + l * GB_LIMB + b * 32 * 8);
          wv[l][b] = *reinterpret_cast<const bf16x8*>(cur + w_off + l * GB_LIMB + b * 32 * 8);
        }
      constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
      st_rd = st_rd + 1 == NS ? 0 : st_rd + 1;
    }
    const int m0 = tm * GB_M, n0 = tn * GB_N;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = n0 + wc * 64 + cb * 32 + li;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      float* op = out + (size_t)(m0 + wr * 64 + 4 * kg) * N + col;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rrel = rb * 32 + (r & 3) + 8 * (r >> 2);
          if (m0 + wr * 64 + 4 * kg + rrel < M) op[(size_t)rrel * N] = acc[rb][cb][r] + bv;
        }
    }
    if (!PERSIST) break;
    tile = ntile_next;
    tm = tm2;
    tn = tn2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no DMA may land after the workgroup's LDS is released
}


// ---------------------------------------------------------------------------------------------------------------
// Epilogue study.  Same main loop as the shipped kernel (f32 A split while staged), different accumulator -> memory paths:
//   EPI 1: operand roles swapped (accumulator rows = output columns n, lanes = token rows m): a lane owns 4 consecutive n
//          -> 16 dwordx4 stores per wave (32 rows x 32 B per instruction)
//   EPI 2: as shipped (rows = tokens), accumulators transposed through the (now free) LDS stages, 16 dwordx4 stores per wave,
//          256 contiguous bytes per row and instruction
// STAG > 0: the first generation of workgroups (the 3 x 256 that are resident at launch) starts staggered: residency slot
// s = blockIdx / 256 waits s * STAG microseconds, so that the three workgroups of a CU are not in the same phase (loop /
// store burst) at the same time; later workgroups start when a predecessor ends and inherit the offset.
template <int EPI, int STAG = 0>
__global__ __launch_bounds__(256, 2)
void gemm_epi_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * GB_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (STAG > 0 && blockIdx.x < 768u) {
    const unsigned slot = blockIdx.x >> 8;
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)slot * STAG * 100ull;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
  }
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  const int ar = tid >> 1, akg = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 8 * akg) * 4) : 0x80000000u;
  const size_t a_base = (size_t)m0 * K * 4;
  const auto asrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + a_base), 0,
                                                        (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs[2][2];
  u32x4 w_regs[2][3];
  auto fetch = [&](int slot, int kt) {
    const unsigned so = (unsigned)kt * (GB_K * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a_regs[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc_t, a_voff + 16 * q, so, 0));
    const __bf16* wk = wsrc + (size_t)kt * 3 * w_limb_stride;
#pragma unroll
    for (int l = 0; l < 3; ++l) w_regs[slot][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
  };
  auto stash = [&](int slot, __bf16* st) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      split2(a_regs[slot][q][0], a_regs[slot][q][1], hh[2 * q], mm[2 * q], ll[2 * q]);
      split2(a_regs[slot][q][2], a_regs[slot][q][3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
    }
    const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    __bf16* pa = st + (akg * GB_M + ar) * 8;
    *reinterpret_cast<u32x4*>(pa) = h;
    *reinterpret_cast<u32x4*>(pa + GB_LIMB) = m;
    *reinterpret_cast<u32x4*>(pa + 2 * GB_LIMB) = l;
    __bf16* pw = st + GB_TILE + (wkg * GB_N + wcol) * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(pw + i * GB_LIMB) = w_regs[slot][i];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KT = K / GB_K;
  const int kg = lane >> 5, li = lane & 31;
  // EPI 1: wave (wr, wc) = (n half, m half); otherwise (m half, n half)
  const int a_off = (kg * GB_M + (EPI == 1 ? wc : wr) * 64 + li) * 8;
  const int w_off = GB_TILE + (kg * GB_N + (EPI == 1 ? wr : wc) * 64 + li) * 8;
  fetch(0, 0);
  stash(0, lds);
  fetch(1, KT > 1 ? 1 : 0);
  fetch(0, KT > 2 ? 2 : KT - 1);
  auto kstep = [&](int kt, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    __syncthreads();
    const __bf16* cur = lds + par * GB_STAGE;
    bf16x8 av[3][2], wv[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = *reinterpret_cast<const bf16x8*>(cur + a_off + l * GB_LIMB + b * 32 * 8);
        wv[l][b] = *reinterpret_cast<const bf16x8*>(cur + w_off + l * GB_LIMB + b * 32 * 8);
      }
    stash(par ^ 1, lds + (par ^ 1) * GB_STAGE);
    fetch(par ^ 1, kt + 3 < KT ? kt + 3 : KT - 1);
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if (EPI == 1)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[PW[p]][rb], av[PA[p]][cb], acc[rb][cb], 0, 0, 0);
          else
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
        }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 2 <= KT; kt += 2) {
    kstep(kt, P0{});
    kstep(kt + 1, P1{});
  }
  if (kt < KT) kstep(kt, P0{});

  if (EPI == 1) {
    // register r of block (rb, cb): n = n0 + wr*64 + 32 rb + (r&3) + 8 (r>>2) + 4 kg,  m = m0 + wc*64 + 32 cb + li
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int m = m0 + wc * 64 + cb * 32 + li;
      if (m >= M) continue;
      float* orow = out + (size_t)m * N;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wr * 64 + rb * 32 + 8 * j + 4 * kg;
          if (n < N) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (bias) b4 = *reinterpret_cast<const f32x4*>(bias + n);
            f32x4 v = {acc[rb][cb][4 * j] + b4[0], acc[rb][cb][4 * j + 1] + b4[1], acc[rb][cb][4 * j + 2] + b4[2],
                       acc[rb][cb][4 * j + 3] + b4[3]};
            *reinterpret_cast<f32x4*>(orow + n) = v;
          }
        }
    }
  } else {
    // transposition through LDS: wave-private 8 KB region, one 32-row half at a time: [32 rows][64 floats]
    __syncthreads();                                     // every wave is past its last fragment reads
    float* reg = reinterpret_cast<float*>(lds) + wave * (32 * 64);
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    const int ncol = n0 + wc * 64 + c4;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ncol < N) b4 = *reinterpret_cast<const f32x4*>(bias + ncol);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + 4 * kg) * 64 + cb * 32 + li] = acc[rb][cb][r];
      // (same wave wrote and reads: no barrier, the compiler's lgkmcnt orders it)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + rr;
        f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * 64 + c4);
        const int m = m0 + wr * 64 + rb * 32 + row;
        if (m < M && ncol < N) {
          v += b4;
          *reinterpret_cast<f32x4*>(out + (size_t)m * N + ncol) = v;
        }
      }
    }
  }
}

template <int EPI, int STAG = 0>
void launch_epi(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL((gemm_epi_kernel<EPI, STAG>), dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}


// ---------------------------------------------------------------------------------------------------------------
// v2: persistent workgroups, ONE continuous stream of K-steps over the workgroup's tiles (the loads of the next tile's first
// K-steps are requested under the last K-steps of the current one: no per-tile prologue), operand roles swapped (accumulator
// rows = output columns: a lane owns 4 consecutive n -> 16-byte stores), and the finished tile's 16 stores issued ONE PER
// K-STEP inside the next tile's K loop (inline asm: invisible to the compiler's vmcnt bookkeeping, so its counted waits for the
// operand loads stay counted).  Bias enters through the accumulator's initial value (staged in LDS once per workgroup).
// Needs K % 32 == 0 and K >= 256 (16 K-steps carry the 16 stores).
template <bool RELU>
__global__ __launch_bounds__(256, 2)
void gemm_v2_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                    float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];           // 2 stages + Npad floats of bias
  float* bias_lds = reinterpret_cast<float*>(lds + 2 * GB_STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;                               // (n half, m half)
  const int KT = K / GB_K;
  const int kg = lane >> 5, li = lane & 31;
  for (int i = tid; i < Npad; i += 256) bias_lds[i] = (bias && i < N) ? bias[i] : 0.f;

  // this workgroup's tiles: XCD x = blockIdx % 8 owns a contiguous chunk of the tile list, its workgroups take every per-th tile
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int cbase = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, ccnt = q + (xcd < r ? 1 : 0);
  const int n_my = ccnt > j ? (ccnt - j + per - 1) / per : 0;
  if (n_my == 0) return;

  // staging roles as in the shipped kernel: A -- thread = (row tid/2, k-group tid%2); W -- (k-group tid/128, column tid%128)
  const int ar = tid >> 1, akg = tid & 1;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;

  // fetch cursor (runs 3 K-steps ahead of the compute cursor)
  int f_tile = cbase + j, f_kt = 0, f_left = n_my;                       // f_left: tiles not yet completely fetched
  unsigned f_avoff = 0, f_asoff = 0;
  const __bf16* f_w = Wp;
  auto fetch_tile_setup = [&]() {
    const int tm = f_tile / tiles_n, tn = f_tile - tm * tiles_n;
    const int m0 = tm * GB_M;
    f_avoff = (m0 + ar < M) ? (unsigned)((ar * K + 8 * akg) * 4) : 0x80000000u;
    f_asoff = (unsigned)((size_t)m0 * K * 4);
    f_w = Wp + ((size_t)wkg * Npad + tn * GB_N + wcol) * 8;
  };
  fetch_tile_setup();
  f32x4 a_regs[2][2];
  u32x4 w_regs[2][3];
  auto fetch = [&](int slot) {                            // next K-step of the stream (past the end: repeats the last one)
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
      a_regs[slot][qq] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, f_avoff + 16 * qq,
                                                                                           f_asoff + (unsigned)f_kt * 64u, 0));
    const __bf16* wk = f_w + (size_t)f_kt * 3 * w_limb_stride;
#pragma unroll
    for (int l = 0; l < 3; ++l) w_regs[slot][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    if (f_left > 0) {
      if (++f_kt == KT) {
        if (--f_left > 0) {
          f_kt = 0;
          f_tile += per;
          fetch_tile_setup();
        } else {
          f_kt = KT - 1;
        }
      }
    }
  };
  auto stash = [&](int slot, __bf16* st) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      split2(a_regs[slot][qq][0], a_regs[slot][qq][1], hh[2 * qq], mm[2 * qq], ll[2 * qq]);
      split2(a_regs[slot][qq][2], a_regs[slot][qq][3], hh[2 * qq + 1], mm[2 * qq + 1], ll[2 * qq + 1]);
    }
    const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    __bf16* pa = st + (akg * GB_M + ar) * 8;
    *reinterpret_cast<u32x4*>(pa) = h;
    *reinterpret_cast<u32x4*>(pa + GB_LIMB) = m;
    *reinterpret_cast<u32x4*>(pa + 2 * GB_LIMB) = l;
    __bf16* pw = st + GB_TILE + (wkg * GB_N + wcol) * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(pw + i * GB_LIMB) = w_regs[slot][i];
  };

  const int a_off = (kg * GB_M + wc * 64 + li) * 8, w_off = GB_TILE + (kg * GB_N + wr * 64 + li) * 8;
  f32x16 acc[2][2], prev[2][2];
  float* pptr[2] = {nullptr, nullptr};                    // this lane's store bases of the parked tile (cb = 0, 1)
  bool pok[2] = {false, false};
  unsigned pnmask = 0;                                    // bit (rb*4 + j): column group inside N

  fetch(0);
  __syncthreads();                                        // bias staged
  stash(0, lds);
  fetch(1);
  fetch(0);

  // one K-step; S = index of the parked tile's store that rides on it (-1: none)
  auto kstep = [&](auto PAR, auto SIDX) {
    constexpr int par = decltype(PAR)::value, S = decltype(SIDX)::value;
    __syncthreads();
    const __bf16* cur = lds + par * GB_STAGE;
    bf16x8 av[3][2], wv[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = *reinterpret_cast<const bf16x8*>(cur + a_off + l * GB_LIMB + b * 32 * 8);
        wv[l][b] = *reinterpret_cast<const bf16x8*>(cur + w_off + l * GB_LIMB + b * 32 * 8);
      }
    stash(par ^ 1, lds + (par ^ 1) * GB_STAGE);
    fetch(par ^ 1);
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PW[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[PW[p]][rb], av[PA[p]][cb], acc[rb][cb], 0, 0, 0);
    if constexpr (S >= 0) {
      constexpr int cb = S >> 3, rb = (S >> 2) & 1, jj = S & 3;
      if (pok[cb] && ((pnmask >> (rb * 4 + jj)) & 1u)) {
        f32x4 v = {prev[rb][cb][4 * jj], prev[rb][cb][4 * jj + 1], prev[rb][cb][4 * jj + 2], prev[rb][cb][4 * jj + 3]};
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        float* const pp = pptr[cb];
        asm volatile("global_store_dwordx4 %0, %1, off offset:%2" ::"v"(pp), "v"(v), "n"((rb * 32 + 8 * jj) * 4));
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  auto park = [&](int tile) {                             // finished accumulators -> prev, store addresses of that tile
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) prev[i][jx] = acc[i][jx];
    const int nb = tn * GB_N + wr * 64 + 4 * kg;
    pnmask = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (nb + (g >> 2) * 32 + 8 * (g & 3) < N) pnmask |= 1u << g;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int m = tm * GB_M + wc * 64 + cb * 32 + li;
      pok[cb] = m < M;
      pptr[cb] = out + (size_t)m * N + nb;
    }
  };
  auto init_acc = [&](int tile) {                         // accumulators start at the bias of their output column
    const int tn = tile % tiles_n;
    const float* bl = bias_lds + tn * GB_N + wr * 64 + 4 * kg;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + rb * 32 + 8 * jj);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rb][0][4 * jj + e] = acc[rb][1][4 * jj + e] = b4[e];
      }
  };
  auto flush = [&]() {                                    // all 16 stores of the parked tile at once
#pragma unroll
    for (int S = 0; S < 16; ++S) {
      const int cb = S >> 3, rb = (S >> 2) & 1, jj = S & 3;
      if (pok[cb] && ((pnmask >> (rb * 4 + jj)) & 1u)) {
        f32x4 v = {prev[rb][cb][4 * jj], prev[rb][cb][4 * jj + 1], prev[rb][cb][4 * jj + 2], prev[rb][cb][4 * jj + 3]};
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(pptr[cb] + rb * 32 + 8 * jj) = v;
      }
    }
  };

  int tile = cbase + j;
  for (int it = 0; it < n_my; ++it, tile += per) {
    init_acc(tile);
    // first 16 K-steps carry the parked tile's stores (none parked during the first tile: pok = false)
    kstep(P0{}, std::integral_constant<int, 0>{});   kstep(P1{}, std::integral_constant<int, 1>{});
    kstep(P0{}, std::integral_constant<int, 2>{});   kstep(P1{}, std::integral_constant<int, 3>{});
    kstep(P0{}, std::integral_constant<int, 4>{});   kstep(P1{}, std::integral_constant<int, 5>{});
    kstep(P0{}, std::integral_constant<int, 6>{});   kstep(P1{}, std::integral_constant<int, 7>{});
    kstep(P0{}, std::integral_constant<int, 8>{});   kstep(P1{}, std::integral_constant<int, 9>{});
    kstep(P0{}, std::integral_constant<int, 10>{});  kstep(P1{}, std::integral_constant<int, 11>{});
    kstep(P0{}, std::integral_constant<int, 12>{});  kstep(P1{}, std::integral_constant<int, 13>{});
    kstep(P0{}, std::integral_constant<int, 14>{});  kstep(P1{}, std::integral_constant<int, 15>{});
#pragma unroll 1
    for (int kt = 16; kt < KT; kt += 2) {
      kstep(P0{}, std::integral_constant<int, -1>{});
      kstep(P1{}, std::integral_constant<int, -1>{});
    }
    park(tile);
  }
  flush();
}

template <bool RELU>
void launch_v2(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const int ntiles = ((M + 127) / 128) * tiles_n;
  const int lds_bytes = 2 * GB_STAGE * 2 + Npad * 4;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_v2_kernel<RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    once = true;
  }
  const unsigned grid = (unsigned)std::min(ntiles, g_cus * 2) & ~7u;
  hipLaunchKernelGGL((gemm_v2_kernel<RELU>), dim3(grid), dim3(256), lds_bytes, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n, ntiles);
}

// f32 (M, K) -> limb tiles (RN split, as split2)
__global__ void split_tiles_kernel(const float* __restrict__ a, __bf16* __restrict__ ap, int M, int K) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one (row, k pair)
  const long long Mp = (long long)(M + 127) / 128 * 128;
  if (idx >= Mp * (K / 2)) return;
  const int kp = (int)(idx % (K / 2));
  const long long row = idx / (K / 2);
  const int k = 2 * kp;
  float a0 = 0.f, a1 = 0.f;
  if (row < M) {
    a0 = a[row * K + k];
    a1 = a[row * K + k + 1];
  }
  unsigned h, m, l;
  split2(a0, a1, h, m, l);
  const long long tm = row / 128;
  const int r = (int)(row % 128), kt = k / 16, kgi = (k % 16) / 8, e = k % 8;
  const int KT = K / 16;
  unsigned* dst = reinterpret_cast<unsigned*>(ap + ((size_t)(tm * KT + kt)) * GB_TILE + (size_t)(kgi * 128 + r) * 8 + e);
  dst[0] = h;
  dst[GB_LIMB / 2] = m;
  dst[GB_LIMB] = l;
}

__global__ void pack_w_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int N, int K, int Npad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Npad * (K / 2);
  if (idx >= total) return;
  const int kp = (int)(idx % (K / 2)), n = (int)(idx / (K / 2));
  const int k = 2 * kp;
  float a0 = 0.f, a1 = 0.f;
  if (n < N) {
    a0 = w[(size_t)n * K + k];
    a1 = w[(size_t)n * K + k + 1];
  }
  unsigned h, m, l;
  split2(a0, a1, h, m, l);
  const int kt = k / GB_K, kg = (k % GB_K) / 8, e = k % 8;
  const size_t limb_stride = (size_t)2 * Npad * 8;
  unsigned* dst = reinterpret_cast<unsigned*>(wp + ((size_t)kt * 3 * limb_stride + ((size_t)kg * Npad + n) * 8 + e));
  dst[0] = h;
  dst[limb_stride / 2] = m;
  dst[limb_stride] = l;
}

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    unsigned y = x * 0x9e3779b9u + 0x7f4a7c15u;
    y ^= y >> 15; y *= 0x2c1b3c6du; y ^= y >> 12;
    // sum of two uniforms in [-1, 1): full mantissa, both signs
    const float u = ((x >> 8) * (1.f / 8388608.f) - 1.f) + ((y >> 8) * (1.f / 8388608.f) - 1.f);
    p[i] = u * scale;
  }
}

__global__ void diff_kernel(const float* a, const float* b, long long n, unsigned long long* nbad, float* maxabs) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  float mx = 0.f;
  for (; i < n; i += stride) {
    const float d = fabsf(a[i] - b[i]);
    if (!(d == 0.f)) ++bad;
    mx = fmaxf(mx, d);
  }
  if (bad) atomicAdd(nbad, bad);
  atomicMax(reinterpret_cast<int*>(maxabs), __float_as_int(mx));
}


// ---------------------------------------------------------------------------------------------------------------
// K = 32 stages for v_mfma_f32_16x16x32_bf16 (the in-situ probe abl21 says the instruction saves 5-9 % through the power
// limit).  One LDS stage of 48 KB per workgroup (three workgroups per CU as shipped): A and W limbs as [limb][k-group 0..3]
// [row][8 bf16]; the next step's operands are prefetched into registers while this step's MFMAs run, split and written
// between two barriers.  Wave tile 64 x 64 = 4 x 4 blocks of 16 x 16.
// ---------------------------------------------------------------------------------------------------------------
constexpr int K32_LIMB = 4 * GB_M * 8;          // bf16 elements of one limb of a 128 x 32 tile
constexpr int K32_TILE = 3 * K32_LIMB;          // 24 KB
__global__ __launch_bounds__(256, 3)
void gemm_k32_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * K32_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  // staging roles: A -- thread (row tid/2, half tid%2) 16 consecutive floats; W -- (k-group tid/128, column tid%128) of both
  // 16-deep sub-steps, 3 limbs each (the packed layout of the shipped kernel: [kt16][limb][kg][Npad][8])
  const int ar = tid >> 1, ah2 = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 16 * ah2) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs[4];
  u32x4 w_regs[2][3];
  auto fetch = [&](int kt) {                                     // kt counts 32-deep steps
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff + 16 * q, so, 0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __bf16* wk = wsrc + (size_t)(2 * kt + j) * 3 * w_limb_stride;
#pragma unroll
      for (int l = 0; l < 3; ++l) w_regs[j][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {                             // k-group 2*ah2 + gq: floats 8 gq .. 8 gq + 7 of this thread
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      __bf16* pa = lds + ((2 * ah2 + gq) * GB_M + ar) * 8;
      *reinterpret_cast<u32x4*>(pa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
      *reinterpret_cast<u32x4*>(pa + K32_LIMB) = u32x4{mm[0], mm[1], mm[2], mm[3]};
      *reinterpret_cast<u32x4*>(pa + 2 * K32_LIMB) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* pw = lds + K32_TILE + ((2 * j + wkg) * GB_N + wcol) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pw + l * K32_LIMB) = w_regs[j][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + (kg4 * GB_M + wr * 64 + l15) * 8;                 // + limb * K32_LIMB + rb * 128
  const __bf16* wfr = lds + K32_TILE + (kg4 * GB_N + wc * 64 + l15) * 8;      // + limb * K32_LIMB + cb * 128
  auto mf = [](bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
  const int KT = K / 32;
  fetch(0);
  stash();
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();                                             // step kt is in LDS
    fetch(kt + 1 < KT ? kt + 1 : KT - 1);
    bf16x8 ahf[4], amf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = *reinterpret_cast<const bf16x8*>(afr + rb * 128);
      amf[rb] = *reinterpret_cast<const bf16x8*>(afr + K32_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
      const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wfr + K32_LIMB + cb * 128);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wfr + 2 * K32_LIMB + cb * 128);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wl, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wh, acc[rb][cb]);
    }
    {                                                            // the (lo, hi) product: A's low limb takes over amf's registers
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = *reinterpret_cast<const bf16x8*>(afr + 2 * K32_LIMB + rb * 128);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
      }
    }
    __syncthreads();                                             // everyone is done reading step kt
    if (kt + 1 < KT) stash();
  }
  const int g4 = lane >> 4;
  const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
  if (full) {
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc[rb][cb][r] + bv[cb];
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = n0 + wc * 64 + cb * 16 + l15;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
        if (row < M) out[(size_t)row * N + col] = acc[rb][cb][r] + bv;
      }
  }
}
__global__ __launch_bounds__(256, 3)
void gemm_k32p2_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * K32_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  // staging roles: A -- thread (row tid/2, half tid%2) 16 consecutive floats; W -- (k-group tid/128, column tid%128) of both
  // 16-deep sub-steps, 3 limbs each (the packed layout of the shipped kernel: [kt16][limb][kg][Npad][8])
  const int ar = tid >> 1, ah2 = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 16 * ah2) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs2[2][4];
  u32x4 w_regs[2][3];
  auto fetchA = [&](int slot, int kt) {
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs2[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff + 16 * q, so, 0));
  };
  auto fetchW = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __bf16* wk = wsrc + (size_t)(2 * kt + j) * 3 * w_limb_stride;
#pragma unroll
      for (int l = 0; l < 3; ++l) w_regs[j][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    }
  };
  auto stash = [&](int slot) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {                             // k-group 2*ah2 + gq: floats 8 gq .. 8 gq + 7 of this thread
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs2[slot][2 * gq + q];
        split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      __bf16* pa = lds + ((2 * ah2 + gq) * GB_M + ar) * 8;
      *reinterpret_cast<u32x4*>(pa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
      *reinterpret_cast<u32x4*>(pa + K32_LIMB) = u32x4{mm[0], mm[1], mm[2], mm[3]};
      *reinterpret_cast<u32x4*>(pa + 2 * K32_LIMB) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* pw = lds + K32_TILE + ((2 * j + wkg) * GB_N + wcol) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pw + l * K32_LIMB) = w_regs[j][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + (kg4 * GB_M + wr * 64 + l15) * 8;                 // + limb * K32_LIMB + rb * 128
  const __bf16* wfr = lds + K32_TILE + (kg4 * GB_N + wc * 64 + l15) * 8;      // + limb * K32_LIMB + cb * 128
  auto mf = [](bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
  const int KT = K / 32;
  fetchA(0, 0);
  fetchW(0);
  stash(0);
  fetchA(1, KT > 1 ? 1 : 0);
  auto step = [&](int kt, auto SL) {
    constexpr int sl = decltype(SL)::value;                     // register slot holding step kt+1's A
    __syncthreads();                                             // step kt is in LDS
    fetchA(sl ^ 1, kt + 2 < KT ? kt + 2 : KT - 1);
    fetchW(kt + 1 < KT ? kt + 1 : KT - 1);
    bf16x8 ahf[4], amf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = *reinterpret_cast<const bf16x8*>(afr + rb * 128);
      amf[rb] = *reinterpret_cast<const bf16x8*>(afr + K32_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
      const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wfr + K32_LIMB + cb * 128);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wfr + 2 * K32_LIMB + cb * 128);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wl, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wh, acc[rb][cb]);
    }
    {                                                            // the (lo, hi) product: A's low limb takes over amf's registers
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = *reinterpret_cast<const bf16x8*>(afr + 2 * K32_LIMB + rb * 128);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
      }
    }
    __syncthreads();                                             // everyone is done reading step kt
    if (kt + 1 < KT) stash(sl);
  };
  {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int kt = 0;
    for (; kt + 2 <= KT; kt += 2) { step(kt, S1{}); step(kt + 1, S0{}); }
    if (kt < KT) step(kt, S1{});
  }
  const int g4 = lane >> 4;
  const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
  if (full) {
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc[rb][cb][r] + bv[cb];
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = n0 + wc * 64 + cb * 16 + l15;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
        if (row < M) out[(size_t)row * N + col] = acc[rb][cb][r] + bv;
      }
  }
}
void launch_k32p2(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_k32p2_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}
__global__ __launch_bounds__(256, 3)
void gemm_k32e_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * K32_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  // staging roles: A -- thread (row tid/2, half tid%2) 16 consecutive floats; W -- (k-group tid/128, column tid%128) of both
  // 16-deep sub-steps, 3 limbs each (the packed layout of the shipped kernel: [kt16][limb][kg][Npad][8])
  const int ar = tid >> 1, ah2 = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 16 * ah2) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs[4];
  u32x4 w_regs[2][3];
  auto fetch = [&](int kt) {                                     // kt counts 32-deep steps
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff + 16 * q, so, 0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __bf16* wk = wsrc + (size_t)(2 * kt + j) * 3 * w_limb_stride;
#pragma unroll
      for (int l = 0; l < 3; ++l) w_regs[j][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    }
  };
  u32x4 limbs[2][3];                                            // next step's A limbs of this thread's two k-groups
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
      limbs[gq][2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  auto write = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + ((2 * ah2 + gq) * GB_M + ar) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pa + l * K32_LIMB) = limbs[gq][l];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* pw = lds + K32_TILE + ((2 * j + wkg) * GB_N + wcol) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pw + l * K32_LIMB) = w_regs[j][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + (kg4 * GB_M + wr * 64 + l15) * 8;                 // + limb * K32_LIMB + rb * 128
  const __bf16* wfr = lds + K32_TILE + (kg4 * GB_N + wc * 64 + l15) * 8;      // + limb * K32_LIMB + cb * 128
  auto mf = [](bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
  const int KT = K / 32;
  fetch(0);
  split();
  write();
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();                                             // step kt is in LDS
    fetch(kt + 1 < KT ? kt + 1 : KT - 1);
    bf16x8 ahf[4], amf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = *reinterpret_cast<const bf16x8*>(afr + rb * 128);
      amf[rb] = *reinterpret_cast<const bf16x8*>(afr + K32_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
      const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wfr + K32_LIMB + cb * 128);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wfr + 2 * K32_LIMB + cb * 128);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wl, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wh, acc[rb][cb]);
    }
    split();                                                     // next step's A (registers): VALU under the MFMAs
    {                                                            // the (lo, hi) product: A's low limb takes over amf's registers
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = *reinterpret_cast<const bf16x8*>(afr + 2 * K32_LIMB + rb * 128);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
      }
    }
    __syncthreads();                                             // everyone is done reading step kt
    if (kt + 1 < KT) write();
  }
  const int g4 = lane >> 4;
  const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
  if (full) {
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc[rb][cb][r] + bv[cb];
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = n0 + wc * 64 + cb * 16 + l15;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
        if (row < M) out[(size_t)row * N + col] = acc[rb][cb][r] + bv;
      }
  }
}
__global__ __launch_bounds__(256, 3)
void gemm_k32f_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * K32_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  // staging roles: A -- thread (row tid/2, half tid%2) 16 consecutive floats; W -- (k-group tid/128, column tid%128) of both
  // 16-deep sub-steps, 3 limbs each (the packed layout of the shipped kernel: [kt16][limb][kg][Npad][8])
  const int ar = tid >> 1, ah2 = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 16 * ah2) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs[4];
  u32x4 w_regs[2][3];
  auto fetch = [&](int kt) {                                     // kt counts 32-deep steps
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff + 16 * q, so, 0));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __bf16* wk = wsrc + (size_t)(2 * kt + j) * 3 * w_limb_stride;
#pragma unroll
      for (int l = 0; l < 3; ++l) w_regs[j][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    }
  };
  u32x4 limbs[2][3];                                            // next step's A limbs of this thread's two k-groups
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
      limbs[gq][2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  auto write = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + ((2 * ah2 + gq) * GB_M + ar) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pa + l * K32_LIMB) = limbs[gq][l];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* pw = lds + K32_TILE + ((2 * j + wkg) * GB_N + wcol) * 8;
#pragma unroll
      for (int l = 0; l < 3; ++l) *reinterpret_cast<u32x4*>(pw + l * K32_LIMB) = w_regs[j][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + (kg4 * GB_M + wr * 64 + l15) * 8;                 // + limb * K32_LIMB + rb * 128
  const __bf16* wfr = lds + K32_TILE + (kg4 * GB_N + wc * 64 + l15) * 8;      // + limb * K32_LIMB + cb * 128
  auto mf = [](bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
  const int KT = K / 32;
  fetch(0);
  split();
  write();
  fetch(KT > 1 ? 1 : 0);
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();                                             // step kt is in LDS
    bf16x8 ahf[4], amf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = *reinterpret_cast<const bf16x8*>(afr + rb * 128);
      amf[rb] = *reinterpret_cast<const bf16x8*>(afr + K32_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
      const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wfr + K32_LIMB + cb * 128);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wfr + 2 * K32_LIMB + cb * 128);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wl, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wh, acc[rb][cb]);
    }
    split();                                                     // next step's A (registers): VALU under the MFMAs
    {                                                            // the (lo, hi) product: A's low limb takes over amf's registers
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = *reinterpret_cast<const bf16x8*>(afr + 2 * K32_LIMB + rb * 128);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
      }
    }
    __syncthreads();                                             // everyone is done reading step kt
    if (kt + 1 < KT) write();
    fetch(kt + 2 < KT ? kt + 2 : KT - 1);
  }
  const int g4 = lane >> 4;
  const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
  if (full) {
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc[rb][cb][r] + bv[cb];
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = n0 + wc * 64 + cb * 16 + l15;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
        if (row < M) out[(size_t)row * N + col] = acc[rb][cb][r] + bv;
      }
  }
}
void launch_k32f(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_k32f_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}
void launch_k32e(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_k32e_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}
void launch_k32(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_k32_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}


// ---------------------------------------------------------------------------------------------------------------
// W straight from L2 into MFMA fragments: the packed weight is fragment-major ([kt32][limb][16-column block][lane][8 bf16], lane =
// (k-group, column)), every wave loads the fragments of its four column blocks itself (the two waves of a workgroup that share
// them hit the vector L1).  LDS then holds only A: two 24 KB buffers, ONE barrier per step, the split / write of step kt+1
// under the MFMAs of step kt.
// ---------------------------------------------------------------------------------------------------------------
__global__ void pack_w_frag_kernel(const float* __restrict__ w, __bf16* __restrict__ wf, int N, int K, int Npad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one (column, k pair)
  const long long total = (long long)Npad * (K / 2);
  if (idx >= total) return;
  const int kp = (int)(idx % (K / 2)), n = (int)(idx / (K / 2));
  const int k = 2 * kp;
  float a0 = 0.f, a1 = 0.f;
  if (n < N) { a0 = w[(size_t)n * K + k]; a1 = w[(size_t)n * K + k + 1]; }
  unsigned h, m, l;
  split2(a0, a1, h, m, l);
  const int kt = k / 32, kg4 = (k % 32) / 8, e = k % 8, nb = n / 16, col = n % 16;
  const size_t limb_stride = (size_t)(Npad / 16) * 64 * 8;                       // elements per (kt, limb)
  unsigned* dst = reinterpret_cast<unsigned*>(wf + ((size_t)kt * 3) * limb_stride + ((size_t)nb * 64 + kg4 * 16 + col) * 8 + e);
  dst[0] = h;
  dst[limb_stride / 2] = m;
  dst[limb_stride] = l;
}

__global__ __launch_bounds__(256, 3)
void gemm_wdirect_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wf, const float* __restrict__ bias,
                         float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * K32_TILE];            // A only, two buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  const int ar = tid >> 1, ah2 = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 16 * ah2) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  f32x4 a_regs[4];
  auto fetchA = [&](int kt) {
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff + 16 * q, so, 0));
  };
  auto stashA = [&](__bf16* buf) {                                // split + write (the write target is the OTHER buffer)
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
        split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
      }
      __bf16* pa = buf + ((2 * ah2 + gq) * GB_M + ar) * 8;
      *reinterpret_cast<u32x4*>(pa) = u32x4{hh[0], hh[1], hh[2], hh[3]};
      *reinterpret_cast<u32x4*>(pa + K32_LIMB) = u32x4{mm[0], mm[1], mm[2], mm[3]};
      *reinterpret_cast<u32x4*>(pa + 2 * K32_LIMB) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  // W fragments: this wave's column blocks nb0 .. nb0+3
  const size_t wf_limb = (size_t)(Npad / 16) * 64 * 8;
  const __bf16* wbase = Wf + ((size_t)((n0 + wc * 64) / 16) * 64 + lane) * 8;
  bf16x8 wq[2][3];                                               // ring of two column blocks x three limbs
  auto fetchW = [&](int slot, int kt, int cb) {
    const __bf16* p = wbase + (size_t)kt * 3 * wf_limb + (size_t)cb * 64 * 8;
#pragma unroll
    for (int l = 0; l < 3; ++l) wq[slot][l] = *reinterpret_cast<const bf16x8*>(p + l * wf_limb);
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const int afr_off = (kg4 * GB_M + wr * 64 + l15) * 8;
  auto mf = [](bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); };
  const int KT = K / 32;
  fetchA(0);
  fetchW(0, 0, 0);
  stashA(lds);
  fetchA(KT > 1 ? 1 : 0);
  auto step = [&](int kt, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    __syncthreads();                                             // A[kt] visible; the other buffer is free
    if (kt + 1 < KT) stashA(lds + (par ^ 1) * K32_TILE);         // next step's A first: its registers die here
    const __bf16* afr = lds + par * K32_TILE + afr_off;
    bf16x8 ahf[4], amf[4], alf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = *reinterpret_cast<const bf16x8*>(afr + rb * 128);
      amf[rb] = *reinterpret_cast<const bf16x8*>(afr + K32_LIMB + rb * 128);
      alf[rb] = *reinterpret_cast<const bf16x8*>(afr + 2 * K32_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      // prefetch the next column block (or the next step's first) into the other ring slot
      if (cb < 3) fetchW((cb + 1) & 1, kt, cb + 1);
      else fetchW(0, kt + 1 < KT ? kt + 1 : KT - 1, 0);
      const bf16x8 wh = wq[cb & 1][0], wm = wq[cb & 1][1], wl = wq[cb & 1][2];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wl, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(alf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wm, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], wh, acc[rb][cb]);
      if (cb == 2) fetchA(kt + 2 < KT ? kt + 2 : KT - 1);        // lands during the rest of this step and the next barrier
    }
  };
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int kt = 0;
    for (; kt + 2 <= KT; kt += 2) { step(kt, P0{}); step(kt + 1, P1{}); }
    if (kt < KT) step(kt, P0{});
  }
  const int g4 = lane >> 4;
  const bool full = (m0 + GB_M <= M) && (n0 + GB_N <= N);
  if (full) {
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[n0 + wc * 64 + cb * 16 + l15] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float* op = out + (size_t)(m0 + wr * 64 + 4 * g4) * N + n0 + wc * 64 + cb * 16 + l15;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(rb * 16 + r) * N] = acc[rb][cb][r] + bv[cb];
    }
    return;
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int col = n0 + wc * 64 + cb * 16 + l15;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + rb * 16 + 4 * g4 + r;
        if (row < M) out[(size_t)row * N + col] = acc[rb][cb][r] + bv;
      }
  }
}
static __bf16* g_wf = nullptr;       // fragment-major pack of the current shape's weights (filled by main)
void launch_wdirect(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL(gemm_wdirect_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, A, g_wf, bias, out, M, N, K, Npad, tiles_n);
}

struct Variant {
  std::string name;
  void (*launch)(const float*, const __bf16*, const __bf16*, const float*, float*, int, int, int);
};


template <int ABL>
void launch_abl(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const long long blocks = (long long)((M + 127) / 128) * tiles_n;
  hipLaunchKernelGGL((gemm_abl_kernel<ABL>), dim3((unsigned)blocks), dim3(256), 0, 0, A, Wp, bias, out, M, N, K, Npad, tiles_n);
}

template <int NS, bool PERSIST, int WPS>
void launch_ps(const float* A, const __bf16* Ap, const __bf16* Wp, const float* bias, float* out, int M, int N, int K) {
  const int Npad = (N + 127) / 128 * 128, tiles_n = Npad / 128;
  const int ntiles = ((M + 127) / 128) * tiles_n;
  const int lds_bytes = NS * GB_STAGE * 2;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ps_kernel<NS, PERSIST, WPS>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    once = true;
  }
  int wg_per_cu = std::min(160 * 1024 / lds_bytes, WPS);
  const unsigned grid = PERSIST ? (unsigned)std::min(ntiles, g_cus * wg_per_cu) : (unsigned)ntiles;
  hipLaunchKernelGGL((gemm_ps_kernel<NS, PERSIST, WPS>), dim3(grid), dim3(256), lds_bytes, 0, Ap, Wp, bias, out, M, N, K, Npad,
                     tiles_n, ntiles);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, g_cus);
  struct Shape { const char* name; int M, N, K; };
  std::vector<Shape> shapes = {{"ffn1", 618240, 1024, 256}, {"ffn2", 618240, 256, 1024}, {"proj", 618240, 544, 256},
                               {"oproj", 618240, 256, 256}};
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  std::vector<Variant> vars = {
      {"shipped", launch_abl<0>},      {"abl1_nosplit", launch_abl<1>}, {"abl2_nostash", launch_abl<2>},
      {"abl3_noload", launch_abl<3>},  {"abl4_mfma_only", launch_abl<4>}, {"abl5_mfma_nostore", launch_abl<5>},
      {"abl10_shipped_timed", launch_abl<10>}, {"abl11_fullpath_dword", launch_abl<11>}, {"abl21_fullpath_mfma16x16x32", launch_abl<21>}, {"pair16_k16stages", launch_abl<22>}, {"k32_single_stage", launch_k32}, {"k32_prefetchA2", launch_k32p2}, {"k32_split_early", launch_k32e}, {"k32_fetch_before_barrier", launch_k32f}, {"wdirect_A2buf", launch_wdirect}, {"abl12_fullpath_ldsT", launch_abl<12>}, {"epi1_swapped_x4", launch_epi<1>}, {"epi2_ldsT_x4", launch_epi<2>},
      {"v2_persist_defer", launch_v2<false>},
      {"ps_ns2_w3", launch_ps<2, false, 3>},
  };
  for (const auto& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    const int Npad = (N + 127) / 128 * 128;
    const long long Mp = (long long)(M + 127) / 128 * 128;
    float *A, *W, *bias, *out, *ref;
    __bf16 *Ap, *Wp;
    CK(hipMalloc(&A, (size_t)M * K * 4));
    CK(hipMalloc(&W, (size_t)N * K * 4));
    CK(hipMalloc(&bias, (size_t)N * 4));
    CK(hipMalloc(&out, (size_t)M * N * 4));
    CK(hipMalloc(&ref, (size_t)M * N * 4));
    CK(hipMalloc(&Ap, (size_t)Mp * K * 6));
    CK(hipMalloc(&Wp, (size_t)Npad * K * 6));
    fill_kernel<<<4096, 256>>>(A, (long long)M * K, 0x1234567u, 1.0f);
    fill_kernel<<<256, 256>>>(W, (long long)N * K, 0x89abcdeu, 1.0f / sqrtf((float)K));
    fill_kernel<<<16, 256>>>(bias, N, 0x5555u, 1.0f);
    split_tiles_kernel<<<(unsigned)((Mp * (K / 2) + 255) / 256), 256>>>(A, Ap, M, K);
    pack_w_kernel<<<(unsigned)(((long long)Npad * (K / 2) + 255) / 256), 256>>>(W, Wp, N, K, Npad);
    if (g_wf) CK(hipFree(g_wf));
    CK(hipMalloc(&g_wf, (size_t)Npad * K * 6));
    pack_w_frag_kernel<<<(unsigned)(((long long)Npad * (K / 2) + 255) / 256), 256>>>(W, g_wf, N, K, Npad);
    CK(hipDeviceSynchronize());
    launch_abl<0>(A, Ap, Wp, bias, ref, M, N, K);
    CK(hipDeviceSynchronize());
    unsigned long long* nbad;
    float* maxabs;
    CK(hipMalloc(&nbad, 8));
    CK(hipMalloc(&maxabs, 4));
    std::vector<std::vector<float>> times(vars.size());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (size_t v = 0; v < vars.size(); ++v) {               // correctness of the non-ablated variants vs shipped
      if (vars[v].name.rfind("abl", 0) == 0 && vars[v].name != "abl10_shipped_timed" && vars[v].name.rfind("abl1", 0) != 0) continue;
      if (vars[v].name == "abl1_nosplit") continue;
      CK(hipMemset(out, 0xff, (size_t)M * N * 4));
      vars[v].launch(A, Ap, Wp, bias, out, M, N, K);
      CK(hipMemset(nbad, 0, 8));
      CK(hipMemset(maxabs, 0, 4));
      diff_kernel<<<2048, 256>>>(out, ref, (long long)M * N, nbad, maxabs);
      unsigned long long hb;
      float hm;
      CK(hipMemcpy(&hb, nbad, 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hm, maxabs, 4, hipMemcpyDeviceToHost));
      printf("check %-6s %-22s differing outputs %llu max|d| %g\n", s.name, vars[v].name.c_str(), hb, hm);
    }
    for (int r = 0; r < rounds; ++r)
      for (size_t v = 0; v < vars.size(); ++v) {
        vars[v].launch(A, Ap, Wp, bias, out, M, N, K);     // warm
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) vars[v].launch(A, Ap, Wp, bias, out, M, N, K);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[v].push_back(ms / 5);
      }
    {
      unsigned long long z[8] = {0}, h[8];
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
      launch_abl<10>(A, Ap, Wp, bias, out, M, N, K);
      CK(hipDeviceSynchronize());
      launch_abl<13>(A, Ap, Wp, bias, out, M, N, K);      // warm
      CK(hipDeviceSynchronize());
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
      launch_abl<13>(A, Ap, Wp, bias, out, M, N, K);
      CK(hipDeviceSynchronize());
      CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)));
      printf("phase %-6s unguarded dword epilogue: prologue %.0f  loop %.0f  epilogue %.0f  (64 stores issued in %.0f, drain %.0f) cycles, %llu full tiles\n",
             s.name, (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], (double)h[4] / h[3], (double)h[6] / h[3], h[3]);
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)));
      launch_abl<10>(A, Ap, Wp, bias, out, M, N, K);
      CK(hipDeviceSynchronize());
      CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)));
      printf("phase %-6s shipped: per tile (wave 0) prologue %.0f  loop %.0f  epilogue(+drain) %.0f cycles  (%llu tiles; MFMA-only loop = %d cycles)\n",
             s.name, (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], h[3], (K / 16) * 24 * 32);
      printf("phase %-6s   epilogue split: first 32 stores issued after %.0f, next 32 after another %.0f, drain (vmcnt 0) %.0f cycles\n",
             s.name, (double)h[4] / h[3], (double)h[5] / h[3], (double)h[6] / h[3]);
    }
    for (size_t v = 0; v < vars.size(); ++v) {
      std::sort(times[v].begin(), times[v].end());
      const float med = times[v][times[v].size() / 2], mn = times[v][0];
      const double fl = 12.0 * M * (double)N * K;
      printf("time  %-6s %-22s median %.3f ms  min %.3f ms  bf16 %.0f TF  frac %.3f\n", s.name, vars[v].name.c_str(), med, mn,
             fl / med / 1e9, fl / med / 1e9 / 2516.6);
    }
    fflush(stdout);
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(out)); CK(hipFree(ref)); CK(hipFree(Ap)); CK(hipFree(Wp));
    CK(hipFree(nbad)); CK(hipFree(maxabs));
  }
  return 0;
}
