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
template <int ABL>
__global__ __launch_bounds__(256, 2)
void gemm_abl_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                     float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * GB_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    if (ABL >= 3) return;
    const unsigned so = (unsigned)kt * (GB_K * 4);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      a_regs[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc_t, a_voff + 16 * q, so, 0));
    const __bf16* wk = wsrc + (size_t)kt * 3 * w_limb_stride;
#pragma unroll
    for (int l = 0; l < 3; ++l) w_regs[slot][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
  };
  auto stash = [&](int slot, __bf16* st) {
    if (ABL >= 3) return;
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
  if (ABL == 4) {   // operands: whatever the (never written) LDS holds once -- random-ish bits from A instead
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = *reinterpret_cast<const bf16x8*>(A + (size_t)(tid * 8 + l * 2 + b) * 4);
        wv[l][b] = *reinterpret_cast<const bf16x8*>(Wp + (size_t)(tid * 8 + l * 2 + b) * 8);
      }
  }
  auto kstep = [&](int kt, auto PAR) {
    constexpr int par = decltype(PAR)::value;
    if (ABL != 4) {
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
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[p]][rb], wv[PW[p]][cb], acc[rb][cb], 0, 0, 0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 2 <= KT; kt += 2) {
    kstep(kt, P0{});
    kstep(kt + 1, P1{});
  }
  if (kt < KT) kstep(kt, P0{});

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

struct Variant {
  std::string name;
  void (*launch)(const float*, const __bf16*, const __bf16*, const float*, float*, int, int, int);
};

static int g_cus = 256;

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
      {"abl3_noload", launch_abl<3>},  {"abl4_mfma_only", launch_abl<4>},
      {"ps_ns2_w3", launch_ps<2, false, 3>}, {"ps_ns2_w4", launch_ps<2, false, 4>}, {"ps_ns3_w2", launch_ps<3, false, 2>},
      {"ps_ns2_w3_persist", launch_ps<2, true, 3>}, {"ps_ns3_w2_persist", launch_ps<3, true, 2>},
      {"ps_ns4_w1_persist", launch_ps<4, true, 1>},
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
      if (vars[v].name.rfind("abl", 0) == 0) continue;
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
