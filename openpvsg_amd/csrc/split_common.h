// Shared by the split-arithmetic kernels (token_gemm.hip, conv1x1_split.hip, conv3x3_halo.hip, bottleneck_tail.hip): f32 results off
// the 16-bit matrix pipe from exact limb splits of every operand.
//
// bf16x3.  a = a_h + a_m + a_l with a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m): both residuals are exact in f32
// and |a - (a_h + a_m + a_l)| <= 2^-27 |a|.  A product a w is the sum of the nine limb products; the six with total order <= 2
// (hh, hm, mh, hl, lh, mm) are kept.  Each limb product is exact in f32 and accumulates in the f32 accumulator of the MFMA: an
// f32-class dot product (tests/test_gemm_bf16x3.py measures it against f64).  f16x2: see the two-limb section below.
// Everything lives in an anonymous namespace: each translation unit gets its own copy of the small pack kernels.
#pragma once
#include "common.h"

// lab switches (scripts/lab/r05_nt_lab.sh): cache-policy bits of the 1x1 convolution's streaming accesses (gfx940+: 1 = sc0, 2 = nt,
// 16 = sc1); the product build uses 0 everywhere
#ifndef PVSG_NT_ST
#define PVSG_NT_ST 0
#endif
#ifndef PVSG_NT_LD
#define PVSG_NT_LD 0
#endif
#ifndef PVSG_ABL
#define PVSG_ABL 0                                               // lab builds only (scripts/lab/abl_split.sh): timing ablations
#endif
#include <stdlib.h>

#include <type_traits>

namespace pvsg {
namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_M = 128, GB_N = 128, GB_K = 16;
constexpr int GB_LIMB = 2 * GB_M * 8;                 // bf16 elements of one limb of a 128 x 16 tile ([kg][row][8])
constexpr int GB_TILE = 3 * GB_LIMB;                  // one operand, three limbs: 6144 bf16 = 12 KB
constexpr int GB_STAGE = 2 * GB_TILE;                 // A and W: 24 KB

// three-limb split of two floats -> packed bf16 pairs (hi, mid, lo)
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
  const bf16x2 hh = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
  const bf16x2 mm = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
  m = __builtin_bit_cast(unsigned, mm);
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

// ------------------------------------------------------------------------------------------------------------------
// Two-limb f16 split (F16 = true in the K = 32 kernels below; csrc header of the f16x2 entry points at the end of the file).
// An f16 carries 11 significant bits, so a = a_h + a_l with a_h = f16(a), a_l = f16(a - a_h) is good to 2^-24 |a| (half an
// ulp of the f32 itself) and a product needs THREE limb products (hh, hl, lh; ll < 2^-24 |a w|) instead of six: half the
// matrix work for an f32-class dot product.  What f16 lacks is exponent range (2^-14 .. 65504), so the low limbs are kept
// away from the subnormals by power-of-two factors that cancel exactly:
//   weights (packed once):  ws = w * 2^e with max|ws| in [2^13, 2^14);  w_h = f16(ws), w_l = f16(ws - w_h)
//   activations (on the fly): a_h = f16(a), a_l' = f16(2^11 (a - a_h))      (a - a_h is exact in f32)
//   acc += a_l' w_h2 + a_h w_l + a_h w_h      (f32 accumulator of v_mfma_f32_16x16x32_f16),   out = 2^-e acc
//   with w_h2 = 2^-11 w_h made from the w_h fragment in registers (four v_pk_mul_f16 per fragment; exact down to the f16
//   subnormals, below which the a_l' w_h2 term is < 2^-39 of |a| max|w|): two 16-bit arrays per operand travel and are staged
// Full accuracy for 2^-13 <= |a| <= 65504 (29 binades; below that the absolute error is <= 2^-36), weights down to 2^-16 of
// the tensor's largest.  |a| > 65504 cannot be represented: every kernel counts such operands into `overflow` (the caller's
// device counter, checked by the host mirror at its next synchronisation point: openpvsg_amd/ops.py).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16X2_LO = 2048.f;                    // 2^11

// PVSG_SPLIT_ISA (lab builds, scripts/lab/split_isa_lab.sh; profiles/r05_split_lab.txt): 0 = the form below as hipcc compiles it
// (v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_mul_f32, v_pk_fma_f32, v_cvt_pk_f16_f32: packed f32 VALU beside the MFMAs);
// 1 = the mixed-precision FMAs read the f16 halves directly: r = a - h by v_fma_mix_f32 (exact), l = f16(2^11 r) by
// v_fma_mixlo_f16 / v_fma_mixhi_f16 -- five VALU instructions per pair, none packed, same bits.
#ifndef PVSG_SPLIT_ISA
#define PVSG_SPLIT_ISA 0
#endif
__device__ __forceinline__ void split2h(float a0, float a1, unsigned& h, unsigned& l, float& amax) {
  const f16x2 hh = __builtin_convertvector(f32x2{a0, a1}, f16x2);
  h = __builtin_bit_cast(unsigned, hh);
#if PVSG_SPLIT_ISA == 1
  float r0, r1;
  unsigned lo = 0u;
  const float k = F16X2_LO;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(a0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(a1));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(r0), "v"(k));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(r1), "v"(k));
  l = lo;
#else
  const float r0 = __builtin_fmaf((float)hh[0], -F16X2_LO, a0 * F16X2_LO), r1 = __builtin_fmaf((float)hh[1], -F16X2_LO, a1 * F16X2_LO);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
#endif
  amax = fmaxf(fmaxf(amax, __builtin_fabsf(a0)), __builtin_fabsf(a1));
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma_k32(u32x4 a, u32x4 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// per-tensor factor of an f16x2-packed weight: the two floats behind its 2 * Npad * K limb elements (amax, 2^-e)
__device__ __forceinline__ float f16x2_unscale(const __bf16* Wp, int Npad, int K) {
  return reinterpret_cast<const float*>(Wp + (size_t)2 * Npad * K)[1];
}
// 2^-11 w_h of a fragment of eight f16
__device__ __forceinline__ u32x4 f16x2_lo_scale(u32x4 wh) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 k = {(_Float16)(1.f / F16X2_LO), (_Float16)(1.f / F16X2_LO)};
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned w = wh[i];               // (never bit_cast a vector ELEMENT expression: hipcc 7.2 then reads element 0)
    r[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, w) * k);
  }
  return r;
}
__device__ __forceinline__ void f16x2_count_overflow(float amax, unsigned* overflow) {
  if (overflow && !(amax <= 65504.f)) atomicAdd(overflow, 1u);          // also counts NaN operands
}

constexpr int K32_LIMB = 4 * GB_M * 8;          // bf16 / f16 elements of one limb of a 128 x 32 tile
constexpr int K32_TILE = 3 * K32_LIMB;          // 24 KB per operand

// W (N, K) f32 -> [k-tile K/16][limb 3][k-group 2][Npad][8] bf16, columns beyond N zero
__global__ void gemm_bf16x3_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int N, int K, int Npad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one (column, k pair)
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

// f16x2 pack: [k-tile K/16][array 2: w_h, w_l][k-group 2][Npad][8] f16, followed by four floats (max|w|, 2^-e, 0, 0).
// max|w| is reduced on the device first (mask embeddings are packed per call).
__global__ void f16x2_amax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ tail) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, __builtin_fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(tail, __builtin_bit_cast(unsigned, m));   // non-negative floats order as uints
}
__device__ __forceinline__ int f16x2_exponent(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
  int e = 13 - ilogbf(amax);                               // max|w| 2^e in [2^13, 2^14)
  return e > 126 ? 126 : (e < -126 ? -126 : e);            // 2^-e stays a normal f32
}
__global__ void gemm_f16x2_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int N, int K, int Npad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // one (column, k pair)
  const long long total = (long long)Npad * (K / 2);
  if (idx >= total) return;
  float* tail = reinterpret_cast<float*>(wp + (size_t)2 * Npad * K);
  const int e = f16x2_exponent(tail[0]);
  if (idx == 0) tail[1] = ldexpf(1.f, -e);
  const int kp = (int)(idx % (K / 2)), n = (int)(idx / (K / 2));
  const int k = 2 * kp;
  float a0 = 0.f, a1 = 0.f;
  if (n < N) {
    a0 = ldexpf(w[(size_t)n * K + k], e);
    a1 = ldexpf(w[(size_t)n * K + k + 1], e);
  }
  const f16x2 hh = __builtin_convertvector(f32x2{a0, a1}, f16x2);
  const f16x2 ll = __builtin_convertvector(f32x2{a0 - (float)hh[0], a1 - (float)hh[1]}, f16x2);
  const int kt = k / GB_K, kg = (k % GB_K) / 8, el = k % 8;
  const size_t limb_stride = (size_t)2 * Npad * 8;
  unsigned* dst = reinterpret_cast<unsigned*>(wp + ((size_t)kt * 2 * limb_stride + ((size_t)kg * Npad + n) * 8 + el));
  dst[0] = __builtin_bit_cast(unsigned, hh);
  dst[limb_stride / 2] = __builtin_bit_cast(unsigned, ll);
}

}  // namespace
}  // namespace pvsg

