// Building blocks of the "row" kernels (decoder_rows.hip, relation_rows.hip): a workgroup of 8 waves owns 16 activation rows in
// LDS and walks a chain of small dense layers over them.  Every GEMM is  X[16 x K] . W^T  on v_mfma_f32_16x16x4_f32 (exact
// f32 products, f32 accumulation), the weights streamed from L2 in MFMA-fragment order (pvsg_pack_rows_weight: one contiguous
// 1 KiB per wave load) through a register ring.
#pragma once
#include "common.h"

namespace pvsg {

constexpr int ROWS_THREADS = 512;         // 8 waves: 8 column groups in the GEMMs

// acc[i] += X[16 x (16*nkc)] . W^T for column tiles t0 + i (packed weights, k chunks kc0 .. kc0+nkc-1 of a
// weight with `wkc` chunks per tile).  A fragment: lane (r = lane&15, g = lane>>4) holds X[r][16kc + 4g + j],
// B fragment: W[16t + r][16kc + 4g + j] -- the k index of MFMA step j is 4g + j on both sides.
template <int NT, int NKC, int DEPTH>
__device__ __forceinline__ void rows_gemm(const float* __restrict__ xl, int ld, const float* __restrict__ wp,
                                          int wkc, int kc0, int t0, f32x4 (&acc)[NT], int lane) {
  // The weights come from L2 / Infinity Cache (0.5 - 2 us away) while one k chunk is only NT x 128 matrix cycles:
  // DEPTH chunks of B fragments are kept in flight in a register ring (fully unrolled: static ring indices).
  const float* xa = xl + (lane & 15) * ld + 4 * (lane >> 4);
  const float* wb = wp + ((long long)t0 * wkc + kc0) * 256 + lane * 4;
  const long long tstride = (long long)wkc * 256;
  float4 ring[DEPTH][NT];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) ring[d][i] = ld4(wb + i * tstride + d * 256);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    const float4 a = *reinterpret_cast<const float4*>(xa + kc * 16);
    float4 cur[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) cur[i] = ring[kc % DEPTH][i];
    if (kc + DEPTH < NKC) {
#pragma unroll
      for (int i = 0; i < NT; ++i) ring[kc % DEPTH][i] = ld4(wb + i * tstride + (kc + DEPTH) * 256);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, cur[i].x, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, cur[i].y, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, cur[i].z, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, cur[i].w, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the ring as written: loads of chunk kc+DEPTH stay behind chunk kc's
  }
}

// ---- the same GEMM on the 16-bit matrix pipe (f16x2 split, split_common.h has the arithmetic's derivation) ------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the f16 rate: a 256 x 256 layer on 16 rows is 3.4 us of matrix time on one CU, and a
// decoder layer chains ~11 of them whatever the clip length.  Here a = a_h + 2^-11 a_l' is split per lane while it is read from
// LDS, the packed weight carries (w_h, w_l) of w 2^e, and two accumulator sets avoid the 2^-11 w_h operand:
//   acc += a_h w_h + a_h w_l,   lo += a_l' w_h,   out = 2^-e (acc + 2^-11 lo)          (rows_finish_h)
// Packed layout (pvsg_pack_rows_weight_f16x2): [max|w|, 2^-e, 0, 0] then per (column tile, 32-wide k block) 2 KiB:
// w_h fragment (lane (r, g): W[16 t + r][32 kb + 8 g + 0..7] as eight f16) | w_l fragment.  Same bytes per weight as the f32 pack.
typedef unsigned rows_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 rows_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rows_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void rows_split_pair(float a0, float a1, unsigned& h, unsigned& l, float& amax) {
  const rows_f16x2 hh = __builtin_convertvector(f32x2{a0, a1}, rows_f16x2);
  h = __builtin_bit_cast(unsigned, hh);
  const float r0 = __builtin_fmaf((float)hh[0], -2048.f, a0 * 2048.f), r1 = __builtin_fmaf((float)hh[1], -2048.f, a1 * 2048.f);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, rows_f16x2));
  amax = fmaxf(fmaxf(amax, __builtin_fabsf(a0)), __builtin_fabsf(a1));
}

__device__ __forceinline__ void rows_count_overflow(float amax, unsigned* overflow) {
  if (overflow && !(amax <= 65504.f)) atomicAdd(overflow, 1u);            // also counts NaN operands (as the split GEMMs do)
}

__device__ __forceinline__ rows_u32x4 rows_ldg16(const rows_u32x4* p) { return *p; }

// acc / lo += X[16 x (32 * nkb)] . W^T for column tiles t0 + i, k blocks kb0 .. kb0 + NKB - 1 of a weight with `wkb` blocks per tile
template <int NT, int NKB, int DEPTH>
__device__ __forceinline__ void rows_gemm_h(const float* __restrict__ xl, int ld, const float* __restrict__ wp, int wkb, int kb0,
                                            int t0, f32x4 (&acc)[NT], f32x4 (&lo)[NT], int lane, float& amax) {
  const float* xa = xl + (lane & 15) * ld + 8 * (lane >> 4);
  // the tile index is uniform in a wave: a SCALAR base + the lane's 16-byte slot keeps the ~50 fragment addresses of a layer in
  // SGPRs (as VGPR pairs they made this kernel spill)
  const int t0s = __builtin_amdgcn_readfirstlane(t0);
  const rows_u32x4* wb = reinterpret_cast<const rows_u32x4*>(wp + 4) + ((long long)t0s * wkb + kb0) * 128 + lane;
  const long long tstride = (long long)wkb * 128;
  rows_u32x4 ring[DEPTH][NT][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      ring[d][i][0] = rows_ldg16(wb + i * tstride + d * 128);
      ring[d][i][1] = rows_ldg16(wb + i * tstride + d * 128 + 64);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const float4 x0 = *reinterpret_cast<const float4*>(xa + kb * 32), x1 = *reinterpret_cast<const float4*>(xa + kb * 32 + 4);
    rows_u32x4 ah, al;
    {
      unsigned h, l;
      rows_split_pair(x0.x, x0.y, h, l, amax); ah[0] = h; al[0] = l;
      rows_split_pair(x0.z, x0.w, h, l, amax); ah[1] = h; al[1] = l;
      rows_split_pair(x1.x, x1.y, h, l, amax); ah[2] = h; al[2] = l;
      rows_split_pair(x1.z, x1.w, h, l, amax); ah[3] = h; al[3] = l;
      asm volatile("" : "+v"(amax));        // the running maximum is taken HERE (sunk to the end, it kept every x alive in scratch)
    }
    rows_u32x4 cur[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      cur[i][0] = ring[kb % DEPTH][i][0];
      cur[i][1] = ring[kb % DEPTH][i][1];
    }
    if (kb + DEPTH < NKB) {
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        ring[kb % DEPTH][i][0] = rows_ldg16(wb + i * tstride + (kb + DEPTH) * 128);
        ring[kb % DEPTH][i][1] = rows_ldg16(wb + i * tstride + (kb + DEPTH) * 128 + 64);
      }
    }
    const rows_f16x8 a_h = __builtin_bit_cast(rows_f16x8, ah), a_l = __builtin_bit_cast(rows_f16x8, al);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const rows_f16x8 w_h = __builtin_bit_cast(rows_f16x8, cur[i][0]), w_l = __builtin_bit_cast(rows_f16x8, cur[i][1]);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, w_h, acc[i], 0, 0, 0);
      lo[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, w_h, lo[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, w_l, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// out = 2^-e (acc + 2^-11 lo); `wp` = the packed weight (its header carries 2^-e)
template <int NT>
__device__ __forceinline__ void rows_finish_h(f32x4 (&acc)[NT], const f32x4 (&lo)[NT], const float* __restrict__ wp) {
  const float un = wp[1];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = (acc[i][e] + lo[i][e] * (1.f / 2048.f)) * un;
}

// One call site for both arithmetic forms: H = false -> rows_gemm (f32 MFMA, packed by pvsg_pack_rows_weight), H = true ->
// rows_gemm_h + rows_finish_h (packed by pvsg_pack_rows_weight_f16x2).  NKC / wkc / kc0 count 16-wide k chunks in both forms.
// `acc` must be ZERO on entry (the f16x2 form scales what it finds in it).
template <int NT>
struct rows_ring_depth {                        // k blocks of B fragments in flight: <= 96 registers of ring per wave
  static constexpr int value = NT >= 4 ? 3 : (NT == 2 ? 6 : 8);
};
template <bool H, int NT, int NKC, int DEPTH>
__device__ __forceinline__ void rows_mm(const float* __restrict__ xl, int ld, const float* __restrict__ wp, int wkc, int kc0,
                                        int t0, f32x4 (&acc)[NT], int lane, float& amax) {
  if constexpr (H) {
    static_assert(NKC % 2 == 0, "rows_mm: the f16x2 form walks 32-wide k blocks");
    constexpr int NKB = NKC / 2, DH = rows_ring_depth<NT>::value < NKB ? rows_ring_depth<NT>::value : NKB;
    f32x4 lo[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) lo[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    rows_gemm_h<NT, NKB, DH>(xl, ld, wp, wkc >> 1, kc0 >> 1, t0, acc, lo, lane, amax);
    rows_finish_h<NT>(acc, lo, wp);
  } else {
    rows_gemm<NT, NKC, DEPTH>(xl, ld, wp, wkc, kc0, t0, acc, lane);
  }
}

// all-reduce over the 64 lanes in the VALU: DPP quad / row permutations for the first four steps,
// v_permlane16_swap / v_permlane32_swap for the last two (no LDS crossbar round trips).
template <bool MAX>
__device__ __forceinline__ float wave_allreduce(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xf, 0xf, false)));  // row_mirror
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(v);
  auto b2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return op(__uint_as_float(b2[0]), __uint_as_float(b2[1]));
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NT]) {
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// One dense layer over the 16 LDS rows: out columns [0, NOUT) split over the 8 waves (NOUT / 128 column tiles each, in passes
// of at most four), epi(tile, acc) receives each finished 16 x 16 tile: acc[e] = row 4 * (lane >> 4) + e, column 16 * tile +
// (lane & 15).
template <int KDIM, int NOUT, typename Epi>
__device__ __forceinline__ void rows_linear(const float* __restrict__ xl, int ld, const float* __restrict__ wp, int w,
                                            int lane, Epi epi) {
  static_assert(KDIM % 16 == 0 && NOUT % 128 == 0, "rows_linear: K a multiple of 16, N a multiple of 128");
  constexpr int TPW = NOUT / 128;                                    // column tiles per wave
  constexpr int NT = TPW % 4 == 0 ? 4 : (TPW % 2 == 0 ? 2 : 1);
  constexpr int NKC = KDIM / 16;
  constexpr int DEPTH = NT == 4 ? 4 : 8;
#pragma unroll 1
  for (int pass = 0; pass < TPW / NT; ++pass) {
    const int t0 = w * TPW + pass * NT;
    f32x4 acc[NT];
    zero_acc(acc);
    // opaque to the optimiser: inside a caller's loop the ~100 fragment addresses of a layer are loop invariants, and hoisting
    // them (as 64-bit VGPR pairs) is what made the first build of relation_rows.hip spill 1 200 registers
    const float* wpl = wp;
    asm volatile("" : "+s"(wpl));
    rows_gemm<NT, NKC, (DEPTH < NKC ? DEPTH : NKC)>(xl, ld, wpl, NKC, 0, t0, acc, lane);
#pragma unroll
    for (int i = 0; i < NT; ++i) epi(t0 + i, acc[i]);
  }
}

// LayerNorm over the D columns of the 16 LDS rows of `x` (row stride D + 4; two-pass statistics like ATen), 32 threads per
// row, thread (r, cl) owns the float4 groups at columns 4 * (32 i + cl).  out: LDS destination or null; g_out: global row
// pointer of row 0 or null, rows `g_stride` floats apart, rows >= valid_rows not written.
template <int D>
__device__ __forceinline__ void rows_layernorm_t(const float* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, float* out,
                                                 float* __restrict__ g_out, long long g_stride, int valid_rows) {
  constexpr int LD = D + 4, NG = D / 128;
  const int r = threadIdx.x >> 5, cl = threadIdx.x & 31;
  float4 v[NG];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    v[i] = *reinterpret_cast<const float4*>(x + r * LD + 4 * (32 * i + cl));
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.f / D) + eps);
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int c = 4 * (32 * i + cl);
    const float4 g = ld4(gamma + c), b = ld4(beta + c);
    const float4 y = make_float4(v[i].x * rstd * g.x + b.x, v[i].y * rstd * g.y + b.y, v[i].z * rstd * g.z + b.z,
                                 v[i].w * rstd * g.w + b.w);
    if (out) *reinterpret_cast<float4*>(out + r * LD + c) = y;
    if (g_out && r < valid_rows) st4(g_out + (long long)r * g_stride + c, y);
  }
}

constexpr int ROWS_PLD = 64 + 4;           // row stride of a wave's probability tile [16 rows][64 keys]

// all-reduce over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15)
template <bool MAX>
__device__ __forceinline__ float row16_allreduce(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xf, 0xf, false)));  // row_mirror
  return v;
}

// Soft-max attention of ONE head (HD = 32 or 128 channels) for the 16 LDS rows of `xq` (scaled queries, row stride ld) over the
// keys of the 64-key chunks c_begin, c_begin + c_step, ... < L, on the matrix cores, by one wave; leaves the UN-normalised output
// tiles O (lane (g, j): rows 4 g + e, channel 16 nt + j), the running maxima M and sums l of rows 4 g + e in registers.
// Key k's row lives at kvb + k * kstride with K at +D and V at +2 D (+ HD h).
//   S = Q K^T: A = q rows from LDS, B = key rows straight from memory (lane (key = lane & 15, g) reads the float4 at channel
//   16 kc + 4 g: the operand layout of rows_gemm with the K matrix in the role of the weight);  the soft-max runs on the
//   accumulators (lane (g, j) holds rows 4 g + e of key-tile column j: a row's keys are one DPP row x the tiles);  P goes through
//   the wave-private LDS tile `pm` [16][ROWS_PLD] to become the A operand of O += P V, V rows read as B (lane = channel).
template <int HD>
__device__ __forceinline__ void rows_attention_core(const float* xq, int ld, float* pm, const float* __restrict__ kvb,
                                                    long long kstride, int D, int L, int h, int lane, int c_begin, int c_step,
                                                    f32x4 (&O)[HD / 16], float (&M)[4], float (&l)[4]) {
  constexpr int NKC = HD / 16;              // 16-channel chunks of a head = float4 fragments per lane
  const int g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) { M[i] = -INFINITY; l[i] = 0.f; }
#pragma unroll
  for (int nt = 0; nt < NKC; ++nt) O[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* qp = xq + j * ld + h * HD + 4 * g;
#pragma unroll 1
  for (int c0 = c_begin; c0 < L; c0 += c_step) {
    const int nk = min(64, L - c0);
    f32x4 S[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      S[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (kt * 16 < nk) {                         // uniform
        const int key = c0 + kt * 16 + j;
        const float* kp = kvb + (long long)min(key, L - 1) * kstride + D + h * HD + 4 * g;
        float4 kb[NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) kb[kc] = ld4(kp + 16 * kc);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          const float4 qa = *reinterpret_cast<const float4*>(qp + 16 * kc);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.x, kb[kc].x, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.y, kb[kc].y, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.z, kb[kc].z, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x4f32(qa.w, kb[kc].w, a, 0, 0, 0);
        }
        if (key < L) S[kt] = a;                   // keys past the sequence stay at -inf
      }
    }
    float alpha[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // row 4 g + i: maximum / sum over its keys = 4 tiles x the 16 lanes of the DPP row
      const float mc = row16_allreduce<true>(fmaxf(fmaxf(S[0][i], S[1][i]), fmaxf(S[2][i], S[3][i])));
      const float mn = fmaxf(M[i], mc);           // finite: every chunk holds at least one key
      alpha[i] = __expf(M[i] - mn);               // first chunk: exp(-inf) = 0
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const float pv = __expf(S[kt][i] - mn);   // exp(-inf) = 0 for the padding keys
        S[kt][i] = pv;
        ps += pv;
      }
      l[i] = l[i] * alpha[i] + row16_allreduce<false>(ps);
      M[i] = mn;
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) pm[(4 * g + i) * ROWS_PLD + kt * 16 + j] = S[kt][i];
#pragma unroll
    for (int nt = 0; nt < NKC; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) O[nt][i] *= alpha[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt * 16 < nk) {                         // uniform
        const float4 ap = *reinterpret_cast<const float4*>(pm + j * ROWS_PLD + kt * 16 + 4 * g);
        const int k0 = c0 + kt * 16 + 4 * g;
        const float* vp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)               // P is 0 for the clamped keys
          vp[u] = kvb + (long long)min(k0 + u, L - 1) * kstride + 2 * D + h * HD + j;
#pragma unroll
        for (int nt = 0; nt < NKC; ++nt) {
          const float v0 = vp[0][16 * nt], v1 = vp[1][16 * nt], v2 = vp[2][16 * nt], v3 = vp[3][16 * nt];
          O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap.x, v0, O[nt], 0, 0, 0);
          O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap.y, v1, O[nt], 0, 0, 0);
          O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap.z, v2, O[nt], 0, 0, 0);
          O[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap.w, v3, O[nt], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one wave = one 32-channel head over all L keys: xo[r][32 h ..] = softmax_k(q_r . k_k) v_k  (keys in chunks of 64 with an online
// soft-max, so any L works)
__device__ __forceinline__ void rows_attention_h32(const float* xq, float* xo, int ld, float* pm, const float* __restrict__ kvb,
                                                   long long kstride, int D, int L, int h, int lane) {
  f32x4 O[2];
  float M[4], l[4];
  rows_attention_core<32>(xq, ld, pm, kvb, kstride, D, L, h, lane, 0, 64, O, M, l);
  const int g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float inv = 1.f / l[i];
    xo[(4 * g + i) * ld + h * 32 + j] = O[0][i] * inv;
    xo[(4 * g + i) * ld + h * 32 + 16 + j] = O[1][i] * inv;
  }
}

}  // namespace pvsg
