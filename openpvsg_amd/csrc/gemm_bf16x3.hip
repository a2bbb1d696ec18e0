// out[M, N] = act(A[M, K] . W[N, K]^T + bias) with f32 inputs and outputs, computed on the bf16 matrix cores from an
// EXACT three-limb split of every operand.
//
// Replaces, on the north-star path, the library f32 GEMMs behind the token-major linear layers of the pixel decoder's
// deformable-attention encoder and of the transformer decoder's key / value projections
//   [3P] mmcv FFN.layers (Linear 256->1024 + ReLU, Linear 1024->256), MultiScaleDeformableAttention.{value_proj,
//        sampling_offsets, attention_weights, output_proj}, MultiheadAttention in_proj (k, v)
// which rocBLAS / hipBLASLt run on the f32 MFMA at 115-150 TFLOP/s (the f32 matrix rate is 1/16 of the bf16 rate).
//
// Arithmetic.  a = a_h + a_m + a_l with a_h = bf16(a), a_m = bf16(a - a_h), a_l = bf16(a - a_h - a_m): both residuals are
// exact in f32 and |a - (a_h + a_m + a_l)| <= 2^-27 |a| (three 8-bit mantissas cover the 24 bits of an f32).  A product
// a w is the sum of the nine limb products; the six with total order <= 2 (hh, hm, mh, hl, lh, mm) are kept, the other
// three are below 2^-25 |a w|.  Each limb product is exact in f32 (8 x 8 bit mantissas) and accumulates in the f32
// accumulator of v_mfma_f32_32x32x16_bf16, so the result is an f32-class dot product (same error class as the library's
// f32 GEMM with a different summation order; tests/test_gemm_bf16x3.py measures both against f64).
//
// Kernel.  Workgroup = 4 waves = 128 x 128 outputs, wave = 64 x 64 (2 x 2 MFMA blocks, 64 accumulator registers),
// K-step = 16: 6 limb pairs x 4 blocks = 24 MFMAs against 12 ds_read_b128 (the three limbs of two row blocks and two
// column blocks).  A is split on the fly while it is staged (global f32 -> 3 x packed bf16 in LDS: 11 VALU instructions
// per element pair, hidden beside the bf16 MFMAs); W is split once by pvsg_gemm_bf16x3_pack into the staging order.
// LDS tiles are [limb][k-group of 8][row][8 bf16]: consecutive lanes read consecutive 16-byte groups.
#include "common.h"

// lab switches (scripts/lab/r05_nt_lab.sh): cache-policy bits of the 1x1 convolution's streaming accesses (gfx940+: 1 = sc0, 2 = nt,
// 16 = sc1); the product build uses 0 everywhere
#ifndef PVSG_NT_ST
#define PVSG_NT_ST 0
#endif
#ifndef PVSG_NT_LD
#define PVSG_NT_LD 0
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

template <bool RELU>
__global__ __launch_bounds__(256, 2)
void gemm_bf16x3_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                        float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * GB_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;      // column tiles of one row tile are neighbours: A from L2
  const int m0 = tm * GB_M, n0 = tn * GB_N;

  // staging: A -- thread = (row tid/2, k-group tid%2), 8 consecutive floats; W -- (k-group tid/128, column tid%128), 3 limbs
  const int ar = tid >> 1, akg = tid & 1;
  const bool a_in = m0 + ar < M;
  const unsigned a_voff = a_in ? (unsigned)((ar * K + 8 * akg) * 4) : 0x80000000u;   // rows beyond M read as 0
  const size_t a_base = (size_t)m0 * K * 4;                       // folded into the pointer below: keeps offsets 32-bit
  const auto asrc_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + a_base), 0,
                                                        (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;              // elements per (k-tile, limb): [kg][Npad][8]
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;

  f32x4 a_regs[2][2];                                 // [fetch slot = K-step & 1][two float4]
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
  const int a_off = (kg * GB_M + wr * 64 + li) * 8, w_off = GB_TILE + (kg * GB_N + wc * 64 + li) * 8;
  // Two LDS stages, one barrier per K-step: stage kt is read right after the barrier that publishes it while stage kt+1
  // is being written; three workgroups per CU (48 KB, <= 168 registers) cover each other's barriers and load latencies.
  // Global loads run two K-steps ahead of their staging (register ring of two slots).
  // (A three-stage variant with the operands of K-step kt+1 prefetched into registers under the MFMAs of kt needs 72 KB
  // and drops to two workgroups per CU: 2.34 vs 1.99 ms on the encoder's first FFN layer.)
  fetch(0, 0);
  stash(0, lds);
  fetch(1, KT > 1 ? 1 : 0);
  fetch(0, KT > 2 ? 2 : KT - 1);
  auto kstep = [&](int kt, auto PAR) {
    constexpr int par = decltype(PAR)::value;          // kt & 1
    __syncthreads();                                   // stage kt visible; stage kt+1's buffer no longer read
    const __bf16* cur = lds + par * GB_STAGE;
    bf16x8 av[3][2], wv[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[l][b] = *reinterpret_cast<const bf16x8*>(cur + a_off + l * GB_LIMB + b * 32 * 8);
        wv[l][b] = *reinterpret_cast<const bf16x8*>(cur + w_off + l * GB_LIMB + b * 32 * 8);
      }
    stash(par ^ 1, lds + (par ^ 1) * GB_STAGE);        // K-step kt+1; past the end: a copy of the last one, never read
    fetch(par ^ 1, kt + 3 < KT ? kt + 3 : KT - 1);
    // small terms first: (m,m) (h,l) (l,h) (h,m) (m,h) (h,h)
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

  // bias / ReLU and store: register r of block (rb, cb) = row (r&3) + 8 (r>>2) + 4 kg, column li of the block.
  // No branch and no memory wait between the 64 stores of a lane: the tile's rows go through a buffer descriptor that ends
  // at row min(m0 + 128, M) (stores beyond it are dropped by the bounds check), lanes of columns >= N get an offset outside
  // every descriptor, and the bias is read once (clamped index) before the first store.  With per-element `if (row < M)`
  // guards the compiler put an `s_waitcnt vmcnt(0)` in front of every store -- each waited for its predecessor's
  // acknowledgement, 600-900 cycles per store, more than the K loop of a K = 256 tile (profiles/r03_lab_gemm_epilogues.txt).
  {
    const int rows = M - m0 < GB_M ? M - m0 : GB_M;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * N, 0, (unsigned)((size_t)rows * N * 4), 0x00020000);
    const unsigned rowpitch = (unsigned)N * 4u;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = n0 + wc * 64 + cb * 32 + li;
      const float bv = bias ? bias[col < N ? col : N - 1] : 0.f;
      const unsigned vbase = col < N ? (unsigned)(wr * 64 + 4 * kg) * rowpitch + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o = acc[rb][cb][r] + bv;
          if (RELU) o = fmaxf(o, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), orsrc,
                                                vbase + (unsigned)(rb * 32 + (r & 3) + 8 * (r >> 2)) * rowpitch, 0, 0);
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_bf16 (default since the end of round 3 for K % 32 == 0).  The bf16 matrix pipe is
// power-limited on real data, and the 16x16x32 instruction spends less energy per flop than 32x32x16 (register-only loops:
// 0.85 vs 0.73 of the roof; swapped into the kernel above on the same operand registers: 5-9 % faster,
// profiles/r03_lab_gemm_ablations.txt).  Its 32-deep fragments need K = 32 stages: ONE 48 KB stage per workgroup (three
// workgroups per CU as before), A and W limbs as [limb][k-group 0..3][row][8 bf16]; the next step's operands travel from
// HBM / L2 into registers while this step's 96 MFMAs run, and are split and written between two barriers -- the other two
// workgroups of the CU cover that window.  Wave tile 64 x 64 = 4 x 4 blocks of 16 x 16; A's hi / mid fragments stay in
// registers across the four column blocks, the low limb takes over the mid limb's registers for the (lo, hi) product, which
// therefore comes last.  The packed weight layout is unchanged (two 16-deep sub-steps per stage).
// Measured against the kernel above: FFN1 1.85 -> 1.63 ms, FFN2 1.63 -> 1.57, 544-wide projection 1.10 -> 1.01.
// ------------------------------------------------------------------------------------------------------------------
constexpr int K32_LIMB = 4 * GB_M * 8;          // bf16 / f16 elements of one limb of a 128 x 32 tile
constexpr int K32_TILE = 3 * K32_LIMB;          // 24 KB per operand
// F16: the two-limb f16 form (see split2h): A as (a_h, a_l'), W as (w_h, w_l, w_h2), three MFMAs per block instead of six,
// 40 KB of LDS; `overflow` counts staged operands beyond the f16 range.
template <bool RELU, bool F16 = false>
__global__ __launch_bounds__(256, 3)
void gemm_bf16x3_k32_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                            float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n, unsigned* __restrict__ overflow = nullptr) {
  constexpr int AL = F16 ? 2 : 3;                                // limbs of the on-the-fly operand
  // A's k-groups are 130 rows apart in LDS (not 128): the staging threads of a wave write (row, k-group) = (lane / 4, lane % 4),
  // and with a 2080-byte k-group stride the eight 16-byte records of a write cycle fall into eight different bank groups
  constexpr int A_KG = (GB_M + 2) * 8, A_LIMB = 4 * A_KG;
  constexpr int WL = F16 ? 2 : 3;                                // arrays of the packed operand
  constexpr int W_AT = AL * A_LIMB;                              // where the weight tile starts
  __shared__ __attribute__((aligned(16))) __bf16 lds[W_AT + WL * K32_LIMB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  // staging: A -- thread = (rows tid/4 and 64 + tid/4, k-group tid%4), 8 consecutive floats of each: four lanes cover one
  // 128-byte line of a row, a load instruction touches 16 lines (with two lanes per row and 64 bytes each it touched 32 and
  // the texture addresser, not the matrix pipe, set the pace: scripts/lab/abl_split.sh); W -- (k-group tid/128, column
  // tid%128) of both 16-deep sub-steps of the packed weight, 3 limbs each
  const int ar = tid >> 2, akg = tid & 3;
  unsigned a_voff[2];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2)                                 // rows beyond M read as 0
    a_voff[p2] = m0 + ar + 64 * p2 < M ? (unsigned)(((ar + 64 * p2) * K + 8 * akg) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  const int wkg = tid >> 7, wcol = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Npad * 8;
  const __bf16* wsrc = Wp + ((size_t)wkg * Npad + n0 + wcol) * 8;
  f32x4 a_regs[4];
  u32x4 w_regs[2][WL];
#ifndef PVSG_ABL
#define PVSG_ABL 0                                               // lab builds only (scripts/lab/abl_split.sh): timing ablations
#endif
  auto fetch = [&](int kt) {                                     // kt counts 32-deep steps
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                // a_regs[2 p + h]: floats 4 h .. 4 h + 3 of row ar + 64 p
      if (PVSG_ABL == 3 || PVSG_ABL == 4) a_regs[q] = f32x4{1.f + so, 2.f, 3.f, 4.f};
      else a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[q >> 1] + 16 * (q & 1), so, 0));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const __bf16* wk = wsrc + (size_t)(2 * kt + j) * WL * w_limb_stride;
#pragma unroll
      for (int l = 0; l < WL; ++l) {
        if (PVSG_ABL == 2 || PVSG_ABL == 4 || PVSG_ABL == 5) w_regs[j][l] = u32x4{0x3c003c00u + (unsigned)kt, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
        else w_regs[j][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
      }
    }
  };
  // the split of step kt+1 (VALU) runs under the MFMAs of step kt, on the registers its loads landed in; between the two
  // barriers only the LDS writes remain
  u32x4 limbs[2][AL];                                           // [row ar + 64 gq][limb]
  float amax = 0.f;
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {                             // the k-group of row ar + 64 gq
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        if constexpr (F16) {
          split2h(v[0], v[1], hh[2 * q], mm[2 * q], amax);
          split2h(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], amax);
        } else {
          split2(v[0], v[1], hh[2 * q], mm[2 * q], ll[2 * q]);
          split2(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], ll[2 * q + 1]);
        }
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
      if constexpr (!F16) limbs[gq][2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  auto write = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + akg * A_KG + (ar + 64 * gq) * 8;
#pragma unroll
      for (int l = 0; l < AL; ++l) *reinterpret_cast<u32x4*>(pa + l * A_LIMB) = limbs[gq][l];
    }
    if (PVSG_ABL == 5) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __bf16* pw = lds + W_AT + ((2 * j + wkg) * GB_N + wcol) * 8;
#pragma unroll
      for (int l = 0; l < WL; ++l) *reinterpret_cast<u32x4*>(pw + l * K32_LIMB) = w_regs[j][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + kg4 * A_KG + (wr * 64 + l15) * 8;                 // + limb * A_LIMB + row block * 128
  const __bf16* wfr = lds + W_AT + (kg4 * GB_N + wc * 64 + l15) * 8;          // + limb * K32_LIMB + column block * 128
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) {
    if (PVSG_ABL == 6) { c[0] += __builtin_bit_cast(float, a[0] ^ b[0]); return c; }
    return mfma_k32<F16>(a, b, c);
  };
  const int KT = K / 32;
  fetch(0);
  split();
  write();
  fetch(KT > 1 ? 1 : 0);                                        // loads run a whole step ahead of their split
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();                                             // step kt is in LDS
    u32x4 ahf[4], amf[4];                                        // A's first two limbs: (hi, mid) or (a_h, a_l')
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = frag(afr + rb * 128);
      amf[rb] = frag(afr + A_LIMB + rb * 128);
    }
    if constexpr (F16) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {                           // small terms first: (l', h2) (h, l) (h, h)
        const u32x4 wh = frag(wfr + cb * 128), wl = frag(wfr + K32_LIMB + cb * 128), wh2 = f16x2_lo_scale(wh);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh2, amf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wl, ahf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, ahf[rb], acc[rb][cb]);
      }
      split();                                                   // next step's A: VALU under the MFMAs still in flight
    } else {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {                           // small terms first: (m,m) (h,l) (h,m) (m,h) (h,h)
        const u32x4 wh = frag(wfr + cb * 128), wm = frag(wfr + K32_LIMB + cb * 128), wl = frag(wfr + 2 * K32_LIMB + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wm, amf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wl, ahf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wm, ahf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, amf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, ahf[rb], acc[rb][cb]);
      }
      split();                                                   // next step's A: VALU under the MFMAs still in flight
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = frag(afr + 2 * A_LIMB + rb * 128);     // A's low limb
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {                           // (l,h)
        const u32x4 wh = frag(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, amf[rb], acc[rb][cb]);
      }
    }
    __syncthreads();                                             // everyone is done reading step kt
    if (kt + 1 < KT) write();
    fetch(kt + 2 < KT ? kt + 2 : KT - 1);                        // registers are free again: step kt+2 starts its trip
  }
  // bias / ReLU and store through a bounded buffer descriptor (see the kernel above).  The MFMAs take the weight fragment as
  // their row operand, so register r of block (rb, cb) = row rb*16 + (lane&15), column cb*16 + 4*(lane>>4) + r of the wave's
  // 64 x 64 tile: a lane owns four consecutive columns of one row -- 16 sixteen-byte stores per lane instead of 64 dword
  // stores (N % 4 == 0; otherwise element by element); no branch, no wait between the stores
  {
    const float unscale = F16 ? f16x2_unscale(Wp, Npad, K) : 1.f;
    const int rows = M - m0 < GB_M ? M - m0 : GB_M;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * N, 0, (unsigned)((size_t)rows * N * 4), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (unsigned)N * 4u : 0u, 0x00020000);
    const unsigned rowpitch = (unsigned)N * 4u;
    const bool vec4 = (N & 3) == 0;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = n0 + wc * 64 + cb * 16 + 4 * kg4;
      f32x4 bv;                                                  // columns >= N read 0 through the descriptor
      if (vec4) bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)col * 4u, 0, 0));
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, (unsigned)(col + r) * 4u, 0, 0));
      const unsigned vbase = (unsigned)(wr * 64 + l15) * rowpitch + (unsigned)col * 4u;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = F16 ? __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]) : acc[rb][cb][r] + bv[r];
          if (RELU) o[r] = fmaxf(o[r], 0.f);
        }
        const unsigned vo = vbase + (unsigned)(rb * 16) * rowpitch;
        if (PVSG_ABL == 1) { if (o[0] == 1.2345e33f) out[0] = o[1]; continue; }
        if (vec4)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, col < N ? vo : 0x80000000u, 0, 0);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float oe = o[r];              // (bit_cast of the vector element itself stored element 0 four times)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oe), orsrc, col + r < N ? vo + 4u * r : 0x80000000u, 0, 0);
          }
      }
    }
  }
  if constexpr (F16) f16x2_count_overflow(amax, overflow);
}

// ------------------------------------------------------------------------------------------------------------------
// The f16x2 GEMM with the packed operand staged by LDS-DMA.  In the kernel above hipcc sinks the weight loads of a step down
// to their LDS writes (the register budget of three workgroups per CU leaves it no room to keep them in flight): every step then
// waits `vmcnt(0)` for an L2 round trip with nothing else to do (scripts/lab/abl_split.sh: "no W loads" -20 %).  Here the
// weight tile of step kt+1 travels global -> LDS (global_load_lds_dwordx4, one 1 KB slab per wave instruction, no registers, no
// ds_write) into the second of two weight buffers while step kt computes; A's loads run two steps ahead in registers as before.
//   LDS: A [2 limbs][4 k-groups][130 rows][8] (16.3 KB) + W [2 buffers][2 arrays][4 k-groups][128 columns][8] (32 KB)
//   per step: wait (everything issued a step ago) -> barrier -> split A(kt+1) -> DMA W(kt+1), load A(kt+2) -> fragments +
//             48 MFMAs -> barrier -> write A(kt+1)
// Past the last step the same addresses are fetched again (nothing reads them).
#if defined(PVSG_ABL) && PVSG_ABL == 7
// lab build (scripts/lab/abl_split.sh 7): where a wave's time goes.  Sums over wave 0 of every workgroup, in s_memtime ticks:
// [0] prologue [1] wait + barrier at the top of a step [2] split / issue / (stage writes) [3] fragments + MFMAs [4] second barrier
// + A writes (128 x 128 kernel) [5] epilogue [6] workgroups [7] steps
__device__ unsigned long long g_split_phase[8];
#define PVSG_TICK(v) const unsigned long long v = __builtin_readcyclecounter()
#define PVSG_PHASE(i, d) do { if (tid == 0) atomicAdd(&g_split_phase[i], (unsigned long long)(d)); } while (0)
#else
#define PVSG_TICK(v) do {} while (0)
#define PVSG_PHASE(i, d) do {} while (0)
#endif
template <bool RELU>
__global__ __launch_bounds__(256, 3)
void gemm_f16x2_dma_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                           float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n, unsigned* __restrict__ overflow) {
  constexpr int A_KG = (GB_M + 2) * 8, A_LIMB = 4 * A_KG;        // (see gemm_bf16x3_k32_kernel: conflict-free staging writes)
  constexpr int W_AT = 2 * A_LIMB, W_BUF = 2 * K32_LIMB;
  __shared__ __attribute__((aligned(16))) __bf16 lds[W_AT + 2 * W_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  PVSG_TICK(tk0);
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * GB_M, n0 = tn * GB_N;
  const int ar = tid >> 2, akg = tid & 3;
  unsigned a_voff[2];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2)                                 // rows beyond M read as 0
    a_voff[p2] = m0 + ar + 64 * p2 < M ? (unsigned)(((ar + 64 * p2) * K + 8 * akg) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)GB_M * K * 4), 0x00020000);
  f32x4 a_regs[4];
  auto loadA = [&](int kt) {
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)                                  // a_regs[2 p + h]: floats 4 h .. 4 h + 3 of row ar + 64 p
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[q >> 1] + 16 * (q & 1), so, 0));
  };
  // weight slabs of a 32-deep step: (array l, k-group kg of 4, column half) = 16 x 1 KB; wave w brings slabs 4 w .. 4 w + 3.
  // packed layout [k-tile of 16][array 2][k-group 2][Npad][8]: k-group kg of the step = k-tile 2 kt + (kg >> 1), group kg & 1
  const size_t w_kg_stride = (size_t)Npad * 8;
  auto dmaW = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sl = wave * 4 + i, l = sl >> 3, kg = (sl >> 1) & 3, half = sl & 1;
      const __bf16* src = Wp + ((((size_t)(2 * kt + (kg >> 1)) * 2 + l) * 2 + (kg & 1)) * w_kg_stride) + (size_t)(n0 + half * 64 + lane) * 8;
      __bf16* dst = lds + W_AT + buf * W_BUF + l * K32_LIMB + (kg * GB_N + half * 64) * 8;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  u32x4 limbs[2][2];                                            // [row ar + 64 gq][limb]
  float amax = 0.f;
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2h(v[0], v[1], hh[2 * q], mm[2 * q], amax);
        split2h(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], amax);
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
    }
    // pin the running maximum here: left to itself the optimiser sinks the max chain below the next loads, the old A registers
    // stay alive, the new loads land in other registers and a copy (with a `vmcnt(0)`) appears at the end of every step
    asm volatile("" : "+v"(amax));
  };
  auto writeA = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + akg * A_KG + (ar + 64 * gq) * 8;
#pragma unroll
      for (int l = 0; l < 2; ++l) *reinterpret_cast<u32x4*>(pa + l * A_LIMB) = limbs[gq][l];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + kg4 * A_KG + (wr * 64 + l15) * 8;                  // + limb * A_LIMB + row block * 128
  const __bf16* wfr0 = lds + W_AT + (kg4 * GB_N + wc * 64 + l15) * 8;          // + buffer * W_BUF + array * K32_LIMB + column block * 128
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<true>(a, b, c); };
  const int KT = K / 32;
  // 16-column blocks of this wave's 64 columns that hold real outputs (wave-uniform): 4 except in a ragged last column tile
  const int ncb = __builtin_amdgcn_readfirstlane(min(4, max(0, (N - (n0 + wc * 64) + 15) >> 4)));
  dmaW(0, 0);
  loadA(0);
  split();
  writeA();
  loadA(KT > 1 ? 1 : 0);
  PVSG_TICK(tk1);
  PVSG_PHASE(0, tk1 - tk0);
  for (int kt = 0; kt < KT; ++kt) {
    PVSG_TICK(ts0);
    // this wave's slabs of step kt, its A rows of step kt (LDS) and of step kt+1 (registers) have arrived -- all were issued a
    // whole step ago.  (Consuming the A registers while newer DMA is in flight would need `vmcnt(4)`; hipcc's own count across
    // the loop's back edge is `vmcnt(0)`, which would wait for the slabs just issued: so the split comes first.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // ... everybody's have; buffer (kt+1)&1 is no longer read
    PVSG_TICK(ts1);
    split();                                                     // A of step kt+1
    __builtin_amdgcn_sched_barrier(0);
    dmaW(kt + 1 < KT ? kt + 1 : KT - 1, (kt + 1) & 1);
    loadA(kt + 2 < KT ? kt + 2 : KT - 1);
    __builtin_amdgcn_sched_barrier(0);
    PVSG_TICK(ts2);
    const __bf16* wfr = wfr0 + (kt & 1) * W_BUF;
    u32x4 ahf[4], alf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = frag(afr + rb * 128);
      alf[rb] = frag(afr + A_LIMB + rb * 128);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {                             // small terms first: (l', 2^-11 h) (h, l) (h, h)
      if (cb >= ncb) continue;                                   // column blocks past N (ragged last tile, e.g. N = 544): no MFMAs
      const u32x4 wh = frag(wfr + cb * 128), wl = frag(wfr + K32_LIMB + cb * 128), wh2 = f16x2_lo_scale(wh);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh2, alf[rb], acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wl, ahf[rb], acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, ahf[rb], acc[rb][cb]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PVSG_TICK(ts3);
    __builtin_amdgcn_s_barrier();                                // everyone is done reading A of step kt
    writeA();
    PVSG_TICK(ts4);
    PVSG_PHASE(1, ts1 - ts0); PVSG_PHASE(2, ts2 - ts1); PVSG_PHASE(3, ts3 - ts2); PVSG_PHASE(4, ts4 - ts3); PVSG_PHASE(7, 1);
  }
  PVSG_TICK(tk2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the repeated last slabs / rows: nothing may land after the end)
  {
    const float unscale = f16x2_unscale(Wp, Npad, K);
    const int rows = M - m0 < GB_M ? M - m0 : GB_M;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * N, 0, (unsigned)((size_t)rows * N * 4), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (unsigned)N * 4u : 0u, 0x00020000);
    const unsigned rowpitch = (unsigned)N * 4u;
    const bool vec4 = (N & 3) == 0;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = n0 + wc * 64 + cb * 16 + 4 * kg4;
      f32x4 bv;                                                  // columns >= N read 0 through the descriptor
      if (vec4) bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)col * 4u, 0, 0));
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, (unsigned)(col + r) * 4u, 0, 0));
      const unsigned vbase = (unsigned)(wr * 64 + l15) * rowpitch + (unsigned)col * 4u;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]);
          if (RELU) o[r] = fmaxf(o[r], 0.f);
        }
        const unsigned vo = vbase + (unsigned)(rb * 16) * rowpitch;
        if (vec4)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, col < N ? vo : 0x80000000u, 0, 0);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float oe = o[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oe), orsrc, col + r < N ? vo + 4u * r : 0x80000000u, 0, 0);
          }
      }
    }
  }
  f16x2_count_overflow(amax, overflow);
#if defined(PVSG_ABL) && PVSG_ABL == 7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PVSG_TICK(tk3);
  PVSG_PHASE(5, tk3 - tk2); PVSG_PHASE(6, 1);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The f16x2 GEMM on 256 x 256 tiles.  With half the matrix work of the bf16 form the 128 x 128 kernels above stop being
// matrix-bound: what they move from L2 into the CUs -- M N K (4 / TN + 4 / TM) bytes, 10 GB for the encoder's first FFN layer,
// every operand element re-read once per tile of the other operand -- sets their time at ~10 TB/s whatever the kernel does
// inside (scripts/lab/abl_split.sh: time falls with every load removed, not with the MFMAs; profiles/r04_split_lab.txt).
// A 256 x 256 tile halves that traffic.  One workgroup of 8 waves per CU (wave tile 64 rows x 128 columns: 128 accumulator
// registers, A's fragments resident across the eight column blocks -- 24 LDS fragment reads per 96 MFMAs where two 64 x 64
// waves need 32), two LDS stages of 64 KB and ONE barrier per 32-deep step:
//   top of step kt: everything issued a step ago has arrived (own slabs of W(kt), own rows of A(kt+1) in registers, own LDS
//   writes of A(kt)) -> barrier -> split A(kt+1) and write it to the other stage, DMA W(kt+1) into it, load A(kt+2) ->
//   fragments + 96 MFMAs of stage kt.
//   LDS stage: A [2 limbs][4 k-groups][258 rows][8] (33 KB; 258: conflict-free staging writes) + W [2 arrays][4][256 columns][8]
// LN (N == 256 == one tile: a workgroup owns whole rows): out = LayerNorm(residual + A W^T + bias) * gamma + beta -- the
// [3P] mmcv encoder layer's `identity + dropout(out)` followed by its `norm` ([3P] BaseTransformerLayer, 'self_attn', 'norm',
// 'ffn', 'norm'), which otherwise costs a separate pass over three (rows, 256) tensors (pvsg_add_layernorm, 0.33 ms x 12 per
// 32-frame clip).  Statistics in two passes like F.layer_norm: row mean, then the centred sum of squares; the four lane
// groups of a wave hold 32 columns of a row each (shuffles), the two waves of a row pair meet through 2 KB of LDS.
template <bool RELU, bool LN = false>
__global__ __launch_bounds__(512)
void gemm_f16x2_t256_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                            float* __restrict__ out, int M, int N, int K, int Npad, int tiles_n, unsigned* __restrict__ overflow,
                            const float* __restrict__ residual = nullptr, const float* __restrict__ gamma = nullptr,
                            const float* __restrict__ beta = nullptr, float eps = 0.f) {
  constexpr int TM = 256, TN = 256;
  constexpr int A_KG = (TM + 2) * 8, A_LIMB = 4 * A_KG, A_STAGE = 2 * A_LIMB;
  constexpr int W_LIMB = 4 * TN * 8, STAGE = A_STAGE + 2 * W_LIMB;
  extern __shared__ __attribute__((aligned(16))) __bf16 lds256[];
  __bf16* lds = lds256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  PVSG_TICK(tk0);
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = logical % tiles_n, tm = logical / tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int ar = tid >> 2, akg = tid & 3;                        // rows ar and ar + 128, k-group akg
  unsigned a_voff[2];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2)                                 // rows beyond M read as 0
    a_voff[p2] = m0 + ar + 128 * p2 < M ? (unsigned)(((ar + 128 * p2) * K + 8 * akg) * 4) : 0x80000000u;
  const auto asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                      (unsigned)((size_t)TM * K * 4), 0x00020000);
  f32x4 a_regs[4];
  auto loadA = [&](int kt) {
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[q >> 1] + 16 * (q & 1), so, 0));
  };
  // weight slabs of a step: (array l, k-group kg of 4, column quarter) = 32 x 1 KB; wave w brings slabs 4 w .. 4 w + 3.  A
  // quarter beyond the packed columns (Npad is a multiple of 128, not of 256) fetches other columns: its outputs are never stored.
  const size_t w_kg_stride = (size_t)Npad * 8;
  auto dmaW = [&](int kt, int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sl = wave * 4 + i, l = sl >> 4, kg = (sl >> 2) & 3, qt = sl & 3;
      const int c0 = n0 + qt * 64 < Npad ? n0 + qt * 64 : Npad - 64;         // (branch-free: such a quarter re-reads valid columns)
      const __bf16* src = Wp + ((((size_t)(2 * kt + (kg >> 1)) * 2 + l) * 2 + (kg & 1)) * w_kg_stride) + (size_t)(c0 + lane) * 8;
      __bf16* dst = lds + st * STAGE + A_STAGE + l * W_LIMB + (kg * TN + qt * 64) * 8;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  u32x4 limbs[2][2];                                            // [row ar + 128 gq][limb]
  float amax = 0.f;
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2h(v[0], v[1], hh[2 * q], mm[2 * q], amax);
        split2h(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], amax);
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
    }
    asm volatile("" : "+v"(amax));                                // (see gemm_f16x2_dma_kernel: keeps the A registers reusable)
  };
  auto writeA = [&](int st) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + st * STAGE + akg * A_KG + (ar + 128 * gq) * 8;
#pragma unroll
      for (int l = 0; l < 2; ++l) *reinterpret_cast<u32x4*>(pa + l * A_LIMB) = limbs[gq][l];
    }
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i >> 3][i & 7] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr0 = lds + kg4 * A_KG + (wr * 64 + l15) * 8;                 // + stage + limb * A_LIMB + row block * 128
  const __bf16* wfr0 = lds + A_STAGE + (kg4 * TN + wc * 128 + l15) * 8;        // + stage + array * W_LIMB + column block * 128
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<true>(a, b, c); };
  const int KT = K / 32;
  dmaW(0, 0);
  loadA(0);
  split();
  writeA(0);
  loadA(KT > 1 ? 1 : 0);
  PVSG_TICK(tk1);
  PVSG_PHASE(0, tk1 - tk0);
  for (int kt = 0; kt < KT; ++kt) {
    PVSG_TICK(ts0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // stage kt & 1 is complete; nobody reads the other stage any more
    PVSG_TICK(ts1);
    const int cur = kt & 1;
    split();                                      // A of step kt+1 (a repeat of the last step past the end: never read)
    __builtin_amdgcn_sched_barrier(0);
    dmaW(kt + 1 < KT ? kt + 1 : KT - 1, cur ^ 1);
    loadA(kt + 2 < KT ? kt + 2 : KT - 1);
    writeA(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    PVSG_TICK(ts2);
    const __bf16* afr = afr0 + cur * STAGE;
    const __bf16* wfr = wfr0 + cur * STAGE;
    u32x4 ahf[4], alf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = frag(afr + rb * 128);
      alf[rb] = frag(afr + A_LIMB + rb * 128);
    }
    // the weight fragments of column block cb+1 are requested before the 12 MFMAs of block cb are issued (left to itself the
    // scheduler reads each pair just in time and waits lgkmcnt(0) in front of every four MFMAs)
    u32x4 wh = frag(wfr), wl = frag(wfr + W_LIMB);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {                             // small terms first: (l', 2^-11 h) (h, l) (h, h)
      u32x4 whn = wh, wln = wl;
      if (cb < 7) {
        whn = frag(wfr + (cb + 1) * 128);
        wln = frag(wfr + W_LIMB + (cb + 1) * 128);
      }
      const u32x4 wh2 = f16x2_lo_scale(wh);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh2, alf[rb], acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wl, ahf[rb], acc[rb][cb]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, ahf[rb], acc[rb][cb]);
      __builtin_amdgcn_sched_barrier(0);
      wh = whn;
      wl = wln;
    }
#if defined(PVSG_ABL) && PVSG_ABL == 7
    asm volatile("" ::"v"(acc[3][7]));
    PVSG_TICK(ts3);
    PVSG_PHASE(1, ts1 - ts0); PVSG_PHASE(2, ts2 - ts1); PVSG_PHASE(3, ts3 - ts2); PVSG_PHASE(7, 1);
#endif
  }
  PVSG_TICK(tk2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the repeated last slabs / rows: nothing may land after the end)
  // register r of block (rb, cb) = row rb*16 + (lane&15), column cb*16 + 4*(lane>>4) + r of the wave's 64 x 128 tile
  if constexpr (LN) {
    const float unscale = f16x2_unscale(Wp, Npad, K);
    const int rows = M - m0 < TM ? M - m0 : TM;
    const unsigned tile_bytes = (unsigned)((size_t)rows * 256 * 4);
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * 256, 0, tile_bytes, 0x00020000);
    const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual) + (size_t)m0 * 256, 0, tile_bytes, 0x00020000);
    // v = residual + acc 2^-e + bias (rows beyond M read 0 and are never stored), in place in the accumulators
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const int col = wc * 128 + cb * 16 + 4 * kg4;
      const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const unsigned vo = (unsigned)(wr * 64 + rb * 16 + l15) * 1024u + (unsigned)col * 4u;
        const f32x4 res = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vo, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]) + res[r];
          acc[rb][cb][r] = v;
          rsum[rb] += v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);                          // (one column block's residual loads in flight at a time: registers)
    }
    // row statistics: lanes l15, l15+16, +32, +48 hold the four 32-column parts of a row of this wave; waves (wr, 0) and (wr, 1)
    // hold the two 128-column halves.  red[pass][wave][64 rows]
    __builtin_amdgcn_s_barrier();                                 // everybody is done with the stages: LDS is free
    float* red = reinterpret_cast<float*>(lds);
    auto row_reduce = [&](float (&part)[4], int pass) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        part[rb] += __shfl_xor(part[rb], 16);
        part[rb] += __shfl_xor(part[rb], 32);
      }
      if (kg4 == 0)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) red[(pass * 8 + wave) * 64 + rb * 16 + l15] = part[rb];
      __syncthreads();
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) part[rb] += red[(pass * 8 + (wave ^ 1)) * 64 + rb * 16 + l15];
    };
    row_reduce(rsum, 0);
    float mean[4], rstd[4], sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) mean[rb] = rsum[rb] * (1.f / 256.f);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[rb][cb][r] - mean[rb];
          acc[rb][cb][r] = d;
          sq[rb] = __builtin_fmaf(d, d, sq[rb]);
        }
    row_reduce(sq, 1);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) rstd[rb] = rsqrtf(sq[rb] * (1.f / 256.f) + eps);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const int col = wc * 128 + cb * 16 + 4 * kg4;
      const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + col), be = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(acc[rb][cb][r] * rstd[rb], gv[r], be[r]);
        const unsigned vo = (unsigned)(wr * 64 + rb * 16 + l15) * 1024u + (unsigned)col * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, vo, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const float unscale = f16x2_unscale(Wp, Npad, K);
    const int rows = M - m0 < TM ? M - m0 : TM;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * N, 0, (unsigned)((size_t)rows * N * 4), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (unsigned)N * 4u : 0u, 0x00020000);
    const unsigned rowpitch = (unsigned)N * 4u;
    const bool vec4 = (N & 3) == 0;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const int col = n0 + wc * 128 + cb * 16 + 4 * kg4;
      f32x4 bv;                                                  // columns >= N read 0 through the descriptor
      if (vec4) bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)col * 4u, 0, 0));
      else
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brsrc, (unsigned)(col + r) * 4u, 0, 0));
      const unsigned vbase = (unsigned)(wr * 64 + l15) * rowpitch + (unsigned)col * 4u;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]);
          if (RELU) o[r] = fmaxf(o[r], 0.f);
        }
        const unsigned vo = vbase + (unsigned)(rb * 16) * rowpitch;
        if (vec4)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, col < N ? vo : 0x80000000u, 0, 0);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float oe = o[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oe), orsrc, col + r < N ? vo + 4u * r : 0x80000000u, 0, 0);
          }
      }
    }
  }
  f16x2_count_overflow(amax, overflow);
#if defined(PVSG_ABL) && PVSG_ABL == 7
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PVSG_TICK(tk3);
  PVSG_PHASE(5, tk3 - tk2); PVSG_PHASE(6, 1);
#endif
}
constexpr int T256_LDS_BYTES = 2 * (2 * 4 * (256 + 2) * 8 + 2 * 4 * 256 * 8) * 2;

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm-fused projection on 128-ROW tiles, two workgroups per CU (round 5).  The 256 x 256 form above owns a CU alone: its
// epilogue -- read 256 KB of residual, two row passes, write 256 KB -- runs with the matrix pipe idle, and its main loop with the
// memory pipe half idle; on the second FFN layer (K = 1024) that serialisation cost what the fusion saved (1.3 ms against
// 1.01 + 0.33 for GEMM + add-LayerNorm launches).  Here a workgroup owns 128 whole rows (tile 128 x 256, four waves of 64 rows x
// 128 columns: the same 128 accumulator registers per lane), keeps A in ONE 16 KB stage (the two-barrier step of
// gemm_f16x2_dma_kernel) and W in two 32 KB LDS-DMA buffers: 80 KB, so TWO workgroups share a CU and one's epilogue runs under
// the other's MFMAs.  Row statistics exactly as above (two passes; lane groups by shuffles, the two column halves of a row
// through 1 KB of LDS).  No row padding in the A stage (80 KB x 2 = the CU's 160 KB to the byte).
// LN = false: the same tile and pipeline as a plain GEMM (bias / ReLU epilogue) for N > 256: `tiles_n` 256-column tiles per row
// block, consecutive workgroups share the row block (A from L2).  Opt-in (PVSG_F16X2_TILE=w256), measured in
// scripts/lab/gemm_tile_ab.py.
// KV = true (round 5): the decoder's key AND value projections of one level in one launch, straight from the encoder's token
// tensor.  Rows = the level's tokens of every frame (row r -> frame r / hw, token start + r % hw of `A` = (frames, S, 256));
// W = [Wk ; Wv] (N = 512): column tile 0 writes keys to `out`, tile 1 values to `residual` (reused as the second output).
// The reference adds level_embed and the positional encoding to the INPUT of the key projection (mask2former_head.py:421-436);
// both are linear terms, so they come in through the epilogue: keys += gamma[r % hw] + beta[(r / hw) % zrows] (two small tables:
// ((pe_yx + level_embed) Wk^T + bk) per cell and (pe_z Wk^T) per frame), values += bias (level_embed Wv^T + bv).  The key / value
// INPUT tensors (4 KB per key written + read by pvsg_decoder_kv_inputs and the two projections) never exist.
template <bool LN, bool RELU, bool KV = false>
__global__ __launch_bounds__(256, 2)
void gemm_f16x2_ln128_kernel(const float* __restrict__ A, const __bf16* __restrict__ Wp, const float* __restrict__ bias,
                             float* __restrict__ out, int M, int K, unsigned* __restrict__ overflow,
                             const float* __restrict__ residual, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float eps, int N = 256, int Npad = 256, int tiles_n = 1,
                             int kv_S = 0, int kv_start = 0, int kv_hw = 1, int kv_zrows = 1) {
  constexpr int TM = 128, TN = 256;
  constexpr int A_KG = TM * 8, A_LIMB = 4 * A_KG;                // f16 elements
  constexpr int W_AT = 2 * A_LIMB, W_LIMB = 4 * TN * 8, W_BUF = 2 * W_LIMB;
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsln[];
  __bf16* lds = ldsln;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tn = LN ? 0 : (int)(logical % (unsigned)tiles_n), n0 = tn * TN;
  const int m0 = (int)(LN ? logical : logical / (unsigned)tiles_n) * TM;
  const int ar = tid >> 2, akg = tid & 3;                        // rows ar and ar + 64, k-group akg
  unsigned a_voff[2];
#pragma unroll
  for (int p2 = 0; p2 < 2; ++p2) {                               // rows beyond M read as 0
    const int r = m0 + ar + 64 * p2;
    if constexpr (KV) {                                          // token row of frame r / hw (the whole tensor is below 4 GB: host check)
      const int f = r / kv_hw, c = r - f * kv_hw;
      a_voff[p2] = r < M ? (unsigned)(((size_t)f * kv_S + kv_start + c) * K + 8 * akg) * 4u : 0xffffffe0u;   // (+16 must not wrap)
    } else {
      a_voff[p2] = r < M ? (unsigned)(((ar + 64 * p2) * K + 8 * akg) * 4) : 0x80000000u;
    }
  }
  const auto asrc = KV ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((size_t)(M / kv_hw) * kv_S * K * 4), 0x00020000)
                       : __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A) + (size_t)m0 * K * 4), 0,
                                                           (unsigned)((size_t)TM * K * 4), 0x00020000);
  f32x4 a_regs[4];
  auto loadA = [&](int kt) {
    const unsigned so = (unsigned)kt * (32 * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a_regs[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc, a_voff[q >> 1] + 16 * (q & 1), so, 0));
  };
  // weight slabs of a step: (array l, k-group kg of 4, column quarter) = 32 x 1 KB; wave w brings slabs 8 w .. 8 w + 7
  const size_t w_kg_stride = (size_t)Npad * 8;
  auto dmaW = [&](int kt, int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int sl = wave * 8 + i, l = sl >> 4, kg = (sl >> 2) & 3, qt = sl & 3;
      const int c0 = n0 + qt * 64 < Npad ? n0 + qt * 64 : Npad - 64;         // (a quarter beyond the packed columns: never stored)
      const __bf16* src = Wp + ((((size_t)(2 * kt + (kg >> 1)) * 2 + l) * 2 + (kg & 1)) * w_kg_stride) + (size_t)(c0 + lane) * 8;
      __bf16* dst = lds + W_AT + buf * W_BUF + l * W_LIMB + (kg * TN + qt * 64) * 8;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  u32x4 limbs[2][2];
  float amax = 0.f;
  auto split = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      unsigned hh[4], mm[4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 v = a_regs[2 * gq + q];
        split2h(v[0], v[1], hh[2 * q], mm[2 * q], amax);
        split2h(v[2], v[3], hh[2 * q + 1], mm[2 * q + 1], amax);
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
    }
    asm volatile("" : "+v"(amax));                                // (see gemm_f16x2_dma_kernel: keeps the A registers reusable)
  };
  auto writeA = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pa = lds + akg * A_KG + (ar + 64 * gq) * 8;
#pragma unroll
      for (int l = 0; l < 2; ++l) *reinterpret_cast<u32x4*>(pa + l * A_LIMB) = limbs[gq][l];
    }
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i >> 3][i & 7] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + kg4 * A_KG + (wr * 64 + l15) * 8;                  // + limb * A_LIMB + row block * 128
  const __bf16* wfr0 = lds + W_AT + (kg4 * TN + wc * 128 + l15) * 8;           // + buffer * W_BUF + array * W_LIMB + column block * 128
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<true>(a, b, c); };
  const int KT = K / 32;
  // 16-column blocks of this wave's 128 columns that exist (plain GEMM with N % 256 != 0: the encoder's 544-wide projection)
  int ncb = 8;
  if constexpr (!LN && !KV) {
    const int left = N - n0 - wc * 128;
    ncb = __builtin_amdgcn_readfirstlane(left >= 128 ? 8 : (left <= 0 ? 0 : (left + 15) >> 4));
  }
  dmaW(0, 0);
  loadA(0);
  split();
  writeA();
  loadA(KT > 1 ? 1 : 0);
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // W(kt), A(kt) complete; buffer (kt+1)&1 is no longer read
    split();                                                     // A of step kt+1
    __builtin_amdgcn_sched_barrier(0);
    dmaW(kt + 1 < KT ? kt + 1 : KT - 1, (kt + 1) & 1);
    loadA(kt + 2 < KT ? kt + 2 : KT - 1);
    __builtin_amdgcn_sched_barrier(0);
    const __bf16* wfr = wfr0 + (kt & 1) * W_BUF;
    u32x4 ahf[4], alf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = frag(afr + rb * 128);
      alf[rb] = frag(afr + A_LIMB + rb * 128);
    }
    u32x4 wh = frag(wfr), wl = frag(wfr + W_LIMB);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {                             // small terms first: (l', 2^-11 h) (h, l) (h, h)
      u32x4 whn = wh, wln = wl;
      if (cb < 7) {
        whn = frag(wfr + (cb + 1) * 128);
        wln = frag(wfr + W_LIMB + (cb + 1) * 128);
      }
      if (cb < ncb) {                                            // (ragged N: column blocks beyond it are never stored)
        const u32x4 wh2 = f16x2_lo_scale(wh);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh2, alf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wl, ahf[rb], acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wh, ahf[rb], acc[rb][cb]);
      }
      __builtin_amdgcn_sched_barrier(0);
      wh = whn;
      wl = wln;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // everyone is done reading A of step kt
    writeA();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the repeated last slabs / rows: nothing may land after the end)
  // register r of block (rb, cb) = row rb*16 + (lane&15), column cb*16 + 4*(lane>>4) + r of the wave's 64 x 128 tile
  const float unscale = f16x2_unscale(Wp, Npad, K);
  const int rows = M - m0 < TM ? M - m0 : TM;
  if constexpr (KV) {
    float* dst = tn == 0 ? out : const_cast<float*>(residual);
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)m0 * 256, 0, (unsigned)((size_t)rows * 1024), 0x00020000);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int rl = wr * 64 + rb * 16 + l15, r = m0 + rl;
      const int f = r / kv_hw, cell = r - f * kv_hw, z = f % kv_zrows;
      const float* ty = gamma + (size_t)cell * 256 + wc * 128 + 4 * kg4;
      const float* tz = beta + (size_t)z * 256 + wc * 128 + 4 * kg4;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        f32x4 add;
        if (tn == 0) {
          if (r < M) {
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(ty + cb * 16), a2 = *reinterpret_cast<const f32x4*>(tz + cb * 16);
            add = a1 + a2;
          } else add = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
          add = *reinterpret_cast<const f32x4*>(bias + wc * 128 + cb * 16 + 4 * kg4);
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(acc[rb][cb][e], unscale, add[e]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc,
                                               (unsigned)rl * 1024u + (unsigned)(wc * 128 + cb * 16 + 4 * kg4) * 4u, 0, 0);
      }
    }
    f16x2_count_overflow(amax, overflow);
    return;
  }
  if constexpr (!LN) {
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * N, 0, (unsigned)((size_t)rows * N * 4), 0x00020000);
    const auto brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (unsigned)N * 4u : 0u, 0x00020000);
    const unsigned rowpitch = (unsigned)N * 4u;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      const int col = n0 + wc * 128 + cb * 16 + 4 * kg4;         // N % 4 == 0 (checked by the host): a float4 is all in or all out
      const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)col * 4u, 0, 0));
      const unsigned vbase = (unsigned)(wr * 64 + l15) * rowpitch + (unsigned)col * 4u;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]);
          if (RELU) o[r] = fmaxf(o[r], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, col < N ? vbase + (unsigned)(rb * 16) * rowpitch : 0x80000000u, 0, 0);
      }
    }
    f16x2_count_overflow(amax, overflow);
    return;
  }
  const unsigned tile_bytes = (unsigned)((size_t)rows * 256 * 4);
  const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)m0 * 256, 0, tile_bytes, 0x00020000);
  const auto rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual) + (size_t)m0 * 256, 0, tile_bytes, 0x00020000);
  const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? 1024u : 0u, 0x00020000);
  const unsigned vrow = (unsigned)(wr * 64 + l15) * 1024u + (unsigned)(wc * 128 + 4 * kg4) * 4u;     // + rb * 16 KiB + cb * 64 B
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    const int col = wc * 128 + cb * 16 + 4 * kg4;
    const f32x4 bv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (unsigned)col * 4u, 0, 0));   // no bias: reads 0
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const f32x4 res = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vrow, rb * 16384 + cb * 64, 0));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __builtin_fmaf(acc[rb][cb][r], unscale, bv[r]) + res[r];
        acc[rb][cb][r] = v;
        rsum[rb] += v;
      }
    }
    __builtin_amdgcn_sched_barrier(0);                            // (one column block's residual loads in flight at a time: registers)
  }
  __builtin_amdgcn_s_barrier();                                   // everybody is done with the stages: LDS is free
  float* red = reinterpret_cast<float*>(lds);                     // red[pass][wave][64 rows]
  auto row_reduce = [&](float (&part)[4], int pass) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      part[rb] += __shfl_xor(part[rb], 16);
      part[rb] += __shfl_xor(part[rb], 32);
    }
    if (kg4 == 0)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) red[(pass * 4 + wave) * 64 + rb * 16 + l15] = part[rb];
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) part[rb] += red[(pass * 4 + (wave ^ 1)) * 64 + rb * 16 + l15];
  };
  row_reduce(rsum, 0);
  float mean[4], rstd[4], sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) mean[rb] = rsum[rb] * (1.f / 256.f);
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[rb][cb][r] - mean[rb];
        acc[rb][cb][r] = d;
        sq[rb] = __builtin_fmaf(d, d, sq[rb]);
      }
  row_reduce(sq, 1);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) rstd[rb] = rsqrtf(sq[rb] * (1.f / 256.f) + eps);
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    const int col = wc * 128 + cb * 16 + 4 * kg4;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + col), be = *reinterpret_cast<const f32x4*>(beta + col);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(acc[rb][cb][r] * rstd[rb], gv[r], be[r]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), orsrc, vrow, rb * 16384 + cb * 64, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  f16x2_count_overflow(amax, overflow);
}
constexpr int LN128_LDS_BYTES = (2 * 4 * 128 * 8 + 2 * 2 * 4 * 256 * 8) * 2;      // 80 KB

// ------------------------------------------------------------------------------------------------------------------
// 1x1 convolution in NCHW on the same arithmetic:  y[b, co, p] = act( (sum_ci w[co, ci] x[b, ci, pin(p)]) * scale[co]
// + shift[co] (+ residual[b, co, p]) ), pin(p) = p (stride 1) or the stride-2 sub-sampled pixel.  Per image a GEMM with
// rows = output channels (the packed weight, same pack as above), columns = pixels, K = input channels.  The pixel
// operand is K-strided in memory: a thread stages (pixel, k-group of 8) with eight coalesced dword loads (64 lanes =
// 64 consecutive pixels of one channel), splits and packs them into the same [limb][kg][column][8] LDS tile -- the
// transposition costs nothing extra.  Columns = pixels also makes the stores pixel-contiguous.
// Replaces [3P] mmdet ResNet Bottleneck.conv1 / conv3 / downsample[0] (+ frozen BN, identity, ReLU) with > 128 input
// channels and the pixel decoder's 1x1 input / lateral / mask-feature convolutions (library GEMM or MIOpen + separate
// BN / bias pass before).
// IN_NORM: the input is normalised on the way in, x' = relu(x * in_scale[b, ci] + in_shift[b, ci]) (a GroupNorm + ReLU
// whose statistics are already known), so that pass never touches HBM.
// BITS (attention-mask mode; rows = queries, one channel tile, no affine): the epilogue thresholds the logits in registers
// (masked <=> sigmoid(x) < 0.5 <=> x < 0) and writes one 128-bit record per key -- bit q = query q masked -- plus the
// "query has an unmasked key" flag words, exactly the format of csrc/mask_gemm.hip (`y` then points at the uint32 records of
// this batch element, image = frame t, key = t * HWo + pixel; `flags` at its 4 flag words).
template <bool RELU, bool RESIDUAL, bool IN_NORM, bool BITS = false>
__global__ __launch_bounds__(256, 2)
void conv1x1_bf16x3_kernel(const float* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ scale,
                           const float* __restrict__ shift, const float* __restrict__ residual,
                           const float* __restrict__ in_scale, const float* __restrict__ in_shift, float* __restrict__ y,
                           int Cin, int Cout, int Cpad, int HWin, int Win, int HWo, int Wo, int stride, int tiles_c,
                           int tiles_p, unsigned* __restrict__ flags = nullptr) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * GB_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tc = logical % tiles_c;                    // channel tiles of one pixel tile are neighbours: x from L2
  logical /= tiles_c;
  const int tp = logical % tiles_p, img = logical / tiles_p;
  const int c0 = tc * GB_M, p0 = tp * GB_N;

  // staging: weights -- (k-group tid/128, row tid%128), 3 limbs; pixels -- (k-group tid/128, pixel tid%128), 8 channels
  const int skg = __builtin_amdgcn_readfirstlane(tid >> 7), srow = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Cpad * 8;
  const __bf16* wsrc = Wp + ((size_t)skg * Cpad + c0 + srow) * 8;
  const int pix = p0 + srow;
  const int pin = stride == 1 ? pix : (2 * (pix / Wo)) * Win + 2 * (pix % Wo);
  const unsigned x_voff = pix < HWo ? (unsigned)pin * 4u : 0x80000000u;            // beyond the map: read as 0
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)img * Cin * HWin, 0,
                                                      (unsigned)((size_t)Cin * HWin * 4), 0x00020000);
  const unsigned plane = (unsigned)HWin * 4u;

  float x_regs[2][8];                                 // [fetch slot = K-step & 1]
  u32x4 w_regs[2][3];
  auto fetch = [&](int slot, int kt) {
    const unsigned so = (unsigned)(kt * GB_K + 8 * skg) * plane;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      x_regs[slot][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, x_voff, so + j * plane, 0));
    const __bf16* wk = wsrc + (size_t)kt * 3 * w_limb_stride;
#pragma unroll
    for (int l = 0; l < 3; ++l) w_regs[slot][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
  };
  auto stash = [&](int slot, __bf16* st, int kt) {
    unsigned hh[4], mm[4], ll[4];
    if (IN_NORM) {                                     // channel kt*16 + 8*skg + j of this image: wave-uniform scalars
      const int ci = img * Cin + kt * GB_K + 8 * skg;
#pragma unroll
      for (int j = 0; j < 8; ++j) x_regs[slot][j] = fmaxf(fmaf(x_regs[slot][j], in_scale[ci + j], in_shift[ci + j]), 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(x_regs[slot][2 * q], x_regs[slot][2 * q + 1], hh[q], mm[q], ll[q]);
    const u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, m = {mm[0], mm[1], mm[2], mm[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
    __bf16* pw = st + (skg * GB_M + srow) * 8;                  // row operand: weights
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(pw + i * GB_LIMB) = w_regs[slot][i];
    __bf16* px = st + GB_TILE + (skg * GB_N + srow) * 8;        // column operand: pixels
    *reinterpret_cast<u32x4*>(px) = h;
    *reinterpret_cast<u32x4*>(px + GB_LIMB) = m;
    *reinterpret_cast<u32x4*>(px + 2 * GB_LIMB) = l;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = Cin / GB_K;
  const int kg = lane >> 5, li = lane & 31;
  const int a_off = (kg * GB_M + wr * 64 + li) * 8, w_off = GB_TILE + (kg * GB_N + wc * 64 + li) * 8;
  fetch(0, 0);
  stash(0, lds, 0);
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
    stash(par ^ 1, lds + (par ^ 1) * GB_STAGE, kt + 1 < KT ? kt + 1 : KT - 1);
    fetch(par ^ 1, kt + 3 < KT ? kt + 3 : KT - 1);
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

  if constexpr (BITS) {
    // register r of block (rb, cb): query 64 wr + 32 rb + 4 kg + (r&3) + 8 (r>>2) = bit (4 kg + (r&3) + 8 (r>>2)) of word 2 wr + rb;
    // key = pixel 64 wc + 32 cb + li.  The two k-group halves of a wave hold complementary bits of the same words.
    unsigned w[2][2];                                    // [cb][rb]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        unsigned v = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (acc[rb][cb][r] < 0.f) v |= 1u << (4 * kg + (r & 3) + 8 * (r >> 2));
        v |= (unsigned)__shfl_xor((int)v, 32);
        w[cb][rb] = v;
      }
    // lane (li, kg) stores the key of column block cb = kg: words 2 wr, 2 wr + 1 of its 16-byte record
    const int p = p0 + wc * 64 + kg * 32 + li;
    const unsigned w0 = kg ? w[1][0] : w[0][0], w1 = kg ? w[1][1] : w[0][1];
    unsigned a0 = 0u, a1 = 0u;
    if (p < HWo) {
      unsigned* rec = reinterpret_cast<unsigned*>(y) + ((size_t)img * HWo + p) * 4 + 2 * wr;
      *reinterpret_cast<uint2*>(rec) = make_uint2(w0, w1);
      a0 = ~w0;
      a1 = ~w1;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      a0 |= (unsigned)__shfl_xor((int)a0, off);
      a1 |= (unsigned)__shfl_xor((int)a1, off);
    }
    // the flag words saturate after the first few workgroups (bits only ever get set): look before the atomic, or thousands
    // of workgroups serialise on four L2 atomics
    if (lane == 0) {
      const unsigned c0w = __hip_atomic_load(flags + 2 * wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned c1w = __hip_atomic_load(flags + 2 * wr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a0 & ~c0w) atomicOr(flags + 2 * wr, a0);
      if (a1 & ~c1w) atomicOr(flags + 2 * wr + 1, a1);
    }
    return;
  }
  // BN affine (+ identity) (+ ReLU): register r of block (rb, cb) = channel (r&3) + 8 (r>>2) + 4 kg of the block, pixel li.
  // Branch-free: scale / shift / identity are read through buffer descriptors (channels >= Cout read 0), the stores go
  // through a descriptor of this image's output (channels >= Cout are dropped by the bounds check) and lanes of pixels
  // beyond the map carry an offset outside every descriptor.  The guarded form was one load(scale, shift, identity) ->
  // `s_waitcnt vmcnt(0)` -> store chain per element: 64 dependent memory round trips per lane.  Needs Cout % 4 == 0 for the
  // float4 reads of scale / shift (checked by the entry point).
  {
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale), 0, scale ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const auto hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shift), 0, shift ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const size_t obase = (size_t)img * Cout * HWo;
    const unsigned img_bytes = (unsigned)((size_t)Cout * HWo * 4);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y + obase, 0, img_bytes, 0x00020000);
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual) + (RESIDUAL ? obase : 0), 0,
                                                       RESIDUAL ? img_bytes : 0u, 0x00020000);
    const unsigned chpitch = (unsigned)HWo * 4u;
    unsigned pvoff[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int p = p0 + wc * 64 + cb * 32 + li;
      pvoff[cb] = p < HWo ? (unsigned)p * 4u : 0x80000000u;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int chb = c0 + wr * 64 + rb * 32 + 4 * kg;               // channels chb + (r&3) + 8 (r>>2)
      f32x4 sc4[4], sh4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc4[j] = scale ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)(chb + 8 * j) * 4u, 0, 0))
                       : f32x4{1.f, 1.f, 1.f, 1.f};
        sh4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)(chb + 8 * j) * 4u, 0, 0));
      }
      float res[2][16];
      if (RESIDUAL) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            res[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rrs, pvoff[cb] + (unsigned)(chb + (r & 3) + 8 * (r >> 2)) * chpitch, 0, 0));
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = fmaf(acc[rb][cb][r], sc4[r >> 2][r & 3], sh4[r >> 2][r & 3]);
          if (RESIDUAL) v += res[cb][r];
          if (RELU) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs,
                                                pvoff[cb] + (unsigned)(chb + (r & 3) + 8 * (r >> 2)) * chpitch, 0, 0);
        }
    }
  }
}

// The 1x1 convolution on v_mfma_f32_16x16x32_bf16 with K = 32 stages (see gemm_bf16x3_k32_kernel): the same operands, epilogues
// and record formats as the kernel above; default for Cin % 32 == 0.
// TM = 64: layers with at most 64 output channels (the bottleneck's reducing 1x1 of layer1).  The packed weight is padded to 128
// rows and is staged whole, but only rows 0..63 are multiplied: the four waves each take 32 pixel columns of the 64 rows, half
// the matrix work of a 128-row tile, which leaves these layers to their HBM traffic.
// TAPS = 9: the 3x3 convolution (pad 1, stride 1 or 2) as an implicit GEMM over K = 9 * Cin, ordered [block of 32 input channels][tap][32 channels] (weight packed from
// pvsg_conv3x3_weight_matrix(w)): a 32-deep step lies inside one tap (Cin % 32 == 0), whose pixel offset replaces the 1x1 one;
// out-of-image taps read 0 through the descriptor's bounds check.
// F16: the two-limb f16 form (see split2h): weights (w_h, w_l, w_h2) resident in registers, pixels as (x_h, x_l').
template <bool RELU, bool RESIDUAL, bool IN_NORM, bool BITS = false, int TM = 128, int TAPS = 1, bool F16 = false>
__global__ __launch_bounds__(256, 3)
void conv1x1_bf16x3_k32_kernel(const float* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ scale,
                               const float* __restrict__ shift, const float* __restrict__ residual,
                               const float* __restrict__ in_scale, const float* __restrict__ in_shift, float* __restrict__ y,
                               int Cin, int Cout, int Cpad, int HWin, int Win, int HWo, int Wo, int stride, int tiles_c,
                               int tiles_p, unsigned* __restrict__ flags = nullptr, unsigned* __restrict__ overflow = nullptr,
                               int imgs_per_w = 0, long long w_batch_stride = 0, double* __restrict__ gn_part = nullptr) {
  constexpr int XL = F16 ? 2 : 3;                                // limbs of the on-the-fly (pixel) operand
  constexpr int WL = F16 ? 2 : 3;                                // arrays of the packed weight
  constexpr int X_AT = WL * K32_LIMB;                            // where the pixel tile starts
  __shared__ __attribute__((aligned(16))) __bf16 lds[X_AT + XL * K32_LIMB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  static_assert(TM == 128 || (TM == 64 && !BITS), "64-row tiles: plain convolution only");
  static_assert(TAPS == 1 || (TAPS == 9 && !BITS && !IN_NORM && !RESIDUAL), "3x3 taps: affine / ReLU epilogue only");
  constexpr int CB = TM == 128 ? 4 : 2;                          // 16-pixel column blocks per wave
  const int wr = TM == 128 ? wave >> 1 : 0, wc = wave & 1;
  const int wcol0 = TM == 128 ? wc * 64 : wave * 32;            // first pixel column of this wave in the tile
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tc = logical % tiles_c;
  logical /= tiles_c;
  const int tp = logical % tiles_p, img = logical / tiles_p;
  const int c0 = tc * GB_M, p0 = tp * GB_N;
  if (BITS && imgs_per_w > 0) {                                  // one launch for a batch: image img belongs to batch element
    const int bb = img / imgs_per_w;                             // img / imgs_per_w, which has its own packed embeddings / flags
    Wp += (long long)bb * w_batch_stride;
    flags += 4 * bb;
  }
  // staging: weights -- (k-group tid/128, row tid%128) of both 16-deep sub-steps, 3 limbs; pixels -- (k-group tid/128, pixel
  // tid%128), the 8 channels of that k-group in each sub-step
  const int skg = __builtin_amdgcn_readfirstlane(tid >> 7), srow = tid & 127;
  const size_t w_limb_stride = (size_t)2 * Cpad * 8;
  const __bf16* wsrc = Wp + ((size_t)skg * Cpad + c0 + srow) * 8;
  const int pix = p0 + srow;
  const int oy = pix / Wo, ox = pix - oy * Wo, Hin = HWin / Win;
  const int pin = stride == 1 ? pix : (2 * oy) * Win + 2 * ox;
  const unsigned x_voff = pix < HWo ? (unsigned)pin * 4u : 0x80000000u;            // beyond the map: read as 0
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)img * Cin * HWin, 0,
                                                      (unsigned)((size_t)Cin * HWin * 4), 0x00020000);
  const unsigned plane = (unsigned)HWin * 4u;
  float x_regs[2][8];                                 // [sub-step]
  u32x4 w_regs[2][WL];
  auto fetch = [&](int kt) {
    int cstep = kt * 32;                               // first input channel of this step
    unsigned voff = x_voff;
    if (TAPS == 9) {                                   // K order: [block of 32 input channels][tap][32 channels]
      const int cib = kt / 9, tap = kt - 9 * cib;
      cstep = cib * 32;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int iy = stride * oy + dy - 1, ix = stride * ox + dx - 1;
      voff = (pix < HWo && iy >= 0 && iy < Hin && ix >= 0 && ix < Win) ? (unsigned)(iy * Win + ix) * 4u : 0x80000000u;
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const unsigned so = (unsigned)(cstep + 16 * gq + 8 * skg) * plane;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#if defined(PVSG_ABL) && (PVSG_ABL == 8 || PVSG_ABL == 9)
        x_regs[gq][j] = 0.5f + (float)(so + j);        // lab build: no pixel loads (timing only)
#else
        x_regs[gq][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, voff, so + j * plane, PVSG_NT_LD));
#endif
      }
      const __bf16* wk = wsrc + (size_t)(2 * kt + gq) * WL * w_limb_stride;
#pragma unroll
      for (int l = 0; l < WL; ++l) w_regs[gq][l] = *reinterpret_cast<const u32x4*>(wk + l * w_limb_stride);
    }
  };
  u32x4 limbs[2][XL];                                  // split of step kt+1 under the MFMAs of step kt (see the GEMM kernel)
  float amax = 0.f;
  auto split = [&](int kt) {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      if (IN_NORM) {                                   // channel kt*32 + 16*gq + 8*skg + j of this image: wave-uniform scalars
        const int ci = img * Cin + kt * 32 + 16 * gq + 8 * skg;
#pragma unroll
        for (int j = 0; j < 8; ++j) x_regs[gq][j] = fmaxf(fmaf(x_regs[gq][j], in_scale[ci + j], in_shift[ci + j]), 0.f);
      }
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#if defined(PVSG_ABL) && PVSG_ABL == 9
        hh[q] = __builtin_bit_cast(unsigned, x_regs[gq][2 * q]); mm[q] = __builtin_bit_cast(unsigned, x_regs[gq][2 * q + 1]); ll[q] = hh[q];   // lab: no split either
#else
        if constexpr (F16) split2h(x_regs[gq][2 * q], x_regs[gq][2 * q + 1], hh[q], mm[q], amax);
        else split2(x_regs[gq][2 * q], x_regs[gq][2 * q + 1], hh[q], mm[q], ll[q]);
#endif
      }
      limbs[gq][0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
      limbs[gq][1] = u32x4{mm[0], mm[1], mm[2], mm[3]};
      if constexpr (!F16) limbs[gq][2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
  };
  auto write = [&]() {
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      __bf16* pw = lds + ((2 * gq + skg) * GB_M + srow) * 8;                  // row operand: weights
#pragma unroll
      for (int i = 0; i < WL; ++i) *reinterpret_cast<u32x4*>(pw + i * K32_LIMB) = w_regs[gq][i];
      __bf16* px = lds + X_AT + ((2 * gq + skg) * GB_N + srow) * 8;           // column operand: pixels
#pragma unroll
      for (int i = 0; i < XL; ++i) *reinterpret_cast<u32x4*>(px + i * K32_LIMB) = limbs[gq][i];
    }
  };
  f32x4 acc[4][CB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* afr = lds + (kg4 * GB_M + wr * 64 + l15) * 8;
  const __bf16* wfr = lds + X_AT + (kg4 * GB_N + wcol0 + l15) * 8;
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<F16>(a, b, c); };
  const int KT = TAPS * Cin / 32;
  fetch(0);
  split(0);
  write();
  for (int kt = 0; kt < KT; ++kt) {
    __syncthreads();
    fetch(kt + 1 < KT ? kt + 1 : KT - 1);     // (issued a step earlier, behind write(), this kernel spills 46 registers)
    u32x4 ahf[4], amf[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      ahf[rb] = frag(afr + rb * 128);
      amf[rb] = frag(afr + K32_LIMB + rb * 128);
    }
    if constexpr (F16) {                       // weights (w_h, w_l, 2^-11 w_h) resident; pixels (x_h, x_l') per column block
      u32x4 a2f[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a2f[rb] = f16x2_lo_scale(ahf[rb]);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 xh = frag(wfr + cb * 128), xl = frag(wfr + K32_LIMB + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(a2f[rb], xl, acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], xh, acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(ahf[rb], xh, acc[rb][cb]);
      }
      split(kt + 1 < KT ? kt + 1 : KT - 1);
    } else {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 wh = frag(wfr + cb * 128), wm = frag(wfr + K32_LIMB + cb * 128), wl = frag(wfr + 2 * K32_LIMB + cb * 128);
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
      split(kt + 1 < KT ? kt + 1 : KT - 1);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) amf[rb] = frag(afr + 2 * K32_LIMB + rb * 128);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 wh = frag(wfr + cb * 128);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(amf[rb], wh, acc[rb][cb]);
      }
    }
    __syncthreads();
    if (kt + 1 < KT) write();
  }
  if constexpr (F16) f16x2_count_overflow(amax, overflow);

  if constexpr (BITS) {
    // register r of block (rb, cb): query 64 wr + 16 rb + 4 kg4 + r = bit 16 (rb & 1) + 4 kg4 + r of word 2 wr + (rb >> 1);
    // key = pixel 64 wc + 16 cb + l15.  The four lane groups of a wave hold complementary bits of the same words.
    unsigned w[4][2];                                    // [cb][word]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int ws = 0; ws < 2; ++ws) {
        unsigned v = 0u;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (acc[2 * ws + h2][cb][r] < 0.f) v |= 1u << (16 * h2 + 4 * kg4 + r);
        v |= (unsigned)__shfl_xor((int)v, 16);
        v |= (unsigned)__shfl_xor((int)v, 32);
        w[cb][ws] = v;
      }
    // lane (l15, kg4) stores the key of column block cb = kg4: words 2 wr, 2 wr + 1 of its 16-byte record
    const int p = p0 + wc * 64 + kg4 * 16 + l15;
    const unsigned w0 = kg4 == 0 ? w[0][0] : kg4 == 1 ? w[1][0] : kg4 == 2 ? w[2][0] : w[3][0];
    const unsigned w1 = kg4 == 0 ? w[0][1] : kg4 == 1 ? w[1][1] : kg4 == 2 ? w[2][1] : w[3][1];
    unsigned a0 = 0u, a1 = 0u;
    if (p < HWo) {
      unsigned* rec = reinterpret_cast<unsigned*>(y) + ((size_t)img * HWo + p) * 4 + 2 * wr;
      *reinterpret_cast<uint2*>(rec) = make_uint2(w0, w1);
      a0 = ~w0;
      a1 = ~w1;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      a0 |= (unsigned)__shfl_xor((int)a0, off);
      a1 |= (unsigned)__shfl_xor((int)a1, off);
    }
    if (lane == 0) {                                     // look before the atomic (see the kernel above)
      const unsigned c0w = __hip_atomic_load(flags + 2 * wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned c1w = __hip_atomic_load(flags + 2 * wr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a0 & ~c0w) atomicOr(flags + 2 * wr, a0);
      if (a1 & ~c1w) atomicOr(flags + 2 * wr + 1, a1);
    }
    return;
  }
  // BN affine (+ identity) (+ ReLU), branch-free through buffer descriptors (see the kernel above): register r of block
  // (rb, cb) = channel 16 rb + 4 kg4 + r, pixel 16 cb + l15 of the wave's 64 x 64 tile
  {
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale), 0, scale ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const auto hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shift), 0, shift ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const size_t obase = (size_t)img * Cout * HWo;
    const unsigned img_bytes = (unsigned)((size_t)Cout * HWo * 4);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y + obase, 0, img_bytes, 0x00020000);
    const auto rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual) + (RESIDUAL ? obase : 0), 0,
                                                       RESIDUAL ? img_bytes : 0u, 0x00020000);
    const unsigned chpitch = (unsigned)HWo * 4u;
    const float unscale = F16 ? f16x2_unscale(Wp, Cpad, TAPS * Cin) : 1.f;
    unsigned pvoff[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int p = p0 + wcol0 + cb * 16 + l15;
      pvoff[cb] = p < HWo ? (unsigned)p * 4u : 0x80000000u;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int chb = c0 + wr * 64 + rb * 16 + 4 * kg4;              // channels chb .. chb + 3
      f32x4 sc4 = scale ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)chb * 4u, 0, 0))
                        : f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (F16) sc4 *= unscale;                             // the packed weight's 2^-e (exact)
      const f32x4 sh4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)chb * 4u, 0, 0));
      float res[CB][4];
      if (RESIDUAL) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            res[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, pvoff[cb] + (unsigned)(chb + r) * chpitch, 0, PVSG_NT_LD));
      }
      float gs = 0.f, gq = 0.f;                                     // GroupNorm statistics of what is stored (gn_part != nullptr)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[rb][cb][r], sc4[r], sh4[r]);
          if (RESIDUAL) v += res[cb][r];
          if (RELU) v = fmaxf(v, 0.f);
          if (pvoff[cb] != 0x80000000u) { gs += v; gq = fmaf(v, v, gq); }
          if (PVSG_ABL == 11 && v != 1.2345e33f) continue;        // lab build: no epilogue stores (timing only)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, pvoff[cb] + (unsigned)(chb + r) * chpitch, 0, PVSG_NT_ST);
        }
      if constexpr (TM == 128) {
        if (gn_part) {
          // groups of 8 channels: lanes (l15 = 0..15, kg4 in {0,1} | {2,3}) x this row block hold one group's values of the wave's
          // 64 pixels.  Fixed-order reduction (bit-reproducible); one (sum, sum of squares) pair per (image, group, pixel tile,
          // wave column) into `gn_part`, summed in f64 by gn_finish_kernel: the statistics pass over the 1.9 GB the convolution
          // has just written never runs.
#pragma unroll
          for (int off = 1; off <= 16; off <<= 1) { gs += __shfl_xor(gs, off); gq += __shfl_xor(gq, off); }
          if (l15 == 0 && (kg4 & 1) == 0 && chb < Cout) {
            const int g = (chb >> 3), G = Cout >> 3;
            double* dst = gn_part + ((((size_t)img * G + g) * tiles_p + tp) * 2 + wc) * 2;
            dst[0] = (double)gs;
            dst[1] = (double)gq;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 convolution (pad 1, stride 1), f16x2, with the input staged ONCE per block of 32 input channels.  The kernel above reads
// and splits the pixel operand again for each of the nine taps; with the pixel loads removed it runs 25-33 % faster
// (scripts/lab/abl_split.sh 8).  Here a workgroup owns 128 output channels x an 8 x 16 block of output pixels; per channel
// block it stages the 10 x 18 input patch (zero halo through the descriptor's bounds check), split into its two f16 limbs, as
// [limb][k-group 4][patch pixel 180][8 channels] -- 22.5 KB -- and runs the nine taps from it: a tap's pixel fragment is the
// patch row (block row + dy), columns dx .. dx + 15, sixteen consecutive 16-byte records.  The weights of a tap (16 KB) arrive by
// LDS-DMA into one of two buffers while the previous tap computes; one barrier per tap, two more per channel block.
//   per channel block and thread: 24 dword loads + 12 split pairs (the tap-by-tap form: 144 + 72)
// CT = 128: 128 output channels x 8 x 16 pixels (wave = 64 channels x 4 pixel rows); CT = 64 (the 64-channel layers): 64 output
// channels x 16 x 16 pixels (wave = all 64 channels x 4 pixel rows), patch 18 x 18.
// NBUF weight buffers: tap t's slabs are asked for NBUF - 1 taps ahead (PVSG_HALO_NBUF, lab: scripts/lab/r05_halo_nbuf.sh)
#ifndef PVSG_HALO_NBUF
#define PVSG_HALO_NBUF 2
#endif
template <int CT>
constexpr int halo_lds_bytes() {
  return (2 * 4 * ((CT == 128 ? 10 : 18) * 18) * 8 + PVSG_HALO_NBUF * 2 * 4 * CT * 8) * 2;
}
template <bool RELU, int CT>
__global__ __launch_bounds__(256, 2)
void conv3x3_f16x2_halo_kernel(const float* __restrict__ x, const __bf16* __restrict__ Wp, const float* __restrict__ scale,
                               const float* __restrict__ shift, float* __restrict__ y, int Cin, int Cout, int Cpad, int H, int W,
                               int tiles_c, int tiles_x, int tiles_y, unsigned* __restrict__ overflow,
                               double* __restrict__ gn_part = nullptr) {
  constexpr int PR = CT == 128 ? 8 : 16;                         // pixel rows of the tile
  constexpr int PH = PR + 2, PW = 18, PP = PH * PW;              // patch
  constexpr int P_KG = PP * 8, P_LIMB = 4 * P_KG;                // elements
  constexpr int W_ARR = 4 * CT * 8;                              // one limb array of a tap: [k-group 4][CT][8]
  constexpr int W_AT = 2 * P_LIMB, W_BUF = 2 * W_ARR;
  constexpr int NR = (4 * PP + 255) / 256;                       // staging rounds
  constexpr int NBUF = PVSG_HALO_NBUF;
  static_assert(halo_lds_bytes<CT>() == (W_AT + NBUF * W_BUF) * 2, "halo LDS size");
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int tc = logical % tiles_c;
  logical /= tiles_c;
  const int tx = logical % tiles_x;
  logical /= tiles_x;
  const int ty = logical % tiles_y, img = logical / tiles_y;
  const int c0 = tc * CT, oy0 = ty * PR, ox0 = tx * 16;
  const int ch_w = CT == 128 ? wr * 64 : 0;                       // the wave's first channel / pixel row inside the tile
  const int pr_w = CT == 128 ? wc * 4 : wr * 8 + wc * 4;
  const int HW = H * W;
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)img * Cin * HW, 0,
                                                      (unsigned)((size_t)Cin * HW * 4), 0x00020000);
  const unsigned plane = (unsigned)HW * 4u;
  // staging items (patch pixel, k-group): item i = round * 256 + tid, pixel i % PP, k-group i / PP; 4 PP items in NR rounds
  unsigned it_voff[NR];
  int it_lds[NR];
  unsigned it_kg[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = r * 256 + tid;
    const int kg = i / PP, pp = i - kg * PP;
    const int py = pp / PW, px = pp - py * PW;
    const int iy = oy0 + py - 1, ix = ox0 + px - 1;
    const bool ok = i < 4 * PP && iy >= 0 && iy < H && ix >= 0 && ix < W;
    it_voff[r] = ok ? (unsigned)(iy * W + ix) * 4u : 0x80000000u;          // halo / beyond the map: read as 0
    it_kg[r] = (unsigned)(kg < 4 ? kg : 3);
    it_lds[r] = i < 4 * PP ? kg * P_KG + pp * 8 : -1;
  }
  float xr[NR][8];
  auto loadX = [&](int cib) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned so = (unsigned)(cib * 32 + 8 * it_kg[r]) * plane;
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, it_voff[r], so + j * plane, 0));
    }
  };
  float amax = 0.f;
  auto stashX = [&]() {                                          // split + write the patch of the channel block in xr
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      unsigned hh[4], ll[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split2h(xr[r][2 * q], xr[r][2 * q + 1], hh[q], ll[q], amax);
      if (it_lds[r] >= 0) {
        *reinterpret_cast<u32x4*>(lds + it_lds[r]) = u32x4{hh[0], hh[1], hh[2], hh[3]};
        *reinterpret_cast<u32x4*>(lds + P_LIMB + it_lds[r]) = u32x4{ll[0], ll[1], ll[2], ll[3]};
      }
    }
    asm volatile("" : "+v"(amax));
  };
  // weight slabs of a tap: (array l, k-group kg of 4, row half) = 16 x 1 KB; wave w brings 4 w .. 4 w + 3.  Packed K order
  // [channel block][tap][32]: 32-deep step index = cib * 9 + tap.
  const size_t w_kg_stride = (size_t)Cpad * 8;
  auto dmaW = [&](int step, int buf) {
    constexpr int HALVES = CT / 64, SLABS = 8 * HALVES / 4;        // (array, k-group, 64-channel half) slabs of 1 KB; per wave
#pragma unroll
    for (int i = 0; i < SLABS; ++i) {
      const int sl = wave * SLABS + i, half = sl % HALVES, kg = (sl / HALVES) & 3, l = sl / (4 * HALVES);
      const __bf16* src = Wp + ((((size_t)(2 * step + (kg >> 1)) * 2 + l) * 2 + (kg & 1)) * w_kg_stride) + (size_t)(c0 + half * 64 + lane) * 8;
      __bf16* dst = lds + W_AT + buf * W_BUF + l * W_ARR + (kg * CT + half * 64) * 8;
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i >> 2][i & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l15 = lane & 15, kg4 = lane >> 4;
  const __bf16* wfr0 = lds + W_AT + (kg4 * CT + ch_w + l15) * 8;               // + buffer + array * W_ARR + row block * 128
  const __bf16* xfr0 = lds + kg4 * P_KG + (pr_w * PW + l15) * 8;               // + limb * P_LIMB + ((cb + dy) * PW + dx) * 8
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<true>(a, b, c); };
  const int NCB = Cin / 32, NSTEP = NCB * 9;
  loadX(0);
  dmaW(0, 0);
  if (NBUF == 3) dmaW(NSTEP > 1 ? 1 : 0, 1);
  stashX();
  int step = 0, buf = 0;                                          // buf = step % NBUF
  constexpr int DMA_OPS = 8 * (CT / 64) / 4, LOAD_OPS = 8 * NR;   // vector-memory operations of one dmaW / loadX per wave
  for (int cib = 0; cib < NCB; ++cib) {
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++step) {
      // own slabs of this tap (and, at tap 0, own patch rows: stashX waited for them) are in LDS.  NBUF == 3: the slabs of the NEXT
      // tap (asked for one tap ago) and, right behind tap 0, the next channel block's patch loads may still be in flight -- loads
      // retire in order, so the count of younger operations is what may remain outstanding
      if (NBUF == 3) {
        if (tap == 1 || tap == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA_OPS + LOAD_OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA_OPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                               // ... everybody's are; the buffer of the previous tap is free
      {
        const int ahead = NBUF - 1, nb = buf + ahead >= NBUF ? buf + ahead - NBUF : buf + ahead;
        dmaW(step + ahead < NSTEP ? step + ahead : NSTEP - 1, nb);
      }
      if (tap == 0) loadX(cib + 1 < NCB ? cib + 1 : cib);         // next channel block's patch: lands under the nine taps
      const __bf16* wfr = wfr0 + buf * W_BUF;
      buf = buf + 1 == NBUF ? 0 : buf + 1;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const __bf16* xfr = xfr0 + (dy * PW + dx) * 8;
      u32x4 whf[4], wlf[4], w2f[4];
      u32x4 xh = frag(xfr), xl = frag(xfr + P_LIMB);               // pixel fragments one column block ahead of their MFMAs
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        whf[rb] = frag(wfr + rb * 128);
        wlf[rb] = frag(wfr + W_ARR + rb * 128);
      }
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) w2f[rb] = f16x2_lo_scale(whf[rb]);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {                             // small terms first: (2^-11 w_h, x_l') (w_l, x_h) (w_h, x_h)
        u32x4 xhn = xh, xln = xl;
        if (cb < 3) {
          xhn = frag(xfr + (cb + 1) * PW * 8);
          xln = frag(xfr + P_LIMB + (cb + 1) * PW * 8);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(w2f[rb], xl, acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(wlf[rb], xh, acc[rb][cb]);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][cb] = mf(whf[rb], xh, acc[rb][cb]);
        __builtin_amdgcn_sched_barrier(0);
        xh = xhn;
        xl = xln;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // everyone is done with this channel block's patch
    if (cib + 1 < NCB) stashX();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (the repeated last slabs: nothing may land after the end)
  // BN affine (+ ReLU): register r of block (rb, cb) = channel ch_w + 16 rb + 4 kg4 + r, pixel (row pr_w + cb, column l15) of the tile
  {
    const auto srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale), 0, scale ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const auto hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shift), 0, shift ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)img * Cout * HW, 0, (unsigned)((size_t)Cout * HW * 4), 0x00020000);
    const unsigned chpitch = (unsigned)HW * 4u;
    const float unscale = f16x2_unscale(Wp, Cpad, 9 * Cin);
    unsigned pvoff[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int oy = oy0 + pr_w + cb, ox = ox0 + l15;
      pvoff[cb] = (oy < H && ox < W) ? (unsigned)(oy * W + ox) * 4u : 0x80000000u;
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const int chb = c0 + ch_w + rb * 16 + 4 * kg4;              // channels chb .. chb + 3 (>= Cout: dropped by the descriptor)
      f32x4 sc4 = scale ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, (unsigned)chb * 4u, 0, 0))
                        : f32x4{1.f, 1.f, 1.f, 1.f};
      sc4 *= unscale;
      const f32x4 sh4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, (unsigned)chb * 4u, 0, 0));
      float gs = 0.f, gq = 0.f;                                     // GroupNorm statistics of what is stored (gn_part != nullptr)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[rb][cb][r], sc4[r], sh4[r]);
          if (RELU) v = fmaxf(v, 0.f);
          if (pvoff[cb] != 0x80000000u) { gs += v; gq = fmaf(v, v, gq); }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrs, pvoff[cb] + (unsigned)(chb + r) * chpitch, 0, 0);
        }
      if constexpr (CT == 128) {
        if (gn_part) {
          // as in conv1x1_bf16x3_k32_kernel: groups of 8 channels = lanes (l15, kg4 in {0,1} | {2,3}) of this row block, the wave's
          // 4 x 16 pixels; one (sum, sum of squares) pair per (image, group, pixel tile, wave pixel half), fixed order
#pragma unroll
          for (int off = 1; off <= 16; off <<= 1) { gs += __shfl_xor(gs, off); gq += __shfl_xor(gq, off); }
          if (l15 == 0 && (kg4 & 1) == 0 && chb < Cout) {
            const int g = (chb >> 3), G = Cout >> 3;
            double* dst = gn_part + ((((size_t)img * G + g) * (tiles_y * tiles_x) + (ty * tiles_x + tx)) * 2 + wc) * 2;
            dst[0] = (double)gs;
            dst[1] = (double)gq;
          }
        }
      }
    }
  }
  f16x2_count_overflow(amax, overflow);
}

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

// ------------------------------------------------------------------------------------------------------------------
// Tail of a 64-plane ResNet bottleneck and the head of the next one in ONE pass over the pixels
// ([3P] mmdet ResNet Bottleneck.forward: out = relu(bn3(conv3(mid)) + identity); next block: relu(bn1(conv1(out)))):
//   y        = relu(conv3(mid) * scale3 + shift3 + identity)       64 -> 256 channels, written once
//   mid_next = relu(conv1n(y) * scale1n + shift1n)                 256 -> 64 channels, from y while it is still in registers
// Separately the pair costs 0.82 + 0.48 ms at 32 x 184 x 320 (4.3 + 2.4 GB); the 1.9 GB re-read of y by the next conv1 is what
// goes away (4.8 GB here).  Both weights are tiny (64 KB of limbs each), so a PERSISTENT workgroup (one per CU, 8 waves) keeps
// them in LDS and every wave walks 32-pixel tiles on its own: no barrier after the fill.
//   lane (l15, kg4) owns pixels p0 + 2 l15, + 1 (column blocks cb = 0 / 1): every access is an 8-byte pair, 128 B per channel row.
//   conv3: B fragments = the split of the mid tile (channel 32 kc + 8 kg4 + e, straight from memory); per chunk of 64 output
//   channels 2 x 4 x 2 x 3 MFMAs, epilogue (affine, identity, ReLU), store.
//   conv1n: the chunk's 64 values per lane ARE two of its B fragments -- register r of row block rb is channel 64 oc + 16 rb +
//   4 kg4 + r, so k position 8 kg4 + e of 32-chunk 2 oc + h takes channel offset 16 (e >> 2) + 4 kg4 + (e & 3): conv1n's weight is
//   packed in that K order (pvsg_bottleneck_next_weight_matrix) and nothing moves between lanes.
typedef unsigned u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
constexpr int BT_W3_ELEMS = 2 * 256 * 64;                 // [k-tile 4][limb 2][k-group 2][256 rows][8]
constexpr int BT_W1_ELEMS = 2 * 64 * 256;                 // [k-tile 16][limb 2][k-group 2][64 rows][8]
constexpr int BT_LDS_BYTES = (BT_W3_ELEMS + BT_W1_ELEMS) * 2;      // 128 KB
// MODE 0: y only; 1: y and mid_next = conv1n(y); 2 ("head" of the stage's first block): y = conv_ds(x) * scale3 + shift3 with NO
// identity / ReLU (the downsample branch) and mid_next = relu(conv1(x) * scale1 + shift1) from the SAME x tile (W1p: plain K order).
template <int MODE>
__global__ __launch_bounds__(512, 1)
void bottleneck_tail64_kernel(const float* __restrict__ mid, const __bf16* __restrict__ W3p, const float* __restrict__ scale3,
                              const float* __restrict__ shift3, const float* __restrict__ identity, float* __restrict__ y,
                              const __bf16* __restrict__ W1p, const float* __restrict__ scale1, const float* __restrict__ shift1,
                              float* __restrict__ mid_next, int HW, int tiles_per_img, long long total_tiles,
                              unsigned* __restrict__ overflow, float* __restrict__ y_s2, int W) {
  extern __shared__ __attribute__((aligned(16))) __bf16 ldsbt[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NX = MODE == 3 ? 128 : 64;               // channels of mid_next; MODE 3: W1n (128 x 256, 128 KB) fills the LDS and
  constexpr int NH = NX / 64;                            // conv3's fragments come straight from global memory (L2-resident 64 KB)
  constexpr int W1_AT = MODE == 3 ? 0 : BT_W3_ELEMS;
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(MODE == 3 ? W1p : W3p);
    u32x4* dst = reinterpret_cast<u32x4*>(ldsbt);
#pragma unroll
    for (int i = 0; i < (MODE == 3 ? 2 * 128 * 256 : BT_W3_ELEMS) / 8 / 512; ++i) dst[i * 512 + tid] = src[i * 512 + tid];
    if (MODE == 1) {                                     // packed with Npad = 128: rows 0..63 of every (k-tile, limb, k-group) slab
      const u32x4* s1 = reinterpret_cast<const u32x4*>(W1p);
      u32x4* d1 = reinterpret_cast<u32x4*>(ldsbt + BT_W3_ELEMS);
#pragma unroll
      for (int i = 0; i < BT_W1_ELEMS / 8 / 512; ++i) {
        const int it = i * 512 + tid, slab = it >> 6, row = it & 63;
        d1[it] = s1[slab * 128 + row];
      }
    } else if (MODE == 2) {                              // (64, 64): 16 slabs
      const u32x4* s1 = reinterpret_cast<const u32x4*>(W1p);
      u32x4* d1 = reinterpret_cast<u32x4*>(ldsbt + BT_W3_ELEMS);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int it = i * 512 + tid, slab = it >> 6, row = it & 63;
        d1[it] = s1[slab * 128 + row];
      }
    }
  }
  __syncthreads();
  const int l15 = lane & 15, kg4 = lane >> 4;
  const float un3 = f16x2_unscale(W3p, 256, 64);
  const float un1 = (MODE == 1 || MODE == 3) ? f16x2_unscale(W1p, 128, 256) : MODE == 2 ? f16x2_unscale(W1p, 128, 64) : 1.f;
  const auto s3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale3), 0, 256u * 4u, 0x00020000);
  const auto h3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shift3), 0, 256u * 4u, 0x00020000);
  const __bf16* w3fr = ldsbt + ((size_t)((kg4 >> 1) * 4 + (kg4 & 1)) * 256 + l15) * 8;      // + (2 kc) k-tiles, limb, row
  const __bf16* w1fr = ldsbt + W1_AT + ((size_t)((kg4 >> 1) * 4 + (kg4 & 1)) * NX + l15) * 8;
  const auto w3rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W3p), 0, (unsigned)BT_W3_ELEMS * 2u, 0x00020000);
  const unsigned w3vo = (unsigned)(((kg4 >> 1) * 4 + (kg4 & 1)) * 256 + l15) * 16u;
  auto frag = [](const __bf16* p) { return *reinterpret_cast<const u32x4*>(p); };
  auto mf = [](u32x4 a, u32x4 b, f32x4 c) { return mfma_k32<true>(a, b, c); };
  // One 32-deep k chunk: 4 row blocks x 2 column blocks x 3 limb products.  v_mfma_f32_16x16x32_f16 reads its A / B registers over
  // several of its 8 passes, and nothing (hardware or hipcc 7.2's hazard recogniser, which guards SrcC only) stops a VALU
  // instruction issued right behind it from overwriting them: with the fragments recycled for the epilogue's v_pk_mul_f32 the last
  // rows of a block (lanes 48..63) came out wrong in ~1 launch of 3 (scripts/lab/bneck_head_repeat.py, round 5).  So: all fragments of the
  // chunk live in their own registers before the first MFMA, and a full MFMA duration of s_nop separates the last MFMA from the
  // next VALU write; the scheduling barriers keep the compiler from moving anything across.
  auto group = [&](auto ld, const u32x4 (&bh)[2], const u32x4 (&bl)[2], f32x4 (&acc)[4][2]) {
    u32x4 whf[4], wlf[4], w2f[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      whf[rb] = ld(rb, 0);
      wlf[rb] = ld(rb, 1);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) w2f[rb] = f16x2_lo_scale(whf[rb]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[rb][cb] = mf(w2f[rb], bl[cb], acc[rb][cb]);
        acc[rb][cb] = mf(wlf[rb], bh[cb], acc[rb][cb]);
        acc[rb][cb] = mf(whf[rb], bh[cb], acc[rb][cb]);
      }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lds_ld = [&](const __bf16* wbase, int limb_off) {     // fragments of 4 row blocks from an LDS-resident weight
    return [=](int rb, int limb) { return frag(wbase + limb * limb_off + rb * 128); };
  };
  auto w3_ld = [&](int kc, int oc) {                           // conv3's: LDS, or (MODE 3) global memory
    return [=](int rb, int limb) {
      const int el = ((2 * kc) * 4 * 256 + 64 * oc + 16 * rb) * 8 + limb * 2 * 256 * 8;
      if constexpr (MODE == 3) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w3rs, w3vo, (unsigned)el * 2u, 0));
      else return frag(w3fr + el);
    };
  };
  const unsigned plane = (unsigned)HW * 4u;
  const int Ho = (HW / W + 1) >> 1, Wo = (W + 1) >> 1;
  const unsigned plane2 = (unsigned)(Ho * Wo) * 4u;
  float amax = 0.f;
  const long long wstride = (long long)gridDim.x * 8;
  for (long long t = (long long)blockIdx.x * 8 + wave; t < total_tiles; t += wstride) {
    const int img = (int)(t / tiles_per_img), p0 = (int)(t - (long long)img * tiles_per_img) * 32;
    const int p = p0 + 2 * l15;
    const unsigned pv = p < HW ? (unsigned)p * 4u : 0x80000000u;                   // (HW is even: a pair is inside or outside)
    const auto mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(mid) + (size_t)img * 64 * HW, 0, 64u * plane, 0x00020000);
    const auto irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(identity) + (MODE == 2 ? 0 : (size_t)img * 256 * HW), 0,
                                                       MODE == 2 ? 0u : 256u * plane, 0x00020000);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)img * 256 * HW, 0, 256u * plane, 0x00020000);
    // ---- the mid tile: 64 channels x 32 pixels, split into the B fragments of conv3
    u32x4 xh[2][2], xl[2][2];
    {
      f32x2 xr[2][8];
      const unsigned vo = pv + (unsigned)(8 * kg4) * plane;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          xr[kc][e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mrs, vo, (unsigned)(32 * kc + e) * plane, 0));
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          unsigned hh[4], ll[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) split2h(xr[kc][2 * q][cb], xr[kc][2 * q + 1][cb], hh[q], ll[q], amax);
          xh[kc][cb] = u32x4{hh[0], hh[1], hh[2], hh[3]};
          xl[kc][cb] = u32x4{ll[0], ll[1], ll[2], ll[3]};
        }
    }
    f32x4 acc1[NH][4][2];
#pragma unroll
    for (int i = 0; i < 8 * NH; ++i) acc1[i >> 3][(i >> 1) & 3][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the stride-2 copy of y for the next stage's downsample convolution (even rows / columns; y_s2 may be NULL)
    unsigned vo2 = 0x80000000u;
    if (y_s2) {
      const int oy = p / W, ox = p - oy * W;
      if (p < HW && !(oy & 1) && !(ox & 1)) vo2 = (unsigned)((oy >> 1) * Wo + (ox >> 1)) * 4u + (unsigned)(4 * kg4) * plane2;
    }
    const auto y2rs = __builtin_amdgcn_make_buffer_rsrc(y_s2 ? y_s2 + (size_t)img * 256 * Ho * Wo : y, 0, y_s2 ? 256u * plane2 : 0u, 0x00020000);
    const unsigned vo4 = pv + (unsigned)(4 * kg4) * plane;                         // channel 4 kg4 of a 16-channel row block
    if constexpr (MODE == 2) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) group(lds_ld(w1fr + ((size_t)(2 * kc) * 4 * 64) * 8, 2 * 64 * 8), xh[kc], xl[kc], acc1[0]);
    }
#pragma unroll 1
    for (int oc = 0; oc < 4; ++oc) {
      f32x2 idn[4][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          idn[rb][r] = MODE == 2 ? f32x2{0.f, 0.f}
                                 : __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(irs, vo4, (unsigned)(64 * oc + 16 * rb + r) * plane, 0));
      f32x4 acc3[4][2];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc3[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) group(w3_ld(kc, oc), xh[kc], xl[kc], acc3);
      // epilogue of the chunk: register r of (rb, cb) = channel 64 oc + 16 rb + 4 kg4 + r, pixel p + cb
      float v[4][2][4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const unsigned cho = (unsigned)(64 * oc + 16 * rb + 4 * kg4) * 4u;
        f32x4 sc4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s3rs, cho, 0, 0));
        sc4 *= un3;
        const f32x4 sh4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h3rs, cho, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const float t = fmaf(acc3[rb][cb][r], sc4[r], sh4[r]) + idn[rb][r][cb];
            v[rb][cb][r] = MODE == 2 ? t : fmaxf(t, 0.f);
          }
          const f32x2 o = {v[rb][0][r], v[rb][1][r]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, o), yrs, vo4, (unsigned)(64 * oc + 16 * rb + r) * plane, 0);
          if (y_s2)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[rb][0][r]), y2rs, vo2, (unsigned)(64 * oc + 16 * rb + r) * plane2, 0);
        }
      }
      if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 yh[2], yl[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            unsigned hh[4], ll[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)                                       // k positions 2 q, 2 q + 1: row block 2 h + (q >> 1)
              split2h(v[2 * h + (q >> 1)][cb][2 * (q & 1)], v[2 * h + (q >> 1)][cb][2 * (q & 1) + 1], hh[q], ll[q], amax);
            yh[cb] = u32x4{hh[0], hh[1], hh[2], hh[3]};
            yl[cb] = u32x4{ll[0], ll[1], ll[2], ll[3]};
          }
#pragma unroll
          for (int half = 0; half < NH; ++half)
            group(lds_ld(w1fr + ((size_t)(2 * (2 * oc + h)) * 4 * NX + 64 * half) * 8, 2 * NX * 8), yh, yl, acc1[half]);
        }
      }
    }
    if constexpr (MODE != 0) {
      const auto nrs = __builtin_amdgcn_make_buffer_rsrc(mid_next + (size_t)img * NX * HW, 0, (unsigned)NX * plane, 0x00020000);
      const auto s1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale1), 0, (unsigned)NX * 4u, 0x00020000);
      const auto h1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(shift1), 0, (unsigned)NX * 4u, 0x00020000);
#pragma unroll
      for (int rbx = 0; rbx < 4 * NH; ++rbx) {
        const int half = rbx >> 2, rb = rbx & 3;
        const unsigned cho = (unsigned)(16 * rbx + 4 * kg4) * 4u;
        f32x4 sc4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1rs, cho, 0, 0));
        sc4 *= un1;
        const f32x4 sh4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h1rs, cho, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x2 o = {fmaxf(fmaf(acc1[half][rb][0][r], sc4[r], sh4[r]), 0.f), fmaxf(fmaf(acc1[half][rb][1][r], sc4[r], sh4[r]), 0.f)};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, o), nrs, vo4, (unsigned)(16 * rbx + r) * plane, 0);
        }
      }
    }
  }
  f16x2_count_overflow(amax, overflow);
}

// (Cn, K) -> the same matrix with the K order conv1n is multiplied in by bottleneck_tail64_kernel (see there)
__global__ void bottleneck_next_matrix_kernel(const float* __restrict__ w, float* __restrict__ m, int K, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long n = i / K;
  const int k = (int)(i - n * K), c = k >> 5, pos = k & 31, g = pos >> 3, e = pos & 7;
  m[i] = w[n * K + 32 * c + 16 * (e >> 2) + 4 * g + (e & 3)];
}

}  // namespace
}  // namespace pvsg

#if defined(PVSG_ABL) && PVSG_ABL == 7
extern "C" int pvsg_lab_split_phase(unsigned long long* host8, int reset) {      // lab builds only (scripts/lab/abl_split.sh 7)
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(pvsg::g_split_phase), 64) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pvsg::g_split_phase), z, 64) != hipSuccess) return 1; }
  return 0;
}
#endif

extern "C" long long pvsg_gemm_f16x2_packed_elems(int N, int K) {
  const long long npad = (N + 127) / 128 * 128;
  return 2LL * npad * K + 8;                              // 16-bit elements; the last 8 hold (max|w|, 2^-e, 0, 0) as floats
}

extern "C" int pvsg_gemm_f16x2_pack(const float* weight, void* w_packed, int N, int K, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && w_packed, "gemm_f16x2_pack: null pointer argument");
  PVSG_REQUIRE(N > 0 && K > 0, "gemm_f16x2_pack: bad shape");
  if (K % 32) return set_err(PVSG_ERR_UNSUPPORTED, "gemm_f16x2: built for K %% 32 == 0 (got %d)", K);
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(w_packed) & 15u), "gemm_f16x2_pack: w_packed must be 16-byte aligned");
  const int Npad = (N + 127) / 128 * 128;
  hipStream_t st = static_cast<hipStream_t>(stream);
  __bf16* wp = static_cast<__bf16*>(w_packed);
  unsigned* tail = reinterpret_cast<unsigned*>(wp + (size_t)2 * Npad * K);
  hipError_t e = zero_words_async(tail, 16, st);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "gemm_f16x2_pack: %s", hipGetErrorString(e));
  const long long n = (long long)N * K;
  const unsigned ablocks = (unsigned)(n / 1024 + 1 < 512 ? n / 1024 + 1 : 512);
  hipLaunchKernelGGL(f16x2_amax_kernel, dim3(ablocks), dim3(256), 0, st, weight, n, tail);
  const long long total = (long long)Npad * (K / 2);
  hipLaunchKernelGGL(gemm_f16x2_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, wp, N, K, Npad);
  PVSG_LAUNCH_CHECK("gemm_f16x2_pack");
  return PVSG_OK;
}

extern "C" long long pvsg_gemm_bf16x3_packed_elems(int N, int K) {
  const long long npad = (N + 127) / 128 * 128;
  return 3LL * npad * K;                                  // bf16 elements
}

extern "C" int pvsg_gemm_bf16x3_pack(const float* weight, void* w_packed, int N, int K, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && w_packed, "gemm_bf16x3_pack: null pointer argument");
  PVSG_REQUIRE(N > 0 && K > 0, "gemm_bf16x3_pack: bad shape");
  if (K % GB_K) return set_err(PVSG_ERR_UNSUPPORTED, "gemm_bf16x3: built for K %% 16 == 0 (got %d)", K);
  const int Npad = (N + 127) / 128 * 128;
  const long long total = (long long)Npad * (K / 2);
  hipLaunchKernelGGL(gemm_bf16x3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), weight, static_cast<__bf16*>(w_packed), N, K, Npad);
  PVSG_LAUNCH_CHECK("gemm_bf16x3_pack");
  return PVSG_OK;
}

static int gemm_split_run(const float* a, const void* w_packed, const float* bias, float* out, long long M, int N, int K,
                          int relu, bool f16, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  const char* nm = f16 ? "gemm_f16x2" : "gemm_bf16x3";
  PVSG_REQUIRE(a && w_packed && out, "%s: null pointer argument", nm);
  PVSG_REQUIRE(M > 0 && N > 0 && K > 0, "%s: bad shape", nm);
  if (K % (f16 ? 32 : GB_K) || M >= (1LL << 31) || (long long)GB_M * K * 4 >= (1LL << 31) || (long long)GB_M * N * 4 >= (1LL << 31))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for K %% %d == 0, M < 2^31, 128 rows < 2 GiB (got M=%lld N=%d K=%d)", nm,
                   f16 ? 32 : GB_K, M, N, K);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w_packed)) & 15u),
               "%s: a and w_packed must be 16-byte aligned", nm);
  const int Npad = (N + 127) / 128 * 128;
  const int tiles_n = Npad / GB_N;
  const long long tiles_m = (M + GB_M - 1) / GB_M;
  const long long blocks = tiles_m * tiles_n;
  PVSG_REQUIRE(blocks < (1LL << 31), "%s: too many blocks", nm);
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* wp = static_cast<const __bf16*>(w_packed);
  const char* sel = getenv("PVSG_GEMM_K32");                    // =0: the 32x32x16 / K = 16 kernel for every shape (A/B tests)
  const bool k32 = K % 32 == 0 && !(sel && sel[0] == '0');
  const char* dsel = getenv("PVSG_F16X2_DMA");                  // =0: the register-staged form (A/B tests)
  const bool dma = !(dsel && dsel[0] == '0');
  if (f16) {
    // PVSG_F16X2_TILE=256: the 256 x 256-tile kernel (half the L2 -> CU traffic).  Opt-in: in a loop of one layer it is 6-10 %
    // faster (1.28 vs 1.42 ms on the encoder's first FFN layer, both at the 1400 W socket limit), inside the step it ties
    // (23.4 vs 23.2 ms over the 48 launches), so the default stays on the 128 x 128 kernels.
    const char* tsel = getenv("PVSG_F16X2_TILE");
    const bool big = tsel && atoi(tsel) == 256;
    // 128-row x 256-column tiles, two workgroups per CU (the LayerNorm-fused kernel's pipeline): default for wide layers
    // (N >= 512, N % 256 == 0: the encoder's first FFN layer 1.39 -> 1.24 ms at 32 x 720p, scripts/lab/gemm_tile_ab.py; ragged N: the
    // wave skips the 16-column blocks beyond N -- the 544-wide projection 0.78 -> 0.76 ms; narrower N stays on 128 x 128).  PVSG_F16X2_TILE=w256 forces it, =128 switches it off.
    // PVSG_W256_RAGGED=1: ragged N >= 512 (the 544-wide projection) on these tiles too, waves skipping the 16-column blocks beyond
    // N: 0.78 -> 0.76 ms in a loop, +0.3 ms inside the step (profiles/r05_w256_ragged.txt) -- opt-in
    const char* rag = getenv("PVSG_W256_RAGGED");
    const bool wide = tsel ? tsel[0] == 'w' : (N >= 512 && (N % 256 == 0 || (rag && rag[0] == '1')));
    if (wide && N % 4 == 0 && (long long)128 * N * 4 < (1LL << 31)) {
      static std::atomic<unsigned long long> dw_r{0}, dw_n{0};
      const int tnw = (N + 255) / 256;
      const dim3 gw((unsigned)(((M + 127) / 128) * tnw)), b256(256);
      hipError_t e = relu ? ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_ln128_kernel<false, true>), LN128_LDS_BYTES, dw_r)
                          : ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_ln128_kernel<false, false>), LN128_LDS_BYTES, dw_n);
      if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: dynamic LDS: %s", nm, hipGetErrorString(e));
      const float* nulf = nullptr;
      if (relu)
        hipLaunchKernelGGL((gemm_f16x2_ln128_kernel<false, true>), gw, b256, LN128_LDS_BYTES, st, a, wp, bias, out, (int)M, K, overflow,
                           nulf, nulf, nulf, 0.f, N, Npad, tnw);
      else
        hipLaunchKernelGGL((gemm_f16x2_ln128_kernel<false, false>), gw, b256, LN128_LDS_BYTES, st, a, wp, bias, out, (int)M, K, overflow,
                           nulf, nulf, nulf, 0.f, N, Npad, tnw);
      PVSG_LAUNCH_CHECK(nm);
      return PVSG_OK;
    }
    if (big && (long long)256 * N * 4 < (1LL << 31) && (long long)256 * K * 4 < (1LL << 31)) {
      static std::atomic<unsigned long long> done_r{0}, done_n{0};
      const int tn256 = (N + 255) / 256;
      const dim3 g256((unsigned)(((M + 255) / 256) * tn256)), b512(512);
      hipError_t e = relu ? ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_t256_kernel<true>), T256_LDS_BYTES, done_r)
                          : ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_t256_kernel<false>), T256_LDS_BYTES, done_n);
      if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: dynamic LDS: %s", nm, hipGetErrorString(e));
      if (relu)
        hipLaunchKernelGGL((gemm_f16x2_t256_kernel<true>), g256, b512, T256_LDS_BYTES, st, a, wp, bias, out, (int)M, N, K, Npad, tn256, overflow);
      else
        hipLaunchKernelGGL((gemm_f16x2_t256_kernel<false>), g256, b512, T256_LDS_BYTES, st, a, wp, bias, out, (int)M, N, K, Npad, tn256, overflow);
      PVSG_LAUNCH_CHECK(nm);
      return PVSG_OK;
    }
  }
  if (f16 && dma && relu)
    hipLaunchKernelGGL((gemm_f16x2_dma_kernel<true>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, overflow);
  else if (f16 && dma)
    hipLaunchKernelGGL((gemm_f16x2_dma_kernel<false>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, overflow);
  else if (f16 && relu)
    hipLaunchKernelGGL((gemm_bf16x3_k32_kernel<true, true>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, overflow);
  else if (f16)
    hipLaunchKernelGGL((gemm_bf16x3_k32_kernel<false, true>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, overflow);
  else if (k32 && relu)
    hipLaunchKernelGGL((gemm_bf16x3_k32_kernel<true>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, nullptr);
  else if (k32)
    hipLaunchKernelGGL((gemm_bf16x3_k32_kernel<false>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n, nullptr);
  else if (relu)
    hipLaunchKernelGGL((gemm_bf16x3_kernel<true>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n);
  else
    hipLaunchKernelGGL((gemm_bf16x3_kernel<false>), grid, block, 0, st, a, wp, bias, out, (int)M, N, K, Npad, tiles_n);
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

extern "C" int pvsg_gemm_bf16x3(const float* a, const void* w_packed, const float* bias, float* out, long long M, int N, int K,
                                int relu, void* stream) {
  return gemm_split_run(a, w_packed, bias, out, M, N, K, relu, false, nullptr, stream);
}

extern "C" int pvsg_gemm_f16x2(const float* a, const void* w_packed, const float* bias, float* out, long long M, int N, int K,
                               int relu, uint32_t* overflow, void* stream) {
  return gemm_split_run(a, w_packed, bias, out, M, N, K, relu, true, overflow, stream);
}

// out = LayerNorm(residual + a w^T + bias) * gamma + beta for N == 256 (one 256-column tile = whole rows per workgroup): the
// [3P] mmcv encoder layer's output_proj / second FFN layer + identity + norm in one launch (see gemm_f16x2_t256_kernel<.., LN>)
extern "C" int pvsg_gemm_f16x2_add_layernorm(const float* a, const void* w_packed, const float* bias, const float* residual,
                                             const float* gamma, const float* beta, float eps, float* out, long long M, int N,
                                             int K, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(a && w_packed && residual && gamma && beta && out, "gemm_f16x2_add_layernorm: null pointer argument");
  PVSG_REQUIRE(M > 0 && K > 0, "gemm_f16x2_add_layernorm: bad shape");
  if (N != 256 || K % 32 || M >= (1LL << 31) || (long long)256 * K * 4 >= (1LL << 31))
    return set_err(PVSG_ERR_UNSUPPORTED, "gemm_f16x2_add_layernorm: built for N == 256, K %% 32 == 0 (got M=%lld N=%d K=%d)", M, N, K);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(bias) |
                  reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                  reinterpret_cast<uintptr_t>(out)) & 15u), "gemm_f16x2_add_layernorm: pointers must be 16-byte aligned");
  // PVSG_LN_TILE=256: the round-4 kernel (256-row tiles, one workgroup per CU); default: 128-row tiles, two workgroups per CU
  const char* tsel = getenv("PVSG_LN_TILE");
  if (!(tsel && tsel[0] == '2')) {
    static std::atomic<unsigned long long> done128{0};
    const hipError_t e1 = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_ln128_kernel<true, false>), LN128_LDS_BYTES, done128);
    if (e1 != hipSuccess) return set_err(PVSG_ERR_HIP, "gemm_f16x2_add_layernorm: dynamic LDS: %s", hipGetErrorString(e1));
    hipLaunchKernelGGL((gemm_f16x2_ln128_kernel<true, false>), dim3((unsigned)((M + 127) / 128)), dim3(256), LN128_LDS_BYTES,
                       static_cast<hipStream_t>(stream), a, static_cast<const __bf16*>(w_packed), bias, out, (int)M, K, overflow,
                       residual, gamma, beta, eps);
    PVSG_LAUNCH_CHECK("gemm_f16x2_add_layernorm");
    return PVSG_OK;
  }
  static std::atomic<unsigned long long> done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_t256_kernel<false, true>), T256_LDS_BYTES, done);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "gemm_f16x2_add_layernorm: dynamic LDS: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((gemm_f16x2_t256_kernel<false, true>), dim3((unsigned)((M + 255) / 256)), dim3(512), T256_LDS_BYTES,
                     static_cast<hipStream_t>(stream), a, static_cast<const __bf16*>(w_packed), bias, out, (int)M, 256, K, 256, 1,
                     overflow, residual, gamma, beta, eps);
  PVSG_LAUNCH_CHECK("gemm_f16x2_add_layernorm");
  return PVSG_OK;
}

// Key and value projections of one decoder level in ONE launch from the encoder's token tensor (see gemm_f16x2_ln128_kernel,
// KV form).  Replaces, per decoder layer, `k = (memory + level_embed + pos) Wk^T + bk`, `v = (memory + level_embed) Wv^T + bv`
// ([3P] nn.MultiheadAttention in_proj on the key / value inputs models/mask2former/mask2former_head.py:421-436,457-468 builds).
//   tokens (frames, S, 256): level rows start .. start + hw of every frame;  w_packed = pvsg_gemm_f16x2_pack([Wk ; Wv]) (512 x 256)
//   tab_cell (hw, 256) = (pe_yx + level_embed) Wk^T + bk;  tab_frame (zrows, 256) = pe_z Wk^T (frame f uses row f % zrows; zrows = 1
//   and zeros for the image head);  bias_v (256) = level_embed Wv^T + bv;  k_out / v_out (frames * hw, 256)
extern "C" int pvsg_decoder_kv_project_f16x2(const float* tokens, int frames, int S, int start, int hw, const void* w_packed,
                                             const float* tab_cell, const float* tab_frame, int zrows, const float* bias_v,
                                             float* k_out, float* v_out, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(tokens && w_packed && tab_cell && tab_frame && bias_v && k_out && v_out, "decoder_kv_project_f16x2: null pointer argument");
  PVSG_REQUIRE(frames > 0 && S > 0 && hw > 0 && start >= 0 && start + hw <= S && zrows > 0, "decoder_kv_project_f16x2: bad shape");
  const long long M = (long long)frames * hw;
  if ((long long)frames * S * 1024 >= 0xffffffe0LL || M >= (1LL << 31))
    return set_err(PVSG_ERR_UNSUPPORTED, "decoder_kv_project_f16x2: the token tensor must stay below 4 GB (frames=%d S=%d)", frames, S);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(tokens) | reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(tab_cell) |
                  reinterpret_cast<uintptr_t>(tab_frame) | reinterpret_cast<uintptr_t>(bias_v) | reinterpret_cast<uintptr_t>(k_out) |
                  reinterpret_cast<uintptr_t>(v_out)) & 15u), "decoder_kv_project_f16x2: pointers must be 16-byte aligned");
  static std::atomic<unsigned long long> done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(gemm_f16x2_ln128_kernel<false, false, true>), LN128_LDS_BYTES, done);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "decoder_kv_project_f16x2: dynamic LDS: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((gemm_f16x2_ln128_kernel<false, false, true>), dim3((unsigned)(((M + 127) / 128) * 2)), dim3(256), LN128_LDS_BYTES,
                     static_cast<hipStream_t>(stream), tokens, static_cast<const __bf16*>(w_packed), bias_v, k_out, (int)M, 256, overflow,
                     v_out, tab_cell, tab_frame, 0.f, 512, 512, 2, S, start, hw, zrows);
  PVSG_LAUNCH_CHECK("decoder_kv_project_f16x2");
  return PVSG_OK;
}

static int conv1x1_split_run(const float* x, const void* w_packed, const float* scale, const float* shift,
                             const float* residual, const float* in_scale, const float* in_shift, float* y, int B, int Cin,
                             int Cout, int H, int W, int stride, int relu, bool f16, uint32_t* overflow, void* stream,
                             double* gn_part = nullptr) {
  using namespace pvsg;
  const char* nm = f16 ? "conv1x1_f16x2" : "conv1x1_bf16x3";
  PVSG_REQUIRE(x && w_packed && y, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", nm);
  PVSG_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "%s: in_scale and in_shift go together", nm);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int Cpad = (Cout + 127) / 128 * 128;
  // the epilogue's 32-bit store offsets: pixel offset + padded channel row * plane must stay below 2^31 (out-of-range pixels
  // carry 0x80000000 and rely on the bounds check)
  if (Cin % (f16 ? 32 : GB_K) || Cout % 4 || (long long)Cin * H * W >= (1LL << 29) || (long long)Cpad * Ho * Wo >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for Cin %% %d == 0, Cout %% 4 == 0, C*H*W < 2^29 (got Cin=%d Cout=%d H=%d W=%d)",
                   nm, f16 ? 32 : GB_K, Cin, Cout, H, W);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u),
               "%s: w_packed, scale and shift must be 16-byte aligned", nm);
  const int tiles_c = Cpad / GB_M, tiles_p = (Ho * Wo + GB_N - 1) / GB_N;
  const long long blocks = (long long)B * tiles_c * tiles_p;
  PVSG_REQUIRE(blocks < (1LL << 31), "%s: too many blocks", nm);
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* wp = static_cast<const __bf16*>(w_packed);
  unsigned* const noflags = nullptr;
  // PVSG_GEMM_K32=0: the 32x32x16 / K = 16 kernel for every shape, =1: the K = 32 kernel wherever Cin allows (A/B tests);
  // default: K = 32 except on the small stride-1 maps (23 x 40 at 720p: layer4 and its input convolution measured 3-5 %
  // slower there, profiles/r03_conv1x1_bf16x3_bench.jsonl).  The f16 form exists on the K = 32 kernel only.
  const char* sel = getenv("PVSG_GEMM_K32");
  const bool k32 = f16 || (Cin % 32 == 0 && !(sel && sel[0] == '0') && ((sel && sel[0] == '1') || stride == 2 || Ho * Wo >= 2048));
  const bool tm64 = k32 && Cout <= 64;
  if (gn_part && (!f16 || tm64 || Cout % 8))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: GroupNorm statistics come with the f16x2 form, Cout > 64, groups of 8 channels", nm);
#define PVSG_C1_K32(R, S, NORM, TMV, F)                                                                                  \
  hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<R, S, NORM, false, TMV, 1, F>), grid, block, 0, st, x, wp, scale, shift,  \
                     residual, in_scale, in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, \
                     noflags, overflow, 0, 0LL, gn_part)
#define PVSG_C1_LAUNCH(R, S)                                                                                            \
  do {                                                                                                                  \
    if (f16) { if (tm64) PVSG_C1_K32(R, S, false, 64, true); else PVSG_C1_K32(R, S, false, 128, true); }                 \
    else if (tm64) PVSG_C1_K32(R, S, false, 64, false);                                                                 \
    else if (k32) PVSG_C1_K32(R, S, false, 128, false);                                                                 \
    else                                                                                                                \
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<R, S, false>), grid, block, 0, st, x, wp, scale, shift, residual,      \
                         in_scale, in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p);       \
  } while (0)
  if (in_scale) {          // normalised input: the pixel decoder's mask-feature convolution (no ReLU / identity behind it)
    if (relu || residual)
      return set_err(PVSG_ERR_UNSUPPORTED, "%s: in_scale / in_shift come without relu / residual", nm);
    if (f16) PVSG_C1_K32(false, false, true, 128, true);
    else if (k32) PVSG_C1_K32(false, false, true, 128, false);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<false, false, true>), grid, block, 0, st, x, wp, scale, shift, residual, in_scale,
                         in_shift, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p);
  } else if (relu) {
    if (residual) PVSG_C1_LAUNCH(true, true); else PVSG_C1_LAUNCH(true, false);
  } else {
    if (residual) PVSG_C1_LAUNCH(false, true); else PVSG_C1_LAUNCH(false, false);
  }
#undef PVSG_C1_LAUNCH
#undef PVSG_C1_K32
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

extern "C" int pvsg_conv1x1_bf16x3(const float* x, const void* w_packed, const float* scale, const float* shift,
                                   const float* residual, const float* in_scale, const float* in_shift, float* y, int B,
                                   int Cin, int Cout, int H, int W, int stride, int relu, void* stream) {
  return conv1x1_split_run(x, w_packed, scale, shift, residual, in_scale, in_shift, y, B, Cin, Cout, H, W, stride, relu, false,
                           nullptr, stream);
}

extern "C" int pvsg_conv1x1_f16x2(const float* x, const void* w_packed, const float* scale, const float* shift,
                                  const float* residual, const float* in_scale, const float* in_shift, float* y, int B,
                                  int Cin, int Cout, int H, int W, int stride, int relu, uint32_t* overflow, void* stream) {
  return conv1x1_split_run(x, w_packed, scale, shift, residual, in_scale, in_shift, y, B, Cin, Cout, H, W, stride, relu, true,
                           overflow, stream);
}

// pvsg_conv1x1_f16x2 that also leaves the GroupNorm statistics of its OUTPUT behind ([3P] mmcv ConvModule(norm_cfg=GN): conv -> GN;
// groups of 8 channels): gn_partials receives B * (Cout / 8) * pvsg_conv1x1_stats_chunks(H, W, stride) pairs of doubles (sum, sum of
// squares), to be turned into per-(image, channel) scale / shift by pvsg_group_norm_finish.
extern "C" int pvsg_conv1x1_stats_chunks(int H, int W, int stride) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  return 2 * ((Ho * Wo + pvsg::GB_N - 1) / pvsg::GB_N);
}
extern "C" int pvsg_conv1x1_f16x2_stats(const float* x, const void* w_packed, const float* scale, const float* shift,
                                        const float* residual, float* y, double* gn_partials, int B, int Cin, int Cout, int H, int W,
                                        int stride, int relu, uint32_t* overflow, void* stream) {
  PVSG_REQUIRE(gn_partials, "conv1x1_f16x2_stats: null pointer argument");
  return conv1x1_split_run(x, w_packed, scale, shift, residual, nullptr, nullptr, y, B, Cin, Cout, H, W, stride, relu, true, overflow,
                           stream, gn_partials);
}

// [3P] mmdet ResNet Bottleneck.conv2 (3x3, pad 1, stride 1 or 2) + frozen BN + ReLU on the split kernel: implicit GEMM over the
// nine taps (K = 9 * Cin).  w_packed = the pack of the (Cout, 9 * Cin) matrix ordered [block of 32 input channels][tap][32] (see ops.conv3x3_bf16x3_pack;
// channel-minor).  Direct-form arithmetic (18 Cin Cout flop per output pixel): it wins where the f32 kernels are weakest -- the
// stride-2 layers (pvsg_conv3x3s2_affine) and, against Winograd (pvsg_conv3x3_winograd), the 64- and 512-channel layers.
static int conv3x3_split_run(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B,
                             int Cin, int Cout, int H, int W, int stride, int relu, bool f16, uint32_t* overflow, void* stream,
                             double* gn_part = nullptr) {
  using namespace pvsg;
  const char* nm = f16 ? "conv3x3_f16x2" : "conv3x3_bf16x3";
  PVSG_REQUIRE(x && w_packed && y, "%s: null pointer argument", nm);
  PVSG_REQUIRE((scale == nullptr) == (shift == nullptr), "%s: scale and shift go together (both NULL = no affine)", nm);
  PVSG_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "%s: bad shape", nm);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int Cpad = (Cout + 127) / 128 * 128;
  if (Cin % 32 || Cout % 4 || (long long)Cin * H * W >= (1LL << 29) || (long long)Cpad * Ho * Wo >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for Cin %% 32 == 0, Cout %% 4 == 0, C*H*W < 2^29 (got Cin=%d Cout=%d H=%d W=%d)",
                   nm, Cin, Cout, H, W);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u),
               "%s: w_packed, scale and shift must be 16-byte aligned", nm);
  const int tiles_c = Cpad / GB_M, tiles_p = (Ho * Wo + GB_N - 1) / GB_N;
  const long long blocks = (long long)B * tiles_c * tiles_p;
  PVSG_REQUIRE(blocks < (1LL << 31), "%s: too many blocks", nm);
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const __bf16* wp = static_cast<const __bf16*>(w_packed);
  const float* nul = nullptr;
  unsigned* const noflags = nullptr;
  {
    const char* hsel = getenv("PVSG_CONV3X3_HALO");               // =0: the tap-by-tap form for stride 1 too (A/B tests)
    if (f16 && stride == 1 && !(hsel && hsel[0] == '0')) {
      const bool wide = Cout > 64;                                // <= 64 output channels: 64 x (16 x 16 pixels) tiles
      const int tc = wide ? tiles_c : (Cout + 63) / 64;
      const int tiles_x = (W + 15) / 16, tiles_y = wide ? (H + 7) / 8 : (H + 15) / 16;
      const long long hb = (long long)B * tc * tiles_x * tiles_y;
      PVSG_REQUIRE(hb < (1LL << 31), "%s: too many blocks", nm);
#define PVSG_HALO_LAUNCH(R, C)                                                                                                     \
  do {                                                                                                                             \
    static std::atomic<unsigned long long> done{0};                                                                               \
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_f16x2_halo_kernel<R, C>), halo_lds_bytes<C>(), done); \
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: dynamic LDS: %s", nm, hipGetErrorString(e));                           \
    hipLaunchKernelGGL((conv3x3_f16x2_halo_kernel<R, C>), dim3((unsigned)hb), block, halo_lds_bytes<C>(), st, x, wp, scale, shift, y, \
                       Cin, Cout, Cpad, H, W, tc, tiles_x, tiles_y, overflow, gn_part);                                            \
  } while (0)
      if (gn_part && (!wide || Cout % 8))
        return set_err(PVSG_ERR_UNSUPPORTED, "%s: the GroupNorm-statistics epilogue needs Cout > 64 in groups of 8 channels", nm);
      if (wide) { if (relu) PVSG_HALO_LAUNCH(true, 128); else PVSG_HALO_LAUNCH(false, 128); }
      else { if (relu) PVSG_HALO_LAUNCH(true, 64); else PVSG_HALO_LAUNCH(false, 64); }
#undef PVSG_HALO_LAUNCH
      PVSG_LAUNCH_CHECK(nm);
      return PVSG_OK;
    }
  }
  if (gn_part) return set_err(PVSG_ERR_UNSUPPORTED, "%s: the GroupNorm-statistics epilogue is built into the stride-1 f16x2 form", nm);
#define PVSG_C3_LAUNCH(R, TMV, F)                                                                                               \
  hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<R, false, false, false, TMV, 9, F>), grid, block, 0, st, x, wp, scale, shift, nul, \
                     nul, nul, y, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, noflags, overflow)
#define PVSG_C3_PICK(F)                                                                       \
  do {                                                                                        \
    if (Cout <= 64) { if (relu) PVSG_C3_LAUNCH(true, 64, F); else PVSG_C3_LAUNCH(false, 64, F); } \
    else { if (relu) PVSG_C3_LAUNCH(true, 128, F); else PVSG_C3_LAUNCH(false, 128, F); }        \
  } while (0)
  if (f16) PVSG_C3_PICK(true); else PVSG_C3_PICK(false);
#undef PVSG_C3_PICK
#undef PVSG_C3_LAUNCH
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

// (Cout, Cin, 3, 3) -> the (Cout, 9 Cin) matrix the 3x3 kernels multiply by, K order [block of 32 input channels][tap ky, kx][32
// channels] (column ((ci / 32) * 9 + ky * 3 + kx) * 32 + ci % 32): what pvsg_gemm_f16x2_pack / pvsg_gemm_bf16x3_pack must be given.
// C callers use this instead of hand-rolling the order (it changed once: tap-major before round 4).
namespace pvsg { namespace {
__global__ void conv3x3_weight_matrix_kernel(const float* __restrict__ w, float* __restrict__ m, int Cin, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int K = 9 * Cin;
  const long long co = i / K;
  const int k = (int)(i - co * K);
  const int blk = k / 288, r = k - blk * 288, tap = r >> 5, c = blk * 32 + (r & 31);
  m[i] = w[(co * Cin + c) * 9 + tap];
}
} }
extern "C" int pvsg_conv3x3_weight_matrix(const float* weight, float* matrix, int Cout, int Cin, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && matrix, "conv3x3_weight_matrix: null pointer argument");
  PVSG_REQUIRE(Cout > 0 && Cin > 0, "conv3x3_weight_matrix: bad shape");
  if (Cin % 32) return set_err(PVSG_ERR_UNSUPPORTED, "conv3x3_weight_matrix: built for Cin %% 32 == 0 (got %d)", Cin);
  const long long total = (long long)Cout * 9 * Cin;
  hipLaunchKernelGGL(conv3x3_weight_matrix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), weight, matrix, Cin, total);
  PVSG_LAUNCH_CHECK("conv3x3_weight_matrix");
  return PVSG_OK;
}

extern "C" int pvsg_conv3x3_bf16x3(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B,
                                   int Cin, int Cout, int H, int W, int stride, int relu, void* stream) {
  return conv3x3_split_run(x, w_packed, scale, shift, y, B, Cin, Cout, H, W, stride, relu, false, nullptr, stream);
}

extern "C" int pvsg_conv3x3_f16x2(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B,
                                  int Cin, int Cout, int H, int W, int stride, int relu, uint32_t* overflow, void* stream) {
  return conv3x3_split_run(x, w_packed, scale, shift, y, B, Cin, Cout, H, W, stride, relu, true, overflow, stream);
}

// [3P] mmdet ResNet Bottleneck (64 planes, stride 1: ResNet-50 layer1): conv3 -> bn3 -> + identity -> ReLU of one block and conv1 ->
// bn1 -> ReLU of the NEXT block in one pass over the pixels (bottleneck_tail64_kernel).  mid (B, 64, H, W) = the block's conv2
// output; identity, y (B, 256, H, W); w3_packed = pvsg_gemm_f16x2_pack of conv3's (256, 64) matrix; w1n_packed = the pack of
// pvsg_bottleneck_next_weight_matrix(next conv1's (64, 256) matrix), scale1n / shift1n its BN, mid_next (B, 64, H, W) -- or all four
// NULL: conv3 + identity + ReLU only (the last block of the stage).  identity == NULL = the HEAD of the stage's first block from one
// read of its input x (passed as `mid`): y = downsample(x) * scale3 + shift3 (no ReLU: Bottleneck.downsample = conv + BN) and mid_next
// = relu(conv1(x) * scale1n + shift1n), w1n_packed = the PLAIN pvsg_gemm_f16x2_pack of conv1's (64, 64) matrix.
// Built for exactly these channel counts and even H * W.
extern "C" int pvsg_bottleneck_next_weight_matrix(const float* weight, float* matrix, int Cn, int K, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && matrix, "bottleneck_next_weight_matrix: null pointer argument");
  PVSG_REQUIRE(Cn > 0 && K > 0, "bottleneck_next_weight_matrix: bad shape");
  if (K % 32) return set_err(PVSG_ERR_UNSUPPORTED, "bottleneck_next_weight_matrix: built for K %% 32 == 0 (got %d)", K);
  const long long total = (long long)Cn * K;
  hipLaunchKernelGGL(bottleneck_next_matrix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     weight, matrix, K, total);
  PVSG_LAUNCH_CHECK("bottleneck_next_weight_matrix");
  return PVSG_OK;
}
extern "C" int pvsg_bottleneck_tail_f16x2(const float* mid, const void* w3_packed, const float* scale3, const float* shift3,
                                          const float* identity, float* y, float* y_stride2, const void* w1n_packed,
                                          const float* scale1n, const float* shift1n, float* mid_next, int B, int Cmid, int Cout,
                                          int Cnext, int H, int W, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  const char* nm = "bottleneck_tail_f16x2";
  PVSG_REQUIRE(mid && w3_packed && scale3 && shift3 && y, "%s: null pointer argument", nm);
  const bool next = w1n_packed != nullptr, head = identity == nullptr;
  PVSG_REQUIRE(!head || next, "%s: identity == NULL (the stage's first block: downsample + conv1 from one read) needs the conv1 arguments", nm);
  PVSG_REQUIRE(next == (scale1n != nullptr) && next == (shift1n != nullptr) && next == (mid_next != nullptr),
               "%s: w1n_packed, scale1n, shift1n and mid_next go together", nm);
  PVSG_REQUIRE(B > 0 && H > 0 && W > 0, "%s: bad shape", nm);
  const long long HW = (long long)H * W;
  const bool wide = next && Cnext == 128;
  if (Cmid != 64 || Cout != 256 || (next && Cnext != 64 && Cnext != 128) || (head && wide) || (HW & 1) || 256 * HW * 4 >= (1LL << 32) ||
      (y_stride2 && (W & 1)))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for 64 -> 256 (-> 64 | 128) channels, even H*W (even W with y_stride2), 256*H*W*4 < 2^32 "
                   "(got %d -> %d -> %d, %d x %d)", nm, Cmid, Cout, Cnext, H, W);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(w3_packed) | reinterpret_cast<uintptr_t>(w1n_packed) | reinterpret_cast<uintptr_t>(scale3) |
                  reinterpret_cast<uintptr_t>(shift3) | reinterpret_cast<uintptr_t>(scale1n) | reinterpret_cast<uintptr_t>(shift1n)) & 15u),
               "%s: packed weights, scale and shift must be 16-byte aligned", nm);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(mid) | reinterpret_cast<uintptr_t>(identity) | reinterpret_cast<uintptr_t>(y) |
                  reinterpret_cast<uintptr_t>(mid_next)) & 7u), "%s: tensors must be 8-byte aligned", nm);
  const int tiles_per_img = (int)((HW + 31) / 32);
  const long long total = (long long)B * tiles_per_img;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long long want = (total + 7) / 8;
  const unsigned grid = (unsigned)(want < cus ? want : cus);
  hipStream_t st = static_cast<hipStream_t>(stream);
  static std::atomic<unsigned long long> done0{0}, done1{0}, done2{0}, done3{0};
  const hipError_t e = head ? ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_tail64_kernel<2>), BT_LDS_BYTES, done2)
                     : wide ? ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_tail64_kernel<3>), BT_LDS_BYTES, done3)
                     : next ? ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_tail64_kernel<1>), BT_LDS_BYTES, done1)
                            : ensure_dynamic_lds(reinterpret_cast<const void*>(bottleneck_tail64_kernel<0>), BT_LDS_BYTES, done0);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: dynamic LDS: %s", nm, hipGetErrorString(e));
  const __bf16* w3 = static_cast<const __bf16*>(w3_packed);
  const __bf16* w1 = static_cast<const __bf16*>(w1n_packed);
#define PVSG_BT_LAUNCH(M)                                                                                                        \
  hipLaunchKernelGGL((bottleneck_tail64_kernel<M>), dim3(grid), dim3(512), BT_LDS_BYTES, st, mid, w3, scale3, shift3, identity, y, w1, \
                     scale1n, shift1n, mid_next, (int)HW, tiles_per_img, total, overflow, y_stride2, W)
  if (head) PVSG_BT_LAUNCH(2); else if (wide) PVSG_BT_LAUNCH(3); else if (next) PVSG_BT_LAUNCH(1); else PVSG_BT_LAUNCH(0);
#undef PVSG_BT_LAUNCH
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

// pvsg_conv3x3_f16x2 (stride 1) that also leaves the GroupNorm statistics of its OUTPUT behind ([3P] mmcv ConvModule(3x3 conv ->
// GN -> ReLU), the FPN output convolution of MSDeformAttnPixelDecoder): gn_partials receives B * (Cout / 8) *
// pvsg_conv3x3_stats_chunks(H, W) pairs of doubles (sum, sum of squares) for pvsg_group_norm_finish.
extern "C" int pvsg_conv3x3_stats_chunks(int H, int W) { return 2 * ((W + 15) / 16) * ((H + 7) / 8); }
extern "C" int pvsg_conv3x3_f16x2_stats(const float* x, const void* w_packed, const float* scale, const float* shift, float* y,
                                        double* gn_partials, int B, int Cin, int Cout, int H, int W, int relu, uint32_t* overflow,
                                        void* stream) {
  PVSG_REQUIRE(gn_partials, "conv3x3_f16x2_stats: null pointer argument");
  return conv3x3_split_run(x, w_packed, scale, shift, y, B, Cin, Cout, H, W, 1, relu, true, overflow, stream, gn_partials);
}

// einsum('bqc,b[t]chw->b[t]qhw') (mask2former_head.py:382, mask2former_video_head.py:344) on the split kernels: per batch
// element a 1x1 "convolution" of the (T, C, N) mask features with the Q mask embeddings as the weight (packed on the fly:
// Q x C is 100 x 256), output (T, Q, N).  Same f32-class arithmetic as above; the f32-MFMA form stays as
// pvsg_mask_logits_forward (csrc/mask_gemm.hip).  w_scratch: B * pvsg_gemm_{bf16x3,f16x2}_packed_elems(Q, C) 16-bit elements.
static int mask_logits_split_run(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B, int T,
                                 int Q, int C, long long N, bool f16, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  const char* nm = f16 ? "mask_logits_f16x2" : "mask_logits_bf16x3";
  PVSG_REQUIRE(mask_embed && mask_feature && w_scratch && out, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  const long long Qpad = (Q + 127) / 128 * 128;
  if (C % (f16 ? 32 : GB_K) || Q % 4 || N >= (1LL << 31) || (long long)C * N >= (1LL << 29) || Qpad * N >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% %d == 0, Q %% 4 == 0, C*N and pad128(Q)*N < 2^29 (got Q=%d C=%d N=%lld)",
                   nm, f16 ? 32 : GB_K, Q, C, N);
  const long long welems = f16 ? pvsg_gemm_f16x2_packed_elems(Q, C) : pvsg_gemm_bf16x3_packed_elems(Q, C);
  for (int b = 0; b < B; ++b) {
    __bf16* wp = static_cast<__bf16*>(w_scratch) + (size_t)b * welems;
    int rc = f16 ? pvsg_gemm_f16x2_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream)
                 : pvsg_gemm_bf16x3_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream);
    if (rc != PVSG_OK) return rc;
    rc = conv1x1_split_run(mask_feature + (size_t)b * T * C * N, wp, nullptr, nullptr, nullptr, nullptr, nullptr,
                           out + (size_t)b * T * Q * N, T, C, Q, 1, (int)N, 1, 0, f16, overflow, stream);
    if (rc != PVSG_OK) return rc;
  }
  return PVSG_OK;
}

extern "C" int pvsg_mask_logits_bf16x3(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B,
                                       int T, int Q, int C, long long N, void* stream) {
  return mask_logits_split_run(mask_embed, mask_feature, w_scratch, out, B, T, Q, C, N, false, nullptr, stream);
}

extern "C" int pvsg_mask_logits_f16x2(const float* mask_embed, const float* mask_feature, void* w_scratch, float* out, int B,
                                      int T, int Q, int C, long long N, uint32_t* overflow, void* stream) {
  return mask_logits_split_run(mask_embed, mask_feature, w_scratch, out, B, T, Q, C, N, true, overflow, stream);
}

// Attention-mask bits of a decoder level straight from the down-sampled mask features (mask2former_head.py:383-393,
// video_head.py:346-357; the all-masked-row test of mask2former_head.py:453-454 becomes the flag words) on the split
// kernels: same record format as pvsg_attn_mask_bits_forward (csrc/mask_gemm.hip), which stays as the f32-MFMA form.
static int attn_mask_bits_split_run(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                    uint32_t* flags, int B, int T, int Q, int C, long long N, bool f16, uint32_t* overflow,
                                    void* stream) {
  using namespace pvsg;
  const char* nm = f16 ? "attn_mask_bits_f16x2" : "attn_mask_bits_bf16x3";
  PVSG_REQUIRE(mask_embed && feature_lowres && w_scratch && bits && flags, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  if (C % (f16 ? 32 : GB_K) || Q > GB_M || (long long)C * N >= (1LL << 29) || (reinterpret_cast<uintptr_t>(bits) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% %d == 0, Q <= 128, C*N < 2^29, 16B-aligned bits (got Q=%d C=%d N=%lld)",
                   nm, f16 ? 32 : GB_K, Q, C, N);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = zero_words_async(flags, (size_t)B * 4 * sizeof(uint32_t), st);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: memset: %s", nm, hipGetErrorString(e));
  const long long welems = f16 ? pvsg_gemm_f16x2_packed_elems(Q, C) : pvsg_gemm_bf16x3_packed_elems(Q, C);
  const int tiles_p = (int)((N + GB_N - 1) / GB_N);
  const float* nul = nullptr;
  for (int b = 0; b < B; ++b) {
    __bf16* wp = static_cast<__bf16*>(w_scratch) + (size_t)b * welems;
    const int rc = f16 ? pvsg_gemm_f16x2_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream)
                       : pvsg_gemm_bf16x3_pack(mask_embed + (size_t)b * Q * C, wp, Q, C, stream);
    if (rc != PVSG_OK) return rc;
    const char* sel = getenv("PVSG_GEMM_K32");
    const float* fl = feature_lowres + (size_t)b * T * C * N;
    float* rec = reinterpret_cast<float*>(bits + (size_t)b * T * N * 4);
    const dim3 grid((unsigned)(T * tiles_p)), block(256);
    if (f16)
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true, 128, 1, true>), grid, block, 0, st, fl, wp, nul, nul,
                         nul, nul, nul, rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4, overflow);
    else if (C % 32 == 0 && !(sel && sel[0] == '0'))
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true>), grid, block, 0, st, fl, wp, nul, nul, nul, nul, nul,
                         rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4, (unsigned*)nullptr);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_kernel<false, false, false, true>), grid, block, 0, st, fl, wp, nul, nul, nul, nul, nul,
                         rec, C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags + (size_t)b * 4);
    PVSG_LAUNCH_CHECK(nm);
  }
  return PVSG_OK;
}

// The bits from embeddings that are ALREADY packed (pvsg_decoder_rows_post writes them in its epilogue, one exact power-of-two
// scale per query row -- the bits are signs) into flag words that are already zero: ONE launch for the whole batch, where the
// entry above issues zero + amax + pack + GEMM per batch element.
extern "C" int pvsg_attn_mask_bits_packed_f16x2(const void* emb_packed, const float* feature_lowres, uint32_t* bits,
                                                uint32_t* flags, int B, int T, int Q, int C, long long N, uint32_t* overflow,
                                                void* stream) {
  using namespace pvsg;
  const char* nm = "attn_mask_bits_packed_f16x2";
  PVSG_REQUIRE(emb_packed && feature_lowres && bits && flags, "%s: null pointer argument", nm);
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "%s: bad shape", nm);
  if (C % 32 || Q > GB_M || (long long)C * N >= (1LL << 29) || (long long)B * T * N >= (1LL << 27) ||
      ((reinterpret_cast<uintptr_t>(bits) | reinterpret_cast<uintptr_t>(emb_packed)) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for C %% 32 == 0, Q <= 128, C*N < 2^29, B*T*N < 2^27, 16B-aligned buffers "
                   "(got B=%d T=%d Q=%d C=%d N=%lld)", nm, B, T, Q, C, N);
  const int tiles_p = (int)((N + GB_N - 1) / GB_N);
  const float* nul = nullptr;
  hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, true, 128, 1, true>), dim3((unsigned)(B * T * tiles_p)), dim3(256),
                     0, static_cast<hipStream_t>(stream), feature_lowres, static_cast<const __bf16*>(emb_packed), nul, nul, nul, nul,
                     nul, reinterpret_cast<float*>(bits), C, Q, GB_M, (int)N, (int)N, (int)N, (int)N, 1, 1, tiles_p, flags, overflow,
                     T, pvsg_gemm_f16x2_packed_elems(Q, C));
  PVSG_LAUNCH_CHECK(nm);
  return PVSG_OK;
}

extern "C" int pvsg_attn_mask_bits_bf16x3(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                          uint32_t* flags, int B, int T, int Q, int C, long long N, void* stream) {
  return attn_mask_bits_split_run(mask_embed, feature_lowres, w_scratch, bits, flags, B, T, Q, C, N, false, nullptr, stream);
}

extern "C" int pvsg_attn_mask_bits_f16x2(const float* mask_embed, const float* feature_lowres, void* w_scratch, uint32_t* bits,
                                         uint32_t* flags, int B, int T, int Q, int C, long long N, uint32_t* overflow, void* stream) {
  return attn_mask_bits_split_run(mask_embed, feature_lowres, w_scratch, bits, flags, B, T, Q, C, N, true, overflow, stream);
}
