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
#include "split_common.h"

namespace pvsg {
namespace {

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
