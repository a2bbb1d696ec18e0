// 3x3 convolution of the split arithmetic: the halo kernel (stride 1: input patch staged once per 32-channel block) and the
// C-ABI entries pvsg_conv3x3_{f16x2,bf16x3}[_stats], pvsg_conv3x3_weight_matrix; stride 2 runs the TAPS = 9 form of
// split_conv1x1.h.
#include "split_conv1x1.h"

namespace pvsg {
namespace {

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
                               double* __restrict__ gn_part = nullptr, int kslices = 1, long long slice_elems = 0) {
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
  // kslices > 1 (small maps, pvsg_conv3x3_f16x2_sliced): channel blocks [cib0, cib1) only, raw sums into slice `ks` of a workspace
  const int ks = kslices > 1 ? (int)(logical % (unsigned)kslices) : 0;
  if (kslices > 1) {
    logical /= (unsigned)kslices;
    y += (size_t)ks * slice_elems;
  }
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
  const int NCBall = Cin / 32;
  const int cib0 = kslices > 1 ? (int)((long long)ks * NCBall / kslices) : 0;
  const int NCB = kslices > 1 ? (int)((long long)(ks + 1) * NCBall / kslices) : NCBall, NSTEP = NCB * 9;
  loadX(cib0);
  dmaW(cib0 * 9, 0);
  if (NBUF == 3) dmaW(cib0 * 9 + 1, 1);
  stashX();
  int step = cib0 * 9, buf = 0;                                   // buf = (step - first step) % NBUF
  constexpr int DMA_OPS = 8 * (CT / 64) / 4, LOAD_OPS = 8 * NR;   // vector-memory operations of one dmaW / loadX per wave
  for (int cib = cib0; cib < NCB; ++cib) {
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

}  // namespace
}  // namespace pvsg

// [3P] mmdet ResNet Bottleneck.conv2 (3x3, pad 1, stride 1 or 2) + frozen BN + ReLU on the split kernel: implicit GEMM over the
// nine taps (K = 9 * Cin).  w_packed = the pack of the (Cout, 9 * Cin) matrix ordered [block of 32 input channels][tap][32] (see ops.conv3x3_bf16x3_pack;
// channel-minor).  Direct-form arithmetic (18 Cin Cout flop per output pixel): it wins where the f32 kernels are weakest -- the
// stride-2 layers (pvsg_conv3x3s2_affine) and, against Winograd (pvsg_conv3x3_winograd), the 64- and 512-channel layers.
static int conv3x3_split_run(const float* x, const void* w_packed, const float* scale, const float* shift, float* y, int B,
                             int Cin, int Cout, int H, int W, int stride, int relu, bool f16, uint32_t* overflow, void* stream,
                             double* gn_part = nullptr, float* ws = nullptr, int slices = 1) {
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
  const long long slice_elems = (long long)B * Cout * Ho * Wo;
  if (slices > 1) {
    if (!f16 || !ws || gn_part || (Ho * Wo) % 4 || slices > (stride == 1 ? 1 : 9) * Cin / 32)
      return set_err(PVSG_ERR_UNSUPPORTED, "%s: K slices need the f16x2 form, a workspace, Ho*Wo %% 4 == 0, slices <= Cin / 32 "
                     "(stride 2: 9 Cin / 32)", nm);
    PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(y)) & 15u), "%s: workspace and y must be 16-byte aligned", nm);
  }
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
      if (slices > 1) {                                           // raw sums of `slices` channel-block ranges, then the fold
        PVSG_REQUIRE(hb * slices < (1LL << 31), "%s: too many blocks", nm);
#define PVSG_HALO_SLICED(C)                                                                                                        \
  do {                                                                                                                             \
    static std::atomic<unsigned long long> done{0};                                                                               \
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(conv3x3_f16x2_halo_kernel<false, C>), halo_lds_bytes<C>(), done); \
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: dynamic LDS: %s", nm, hipGetErrorString(e));                           \
    hipLaunchKernelGGL((conv3x3_f16x2_halo_kernel<false, C>), dim3((unsigned)(hb * slices)), block, halo_lds_bytes<C>(), st, x, wp, nul, \
                       nul, ws, Cin, Cout, Cpad, H, W, tc, tiles_x, tiles_y, overflow, (double*)nullptr, slices, slice_elems);    \
  } while (0)
        if (wide) PVSG_HALO_SLICED(128); else PVSG_HALO_SLICED(64);
#undef PVSG_HALO_SLICED
        PVSG_LAUNCH_CHECK(nm);
        launch_conv_slices_finish(ws, slices, slice_elems, scale, shift, nul, y, Cout, Ho * Wo, relu, st);
        PVSG_LAUNCH_CHECK(nm);
        return PVSG_OK;
      }
      if (wide) { if (relu) PVSG_HALO_LAUNCH(true, 128); else PVSG_HALO_LAUNCH(false, 128); }
      else { if (relu) PVSG_HALO_LAUNCH(true, 64); else PVSG_HALO_LAUNCH(false, 64); }
#undef PVSG_HALO_LAUNCH
      PVSG_LAUNCH_CHECK(nm);
      return PVSG_OK;
    }
  }
  if (gn_part) return set_err(PVSG_ERR_UNSUPPORTED, "%s: the GroupNorm-statistics epilogue is built into the stride-1 f16x2 form", nm);
  if (slices > 1) {                                               // tap-by-tap form (stride 2), K-steps [kt0, kt1) per slice
    if (slices > 9 * Cin / 32 || blocks * slices >= (1LL << 31))
      return set_err(PVSG_ERR_UNSUPPORTED, "%s: too many K slices", nm);
    const dim3 gs((unsigned)(blocks * slices));
    if (Cout <= 64)
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, false, 64, 9, true>), gs, block, 0, st, x, wp, nul, nul, nul, nul,
                         nul, ws, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, noflags, overflow, 0, 0LL,
                         (double*)nullptr, slices, slice_elems);
    else
      hipLaunchKernelGGL((conv1x1_bf16x3_k32_kernel<false, false, false, false, 128, 9, true>), gs, block, 0, st, x, wp, nul, nul, nul, nul,
                         nul, ws, Cin, Cout, Cpad, H * W, W, Ho * Wo, Wo, stride, tiles_c, tiles_p, noflags, overflow, 0, 0LL,
                         (double*)nullptr, slices, slice_elems);
    PVSG_LAUNCH_CHECK(nm);
    launch_conv_slices_finish(ws, slices, slice_elems, scale, shift, nul, y, Cout, Ho * Wo, relu, st);
    PVSG_LAUNCH_CHECK(nm);
    return PVSG_OK;
  }
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

// K-sliced form for small maps (see pvsg_conv_slices, csrc/conv1x1_split.hip): workspace of slices * B * Cout * Ho * Wo floats
extern "C" int pvsg_conv3x3_f16x2_sliced(const float* x, const void* w_packed, const float* scale, const float* shift, float* y,
                                         float* workspace, int slices, int B, int Cin, int Cout, int H, int W, int stride, int relu,
                                         uint32_t* overflow, void* stream) {
  PVSG_REQUIRE(slices >= 1, "conv3x3_f16x2_sliced: slices must be >= 1");
  return conv3x3_split_run(x, w_packed, scale, shift, y, B, Cin, Cout, H, W, stride, relu, true, overflow, stream, nullptr, workspace,
                           slices);
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
