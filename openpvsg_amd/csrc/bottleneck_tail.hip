// ResNet-50 layer1 bottleneck tails on the f16x2 split arithmetic (split_common.h): persistent workgroups, weights resident in LDS.
#include "split_common.h"

namespace pvsg {
namespace {

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
