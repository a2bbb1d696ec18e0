// Per-query mask-logit projection  out[g,t,q,n] = sum_c E[g,q,c] * F[g,t,c,n]  on the gfx950
// f32-input matrix cores, plus the kernels that turn mask logits into attention-mask bits.
//
// Replaces (reference, fp32):
//   torch.einsum('bqc,bchw->bqhw')    models/mask2former/mask2former_head.py:382
//   torch.einsum('bqc,btchw->btqhw')  models/mask2former_vps/mask2former_video_head.py:344
//   F.interpolate(..., bilinear) + repeat(num_heads) + sigmoid() < 0.5
//                                     mask2former_head.py:383-393 / video_head.py:346-357
//   attn_mask[where(attn_mask.sum(-1) == K)] = False      mask2former_head.py:453-454
//
// Design (MI355X):
//  * v_mfma_f32_16x16x4_f32: exact f32 products/accumulation (the `sigmoid < 0.5` threshold on
//    these logits is a hard decision; 16-bit inputs would flip bits).  Q=100 pads to 7 row tiles of
//    16 (112) instead of 4 tiles of 32 (128): 89% instead of 78% of the 157 TF f32 matrix peak.
//  * One wave owns a 112 x 64 output tile.  B operand = F read straight from HBM as one float4 per
//    lane per k-step (lanes 0-15 = 256 contiguous bytes of one channel row; the 4 floats of a lane
//    are the B values of 4 column tiles, so the lane ends up holding 4 CONSECUTIVE pixels of each
//    of its rows -> float4 stores).  F is touched exactly once: 4*C*N bytes, the algorithmic minimum.
//  * A operand = E (100 KB) staged once per workgroup in LDS in MFMA-fragment order
//    [k-block][row tile][lane][4 k-steps], so every A fetch is a conflict-free ds_read_b128.
//  * Attention-mask mode: the epilogue thresholds in registers and emits ONE BIT per (query,key),
//    key-major (16 bytes per key, shared by all 8 heads) plus a 128-bit "query has an unmasked
//    key" flag word per batch element: the (B*8, Q, K) bool tensor of the reference (0.38 GB at
//    T=32) becomes K*16 bytes, and the all-masked-row reset becomes a flag test in the attention
//    kernel.
//  * Bilinear down-sampling by an integer factor s in {2,4,8} with align_corners=False is the mean
//    of the 2x2 centre taps; it is linear, so mask bits at a level come from a GEMM over the
//    down-sampled features (downsample kernel below, F read once for all three levels).
#include "common.h"

namespace pvsg {

constexpr int QT = 7;        // row tiles of 16 -> up to 112 queries
constexpr int TILE_N = 64;   // pixels per wave tile

enum { MODE_LOGITS = 0, MODE_BITS = 1 };

template <int MODE, bool VEC>
__global__ __launch_bounds__(512) void mask_gemm_kernel(
    const float* __restrict__ E, const float* __restrict__ F, float* __restrict__ out,
    uint32_t* __restrict__ bits, uint32_t* __restrict__ flags, int Q, int C, int N, int T,
    int tiles_per_img, int wgs_per_b) {
  extern __shared__ __attribute__((aligned(16))) float elds[];  // [C/16][QT][64][4]
  const int b = blockIdx.x / wgs_per_b;
  const int wg = blockIdx.x - b * wgs_per_b;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, g = lane >> 4;

  // ---- stage E[b] into LDS in fragment order (coalesced global reads) ----------------------
  {
    const float* Eb = E + (long long)b * Q * C;
    const int total = QT * 16 * C;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int q = i / C, c = i - q * C;
      const float v = (q < Q) ? Eb[i] : 0.f;
      const int kb = c >> 4, cc = c & 15;
      const int ln = (cc & 3) * 16 + (q & 15);
      elds[(((kb * QT + (q >> 4)) * 64 + ln) << 2) + (cc >> 2)] = v;
    }
  }
  __syncthreads();

  const int ntiles = T * tiles_per_img;
  const int nkb = C >> 4;
  uint32_t allowed_or[4] = {0u, 0u, 0u, 0u};

  for (int tile = wg * 8 + wave; tile < ntiles; tile += wgs_per_b * 8) {
    const int t = tile / tiles_per_img;
    const int n = (tile - t * tiles_per_img) * TILE_N + 4 * j16;
    // VEC: N % 4 == 0, a lane's 4 pixels are all in range or all out.  Otherwise (odd level
    // sizes) every pixel is bounds-checked and moved with scalar accesses.
    const bool valid = VEC ? (n + 3 < N) : (n < N);
    const long long img = (long long)b * T + t;
    const float* Fp = F + img * C * N + (long long)g * N + (valid ? n : 0);
    auto ldrow = [&](int crow) -> float4 {
      const float* p = Fp + (long long)crow * N;
      if constexpr (VEC) {
        return valid ? ld4(p) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) v.x = p[0];
        if (n + 1 < N) v.y = p[1];
        if (n + 2 < N) v.z = p[2];
        if (n + 3 < N) v.w = p[3];
        return v;
      }
    };

    f32x4 acc[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[qt][x] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 bv[4], bn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = ldrow(4 * s);

    for (int kb = 0; kb < nkb; ++kb) {
      if (kb + 1 < nkb) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          bn[s] = ldrow((kb + 1) * 16 + 4 * s);
      }
      float4 a[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        a[qt] = *reinterpret_cast<const float4*>(&elds[((kb * QT + qt) * 64 + lane) << 2]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b0 = bv[s].x, b1 = bv[s].y, b2 = bv[s].z, b3 = bv[s].w;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float av = (s == 0) ? a[qt].x : (s == 1) ? a[qt].y : (s == 2) ? a[qt].z : a[qt].w;
          acc[qt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[qt][0], 0, 0, 0);
          acc[qt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[qt][1], 0, 0, 0);
          acc[qt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc[qt][2], 0, 0, 0);
          acc[qt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3, acc[qt][3], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = bn[s];
    }

    if constexpr (MODE == MODE_LOGITS) {
      // lane holds rows q = qt*16 + g*4 + r, pixels n..n+3 (one per column tile)
      if (valid) {
        float* op = out + img * Q * N + n;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g * 4 + r;
            if (q < Q) {
              float* o = op + (long long)q * N;
              if constexpr (VEC) {
                st4(o, make_float4(acc[qt][0][r], acc[qt][1][r], acc[qt][2][r], acc[qt][3][r]));
              } else {
                o[0] = acc[qt][0][r];
                if (n + 1 < N) o[1] = acc[qt][1][r];
                if (n + 2 < N) o[2] = acc[qt][2][r];
                if (n + 3 < N) o[3] = acc[qt][3][r];
              }
            }
          }
      }
    } else {
      // masked <=> sigmoid(x) < 0.5 <=> x < 0.  bit q of the key's 128-bit word = masked.
      uint32_t w[4][4];  // [pixel x][word]
#pragma unroll
      for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[x][k] = 0u;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (acc[qt][x][r] < 0.f) w[x][qt >> 1] |= 1u << ((qt & 1) * 16 + g * 4 + r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          w[x][k] |= __shfl_xor(w[x][k], 16);
          w[x][k] |= __shfl_xor(w[x][k], 32);
        }
      }
      // lane group g stores pixel x = g: the wave writes 64 consecutive keys * 16 B
      uint4 mine;
      mine.x = (g == 0) ? w[0][0] : (g == 1) ? w[1][0] : (g == 2) ? w[2][0] : w[3][0];
      mine.y = (g == 0) ? w[0][1] : (g == 1) ? w[1][1] : (g == 2) ? w[2][1] : w[3][1];
      mine.z = (g == 0) ? w[0][2] : (g == 1) ? w[1][2] : (g == 2) ? w[2][2] : w[3][2];
      mine.w = (g == 0) ? w[0][3] : (g == 1) ? w[1][3] : (g == 2) ? w[2][3] : w[3][3];
      if (VEC ? valid : (n + g < N)) {
        const long long key = (long long)t * N + n + g;
        *reinterpret_cast<uint4*>(bits + ((long long)b * T * N + key) * 4) = mine;
        allowed_or[0] |= ~mine.x; allowed_or[1] |= ~mine.y;
        allowed_or[2] |= ~mine.z; allowed_or[3] |= ~mine.w;
      }
    }
  }

  if constexpr (MODE == MODE_BITS) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t v = allowed_or[k];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
      if (lane == 0 && v) atomicOr(flags + b * 4 + k, v);
    }
  }
}

// ---- general path: bits from already-resized low-resolution logits -----------------------------
// low: (B, T, Q, HW)  ->  bits (B, T*HW, 4) + flags (B, 4).  One lane per key.
__global__ __launch_bounds__(256) void attn_mask_pack_kernel(const float* __restrict__ low,
                                                            uint32_t* __restrict__ bits,
                                                            uint32_t* __restrict__ flags, int T,
                                                            int Q, int HW) {
  const int b = blockIdx.y;
  const long long K = (long long)T * HW;
  uint32_t allowed[4] = {0u, 0u, 0u, 0u};
  for (long long key = (long long)blockIdx.x * blockDim.x + threadIdx.x; key < K;
       key += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(key / HW);
    const int p = (int)(key - (long long)t * HW);
    const float* lp = low + ((long long)(b * T + t) * Q) * HW + p;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int q = 0; q < Q; ++q)
      if (lp[(long long)q * HW] < 0.f) w[q >> 5] |= 1u << (q & 31);
    *reinterpret_cast<uint4*>(bits + ((long long)b * K + key) * 4) = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) allowed[k] |= ~w[k];
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t v = allowed[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v |= __shfl_xor(v, off);
    if (lane == 0 && v) atomicOr(flags + b * 4 + k, v);
  }
}

// ---- centre-tap down-sampling of the mask features to the three decoder levels -----------------
// F (G, H, W) planes (G = B*T*C), H % 8 == 0, W % 8 == 0 ->
//   d2 (G, H/2, W/2), d4 (G, H/4, W/4), d8 (G, H/8, W/8)
// == F.interpolate(F, size, mode='bilinear', align_corners=False) for these exact factors.
// One lane owns an 8x8 block (two float4 per row): F is read once for all three outputs.
__global__ __launch_bounds__(256) void center_downsample_kernel(const float* __restrict__ F,
                                                               float* __restrict__ d2,
                                                               float* __restrict__ d4,
                                                               float* __restrict__ d8, int H, int W,
                                                               long long nblocks_total) {
  const int bw = W >> 3, bh = H >> 3;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks_total;
       i += (long long)gridDim.x * blockDim.x) {
    const int bx = (int)(i % bw);
    const int by = (int)((i / bw) % bh);
    const long long plane = i / ((long long)bw * bh);
    const float* fp = F + plane * H * W + (long long)(by * 8) * W + bx * 8;
    float r[8][8];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const float4 a = ld4(fp + (long long)y * W), c = ld4(fp + (long long)y * W + 4);
      r[y][0] = a.x; r[y][1] = a.y; r[y][2] = a.z; r[y][3] = a.w;
      r[y][4] = c.x; r[y][5] = c.y; r[y][6] = c.z; r[y][7] = c.w;
    }
    // bilinear weights are exactly 0.25 each; keep ATen's accumulation order:
    // w00*a + w01*b + w10*c + w11*d
#define TAP4(y0, x0) (0.25f * r[y0][x0] + 0.25f * r[y0][(x0) + 1] + 0.25f * r[(y0) + 1][x0] + 0.25f * r[(y0) + 1][(x0) + 1])
    float* o2 = d2 + plane * (H >> 1) * (W >> 1) + (long long)(by * 4) * (W >> 1) + bx * 4;
#pragma unroll
    for (int y = 0; y < 4; ++y)
      st4(o2 + (long long)y * (W >> 1),
          make_float4(TAP4(2 * y, 0), TAP4(2 * y, 2), TAP4(2 * y, 4), TAP4(2 * y, 6)));
    float* o4 = d4 + plane * (H >> 2) * (W >> 2) + (long long)(by * 2) * (W >> 2) + bx * 2;
    *reinterpret_cast<float2*>(o4) = make_float2(TAP4(1, 1), TAP4(1, 5));
    *reinterpret_cast<float2*>(o4 + (W >> 2)) = make_float2(TAP4(5, 1), TAP4(5, 5));
    d8[plane * (H >> 3) * (W >> 3) + (long long)by * (W >> 3) + bx] = TAP4(3, 3);
#undef TAP4
  }
}

constexpr size_t MASK_GEMM_MAX_LDS = (size_t)(320 / 16) * QT * 64 * 4 * sizeof(float);   // C <= 320 (checked by the callers)

static int launch_cfg(int B, int T, int N, int* tiles_per_img, int* wgs_per_b) {
  *tiles_per_img = (N + TILE_N - 1) / TILE_N;
  const long long ntiles = (long long)T * *tiles_per_img;
  long long per_b = (ntiles + 7) / 8;            // one workgroup pass = 8 wave tiles
  long long cap = (256 + B - 1) / B;             // ~1 resident workgroup per CU (112 KB LDS each)
  if (cap < 1) cap = 1;
  *wgs_per_b = (int)(per_b < cap ? per_b : cap);
  if (*wgs_per_b < 1) *wgs_per_b = 1;
  return 0;
}

}  // namespace pvsg

extern "C" int pvsg_mask_logits_forward(const float* mask_embed, const float* mask_feature,
                                        float* out, int B, int T, int Q, int C, int N,
                                        hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(mask_embed && mask_feature && out, "mask_logits_forward: null pointer argument");
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "mask_logits_forward: non-positive dimension");
  if (Q > QT * 16 || (C & 15) || C > 320)
    return set_err(PVSG_ERR_UNSUPPORTED,
                   "mask_logits_forward: needs Q<=112, C%%16==0, C<=320 (got Q=%d C=%d N=%d)", Q, C, N);
  const bool vec = !(N & 3) &&
      !((reinterpret_cast<uintptr_t>(mask_feature) | reinterpret_cast<uintptr_t>(out)) & 15u);
  int tpi, wpb;
  launch_cfg(B, T, N, &tpi, &wpb);
  const size_t lds = (size_t)(C / 16) * QT * 64 * 4 * sizeof(float);
  if (vec) {
    static std::atomic<unsigned long long> attr_done_1;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&mask_gemm_kernel<MODE_LOGITS, true>), (int)MASK_GEMM_MAX_LDS, attr_done_1) != hipSuccess)
      return set_err(PVSG_ERR_HIP, "mask_gemm: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL((mask_gemm_kernel<MODE_LOGITS, true>), dim3(B * wpb), dim3(512), lds, stream,
                       mask_embed, mask_feature, out, (uint32_t*)nullptr, (uint32_t*)nullptr, Q, C, N,
                       T, tpi, wpb);
  } else {
    static std::atomic<unsigned long long> attr_done_2;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&mask_gemm_kernel<MODE_LOGITS, false>), (int)MASK_GEMM_MAX_LDS, attr_done_2) != hipSuccess)
      return set_err(PVSG_ERR_HIP, "mask_gemm: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL((mask_gemm_kernel<MODE_LOGITS, false>), dim3(B * wpb), dim3(512), lds, stream,
                       mask_embed, mask_feature, out, (uint32_t*)nullptr, (uint32_t*)nullptr, Q, C, N,
                       T, tpi, wpb);
  }
  PVSG_LAUNCH_CHECK("mask_logits_forward");
  return PVSG_OK;
}

extern "C" int pvsg_attn_mask_bits_forward(const float* mask_embed, const float* feature_lowres,
                                           uint32_t* bits, uint32_t* flags, int B, int T, int Q,
                                           int C, int N, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(mask_embed && feature_lowres && bits && flags, "attn_mask_bits_forward: null pointer argument");
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && C > 0 && N > 0, "attn_mask_bits_forward: non-positive dimension");
  if (Q > QT * 16 || (C & 15) || C > 320 || (reinterpret_cast<uintptr_t>(bits) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED,
                   "attn_mask_bits_forward: needs Q<=112, C%%16==0, C<=320, 16B-aligned bits "
                   "(got Q=%d C=%d N=%d)", Q, C, N);
  const bool vec = !(N & 3) && !(reinterpret_cast<uintptr_t>(feature_lowres) & 15u);
  hipError_t e = zero_words_async(flags, (size_t)B * 4 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "attn_mask_bits_forward: memset: %s", hipGetErrorString(e));
  int tpi, wpb;
  launch_cfg(B, T, N, &tpi, &wpb);
  const size_t lds = (size_t)(C / 16) * QT * 64 * 4 * sizeof(float);
  if (vec) {
    static std::atomic<unsigned long long> attr_done_3;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&mask_gemm_kernel<MODE_BITS, true>), (int)MASK_GEMM_MAX_LDS, attr_done_3) != hipSuccess)
      return set_err(PVSG_ERR_HIP, "mask_gemm: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL((mask_gemm_kernel<MODE_BITS, true>), dim3(B * wpb), dim3(512), lds, stream,
                       mask_embed, feature_lowres, (float*)nullptr, bits, flags, Q, C, N, T, tpi, wpb);
  } else {
    static std::atomic<unsigned long long> attr_done_4;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(&mask_gemm_kernel<MODE_BITS, false>), (int)MASK_GEMM_MAX_LDS, attr_done_4) != hipSuccess)
      return set_err(PVSG_ERR_HIP, "mask_gemm: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL((mask_gemm_kernel<MODE_BITS, false>), dim3(B * wpb), dim3(512), lds, stream,
                       mask_embed, feature_lowres, (float*)nullptr, bits, flags, Q, C, N, T, tpi, wpb);
  }
  PVSG_LAUNCH_CHECK("attn_mask_bits_forward");
  return PVSG_OK;
}

extern "C" int pvsg_attn_mask_pack(const float* logits_lowres, uint32_t* bits, uint32_t* flags,
                                   int B, int T, int Q, int HW, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(logits_lowres && bits && flags, "attn_mask_pack: null pointer argument");
  PVSG_REQUIRE(B > 0 && T > 0 && Q > 0 && HW > 0, "attn_mask_pack: non-positive dimension");
  PVSG_REQUIRE(Q <= 128, "attn_mask_pack: at most 128 queries (got %d)", Q);
  PVSG_REQUIRE((reinterpret_cast<uintptr_t>(bits) & 15u) == 0, "attn_mask_pack: bits must be 16B aligned");
  hipError_t e = zero_words_async(flags, (size_t)B * 4 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "attn_mask_pack: memset: %s", hipGetErrorString(e));
  const long long K = (long long)T * HW;
  long long nb = (K + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(attn_mask_pack_kernel, dim3((unsigned)nb, B), dim3(256), 0, stream, logits_lowres,
                     bits, flags, T, Q, HW);
  PVSG_LAUNCH_CHECK("attn_mask_pack");
  return PVSG_OK;
}

extern "C" int pvsg_center_downsample(const float* feature, float* d2, float* d4, float* d8,
                                      long long planes, int H, int W, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(feature && d2 && d4 && d8, "center_downsample: null pointer argument");
  PVSG_REQUIRE(planes > 0 && H > 0 && W > 0, "center_downsample: non-positive dimension");
  if ((H & 7) || (W & 7) || (reinterpret_cast<uintptr_t>(feature) & 15u) ||
      (reinterpret_cast<uintptr_t>(d2) & 15u) || (reinterpret_cast<uintptr_t>(d4) & 7u))
    return set_err(PVSG_ERR_UNSUPPORTED, "center_downsample: needs H%%8==0 and W%%8==0 (got %dx%d)", H, W);
  const long long nblk = planes * (H >> 3) * (W >> 3);
  long long nb = (nblk + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(center_downsample_kernel, dim3((unsigned)nb), dim3(256), 0, stream, feature, d2, d4,
                     d8, H, W, nblk);
  PVSG_LAUNCH_CHECK("center_downsample");
  return PVSG_OK;
}
