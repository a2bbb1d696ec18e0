// ResNet stem in one kernel: 7x7 / stride 2 / pad 3 convolution (3 -> 64 channels) -> frozen BatchNorm -> ReLU ->
// MaxPool2d(3, stride 2, pad 1), NCHW f32.
//
// Replaces, on the north-star path, [3P] mmdet ResNet.conv1 -> norm1 -> relu -> maxpool (style='pytorch'), which ran as
// a MIOpen convolution (2.8 ms at 32 x 736 x 1280) + the BN / ReLU / pool pass of fpn_fuse.hip (0.6 ms) with the
// 1.9 GB convolution output written and read back in between.
//
// A workgroup produces an 8 x 16 block of POOLED pixels of one image for all 64 channels: it needs the 17 x 33 convolution
// outputs under those pooling windows (10 % recomputed along the block edges), i.e. a 39 x 72 input patch per colour,
// staged into LDS once (out-of-image elements read as 0 through the buffer bounds check).  The convolution is a direct
// convolution on v_mfma_f32_32x32x2_f32, rows = 32 output channels, columns = 32 of the 561 convolution pixels, K = the
// 3 x 7 x 8 taps (the 7-wide rows padded to 8 with a zero weight, so that the two K lanes read columns 2j and 2j+1 of the
// window: 64 consecutive LDS words per instruction, no address arithmetic).  The loop over the 21 (colour, row) groups
// is fully unrolled -- every LDS offset is an immediate, the weights arrive lane-major from L2 (pvsg_stem7x7_pack) -- and
// holds nothing but MFMAs, LDS reads and weight loads.  Then BN + ReLU on the accumulators, the 17 x 33 maps of 32
// channels at a time through LDS (over the dead input patch), and the 3 x 3 / 2 maximum.
#include "common.h"

namespace pvsg {
namespace {

constexpr int ST_PH = 8, ST_PW = 16;                  // pooled block
constexpr int ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1;   // 17 x 33 convolution outputs
constexpr int ST_CPIX = ST_CH * ST_CW;                // 561
constexpr int ST_BLK = (ST_CPIX + 31) / 32;           // 18 blocks of 32 pixels
constexpr int ST_IH = 2 * (ST_CH - 1) + 7;            // 39 input rows
constexpr int ST_PITCH = 72;                          // 2 * 32 + 7 input columns + the zero-weight tap
constexpr int ST_PLANE = ST_IH * ST_PITCH;            // 2808
constexpr int ST_PATCH = 3 * ST_PLANE;                // 8424 floats
constexpr int ST_CPITCH = 564;                        // convolution map of one channel in LDS
constexpr int ST_LDS_FLOATS = 32 * ST_CPITCH;         // 18048 floats = 72 192 B (>= the patch)

template <int NB>
__device__ __forceinline__ void stem_tile(float* lds, const float* __restrict__ wp, const float* __restrict__ scale,
                                          const float* __restrict__ shift, float* __restrict__ out, int wave, int lane, int tid,
                                          int img, int py0, int px0, int cy0, int cx0, int Hc, int Wc, int Hp, int Wp) {
  const int k = lane >> 5, n = lane & 31;
  int boff[NB];                                          // LDS word offset of the lane's window origin (+ its K lane)
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int p = (wave + 4 * i) * 32 + n;
    p = p < ST_CPIX ? p : ST_CPIX - 1;                    // padding pixels of the last block: any valid window
    const int cy = p / ST_CW, cx = p - cy * ST_CW;
    boff[i] = 2 * cy * ST_PITCH + 2 * cx + k;
  }

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][i][r] = 0.f;

  const f32x4* wl = reinterpret_cast<const f32x4*>(wp) + lane * 2;     // [group 21][lane 64][channel block 2][K pair 4]

#pragma unroll
  for (int g = 0; g < 21; ++g) {
    const int c = g / 7, u = g % 7;
    const f32x4 a0 = wl[g * 128], a1 = wl[g * 128 + 1];
    float bv[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float* q = lds + boff[i] + c * ST_PLANE + u * ST_PITCH;
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[i][j] = q[2 * j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bv[i][j], acc[0][i], 0, 0, 0);
        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bv[i][j], acc[1][i], 0, 0, 0);
      }
  }

  // ---- BN + ReLU, 32 channels at a time through LDS, 3x3 / 2 maximum ----
  const int pq = tid & 127, half = tid >> 7;             // pooled pixel of the block, 16-channel half
  const int q = pq >> 4, s = pq & 15;
  const int py = py0 + q, px = px0 + s;
  bool rv[3], cv[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    rv[d] = cy0 + 2 * q + d >= 0 && cy0 + 2 * q + d < Hc;
    cv[d] = cx0 + 2 * s + d >= 0 && cx0 + 2 * s + d < Wc;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();                                     // the patch (a = 0) / the previous maps (a = 1) are no longer read
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = (wave + 4 * i) * 32 + n;
      if (p < ST_CPIX) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = (r & 3) + 8 * (r >> 2) + 4 * k;
          const float v = fmaf(acc[a][i][r], scale[a * 32 + cl], shift[a * 32 + cl]);
          lds[cl * ST_CPITCH + p] = fmaxf(v, 0.f);
        }
      }
    }
    __syncthreads();
    if (py < Hp && px < Wp) {
#pragma unroll 4
      for (int cc = 0; cc < 16; ++cc) {
        const float* m = lds + (half * 16 + cc) * ST_CPITCH + 2 * q * ST_CW + 2 * s;
        float best = 0.f;                                // ReLU output >= 0 and every window holds a valid cell
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (rv[d] && cv[e]) best = fmaxf(best, m[d * ST_CW + e]);
        out[(((size_t)img * 64 + a * 32 + half * 16 + cc) * Hp + py) * Wp + px] = best;
      }
    }
  }
}


__global__ __launch_bounds__(256, 2)
void stem7x7_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                    const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int Hc, int Wc, int Hp,
                    int Wp, int TY, int TX) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int bx = logical % TX;
  logical /= TX;
  const int by = logical % TY, img = logical / TY;
  const int py0 = by * ST_PH, px0 = bx * ST_PW;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;       // first convolution output of the block
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;       // first input element of the patch

  // ---- stage the input patch: 3 x 39 x 72 elements, 33 per thread ----
  const size_t HW = (size_t)H * W;
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)img * 3 * HW, 0, (unsigned)(3 * HW * 4), 0x00020000);
  {
    constexpr int PER = (ST_PATCH + 255) / 256;          // all loads in flight before the first LDS write
    float tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      const int c = e / ST_PLANE, rem = e - c * ST_PLANE;
      const int r = rem / ST_PITCH, col = rem - r * ST_PITCH;
      const int iy = iy0 + r, ix = ix0 + col;
      const bool ok = e < ST_PATCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const unsigned off = ok ? (unsigned)(((size_t)c * HW + (size_t)iy * W + ix) * 4) : 0x80000000u;
      tmp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (tid + 256 * i < ST_PATCH) lds[tid + 256 * i] = tmp[i];
  }

  __syncthreads();
  // ---- this wave's pixel blocks: wave, wave + 4, ... -- five for waves 0 and 1, four for waves 2 and 3 (18 blocks) ----
  if (wave < 2)
    stem_tile<5>(lds, wp, scale, shift, out, wave, lane, tid, img, py0, px0, cy0, cx0, Hc, Wc, Hp, Wp);
  else
    stem_tile<4>(lds, wp, scale, shift, out, wave, lane, tid, img, py0, px0, cy0, cx0, Hc, Wc, Hp, Wp);
}

// w (64, 3, 7, 7) -> [group = colour * 7 + row][lane 64][channel block 2][K pair 4]: lane (m, k) holds w[32 blk + m][c][u][2 j + k]
__global__ void stem7x7_pack_kernel(const float* __restrict__ w, float* __restrict__ wp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 21 * 64 * 8) return;
  const int j = idx & 3, blk = (idx >> 2) & 1, lane = (idx >> 3) & 63, g = idx >> 9;
  const int m = lane & 31, k = lane >> 5, c = g / 7, u = g % 7, v = 2 * j + k;
  wp[idx] = v < 7 ? w[(((size_t)(blk * 32 + m) * 3 + c) * 7 + u) * 7 + v] : 0.f;
}


// ------------------------------------------------------------------------------------------------------------------
// The same stem on the f16 matrix pipe (two-limb split, csrc/split_common.h section "Two-limb f16 split"): round 5.
// v_mfma_f32_32x32x2_f32 issues the stem's 168 padded taps in 84 K-steps of 64 cycles -- 2.0 ms at 32 x 720p, matrix-bound.
// Here the patch is split ONCE while it is staged (a_h = f16(a), a_l' = f16(2^11 (a - a_h)), two 16-bit planes in the space of the
// f32 patch) and the convolution runs on v_mfma_f32_16x16x32_f16: rows = 16 output channels, columns = 16 convolution pixels,
// K = 192 = 24 (colour, row) groups of 8 taps (21 real groups, the 8th tap and the last three groups have zero weights): k-step s
// holds groups 4 s + kg4, so a lane's B fragment is 8 consecutive patch columns of one row -- four 4-byte LDS reads per limb, no
// arithmetic at all in the loop.  Three limb products per multiply (w_h2 x_l', w_l x_h, w_h x_h): 72 MFMAs of 16 cycles per 16
// pixels x 64 channels against 168 of 64 cycles per 32 pixels.  Weights: pvsg_gemm_f16x2_pack of the (64, 192) matrix
// M[ch][8 g + e] = w[ch][g / 7][g % 7][e] (0 for e = 7 or g >= 21), fragments straight from L2.  Epilogue (BN, ReLU, 3x3/2 max
// through LDS) as above.  Inputs beyond the f16 range are counted into `overflow` (the caller re-runs on the f32 kernel).
typedef _Float16 st_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned st_u32x4 __attribute__((ext_vector_type(4)));
constexpr int ST16_BLK = (ST_CPIX + 15) / 16;          // 36 blocks of 16 pixels, 9 per wave

__global__ __launch_bounds__(256, 2)
void stem7x7_f16x2_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp, const float* __restrict__ scale,
                          const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int Hc, int Wc, int Hp,
                          int Wp, int TY, int TX, unsigned* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* ph = reinterpret_cast<unsigned short*>(lds);            // x_h plane, then x_l' plane (ST_PATCH halves each)
  unsigned short* pl = ph + ST_PATCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int bx = logical % TX;
  logical /= TX;
  const int by = logical % TY, img = logical / TY;
  const int py0 = by * ST_PH, px0 = bx * ST_PW;
  const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;
  const size_t HW = (size_t)H * W;
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)img * 3 * HW, 0, (unsigned)(3 * HW * 4), 0x00020000);
  {
    constexpr int PER = (ST_PATCH + 255) / 256;
    float tmp[PER];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      const int c = e / ST_PLANE, rem = e - c * ST_PLANE;
      const int r = rem / ST_PITCH, col = rem - r * ST_PITCH;
      const int iy = iy0 + r, ix = ix0 + col;
      const bool ok = e < ST_PATCH && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const unsigned off = ok ? (unsigned)(((size_t)c * HW + (size_t)iy * W + ix) * 4) : 0x80000000u;
      tmp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const float a = tmp[i];
      const _Float16 h = (_Float16)a;
      const _Float16 l = (_Float16)((a - (float)h) * 2048.f);
      amax = fmaxf(amax, __builtin_fabsf(a));
      if (tid + 256 * i < ST_PATCH) {
        ph[tid + 256 * i] = __builtin_bit_cast(unsigned short, h);
        pl[tid + 256 * i] = __builtin_bit_cast(unsigned short, l);
      }
    }
    if (overflow && !(amax <= 65504.f)) atomicAdd(overflow, 1u);
  }
  __syncthreads();

  const int l15 = lane & 15, kg4 = lane >> 4;
  constexpr int NB = ST16_BLK / 4;                     // 9
  int boff[NB];                                        // element offset of the lane's window origin in a plane
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int p = (wave + 4 * i) * 16 + l15;
    p = p < ST_CPIX ? p : ST_CPIX - 1;
    const int cy = p / ST_CW, cx = p - cy * ST_CW;
    boff[i] = 2 * cy * ST_PITCH + 2 * cx;
  }
  f32x4 acc[4][NB];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[rb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // packed weight (Npad = 128, K = 192): element (((kt * 2 + limb) * 2 + kg) * 128 + row) * 8, kt = 2 s + (kg4 >> 1), kg = kg4 & 1
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wp), 0, 2u * 128u * 192u * 2u, 0x00020000);
  const unsigned wvo = (unsigned)((((kg4 >> 1) * 4 + (kg4 & 1)) * 128 + l15) * 8) * 2u;
  const float unscale = reinterpret_cast<const float*>(wp + 2 * 128 * 192)[1];
  auto mf = [](st_u32x4 a, st_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(st_f16x8, a), __builtin_bit_cast(st_f16x8, b), c, 0, 0, 0);
  };
#pragma unroll 1
  for (int s = 0; s < 6; ++s) {
    const int g = 4 * s + kg4, gc = g < 21 ? g : 20;   // groups 21..23: zero weights, any valid window
    const int goff = (gc / 7) * ST_PLANE + (gc % 7) * ST_PITCH;
    st_u32x4 whf[4], wlf[4], w2f[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const unsigned so = (unsigned)((2 * s) * 4096 + rb * 128) * 2u;
      whf[rb] = __builtin_bit_cast(st_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0));
      wlf[rb] = __builtin_bit_cast(st_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so + 2048u * 2u, 0));
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const st_f16x2 k = {(_Float16)(1.f / 2048.f), (_Float16)(1.f / 2048.f)};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned w = whf[rb][q];
        w2f[rb][q] = __builtin_bit_cast(unsigned, __builtin_bit_cast(st_f16x2, w) * k);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const unsigned* qh = reinterpret_cast<const unsigned*>(ph + boff[i] + goff);       // 4-byte aligned: even element offsets
      const unsigned* ql = reinterpret_cast<const unsigned*>(pl + boff[i] + goff);
      const st_u32x4 xh = {qh[0], qh[1], qh[2], qh[3]}, xl = {ql[0], ql[1], ql[2], ql[3]};
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][i] = mf(w2f[rb], xl, acc[rb][i]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][i] = mf(wlf[rb], xh, acc[rb][i]);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) acc[rb][i] = mf(whf[rb], xh, acc[rb][i]);
    }
  }

  // ---- BN + ReLU, 32 channels at a time through LDS, 3x3 / 2 maximum (register r of (rb, i): channel 16 rb + 4 kg4 + r) ----
  const int pq = tid & 127, half = tid >> 7;
  const int q = pq >> 4, sx = pq & 15;
  const int py = py0 + q, px = px0 + sx;
  bool rv[3], cv[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    rv[d] = cy0 + 2 * q + d >= 0 && cy0 + 2 * q + d < Hc;
    cv[d] = cx0 + 2 * sx + d >= 0 && cx0 + 2 * sx + d < Wc;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int p = (wave + 4 * i) * 16 + l15;
      if (p < ST_CPIX) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int cl = 16 * rr + 4 * kg4 + r;
            const float v = fmaf(acc[2 * a + rr][i][r], scale[a * 32 + cl] * unscale, shift[a * 32 + cl]);
            lds[cl * ST_CPITCH + p] = fmaxf(v, 0.f);
          }
      }
    }
    __syncthreads();
    if (py < Hp && px < Wp) {
#pragma unroll 4
      for (int cc = 0; cc < 16; ++cc) {
        const float* m = lds + (half * 16 + cc) * ST_CPITCH + 2 * q * ST_CW + 2 * sx;
        float best = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (rv[d] && cv[e]) best = fmaxf(best, m[d * ST_CW + e]);
        out[(((size_t)img * 64 + a * 32 + half * 16 + cc) * Hp + py) * Wp + px] = best;
      }
    }
  }
}

// (64, 3, 7, 7) -> the (64, 192) matrix stem7x7_f16x2_kernel multiplies by: column 8 g + e = w[ch][g / 7][g % 7][e], 0 for e = 7, g >= 21
__global__ void stem7x7_matrix_kernel(const float* __restrict__ w, float* __restrict__ m) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 192) return;
  const int ch = idx / 192, k = idx - ch * 192, g = k >> 3, e = k & 7;
  m[idx] = (g < 21 && e < 7) ? w[((size_t)ch * 21 + g) * 7 + e] : 0.f;
}
}  // namespace
}  // namespace pvsg

extern "C" int pvsg_stem7x7_pack(const float* weight, float* w_packed, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && w_packed, "stem7x7_pack: null pointer argument");
  hipLaunchKernelGGL(stem7x7_pack_kernel, dim3((21 * 64 * 8 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), weight,
                     w_packed);
  PVSG_LAUNCH_CHECK("stem7x7_pack");
  return PVSG_OK;
}

extern "C" int pvsg_stem7x7_bn_relu_pool(const float* x, const float* w_packed, const float* scale, const float* shift,
                                         float* out, int N, int H, int W, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && w_packed && scale && shift && out, "stem7x7_bn_relu_pool: null pointer argument");
  PVSG_REQUIRE(N > 0 && H > 0 && W > 0, "stem7x7_bn_relu_pool: bad shape");
  if ((long long)3 * H * W >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "stem7x7_bn_relu_pool: image too large (H=%d W=%d)", H, W);
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(w_packed) & 15u), "stem7x7_bn_relu_pool: w_packed must be 16-byte aligned");
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;          // 7x7 / 2, pad 3
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;        // 3x3 / 2, pad 1
  const int TY = (Hp + ST_PH - 1) / ST_PH, TX = (Wp + ST_PW - 1) / ST_PW;
  const long long blocks = (long long)N * TY * TX;
  PVSG_REQUIRE(blocks < (1LL << 31), "stem7x7_bn_relu_pool: too many blocks");
  const int lds_bytes = ST_LDS_FLOATS * (int)sizeof(float);
  static std::atomic<unsigned long long> attr_done;
  {
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(stem7x7_kernel), lds_bytes, attr_done);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "stem7x7_bn_relu_pool: LDS attribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(stem7x7_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), x, w_packed,
                     scale, shift, out, N, H, W, Hc, Wc, Hp, Wp, TY, TX);
  PVSG_LAUNCH_CHECK("stem7x7_bn_relu_pool");
  return PVSG_OK;
}

// The stem on the f16 matrix pipe (stem7x7_f16x2_kernel above).  w_packed = pvsg_gemm_f16x2_pack(N = 64, K = 192) of the matrix
// pvsg_stem7x7_f16x2_matrix writes; `overflow` as for the other f16x2 entries (inputs beyond +-65504 are counted, results then invalid).
extern "C" int pvsg_stem7x7_f16x2_matrix(const float* weight, float* matrix, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && matrix, "stem7x7_f16x2_matrix: null pointer argument");
  hipLaunchKernelGGL(stem7x7_matrix_kernel, dim3((64 * 192 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), weight, matrix);
  PVSG_LAUNCH_CHECK("stem7x7_f16x2_matrix");
  return PVSG_OK;
}

extern "C" int pvsg_stem7x7_f16x2_bn_relu_pool(const float* x, const void* w_packed, const float* scale, const float* shift,
                                               float* out, int N, int H, int W, uint32_t* overflow, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && w_packed && scale && shift && out, "stem7x7_f16x2_bn_relu_pool: null pointer argument");
  PVSG_REQUIRE(N > 0 && H > 0 && W > 0, "stem7x7_f16x2_bn_relu_pool: bad shape");
  if ((long long)3 * H * W >= (1LL << 29))
    return set_err(PVSG_ERR_UNSUPPORTED, "stem7x7_f16x2_bn_relu_pool: image too large (H=%d W=%d)", H, W);
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(w_packed) & 15u), "stem7x7_f16x2_bn_relu_pool: w_packed must be 16-byte aligned");
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
  const int TY = (Hp + ST_PH - 1) / ST_PH, TX = (Wp + ST_PW - 1) / ST_PW;
  const long long blocks = (long long)N * TY * TX;
  PVSG_REQUIRE(blocks < (1LL << 31), "stem7x7_f16x2_bn_relu_pool: too many blocks");
  const int lds_bytes = ST_LDS_FLOATS * (int)sizeof(float);
  static std::atomic<unsigned long long> attr_done;
  {
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(stem7x7_f16x2_kernel), lds_bytes, attr_done);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "stem7x7_f16x2_bn_relu_pool: LDS attribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(stem7x7_f16x2_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), x,
                     static_cast<const unsigned short*>(w_packed), scale, shift, out, N, H, W, Hc, Wc, Hp, Wp, TY, TX, overflow);
  PVSG_LAUNCH_CHECK("stem7x7_f16x2_bn_relu_pool");
  return PVSG_OK;
}
