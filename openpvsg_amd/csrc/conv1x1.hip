// 1x1 convolution + frozen BatchNorm (+ residual) (+ ReLU) in one pass, for the HBM-bound layers of the backbone.
//
// [3P] mmdet ResNet bottleneck (norm_eval): conv3 (1x1) -> BatchNorm -> + identity -> ReLU, and the stride-1
// downsample conv -> BatchNorm.  In layer1 / layer2 these GEMMs have 64 / 128 input channels: 25-50 flop per byte, far
// below the f32 matrix-core ridge, so the separate BN/ReLU pass (read conv output + identity, write) costs more than
// the convolution itself (0.63 ms + 1.09 ms per layer1 block at 32 x 720p).  Here the BN scale/shift, the identity and
// the ReLU are applied to the accumulators before the only store: traffic = x + identity + out.
//
// Same matrix-core scheme as mask_gemm.hip: out[b] (Cout x HW) = W (Cout x Cin) @ x[b] (Cin x HW),
// v_mfma_f32_16x16x4_f32, a wave owns 32 output channels x 64 pixels (the waves of a workgroup that share a pixel tile
// read the same x rows: served by L1/L2); B operand = one float4 of x per lane and
// k-step straight from HBM (256-byte runs, each lane ends up with 4 consecutive pixels -> float4 stores); A operand =
// the workgroup's weight rows (up to 64 KB) staged once in LDS in fragment order (conflict-free ds_read_b128).  The
// identity tile is fetched before the K loop and the first x rows of the wave's NEXT tile under the last MFMA block, so
// that with only Cin/16 = 4..16 K-steps per tile the matrix pipe does not drain between tiles.
#include "common.h"

namespace pvsg {

// CMT row tiles of 16 output channels per wave tile
template <int CMT>
__global__ __launch_bounds__(512) void conv1x1_affine_kernel(
    const float* __restrict__ Wt, const float* __restrict__ X, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ out, int Cout, int Cin,
    int HW, int relu, int mw, int mgroups, int wgs_per_bm) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];  // [mw][Cin/16][CMT][64][4]
  const int bm = blockIdx.x / wgs_per_bm;
  const int wg = blockIdx.x - bm * wgs_per_bm;
  const int b = bm / mgroups, mg = bm - b * mgroups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, g = lane >> 4;
  const int rows = mw * (CMT * 16);
  const int mbase = mg * rows;
  {
    const float* Wp = Wt + (long long)mbase * Cin;
    const int total = rows * Cin;
    const int panel = CMT * 16 * Cin;                 // floats per 64-row panel
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int q = i / Cin, c = i - q * Cin;
      const int mi = q / (CMT * 16), ql = q - mi * (CMT * 16);
      const int kb = c >> 4, cc = c & 15;
      const int ln = (cc & 3) * 16 + (ql & 15);
      wlds[mi * panel + (((kb * CMT + (ql >> 4)) * 64 + ln) << 2) + (cc >> 2)] = Wp[i];
    }
  }
  __syncthreads();
  const int ppw = 8 / mw;                             // pixel tiles per workgroup pass
  const int mi = wave % mw, pj = wave / mw;
  const float* wl = wlds + mi * (CMT * 16 * Cin);
  const int m0 = mbase + mi * (CMT * 16);
  const int ntiles = (HW + 63) >> 6;
  const int nkb = Cin >> 4;
  const float* Xb = X + (long long)b * Cin * HW;
  float sc[CMT][4], sh[CMT][4];
#pragma unroll
  for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[qt][r] = scale[m0 + qt * 16 + g * 4 + r];
      sh[qt][r] = shift[m0 + qt * 16 + g * 4 + r];
    }
  const int stride = wgs_per_bm * ppw;
  int tile = wg * ppw + pj;
  float4 bv[4], bn[4];
  {
    const int n = tile * 64 + 4 * j16;
    const bool valid = tile < ntiles && n + 3 < HW;
    const float* Fp = Xb + (long long)g * HW + (valid ? n : 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = valid ? ld4(Fp + (long long)(4 * s) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (; tile < ntiles; tile += stride) {
    const int n = tile * 64 + 4 * j16;
    const bool valid = n + 3 < HW;
    const float* Fp = Xb + (long long)g * HW + (valid ? n : 0);
    const int nn = (tile + stride) * 64 + 4 * j16;                  // the wave's next tile: its first rows are
    const bool nvalid = tile + stride < ntiles && nn + 3 < HW;      // fetched under this tile's last MFMA block
    const float* Fn = Xb + (long long)g * HW + (nvalid ? nn : 0);
    // lane holds channels m0 + qt*16 + g*4 + r, pixels n..n+3 (one per column tile)
    const long long base = ((long long)b * Cout + m0 + g * 4) * HW + (valid ? n : 0);
    float4 rr[CMT][4];
    if (residual) {
#pragma unroll
      for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rr[qt][r] = valid ? ld4_stream(residual + base + (long long)(qt * 16 + r) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[CMT][4];
#pragma unroll
    for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[qt][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
      if (kb + 1 < nkb) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          bn[s] = valid ? ld4(Fp + (long long)((kb + 1) * 16 + 4 * s) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          bn[s] = nvalid ? ld4(Fn + (long long)(4 * s) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float4 a[CMT];
#pragma unroll
      for (int qt = 0; qt < CMT; ++qt)
        a[qt] = *reinterpret_cast<const float4*>(&wl[((kb * CMT + qt) * 64 + lane) << 2]);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b0 = bv[s].x, b1 = bv[s].y, b2 = bv[s].z, b3 = bv[s].w;
#pragma unroll
        for (int qt = 0; qt < CMT; ++qt) {
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
    if (valid) {
#pragma unroll
      for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = sc[qt][r], s0 = sh[qt][r];
          float4 v = make_float4(acc[qt][0][r] * s1 + s0, acc[qt][1][r] * s1 + s0, acc[qt][2][r] * s1 + s0,
                                 acc[qt][3][r] * s1 + s0);
          if (residual) { v.x += rr[qt][r].x; v.y += rr[qt][r].y; v.z += rr[qt][r].z; v.w += rr[qt][r].w; }
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          st4(out + base + (long long)(qt * 16 + r) * HW, v);
        }
    }
  }
}

// Same tile scheme with the x operand four K-blocks ahead (Cin % 64 == 0): a K-block is 32 MFMAs (~0.4 us) per wave
// but an HBM round trip under load is ~2 us, so with a one-block look-ahead the wave waits on memory in every block
// and matrix time and memory time add up instead of overlapping.  Here a ring of four blocks (16 float4 per lane) is
// always in flight; the ring rolls over into the wave's NEXT tile, so the pipe stays fed across tiles too.
template <int CMT>
__global__ __launch_bounds__(512) void conv1x1_affine_deep_kernel(
    const float* __restrict__ Wt, const float* __restrict__ X, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ out, int Cout, int Cin,
    int HW, int relu, int mw, int mgroups, int wgs_per_bm) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int bm = blockIdx.x / wgs_per_bm;
  const int wg = blockIdx.x - bm * wgs_per_bm;
  const int b = bm / mgroups, mg = bm - b * mgroups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j16 = lane & 15, g = lane >> 4;
  const int rows = mw * (CMT * 16);
  const int mbase = mg * rows;
  {
    const float* Wp = Wt + (long long)mbase * Cin;
    const int total = rows * Cin;
    const int panel = CMT * 16 * Cin;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
      const int q = i / Cin, c = i - q * Cin;
      const int mi = q / (CMT * 16), ql = q - mi * (CMT * 16);
      const int kb = c >> 4, cc = c & 15;
      const int ln = (cc & 3) * 16 + (ql & 15);
      wlds[mi * panel + (((kb * CMT + (ql >> 4)) * 64 + ln) << 2) + (cc >> 2)] = Wp[i];
    }
  }
  __syncthreads();
  const int ppw = 8 / mw;
  const int mi = wave % mw, pj = wave / mw;
  const float* wl = wlds + mi * (CMT * 16 * Cin);
  const int m0 = mbase + mi * (CMT * 16);
  const int ntiles = (HW + 63) >> 6;
  const int nkb = Cin >> 4;
  const float* Xb = X + (long long)b * Cin * HW + (long long)g * HW;
  float sc[CMT][4], sh[CMT][4];
#pragma unroll
  for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[qt][r] = scale[m0 + qt * 16 + g * 4 + r];
      sh[qt][r] = shift[m0 + qt * 16 + g * 4 + r];
    }
  const int stride = wgs_per_bm * ppw;
  int tile = wg * ppw + pj;
  float4 bq[4][4];
  {
    const int n = tile * 64 + 4 * j16;
    const bool valid = tile < ntiles && n + 3 < HW;
    const float* Fp = Xb + (valid ? n : 0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        bq[u][s] = valid ? ld4(Fp + (long long)(u * 16 + 4 * s) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (; tile < ntiles; tile += stride) {
    const int n = tile * 64 + 4 * j16;
    const bool valid = n + 3 < HW;
    const float* Fp = Xb + (valid ? n : 0);
    const int nn = (tile + stride) * 64 + 4 * j16;
    const bool nvalid = tile + stride < ntiles && nn + 3 < HW;
    const float* Fn = Xb + (nvalid ? nn : 0);
    const long long base = ((long long)b * Cout + m0 + g * 4) * HW + (valid ? n : 0);
    float4 rr[CMT][4];
    if (residual) {
#pragma unroll
      for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rr[qt][r] = valid ? ld4_stream(residual + base + (long long)(qt * 16 + r) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[CMT][4];
#pragma unroll
    for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
      for (int x = 0; x < 4; ++x) acc[qt][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb4 = 0; kb4 < nkb; kb4 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kb = kb4 + u;
        float4 a[CMT];
#pragma unroll
        for (int qt = 0; qt < CMT; ++qt)
          a[qt] = *reinterpret_cast<const float4*>(&wl[((kb * CMT + qt) * 64 + lane) << 2]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float b0 = bq[u][s].x, b1 = bq[u][s].y, b2 = bq[u][s].z, b3 = bq[u][s].w;
#pragma unroll
          for (int qt = 0; qt < CMT; ++qt) {
            const float av = (s == 0) ? a[qt].x : (s == 1) ? a[qt].y : (s == 2) ? a[qt].z : a[qt].w;
            acc[qt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[qt][0], 0, 0, 0);
            acc[qt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[qt][1], 0, 0, 0);
            acc[qt][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc[qt][2], 0, 0, 0);
            acc[qt][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3, acc[qt][3], 0, 0, 0);
          }
        }
        // refill this ring slot: four K-blocks ahead, rolling over into the next tile
        const int nk = kb + 4;
        const bool cur = nk < nkb;
        const float* src = cur ? Fp + (long long)(nk * 16) * HW : Fn + (long long)((nk - nkb) * 16) * HW;
        const bool ok = cur ? valid : nvalid;
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[u][s] = ok ? ld4(src + (long long)(4 * s) * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (valid) {
#pragma unroll
      for (int qt = 0; qt < CMT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s1 = sc[qt][r], s0 = sh[qt][r];
          float4 v = make_float4(acc[qt][0][r] * s1 + s0, acc[qt][1][r] * s1 + s0, acc[qt][2][r] * s1 + s0,
                                 acc[qt][3][r] * s1 + s0);
          if (residual) { v.x += rr[qt][r].x; v.y += rr[qt][r].y; v.z += rr[qt][r].z; v.w += rr[qt][r].w; }
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          st4(out + base + (long long)(qt * 16 + r) * HW, v);
        }
    }
  }
}

}  // namespace pvsg

extern "C" int pvsg_conv1x1_affine(const float* weight, const float* x, const float* scale, const float* shift,
                                   const float* residual, float* out, int B, int Cout, int Cin, long long HW, int relu,
                                   hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && x && scale && shift && out, "conv1x1_affine: null pointer argument");
  PVSG_REQUIRE(B > 0 && Cout > 0 && Cin > 0 && HW > 0 && HW < (1LL << 31), "conv1x1_affine: bad shape");
  if (Cout % 32 || Cin % 16 || Cin > 256 || (HW & 3))
    return set_err(PVSG_ERR_UNSUPPORTED,
                   "conv1x1_affine: built for Cout %% 32 == 0, Cin %% 16 == 0, Cin <= 256, HW %% 4 == 0 (got %d %d %lld)",
                   Cout, Cin, HW);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(residual)) & 15u),
               "conv1x1_affine: 16-byte alignment required");
  // panels of CMT*16 rows per workgroup: a divisor of 8 (waves) and of the panel count, weights <= `lds_cap` in LDS
  // (measured: 32-row wave tiles with <= 64 KB of weights per workgroup -- two workgroups per CU -- beat 64-row tiles)
  constexpr int CMT = 2;
  const int prow = CMT * 16;
  int mw = 8;
  while (mw > 1 && ((Cout / prow) % mw || (long long)mw * prow * Cin * 4 > 64 * 1024)) mw >>= 1;
  const int mgroups = Cout / (mw * prow);
  const int ppw = 8 / mw;
  const int ntiles = (int)((HW + 63) / 64);
  int wgs = (1024 + B * mgroups - 1) / (B * mgroups);
  const int maxw = (ntiles + ppw * 4 - 1) / (ppw * 4);      // at least ~4 tiles per wave
  if (wgs > maxw) wgs = maxw;
  if (wgs < 1) wgs = 1;
  const size_t lds = (size_t)mw * prow * Cin * sizeof(float);
  PVSG_REQUIRE(lds <= 128 * 1024, "conv1x1_affine: weight panel does not fit LDS");
  const bool deep = (Cin % 64) == 0;
  const auto kern = deep ? conv1x1_affine_deep_kernel<CMT> : conv1x1_affine_kernel<CMT>;
  static std::atomic<unsigned long long> attr_done[2];
  {
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 128 * 1024, attr_done[deep]);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "conv1x1_affine: LDS attribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * mgroups * wgs)), dim3(512), lds, stream, weight, x, scale, shift, residual,
                     out, Cout, Cin, (int)HW, relu, mw, mgroups, wgs);
  PVSG_LAUNCH_CHECK("conv1x1_affine");
  return PVSG_OK;
}
