// 3x3 / stride 2 / pad 1 convolution in NCHW as a direct (implicit-GEMM) convolution on the f32 matrix cores, with the
// frozen-BN affine + ReLU in the epilogue.
//
// Replaces, on the north-star path, the library call behind the first bottleneck of ResNet layers 2-4
//   [3P] mmdet ResNet Bottleneck.conv2 (stride 2, style='pytorch') -> bn2 -> relu
// which MIOpen serves with miopenSp3AsmConv ... f3x2_stride2 at 2.3-3.1 ms per layer for 0.14 TFLOP (45-60 TFLOP/s).
//
// Mapping.  out[co, p] = sum_{ci, u, v} w[co, ci, u, v] x[ci, 2 py + u - 1, 2 px + v - 1]: for every (channel pair, tap)
// one v_mfma_f32_32x32x2_f32 per (32 output channels x 32 output pixels) block, K = the two channels of the pair.
// The lane that supplies B[k][j] for (channel k, pixel j) reads that pixel's 3x3 window of that channel from LDS (three
// ds_read2_b64: the rows of the window, the same halo geometry and bank mapping as the 4x4 patches of winograd3x3.hip)
// and feeds one window element per tap: no arithmetic besides the MFMAs in the loop.  A wave owns 64 channels x 64 pixels
// (2 x 2 blocks, 64 accumulator registers; every A value is used for two pixel blocks, every window element for two
// channel blocks), a workgroup of 4 waves 128 channels x 128 pixels (16 wide x 8 high), two workgroups per CU.
// Weights arrive pre-packed lane-major (pvsg_conv3x3s2_pack: 9 taps + 3 pad per lane, channel pair and 32-channel block),
// global -> register two channel pairs ahead; the input goes through LDS in stages of 8 channels x 17 x 33 halo
// elements (three buffers, one barrier per stage, out-of-image elements read as 0 through the buffer bounds check).
#include "common.h"

#include <type_traits>

namespace pvsg {
namespace {

constexpr int S2_KC = 8;                         // input channels per LDS stage (4 channel pairs)
constexpr int S2_ROWS = 17, S2_COLS = 33;        // halo block of an 8 x 16 output block
constexpr int S2_PITCH = 40;                     // LDS row pitch (floats): 2 rows = 80 = 16 mod 64 banks
constexpr int S2_PLANE = S2_ROWS * S2_PITCH;     // 680
constexpr int S2_STAGE = S2_KC * S2_PLANE;       // 5440 floats; three stages = 65280 B of static LDS
constexpr int S2_POS = S2_ROWS * S2_COLS;        // 561 halo positions, 3 per thread (the last round partly idle)

template <bool RELU>
__global__ __launch_bounds__(256, 2)
void conv3x3s2_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ scale,
                      const float* __restrict__ shift, float* __restrict__ y, int N, int Cin, int Cout, int H, int W, int Ho,
                      int Wo, int TY, int TX) {
  __shared__ __attribute__((aligned(16))) float lds[3 * S2_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cw = wave >> 1, pw = wave & 1;                        // 64-channel half / 8-column half of this wave
  const int k = lane >> 5, n = lane & 31;
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int bx = logical % TX;
  logical /= TX;
  const int by = logical % TY;
  logical /= TY;
  const int img = logical % N, cg = logical / N;
  const int oy0 = by * 8, ox0 = bx * 16;
  const size_t HW = (size_t)H * W;
  const float* xn = x + (size_t)img * Cin * HW;
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (unsigned)(Cin * HW * 4), 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, (unsigned)((size_t)Cin * Cout * 48), 0x00020000);

  // staging plan: halo positions tid, tid+256, tid+512 of every channel of the stage (channel = scalar offset)
  unsigned goff[3];
  int loff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = tid + 256 * i;
    const int r = p / S2_COLS, c = p - r * S2_COLS;
    const int iy = 2 * oy0 - 1 + r, ix = 2 * ox0 - 1 + c;
    const bool real = p < S2_POS, inside = real && iy >= 0 && iy < H && ix >= 0 && ix < W;
    loff[i] = real ? r * S2_PITCH + c : S2_COLS;                  // idle slots: a pad column nobody reads
    goff[i] = inside ? 4u * (unsigned)(iy * W + ix) : 0x80000000u;  // outside the image: beyond the descriptor -> 0
  }
  float hold[S2_KC][3];
  const unsigned plane_bytes = (unsigned)(HW * 4);
  auto fetch = [&](int s) {
#pragma unroll
    for (int cl = 0; cl < S2_KC; ++cl) {
      const unsigned so = (unsigned)(s * S2_KC + cl) * plane_bytes;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        hold[cl][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, goff[i], so, 0));
    }
  };
  auto stash = [&](float* buf) {
#pragma unroll
    for (int cl = 0; cl < S2_KC; ++cl)
#pragma unroll
      for (int i = 0; i < 3; ++i) buf[cl * S2_PLANE + loff[i]] = hold[cl][i];
  };

  // weights: [channel pair][32-channel block][lane][12] (9 taps + pad), 48 B per lane, block and pair
  const unsigned wstride = (unsigned)Cout * 96;                                              // bytes per channel pair
  const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane((cg * 4 + cw * 2) * 3072);  // first block of this wave
  const unsigned wlane = lane * 48;
  f32x4 a[2][2][3];                                    // [ring slot][channel block][3 x 4 taps]
  auto fetch_w = [&](int slot, int pair) {
    const unsigned so = wbase + (unsigned)pair * wstride;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        a[slot][cb][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wsrc, wlane + cb * 3072 + 16 * q, so, 0));
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;

  const int S = Cin / S2_KC, P = Cin / 2;
  const float* const lbase = lds + k * S2_PLANE + 2 * (n >> 3) * S2_PITCH + 2 * (8 * pw + (n & 7));
  // window of this lane's pixel in the two pixel blocks (rows 4 pb + n/8): 3 rows x 4 floats (3 used), one pair ahead
  f32x4 win[2][2][3];
  auto read_window = [&](int slot, const float* plane) {
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const float* p = plane + (8 * pb + u) * S2_PITCH;
        const float2 lo = *reinterpret_cast<const float2*>(p);
        const float2 hi = *reinterpret_cast<const float2*>(p + 2);
        win[slot][pb][u] = f32x4{lo.x, lo.y, hi.x, hi.y};
      }
  };

  // issue order of the prologue = issue order of one loop iteration (see winograd3x3.hip)
  fetch(0);
  stash(lds);
  fetch_w(0, 0);
  fetch(S > 1 ? 1 : 0);
  fetch_w(1, P > 1 ? 1 : 0);
  __syncthreads();
  read_window(0, lbase);

  // Three staging buffers, as in winograd3x3.hip: stage s+1 is written during the first pair of iteration s, published by
  // the barrier before the third, first read (window prefetch of its first pair) during the fourth.
  auto stage = [&](int s, auto BC, auto BN) {
    constexpr int b_cur = decltype(BC)::value, b_nxt = decltype(BN)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == 2) __syncthreads();
      if (j == 0) stash(lds + b_nxt * S2_STAGE);
      if (j == 1) fetch(s + 2 < S ? s + 2 : S - 1);
      read_window((j + 1) & 1, lbase + (j < 3 ? b_cur * S2_STAGE + 2 * (j + 1) * S2_PLANE : b_nxt * S2_STAGE));
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j & 1][cb][t >> 2][t & 3], win[j & 1][pb][t / 3][t % 3],
                                                               acc[cb][pb], 0, 0, 0);
      const int pn = 4 * s + j + 2;
      fetch_w(j & 1, pn < P ? pn : P - 1);
      // one memory instruction after each MFMA: the window reads first, then LDS writes / buffer loads
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 30; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x210, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using B2 = std::integral_constant<int, 2>;
  int s = 0;
  for (; s + 3 <= S; s += 3) {
    stage(s, B0{}, B1{});
    stage(s + 1, B1{}, B2{});
    stage(s + 2, B2{}, B0{});
  }
  if (s < S) {
    stage(s, B0{}, B1{});
    if (s + 1 < S) stage(s + 1, B1{}, B2{});
  }

  // BN / ReLU and store: accumulator register r of block (cb, pb) = channel (r&3) + 8 (r>>2) + 4 k of the block,
  // pixel (4 pb + n/8, 8 pw + n%8) of the output block
  const int ox = ox0 + 8 * pw + (n & 7);
  if (ox >= Wo) return;
  const size_t HWo = (size_t)Ho * Wo;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int ch0 = cg * 128 + cw * 64 + cb * 32 + 4 * k;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int oy = oy0 + 4 * pb + (n >> 3);
      if (oy >= Ho) continue;
      float* yp = y + ((size_t)img * Cout + ch0) * HWo + (size_t)oy * Wo + ox;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = (r & 3) + 8 * (r >> 2);
        float o = fmaf(acc[cb][pb][r], scale[ch0 + co], shift[ch0 + co]);
        if (RELU) o = fmaxf(o, 0.f);
        yp[(size_t)co * HWo] = o;
      }
    }
  }
}

__global__ void conv3x3s2_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int cin = idx % Cin, cout = idx / Cin;
  const float* g = w + (size_t)idx * 9;
  const int pair = cin >> 1, k = cin & 1, cb = cout >> 5, m = cout & 31;
  float* dst = wp + (((size_t)pair * (Cout / 32) + cb) * 64 + k * 32 + m) * 12;
  for (int t = 0; t < 9; ++t) dst[t] = g[t];
  for (int t = 9; t < 12; ++t) dst[t] = 0.f;
}

}  // namespace
}  // namespace pvsg

extern "C" int pvsg_conv3x3s2_pack(const float* weight, float* w_packed, int Cin, int Cout, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && w_packed, "conv3x3s2_pack: null pointer argument");
  PVSG_REQUIRE(Cin > 0 && Cout > 0, "conv3x3s2_pack: bad shape");
  if (Cin % S2_KC || Cout % 128)
    return set_err(PVSG_ERR_UNSUPPORTED, "conv3x3s2: built for Cin %% 8 == 0 and Cout %% 128 == 0 (got %d %d)", Cin, Cout);
  const int total = Cin * Cout;
  hipLaunchKernelGGL(conv3x3s2_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), weight,
                     w_packed, Cin, Cout);
  PVSG_LAUNCH_CHECK("conv3x3s2_pack");
  return PVSG_OK;
}

extern "C" int pvsg_conv3x3s2_affine(const float* x, const float* w_packed, const float* scale, const float* shift, float* y,
                                     int N, int Cin, int Cout, int H, int W, int relu, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && w_packed && scale && shift && y, "conv3x3s2_affine: null pointer argument");
  PVSG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3s2_affine: bad shape");
  if (Cin % S2_KC || Cout % 128 || (long long)Cin * H * W >= (1LL << 29) || (long long)Cin * Cout >= (1LL << 25))
    return set_err(PVSG_ERR_UNSUPPORTED,
                   "conv3x3s2_affine: built for Cin %% 8 == 0, Cout %% 128 == 0, Cin*H*W < 2^29 (got Cin=%d Cout=%d H=%d W=%d)", Cin,
                   Cout, H, W);
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(w_packed) & 15u), "conv3x3s2_affine: w_packed must be 16-byte aligned");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int TY = (Ho + 7) / 8, TX = (Wo + 15) / 16;
  const long long blocks = (long long)N * TY * TX * (Cout / 128);
  PVSG_REQUIRE(blocks < (1LL << 31), "conv3x3s2_affine: too many blocks");
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (relu)
    hipLaunchKernelGGL((conv3x3s2_kernel<true>), grid, block, 0, st, x, w_packed, scale, shift, y, N, Cin, Cout, H, W, Ho, Wo, TY, TX);
  else
    hipLaunchKernelGGL((conv3x3s2_kernel<false>), grid, block, 0, st, x, w_packed, scale, shift, y, N, Cin, Cout, H, W, Ho, Wo, TY, TX);
  PVSG_LAUNCH_CHECK("conv3x3s2_affine");
  return PVSG_OK;
}
