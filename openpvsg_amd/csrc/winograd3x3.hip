// 3x3 / stride 1 / pad 1 convolution in NCHW as Winograd F(2x2, 3x3) on the f32 matrix cores, with the frozen-BN
// affine + ReLU of the ResNet bottleneck (or nothing, for the pixel decoder's FPN output convolution) in the epilogue.
//
// Replaces, on the north-star path, the library call behind
//   [3P] mmdet ResNet Bottleneck.conv2 -> bn2 -> relu      (configs/mask2former/*: backbone=dict(type='ResNet', depth=50))
//   [3P] MSDeformAttnPixelDecoder.output_convs[i].conv     (models/mask2former/... pixel_decoder, 3x3 ConvModule + GN)
// which MIOpen serves with its VALU Winograd (miopenSp3AsmConv ... f2x3_stride1, 53 TFLOP/s of real multiplies on this
// shape set = 34 % of the f32 rate).  Same arithmetic class as that kernel (F(2,3) transforms, f32 accumulation), so the
// parity tolerance of the convolution stack is unchanged.
//
// Mapping.  Y = A^T [ (G g G^T) . (B^T d B) ] A : for each of the 16 transform positions xi an independent GEMM
//   M_xi[cout, tile] = sum_cin U_xi[cout, cin] V_xi[cin, tile]
// runs on v_mfma_f32_32x32x2_f32 (rows = 32 output channels, columns = 32 tiles, K = 2 input channels).  The operand
// and accumulator layouts of that instruction are the same for every xi, so BOTH transforms are lane-local:
//   * the lane that supplies B[k][j] for (channel k, tile j) loads that tile's 4x4 input patch of that channel, applies
//     B^T d B in registers (32 adds) and owns the 16 V_xi values -- one per MFMA of the K-step;
//   * accumulator register r of the 16 MFMAs holds M_xi[cout(r), tile(lane)] for all 16 xi in the SAME lane, so
//     A^T M A (24 adds per output channel) and the BN/ReLU epilogue need no exchange at all.
// A wave owns 16 xi x (32 couts x 32 tiles) = 256 accumulator registers, i.e. one wave per SIMD; the four waves of a
// workgroup cover 64 output channels x 64 tiles (a 16 x 16 output block, 8 x 8 tiles).  Per K-step a lane issues 16 MFMAs
// (1024 cycles) against 4 x global_load_dwordx4 (its 16 U values, pre-packed lane-major by pvsg_conv3x3_winograd_pack),
// 8 x ds_read_b64 (its patch) and 32 VALU adds.
//
// Input staging: 8 channels x 18 x 18 halo patches per stage through LDS (three buffers, borders zero-filled at
// staging time, one barrier per stage); patches are read one K-step, the U operands (global -> register) four K-steps
// ahead of their use.
// Launch order: output-channel group major, so that an XCD's L2 holds the 16 x Cin x 64 slice of U it is using
// (1 MB at Cin = 256) while the input streams.
#include "common.h"

#include <type_traits>

namespace pvsg {
namespace {

constexpr int WINO_KC = 8;                          // input channels per LDS stage (4 K-steps of 2)
constexpr int WINO_ROWS = 18;                       // 16 output rows/cols + halo
constexpr int WINO_PITCH = 24;                      // LDS row pitch in floats: the 4 tile rows of a wave land on disjoint bank groups
constexpr int WINO_PLANE = WINO_ROWS * WINO_PITCH;  // floats per staged channel
constexpr int WINO_STAGE = WINO_KC * WINO_PLANE;
constexpr int WINO_ELEMS = WINO_KC * WINO_ROWS * WINO_ROWS;
constexpr int WINO_LD = (WINO_ELEMS + 255) / 256;   // staged elements per thread

template <bool AFFINE, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void winograd_f2x3_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ scale,
                          const float* __restrict__ shift, float* __restrict__ y, int N, int Cin, int Cout, int H, int W,
                          int TY, int TX) {
  __shared__ __attribute__((aligned(16))) float lds[3 * WINO_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cbw = wave >> 1, tg = wave & 1;                       // 32-channel block / 32-tile group of this wave
  const int k = lane >> 5, tx = lane & 7, ty = tg * 4 + ((lane & 31) >> 3);
  unsigned logical = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int bx = logical % TX;
  logical /= TX;
  const int by = logical % TY;
  logical /= TY;
  const int n = logical % N, cg = logical / N;
  const int oy0 = by * 16, ox0 = bx * 16;
  const size_t HW = (size_t)H * W;
  const float* xn = x + (size_t)n * Cin * HW;

  // Every memory operand of the loop is a buffer load with a wave-uniform scalar offset: no per-load address
  // arithmetic on the VALU (which shares the SIMD with the f32 MFMAs), and out-of-image halo elements are fetched at an
  // offset beyond the descriptor's range, i.e. read as 0 by the bounds check -- no select either.
  const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (unsigned)(Cin * HW * 4), 0x00020000);
  const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (unsigned)((size_t)Cin * Cout * 64), 0x00020000);

  // staging plan of this thread: WINO_LD elements of the (8 x 18 x 18) halo block, the same for every stage
  unsigned goff[WINO_LD];
  int loff[WINO_LD];
#pragma unroll
  for (int i = 0; i < WINO_LD; ++i) {
    const int e = tid + 256 * i;
    const int cl = e / (WINO_ROWS * WINO_ROWS), rem = e - cl * (WINO_ROWS * WINO_ROWS);
    const int r = rem / WINO_ROWS, c = rem - r * WINO_ROWS;
    const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
    const bool real = e < WINO_ELEMS, inside = real && iy >= 0 && iy < H && ix >= 0 && ix < W;
    loff[i] = real ? cl * WINO_PLANE + r * WINO_PITCH + c : WINO_ROWS;      // surplus threads: a pad column nobody reads
    goff[i] = inside ? 4u * (unsigned)(cl * (int)HW + iy * W + ix) : 0x80000000u;
  }
  float hold[WINO_LD];
  const unsigned stage_bytes = (unsigned)(WINO_KC * HW * 4);
  auto fetch = [&](int s) {
    const unsigned so = (unsigned)s * stage_bytes;
#pragma unroll
    for (int i = 0; i < WINO_LD; ++i)
      hold[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xsrc, goff[i], so, 0));
  };
  auto stash = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < WINO_LD; ++i) buf[loff[i]] = hold[i];
  };

  // U operands: [cin pair][32-channel block][lane][16 xi], 64 B per lane and K-step
  const unsigned ustride = (unsigned)Cout * 128;                                           // bytes per cin pair
  const unsigned ubase = (unsigned)__builtin_amdgcn_readfirstlane((cg * 2 + cbw) * 4096);  // this wave's channel block
  const unsigned ulane = lane * 64;
  f32x4 a[4][4];
  auto fetch_u = [&](int j, int pair) {
    const unsigned so = ubase + (unsigned)pair * ustride;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      a[j][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, ulane + 16 * q, so, 0));
  };

  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int S = Cin / WINO_KC;
  const float* const lbase = lds + k * WINO_PLANE + 2 * ty * WINO_PITCH + 2 * tx;
  // The f32 MFMAs and the VALU share the SIMD.  Measured on the FPN output convolution (32 x 256 x 184 x 320):
  //   all non-MFMA instructions in front of each K-step's MFMA chain            10.8 ms
  //   memory instructions (LDS, buffer loads) handed out one per MFMA            9.0 ms   <- this schedule
  //   ... and the transform's adds as well, two per MFMA                        10.5 ms   (VALU between MFMAs costs a
  //                                                                                        pipeline switch each time)
  //   packed adds (v_pk_add_f32 with half selects) instead of scalar ones       no change
  // So each K-step is one VALU block (the 32 adds of B^T d B for this K-step, on the patch read during the previous
  // one), then the MFMA chain with one LDS / buffer instruction after each MFMA.
  float d[2][16];
  auto read_patch = [&](int slot, const float* plane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 lo = *reinterpret_cast<const float2*>(plane + r * WINO_PITCH);
      const float2 hi = *reinterpret_cast<const float2*>(plane + r * WINO_PITCH + 2);
      d[slot][r * 4 + 0] = lo.x; d[slot][r * 4 + 1] = lo.y; d[slot][r * 4 + 2] = hi.x; d[slot][r * 4 + 3] = hi.y;
    }
  };
  // same issue order as one loop iteration (staging loads, then the four U loads), so that the load counts the
  // compiler assumes at the loop head are those of the steady state
  fetch(0);
  stash(lds);
  fetch(S > 1 ? 1 : 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) fetch_u(j, j);
  __syncthreads();
  read_patch(0, lbase);

  // Three staging buffers: stage s+1 is written at the top of iteration s (fetched during iteration s-1), published by
  // the barrier in the middle of iteration s and first read by the last K-step of iteration s (the patch prefetch of
  // the next stage's first K-step); its buffer was last read in iteration s-2, which every wave has left by then.
  // Past the last stage the body re-stages / re-loads the last one into buffers and registers nobody reads: no branches
  // inside it, so the compiler knows how many loads are in flight at every wait.  The buffer index is a compile-time
  // constant of each copy of the body (LDS addresses = immediate offsets).
  auto stage = [&](int s, auto BC, auto BN) {
    constexpr int b_cur = decltype(BC)::value, b_nxt = decltype(BN)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == 2) __syncthreads();
      const float* dd = d[j & 1];
      float t[4][4], v[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        t[0][c] = dd[0 + c] - dd[8 + c];
        t[1][c] = dd[4 + c] + dd[8 + c];
        t[2][c] = dd[8 + c] - dd[4 + c];
        t[3][c] = dd[4 + c] - dd[12 + c];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r * 4 + 0] = t[r][0] - t[r][2];
        v[r * 4 + 1] = t[r][1] + t[r][2];
        v[r * 4 + 2] = t[r][2] - t[r][1];
        v[r * 4 + 3] = t[r][1] - t[r][3];
      }
      if (j == 0) {
        stash(lds + b_nxt * WINO_STAGE);
        fetch(s + 2 < S ? s + 2 : S - 1);
      }
      read_patch((j + 1) & 1, lbase + (j < 3 ? b_cur * WINO_STAGE + 2 * (j + 1) * WINO_PLANE : b_nxt * WINO_STAGE));
#pragma unroll
      for (int xi = 0; xi < 16; ++xi)
        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][xi >> 2][xi & 3], v[xi], acc[xi], 0, 0, 0);
      fetch_u(j, s + 1 < S ? 4 * (s + 1) + j : j);
      __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);         // the transform
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // the next patch first
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x210, 1, 0);        // then one buffer load / LDS write per MFMA
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

  // A^T M A per (output channel, tile), then BN / ReLU, then two float2 rows per channel
  const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
  if (ox >= W || oy >= H) return;
  const bool row1 = oy + 1 < H;
  float* yn = y + ((size_t)n * Cout + cg * 64 + cbw * 32 + 4 * k) * HW + (size_t)oy * W + ox;
  // BN scale / shift of the lane's 16 channels requested together, ahead of the store loop: fetched per element inside the
  // loop the compiler puts an s_waitcnt vmcnt(0) in front of every store and each store then also waits for its predecessor
  f32x4 sc4[4], sh4[4];
  if (AFFINE) {
    const int chb = cg * 64 + cbw * 32 + 4 * k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc4[j] = *reinterpret_cast<const f32x4*>(scale + chb + 8 * j);
      sh4[j] = *reinterpret_cast<const f32x4*>(shift + chb + 8 * j);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = (r & 3) + 8 * (r >> 2);                    // + 4 k + block base: accumulator row of register r
    float s0[4], s1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s0[c] = acc[c][r] + acc[4 + c][r] + acc[8 + c][r];
      s1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
    }
    float o00 = s0[0] + s0[1] + s0[2], o01 = s0[1] - s0[2] - s0[3];
    float o10 = s1[0] + s1[1] + s1[2], o11 = s1[1] - s1[2] - s1[3];
    if (AFFINE) {
      const float sc = sc4[r >> 2][r & 3], sh = sh4[r >> 2][r & 3];
      o00 = fmaf(o00, sc, sh); o01 = fmaf(o01, sc, sh); o10 = fmaf(o10, sc, sh); o11 = fmaf(o11, sc, sh);
    }
    if (RELU) {
      o00 = fmaxf(o00, 0.f); o01 = fmaxf(o01, 0.f); o10 = fmaxf(o10, 0.f); o11 = fmaxf(o11, 0.f);
    }
    float* p = yn + (size_t)co * HW;
    *reinterpret_cast<float2*>(p) = make_float2(o00, o01);
    if (row1) *reinterpret_cast<float2*>(p + W) = make_float2(o10, o11);
  }
}

// U = G g G^T per (cout, cin), written lane-major for the kernel above.  One-off (weights are static): f64 arithmetic.
__global__ void winograd_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int cin = idx % Cin, cout = idx / Cin;
  const float* g = w + (size_t)idx * 9;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  double t[4][3];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 3; ++c) t[r][c] = G[r][0] * g[c] + G[r][1] * g[3 + c] + G[r][2] * g[6 + c];
  const int pair = cin >> 1, k = cin & 1, cb = cout >> 5, m = cout & 31;
  float* dst = u + (((size_t)pair * (Cout / 32) + cb) * 64 + k * 32 + m) * 16;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) dst[r * 4 + c] = (float)(t[r][0] * G[c][0] + t[r][1] * G[c][1] + t[r][2] * G[c][2]);
}

}  // namespace
}  // namespace pvsg

extern "C" int pvsg_conv3x3_winograd_pack(const float* weight, float* u_packed, int Cin, int Cout, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(weight && u_packed, "conv3x3_winograd_pack: null pointer argument");
  PVSG_REQUIRE(Cin > 0 && Cout > 0, "conv3x3_winograd_pack: bad shape");
  if (Cin % WINO_KC || Cout % 64)
    return set_err(PVSG_ERR_UNSUPPORTED, "conv3x3_winograd: built for Cin %% 8 == 0 and Cout %% 64 == 0 (got %d %d)", Cin, Cout);
  const int total = Cin * Cout;
  hipLaunchKernelGGL(winograd_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), weight,
                     u_packed, Cin, Cout);
  PVSG_LAUNCH_CHECK("conv3x3_winograd_pack");
  return PVSG_OK;
}

extern "C" int pvsg_conv3x3_winograd(const float* x, const float* u_packed, const float* scale, const float* shift, float* y,
                                     int N, int Cin, int Cout, int H, int W, int relu, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && u_packed && y, "conv3x3_winograd: null pointer argument");
  PVSG_REQUIRE((scale == nullptr) == (shift == nullptr), "conv3x3_winograd: scale and shift go together");
  PVSG_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd: bad shape");
  if (Cin % WINO_KC || Cout % 64 || (W & 1) || (long long)Cin * H * W >= (1LL << 29) || (long long)Cin * Cout >= (1LL << 25))
    return set_err(PVSG_ERR_UNSUPPORTED,
                   "conv3x3_winograd: built for Cin %% 8 == 0, Cout %% 64 == 0, even W, Cin*H*W < 2^29 (got Cin=%d Cout=%d H=%d W=%d)", Cin, Cout,
                   H, W);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15u),
               "conv3x3_winograd: scale and shift must be 16-byte aligned");
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(u_packed) & 15u) | (reinterpret_cast<uintptr_t>(y) & 7u)),
               "conv3x3_winograd: u_packed must be 16-byte and y 8-byte aligned");
  const int TY = (H + 15) / 16, TX = (W + 15) / 16;
  const long long blocks = (long long)N * TY * TX * (Cout / 64);
  PVSG_REQUIRE(blocks < (1LL << 31), "conv3x3_winograd: too many blocks");
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (scale) {
    if (relu)
      hipLaunchKernelGGL((winograd_f2x3_kernel<true, true>), grid, block, 0, st, x, u_packed, scale, shift, y, N, Cin, Cout, H, W, TY, TX);
    else
      hipLaunchKernelGGL((winograd_f2x3_kernel<true, false>), grid, block, 0, st, x, u_packed, scale, shift, y, N, Cin, Cout, H, W, TY, TX);
  } else {
    if (relu)
      hipLaunchKernelGGL((winograd_f2x3_kernel<false, true>), grid, block, 0, st, x, u_packed, scale, shift, y, N, Cin, Cout, H, W, TY, TX);
    else
      hipLaunchKernelGGL((winograd_f2x3_kernel<false, false>), grid, block, 0, st, x, u_packed, scale, shift, y, N, Cin, Cout, H, W, TY, TX);
  }
  PVSG_LAUNCH_CHECK("conv3x3_winograd");
  return PVSG_OK;
}
