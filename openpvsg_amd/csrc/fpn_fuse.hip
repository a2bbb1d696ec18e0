// HBM-streaming glue around the library convolutions (each replaces 2-4 separate torch passes):
//
//  fpn_merge_up2x      [3P] mmdet MSDeformAttnPixelDecoder.forward, FPN step:
//                        y = GroupNorm(lateral_conv(C2)) + F.interpolate(top, size=2x, bilinear, align_corners=False)
//                      GroupNorm enters as per-(image, channel) scale/shift; the x2 resize is evaluated in place
//                      (torch's upsample kernel alone ran at 0.6 TB/s): one read of lateral, a quarter-size read of
//                      top, one write.
//  stem_bn_relu_pool   [3P] mmdet ResNet stem: BatchNorm(eval) -> ReLU -> MaxPool2d(3, stride 2, pad 1) in one pass
//                      (reads the 7x7 conv output once, writes the quarter-size map).
//  nchw_to_tokens      pixel-decoder hand-off: GroupNorm(input_conv(C_l)).flatten(2).transpose(1,2) written straight
//                      into its slice of the (B, S, C) token tensor (LDS-tiled transpose; torch's cat of strided
//                      views ran at 0.6 TB/s).
#include "common.h"

namespace pvsg {

// thread -> top cell (i, 2j..2j+1): writes the 2x4 output patch rows 2i,2i+1 / cols 4j..4j+3
__global__ __launch_bounds__(256) void fpn_merge_up2x_kernel(const float* __restrict__ lat,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ top, float* __restrict__ out,
                                                            int h, int w) {
  const long long plane = blockIdx.x;
  const float sc = scale ? scale[plane] : 1.f, sh = shift ? shift[plane] : 0.f;
  const float* tp = top + plane * (long long)h * w;
  const float* lp = lat + plane * (long long)h * w * 4;
  float* op = out + plane * (long long)h * w * 4;
  const int w2 = w >> 1, W = 2 * w;
  const int cells = h * w2;
  for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < cells; t += gridDim.y * blockDim.x) {
    const int i = t / w2, j = t - i * w2;
    const int im = max(i - 1, 0), ip = min(i + 1, h - 1);
    const int c0 = max(2 * j - 1, 0), c1 = 2 * j, c2 = 2 * j + 1, c3 = min(2 * j + 2, w - 1);
    float r[3][4];
    const int rows[3] = {im, i, ip};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float* q = tp + (long long)rows[k] * w;
      r[k][0] = q[c0]; r[k][1] = q[c1]; r[k][2] = q[c2]; r[k][3] = q[c3];
    }
    // torch upsample_bilinear2d, scale 0.5, align_corners=False: dst 2a -> (a-1: .25, a: .75) except dst 0 -> (0: 1);
    // dst 2a+1 -> (a: .75, a+1: .25) with a+1 clamped to the last cell
    const float wx0a = (j == 0) ? 1.f : 0.25f, wx0b = (j == 0) ? 0.f : 0.75f;   // out col 4j   : taps c0, c1  (c0==c1==0 at j=0)
    float hx[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      hx[k][0] = (j == 0) ? (1.f * r[k][1] + 0.f * r[k][2]) : (wx0a * r[k][0] + wx0b * r[k][1]);
      hx[k][1] = 0.75f * r[k][1] + 0.25f * r[k][2];
      hx[k][2] = 0.25f * r[k][1] + 0.75f * r[k][2];
      hx[k][3] = 0.75f * r[k][2] + 0.25f * r[k][3];
    }
    float4 o0, o1;
    if (i == 0) {
      o0 = make_float4(1.f * hx[1][0] + 0.f * hx[2][0], 1.f * hx[1][1] + 0.f * hx[2][1], 1.f * hx[1][2] + 0.f * hx[2][2],
                       1.f * hx[1][3] + 0.f * hx[2][3]);
    } else {
      o0 = make_float4(0.25f * hx[0][0] + 0.75f * hx[1][0], 0.25f * hx[0][1] + 0.75f * hx[1][1],
                       0.25f * hx[0][2] + 0.75f * hx[1][2], 0.25f * hx[0][3] + 0.75f * hx[1][3]);
    }
    o1 = make_float4(0.75f * hx[1][0] + 0.25f * hx[2][0], 0.75f * hx[1][1] + 0.25f * hx[2][1],
                     0.75f * hx[1][2] + 0.25f * hx[2][2], 0.75f * hx[1][3] + 0.25f * hx[2][3]);
    const long long b0 = (long long)(2 * i) * W + 4 * j, b1 = b0 + W;
    const float4 l0 = ld4_stream(lp + b0), l1 = ld4_stream(lp + b1);
    o0.x += l0.x * sc + sh; o0.y += l0.y * sc + sh; o0.z += l0.z * sc + sh; o0.w += l0.w * sc + sh;
    o1.x += l1.x * sc + sh; o1.y += l1.y * sc + sh; o1.z += l1.z * sc + sh; o1.w += l1.w * sc + sh;
    st4(op + b0, o0);
    st4(op + b1, o1);
  }
}

// thread -> output cells (y, 2k), (y, 2k+1): input cols 4k-1 .. 4k+3, rows 2y-1 .. 2y+1
__global__ __launch_bounds__(256) void stem_bn_relu_pool_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* __restrict__ out,
                                                               int C, int H, int W, int Ho, int Wo, bool vec) {
  const long long plane = blockIdx.x;
  const int c = (int)(plane % C);
  const float sc = scale[c], sh = shift[c];
  const float* xp = x + plane * (long long)H * W;
  float* op = out + plane * (long long)Ho * Wo;
  const int wp = (Wo + 1) >> 1;
  const int cells = Ho * wp;
  for (int t = blockIdx.y * blockDim.x + threadIdx.x; t < cells; t += gridDim.y * blockDim.x) {
    const int y = t / wp, k = t - y * wp;
    float m0 = 0.f, m1 = 0.f;            // ReLU output is >= 0 and every window holds a valid cell
    const int x0 = 4 * k;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = 2 * y + dy;
      if (yy < 0 || yy >= H) continue;
      const float* q = xp + (long long)yy * W;
      float v[5];
      v[0] = (x0 > 0) ? q[x0 - 1] * sc + sh : 0.f;
      if (vec && x0 + 3 < W) {
        const float4 f = ld4_stream(q + x0);
        v[1] = f.x * sc + sh; v[2] = f.y * sc + sh; v[3] = f.z * sc + sh; v[4] = f.w * sc + sh;
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) v[1 + d] = (x0 + d < W) ? q[x0 + d] * sc + sh : 0.f;
      }
      m0 = fmaxf(m0, fmaxf(v[0], fmaxf(v[1], v[2])));
      m1 = fmaxf(m1, fmaxf(v[2], fmaxf(v[3], v[4])));
    }
    op[(long long)y * Wo + 2 * k] = m0;
    if (2 * k + 1 < Wo) op[(long long)y * Wo + 2 * k + 1] = m1;
  }
}

// src (B, C, HW) -> dst[b, start + p, c] = src[b, c, p] * scale[b*C+c] + shift[b*C+c];  32x32 tiles through LDS
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const float* __restrict__ src,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ dst,
                                                            int C, int HW, long long dst_batch_stride) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* sp = src + ((long long)b * C) * HW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, p = p0 + tx;
    if (c < C && p < HW) {
      float v = sp[(long long)c * HW + p];
      if (scale) v = v * scale[(long long)b * C + c] + shift[(long long)b * C + c];
      tile[ty + 8 * r][tx] = v;
    }
  }
  __syncthreads();
  float* dp = dst + (long long)b * dst_batch_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = p0 + ty + 8 * r, c = c0 + tx;
    if (c < C && p < HW) dp[(long long)p * C + c] = tile[tx][ty + 8 * r];
  }
}

// inverse hand-off: dst[b, c, p] = src[b, start + p, c]  (encoder memory of one level -> NCHW for the FPN branch)
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int C, int HW, long long src_batch_stride) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* sp = src + (long long)b * src_batch_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = p0 + ty + 8 * r, c = c0 + tx;
    if (c < C && p < HW) tile[ty + 8 * r][tx] = sp[(long long)p * C + c];
  }
  __syncthreads();
  float* dp = dst + ((long long)b * C) * HW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + ty + 8 * r, p = p0 + tx;
    if (c < C && p < HW) dp[(long long)c * HW + p] = tile[tx][ty + 8 * r];
  }
}

// Decoder key / value inputs of one level (mask2former_head.py:421-436 / mask2former_video_head.py:392-410):
//   value = memory_l.flatten(2).permute(..) + level_embed[l],  key = value + positional_encoding_l
// straight from the encoder's token tensor (frames x S x C, level l = rows start .. start+hw of every frame): one pass
// that writes both tensors (the torch form is a strided gather copy + two adds = 7 tensor passes instead of 4).
__global__ __launch_bounds__(256) void decoder_kv_inputs_kernel(const float* __restrict__ tokens,
                                                               const float* __restrict__ level_embed,
                                                               const float* __restrict__ pe, float* __restrict__ v_out,
                                                               float* __restrict__ k_out, long long rows, int hw,
                                                               long long frame_stride, long long pe_rows) {
  const int lane = threadIdx.x & 63;
  const float4 le = ld4(level_embed + lane * 4);
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
    const long long t = r / hw, p = r - t * hw;
    float4 x = ld4_stream(tokens + t * frame_stride + p * 256 + lane * 4);
    x.x += le.x; x.y += le.y; x.z += le.z; x.w += le.w;
    st4(v_out + r * 256 + lane * 4, x);
    const float4 e = ld4(pe + (r % pe_rows) * 256 + lane * 4);
    x.x += e.x; x.y += e.y; x.z += e.z; x.w += e.w;
    st4(k_out + r * 256 + lane * 4, x);
  }
}


// GroupNorm statistics -> per-(image, channel) scale / shift, so that GroupNorm(x) == x * scale + shift can ride in a
// consumer (fpn_merge_up2x, nchw_to_tokens, the mask-feature convolution's operand staging).  [3P] torch.nn.GroupNorm as
// used by mmdet ConvModule(norm_cfg=GN): biased variance over the group's channels x pixels.  Two launches, fixed
// reduction order (bitwise reproducible): partial (sum, sum of squares) of 64 chunks per (image, group) -- f32 per
// thread over <= a few hundred elements, f64 from there on -- then one thread per (image, group).
constexpr int GN_CHUNKS = 64;

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long long n_per_group) {
  const long long bg = blockIdx.x;
  const int chunk = blockIdx.y;
  const float* p = x + bg * n_per_group;
  const long long n4 = n_per_group >> 2;
  const long long per = (n4 + GN_CHUNKS - 1) / GN_CHUNKS;
  const long long lo = chunk * per, hi = (lo + per < n4) ? lo + per : n4;
  double s = 0.0, q = 0.0;
  float fs = 0.f, fq = 0.f;
  int cnt = 0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float4 v = ld4_stream(p + 4 * i);
    fs += (v.x + v.y) + (v.z + v.w);
    fq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    if (++cnt == 64) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }
  }
  s += fs; q += fq;
  if (chunk == GN_CHUNKS - 1)                                    // the < 4 trailing elements
    for (long long i = 4 * n4 + threadIdx.x; i < n_per_group; i += 256) { const double v = p[i]; s += v; q += v * v; }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[(bg * GN_CHUNKS + chunk) * 2] = red[0][0];
    part[(bg * GN_CHUNKS + chunk) * 2 + 1] = red[1][0];
  }
}

__global__ void gn_finish_kernel(const double* __restrict__ part, const float* __restrict__ weight, const float* __restrict__ bias,
                                 float* __restrict__ scale, float* __restrict__ shift, int B, int C, int G, long long n_per_group,
                                 float eps, int nchunks = GN_CHUNKS) {
  const int bg = blockIdx.x * blockDim.x + threadIdx.x;
  if (bg >= B * G) return;
  double s = 0.0, q = 0.0;
  for (int c = 0; c < nchunks; ++c) {
    s += part[((long long)bg * nchunks + c) * 2];
    q += part[((long long)bg * nchunks + c) * 2 + 1];
  }
  const double mean = s / (double)n_per_group;
  double var = q / (double)n_per_group - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int b = bg / G, g = bg - b * G, cpg = C / G;
  for (int i = 0; i < cpg; ++i) {
    const int ch = g * cpg + i;
    const float sc = rstd * (weight ? weight[ch] : 1.f);
    scale[b * C + ch] = sc;
    shift[b * C + ch] = (bias ? bias[ch] : 0.f) - (float)mean * sc;
  }
}

// The same with one WAVE per (image, group): lane l sums chunks l, l + 64, ... in f64, then a fixed-order butterfly -- for the many
// chunks a convolution's epilogue leaves behind (920 per group for the 184 x 320 lateral convolution: 61 us with one thread per group).
__global__ __launch_bounds__(64) void gn_finish_wave_kernel(const double* __restrict__ part, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float* __restrict__ scale,
                                                            float* __restrict__ shift, int C, int G, long long n_per_group, float eps,
                                                            int nchunks) {
  const int bg = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int c = lane; c < nchunks; c += 64) {
    s += part[((long long)bg * nchunks + c) * 2];
    q += part[((long long)bg * nchunks + c) * 2 + 1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
  const double mean = s / (double)n_per_group;
  double var = q / (double)n_per_group - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int b = bg / G, g = bg - b * G, cpg = C / G;
  if (lane < cpg) {
    const int ch = g * cpg + lane;
    const float sc = rstd * (weight ? weight[ch] : 1.f);
    scale[b * C + ch] = sc;
    shift[b * C + ch] = (bias ? bias[ch] : 0.f) - (float)mean * sc;
  }
}

}  // namespace pvsg

extern "C" int pvsg_fpn_merge_up2x(const float* lateral, const float* scale, const float* shift, const float* top,
                                   float* out, long long planes, int h, int w, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(lateral && top && out, "fpn_merge_up2x: null pointer argument");
  PVSG_REQUIRE((scale == nullptr) == (shift == nullptr), "fpn_merge_up2x: scale and shift come together");
  PVSG_REQUIRE(planes > 0 && planes < (1LL << 31) && h > 0 && w > 0, "fpn_merge_up2x: bad shape");
  PVSG_REQUIRE(!(w & 1), "fpn_merge_up2x: the low-resolution width must be even (got %d)", w);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(lateral) | reinterpret_cast<uintptr_t>(out)) & 15u),
               "fpn_merge_up2x: 16-byte alignment required");
  int by = (h * (w / 2) + 255) / 256;
  if (by > 32) by = 32;
  hipLaunchKernelGGL(fpn_merge_up2x_kernel, dim3((unsigned)planes, by), dim3(256), 0, stream, lateral, scale, shift, top,
                     out, h, w);
  PVSG_LAUNCH_CHECK("fpn_merge_up2x");
  return PVSG_OK;
}

extern "C" int pvsg_stem_bn_relu_pool(const float* x, const float* scale, const float* shift, float* out,
                                      long long planes, int C, int H, int W, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && scale && shift && out, "stem_bn_relu_pool: null pointer argument");
  PVSG_REQUIRE(planes > 0 && planes < (1LL << 31) && C > 0 && planes % C == 0 && H > 0 && W > 0,
               "stem_bn_relu_pool: bad shape");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;      // floor((n + 2 - 3) / 2) + 1
  int by = (Ho * ((Wo + 1) / 2) + 255) / 256;
  if (by > 64) by = 64;
  const bool vec = !(W & 3) && !(reinterpret_cast<uintptr_t>(x) & 15u);
  hipLaunchKernelGGL(stem_bn_relu_pool_kernel, dim3((unsigned)planes, by), dim3(256), 0, stream, x, scale, shift, out, C,
                     H, W, Ho, Wo, vec);
  PVSG_LAUNCH_CHECK("stem_bn_relu_pool");
  return PVSG_OK;
}

extern "C" int pvsg_nchw_to_tokens(const float* src, const float* scale, const float* shift, float* dst, int B, int C,
                                   int HW, long long dst_batch_stride, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(src && dst, "nchw_to_tokens: null pointer argument");
  PVSG_REQUIRE((scale == nullptr) == (shift == nullptr), "nchw_to_tokens: scale and shift come together");
  PVSG_REQUIRE(B > 0 && B < 65536 && C > 0 && HW > 0 && dst_batch_stride >= (long long)HW * C, "nchw_to_tokens: bad shape");
  PVSG_REQUIRE((C + 31) / 32 < 65536, "nchw_to_tokens: too many channels");
  hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, stream, src, scale,
                     shift, dst, C, HW, dst_batch_stride);
  PVSG_LAUNCH_CHECK("nchw_to_tokens");
  return PVSG_OK;
}

extern "C" int pvsg_tokens_to_nchw(const float* src, float* dst, int B, int C, int HW, long long src_batch_stride,
                                   hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(src && dst, "tokens_to_nchw: null pointer argument");
  PVSG_REQUIRE(B > 0 && B < 65536 && C > 0 && HW > 0 && src_batch_stride >= (long long)HW * C, "tokens_to_nchw: bad shape");
  PVSG_REQUIRE((C + 31) / 32 < 65536, "tokens_to_nchw: too many channels");
  hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, stream, src, dst, C, HW,
                     src_batch_stride);
  PVSG_LAUNCH_CHECK("tokens_to_nchw");
  return PVSG_OK;
}

extern "C" int pvsg_decoder_kv_inputs(const float* tokens, const float* level_embed, const float* pos_enc, float* v_out,
                                      float* k_out, long long frames, int hw, int C, long long frame_stride,
                                      long long pe_rows, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(tokens && level_embed && pos_enc && v_out && k_out, "decoder_kv_inputs: null pointer argument");
  PVSG_REQUIRE(frames > 0 && hw > 0 && pe_rows > 0 && frame_stride >= (long long)hw * C, "decoder_kv_inputs: bad shape");
  if (C != 256) return set_err(PVSG_ERR_UNSUPPORTED, "decoder_kv_inputs: built for 256 channels (got %d)", C);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(tokens) | reinterpret_cast<uintptr_t>(level_embed) |
                  reinterpret_cast<uintptr_t>(pos_enc) | reinterpret_cast<uintptr_t>(v_out) |
                  reinterpret_cast<uintptr_t>(k_out)) & 15u) && !(frame_stride & 3),
               "decoder_kv_inputs: 16-byte alignment required");
  const long long rows = frames * hw;
  long long nb = (rows + 3) / 4;
  if (nb > 256 * 16) nb = 256 * 16;
  hipLaunchKernelGGL(decoder_kv_inputs_kernel, dim3((unsigned)nb), dim3(256), 0, stream, tokens, level_embed, pos_enc,
                     v_out, k_out, rows, hw, frame_stride, pe_rows);
  PVSG_LAUNCH_CHECK("decoder_kv_inputs");
  return PVSG_OK;
}

// The second half of pvsg_group_norm_affine on partial sums somebody else produced: pvsg_conv1x1_f16x2_stats writes one (sum, sum
// of squares) pair per (image, group, chunk) from its epilogue, so the statistics pass over the convolution's output never runs.
extern "C" int pvsg_group_norm_finish(const double* partials, int nchunks, const float* weight, const float* bias, float* scale,
                                      float* shift, int B, int C, int G, long long HW, float eps, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(partials && scale && shift, "group_norm_finish: null pointer argument");
  PVSG_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0 && nchunks > 0, "group_norm_finish: bad shape");
  if (C / G <= 64)
    hipLaunchKernelGGL(gn_finish_wave_kernel, dim3((unsigned)(B * G)), dim3(64), 0, stream, partials, weight, bias, scale, shift, C, G,
                       (long long)(C / G) * HW, eps, nchunks);
  else
    hipLaunchKernelGGL(gn_finish_kernel, dim3((unsigned)((B * G + 63) / 64)), dim3(64), 0, stream, partials, weight, bias, scale, shift,
                       B, C, G, (long long)(C / G) * HW, eps, nchunks);
  PVSG_LAUNCH_CHECK("group_norm_finish");
  return PVSG_OK;
}

extern "C" int pvsg_group_norm_affine(const float* x, const float* weight, const float* bias, double* workspace, float* scale,
                                      float* shift, int B, int C, int G, long long HW, float eps, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && workspace && scale && shift, "group_norm_affine: null pointer argument");
  PVSG_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && HW > 0, "group_norm_affine: bad shape");
  const long long npg = (long long)(C / G) * HW;
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(x) & 15u) && !(npg & 3), "group_norm_affine: 16-byte aligned groups required");
  hipLaunchKernelGGL(gn_partial_kernel, dim3((unsigned)(B * G), GN_CHUNKS), dim3(256), 0, stream, x, workspace, npg);
  PVSG_LAUNCH_CHECK("group_norm_affine");
  hipLaunchKernelGGL(gn_finish_kernel, dim3((unsigned)((B * G + 63) / 64)), dim3(64), 0, stream, workspace, weight, bias, scale, shift,
                     B, C, G, npg, eps);
  PVSG_LAUNCH_CHECK("group_norm_affine");
  return PVSG_OK;
}
