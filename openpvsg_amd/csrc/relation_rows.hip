// The relation head's transformer encoders and temporal models as fused row kernels (rows a10 / a13 of the scope table).
//
// Replaces the library calls behind
//   models/relation_head/base.py:26-40        ObjectEncoder = nn.TransformerEncoder(2 x TransformerEncoderLayer(256, 8 heads, ff 512)),
//                                              batch_first=False on feats [N, T, 256]: attention ACROSS OBJECTS, batch = frames
//   models/relation_head/transformer.py:7-56  TemporalTransformer: + pe, TransformerEncoderLayer(512, 4 heads, ff 512) over the
//                                              T frames of a pair, LayerNorm, fc1 / fc2 / span_head / pred_head, max over frames
//   models/relation_head/convolution.py:6-75  HandcraftedFilter (5-tap depthwise filter along T), Learnable1DConv (Conv1d k = 5)
//   models/relation_head/base.py:6-23         VanillaModel (the tail alone)
// (about 80 Tensile / AOTriton / elementwise launches per video) by
//   rel_qkv_kernel      [gather (subject, object) rows of the selected pairs | + positional table] -> in_proj (q scaled) of layer 0
//   rel_layer_kernel    self-attention over a sequence + out_proj + LayerNorm + FFN + LayerNorm [+ the next layer's in_proj]
//   rel_conv5_kernel    Conv1d(512, 512, 5, padding 2) + ReLU along T as five shifted row GEMMs
//   rel_tail_kernel     [5-tap filter] [LayerNorm] fc1 + ReLU, fc2 + ReLU, span_head per frame, max over frames of pred_head
// A workgroup owns 16 rows (objects of one frame / frames of one pair) with the activations in LDS; GEMMs are exact f32 on
// v_mfma_f32_16x16x4_f32 with weights streamed from L2 in fragment order (rows_common.h), the post-norm layer follows
// torch.nn.TransformerEncoderLayer(norm_first=False, activation=relu) in eval mode.  Soft-max over the keys is online in chunks,
// so any sequence length works (N tubes per frame, T frames per pair).
#include <stdlib.h>

#include "rows_common.h"

#include "../../include/openpvsg_hip.h"

namespace pvsg {

template <int D> struct RelCfg;
template <> struct RelCfg<256> {           // ObjectEncoder: 8 heads x 32, one wave per head over the tile's 16 rows
  static constexpr int H = 8, HD = 32;
};
template <> struct RelCfg<512> {           // TemporalTransformer: 4 heads x 128, two waves per head (alternate 64-key chunks)
  static constexpr int H = 4, HD = 128;
};
constexpr int REL_F = 512;                 // dim_feedforward of both reference modules
constexpr int REL_LDH = REL_F + 4;

constexpr int REL_BIG = 8 * 16 * ROWS_PLD > 16 * REL_LDH ? 8 * 16 * ROWS_PLD : 16 * REL_LDH;      // floats
template <int D>
constexpr size_t rel_layer_lds() {
  return (size_t)(2 * 16 * (D + 4) + REL_BIG) * sizeof(float);
}
static_assert(4 * 16 * 128 <= 16 * (512 + 4), "the heads' exchange tiles must fit in the q tile");


// ------------------------------------------------------------------------------------------------
// in_proj of the FIRST layer (later layers get theirs from the previous layer's kernel)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(ROWS_THREADS) void rel_qkv_kernel(
    const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ x, const float* __restrict__ g0, const float* __restrict__ g1,
    const long long* __restrict__ gpairs, const float* __restrict__ pe, float* __restrict__ x0_out,
    float* __restrict__ qkv, long long rows, int L, float qscale, int esplit) {
  constexpr int LD = D + 4;
  __shared__ __attribute__((aligned(16))) float xa[16 * LD];
  int e = blockIdx.y, bx = blockIdx.x;
  if (esplit) {                            // encoder by XCD half (see rel_layer_body)
    e = (bx >> 2) & 1;
    bx = ((bx >> 3) << 2) | (bx & 3);
    if (bx >= esplit) return;
  }
  const long long row0 = (long long)bx * 16;
  const int valid = (int)min((long long)16, rows - row0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < D / 128; ++it) {
    const int idx = threadIdx.x + it * ROWS_THREADS;
    const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < valid) {
      const long long row = row0 + r;
      if (gpairs) {                          // row (p, t) = [ g0[pairs[p][0], t, :] | g1[pairs[p][1], t, :] ]
        const long long p = row / L;
        const int t = (int)(row - p * L), half = c >= D / 2 ? 1 : 0;
        const long long oi = gpairs[2 * p + half];
        v = ld4((half ? g1 : g0) + (oi * L + t) * (D / 2) + (c - half * (D / 2)));
      } else {
        v = ld4(x + row * D + c);
      }
      if (pe) {
        const float4 q = ld4(pe + (row % L) * D + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (x0_out && e == 0) st4(x0_out + row * D + c, v);
    }
    *reinterpret_cast<float4*>(xa + r * LD + c) = v;
  }
  __syncthreads();
  const float* wp = e ? w1 : w0;
  const float* bp = e ? b1 : b0;
  float* out = qkv + (long long)e * rows * (3 * D);
  const int g = lane >> 4, j = lane & 15;
  rows_linear<D, 3 * D>(xa, LD, wp, w, lane, [&](int t, const f32x4& acc) {
    const int col = t * 16 + j;
    const float bv = bp[col], sc = col < D ? qscale : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * g + i;
      if (r < valid) out[(row0 + r) * (3 * D) + col] = (acc[i] + bv) * sc;
    }
  });
}

// ------------------------------------------------------------------------------------------------
// one post-norm encoder layer for a 16-row tile of one sequence (and one of up to two encoders: blockIdx.y)
// ------------------------------------------------------------------------------------------------
#define REL_SEL(f) (e ? L1.f : L0.f)

template <int D>
__device__ __forceinline__ void rel_layer_body(
    const pvsg_encoder_layer& L0, const pvsg_encoder_layer& L1, const float* __restrict__ nw0, const float* __restrict__ nb0,
    const float* __restrict__ nw1, const float* __restrict__ nb1, const float* __restrict__ x, long long x_estride,
    const float* __restrict__ qkv, float* __restrict__ y, float* __restrict__ qkv_next, long long rows, int L,
    long long seq_stride, long long pos_stride, int tiles_per_seq, float qscale, int esplit) {
  using C = RelCfg<D>;
  constexpr int LD = D + 4, HD = C::HD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xa = smem;                        // attention output -> x1 -> y
  float* xb = xa + 16 * LD;                // q rows -> pre-norm sums
  float* big = xb + 16 * LD;               // attention probabilities [8 waves][16][ROWS_PLD]  /  FFN hidden [16][REL_LDH]
  // two encoders: workgroup b runs on XCD b % 8 -- XCDs 0-3 take encoder 0, XCDs 4-7 encoder 1, so that an XCD's 4 MB L2 holds
  // ONE encoder's 2 MB of layer weights (1-D grid of 8 * ceil(n / 4) workgroups, n per encoder)
  int e = blockIdx.y, bx = blockIdx.x;
  if (esplit) {
    e = (bx >> 2) & 1;
    bx = ((bx >> 3) << 2) | (bx & 3);
    if (bx >= esplit) return;
  }
  const int s = bx / tiles_per_seq, tile = bx - s * tiles_per_seq;
  const int p0 = tile * 16;
  const int valid = min(16, L - p0);
  const long long row0 = (long long)s * seq_stride + (long long)p0 * pos_stride;
  const float* qkv_e = qkv + (long long)e * rows * (3 * D);
  const float* xe = x + (long long)e * x_estride;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;

#pragma unroll
  for (int it = 0; it < D / 128; ++it) {                        // scaled q rows of the tile -> xb
    const int idx = threadIdx.x + it * ROWS_THREADS;
    const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
    const float4 v = r < valid ? ld4(qkv_e + (row0 + (long long)r * pos_stride) * (3 * D) + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(xb + r * LD + c) = v;
  }
  __syncthreads();
  if constexpr (HD == 32) {
    // ---- self-attention on the matrix cores: wave = head, 16 rows x (chunks of 64 keys): rows_common.h --------------------
    rows_attention_h32(xb, xa, LD, big + w * (16 * ROWS_PLD), qkv_e + (long long)s * seq_stride * (3 * D), pos_stride * (3 * D), D,
                       L, w, lane);
  } else {
    // ---- 4 heads x 128 channels: two waves per head.  Wave (h, half) takes the 64-key chunks half, half + 2, ...; its
    // un-normalised output, running maximum and sum meet the other half's through LDS (the merge of two key ranges) ----------
    const int h = w & 3, half = w >> 2;
    float* pm = big + w * (16 * ROWS_PLD);
    f32x4 O[8];
    float M[4], l[4];
    rows_attention_core<128>(xb, LD, pm, qkv_e + (long long)s * seq_stride * (3 * D), pos_stride * (3 * D), D, L, h, lane, half * 64,
                             128, O, M, l);
    __syncthreads();                              // every wave is done with the q rows: xb becomes the exchange buffer
    float* ox = xb + h * (16 * 128);              // [16 rows][128 channels] of this head (4 x 2048 floats <= 16 x LD)
    if (half == 1) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ox[(4 * g + i) * 128 + nt * 16 + j] = O[nt][i];
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pm[(4 * g + i) * 2] = M[i];
          pm[(4 * g + i) * 2 + 1] = l[i];
        }
      }
    }
    __syncthreads();
    if (half == 0) {
      const float* p1 = big + (w + 4) * (16 * ROWS_PLD);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m1 = p1[(4 * g + i) * 2], l1 = p1[(4 * g + i) * 2 + 1];
        const float mn = fmaxf(M[i], m1);         // M is finite (chunk 0 holds a key); a range without keys has m1 = -inf, l1 = 0
        const float a0 = __expf(M[i] - mn), a1 = __expf(m1 - mn);
        const float inv = 1.f / (l[i] * a0 + l1 * a1);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          xa[(4 * g + i) * LD + h * 128 + nt * 16 + j] = (O[nt][i] * a0 + ox[(4 * g + i) * 128 + nt * 16 + j] * a1) * inv;
      }
    }
  }
  __syncthreads();
  // ---- out_proj + identity -> xb;  x1 = norm1 -> xa ----------------------------------------------------------------
  {
    const float* ob = REL_SEL(out_b);
    rows_linear<D, D>(xa, LD, REL_SEL(out_w), w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = ob[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const float res = r < valid ? xe[(row0 + (long long)r * pos_stride) * D + col] : 0.f;
        xb[r * LD + col] = acc[i] + bv + res;
      }
    });
  }
  __syncthreads();
  rows_layernorm_t<D>(xb, REL_SEL(n1_g), REL_SEL(n1_b), REL_SEL(eps1), xa, nullptr, 0, valid);
  __syncthreads();
  // ---- FFN: relu(x1 W1^T + b1) -> big;  . W2^T + b2 + x1 -> xb;  y = norm2 -> xa + global -------------------------------
  {
    const float* fb = REL_SEL(f1_b);
    rows_linear<D, REL_F>(xa, LD, REL_SEL(f1_w), w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = fb[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) big[(4 * g + i) * REL_LDH + col] = fmaxf(acc[i] + bv, 0.f);
    });
  }
  __syncthreads();
  {
    const float* fb = REL_SEL(f2_b);
    rows_linear<REL_F, D>(big, REL_LDH, REL_SEL(f2_w), w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = fb[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        xb[r * LD + col] = acc[i] + bv + xa[r * LD + col];
      }
    });
  }
  __syncthreads();
  rows_layernorm_t<D>(xb, REL_SEL(n2_g), REL_SEL(n2_b), REL_SEL(eps2), xa, y + ((long long)e * rows + row0) * D,
                      pos_stride * D, valid);
  if (qkv_next) {                          // the next layer's in_proj on the rows just finished
    __syncthreads();
    const float* nw = e ? nw1 : nw0;
    const float* nb = e ? nb1 : nb0;
    float* qn = qkv_next + (long long)e * rows * (3 * D);
    rows_linear<D, 3 * D>(xa, LD, nw, w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = nb[col], sc = col < D ? qscale : 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        if (r < valid) qn[(row0 + (long long)r * pos_stride) * (3 * D) + col] = (acc[i] + bv) * sc;
      }
    });
  }
}

#define REL_LAYER_ARGS                                                                                                         \
  pvsg_encoder_layer L0, pvsg_encoder_layer L1, const float *__restrict__ nw0, const float *__restrict__ nb0,                  \
      const float *__restrict__ nw1, const float *__restrict__ nb1, const float *__restrict__ x, long long x_estride,          \
      const float *__restrict__ qkv, float *__restrict__ y, float *__restrict__ qkv_next, long long rows, int L,               \
      long long seq_stride, long long pos_stride, int tiles_per_seq, float qscale, int esplit
#define REL_LAYER_PASS L0, L1, nw0, nb0, nw1, nb1, x, x_estride, qkv, y, qkv_next, rows, L, seq_stride, pos_stride, tiles_per_seq, qscale, esplit
// d_model 256: 66 KB of LDS and at most 128 VGPRs -> two workgroups per CU (T x ceil(N / 16) x 2 encoders = 448 workgroups at
// N = 100, T = 32 are all resident at once); d_model 512: 99 KB, one workgroup per CU
__global__ __launch_bounds__(ROWS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void rel_layer256_kernel(REL_LAYER_ARGS) {
  rel_layer_body<256>(REL_LAYER_PASS);
}
__global__ __launch_bounds__(ROWS_THREADS) void rel_layer512_kernel(REL_LAYER_ARGS) { rel_layer_body<512>(REL_LAYER_PASS); }

// ------------------------------------------------------------------------------------------------
// soft-max(q k^T / sqrt(hd)) v alone, for the LONG-video route (relation.py): there the linear layers of an encoder layer run as
// token GEMMs on the 16-bit matrix pipe (csrc/token_gemm.hip, 4 - 5 x the f32 MFMA rate once the rows fill 128-row tiles) and only
// the attention stays a row kernel.  qkv (rows, 3 D) with UN-scaled q; out (rows, D) = the concatenated heads.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(ROWS_THREADS) void rel_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int L,
                                                                     long long seq_stride, long long pos_stride, int tiles_per_seq,
                                                                     float qscale) {
  constexpr int LD = D + 4, HD = RelCfg<D>::HD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xq = smem;                        // [16][LD] scaled q rows (D = 512: afterwards the heads' exchange tiles)
  float* xo = xq + 16 * LD;                // [16][LD] output rows
  float* pmb = xo + 16 * LD;               // [8 waves][16][ROWS_PLD]
  const int s = blockIdx.x / tiles_per_seq, tile = blockIdx.x - s * tiles_per_seq;
  const int p0 = tile * 16;
  const int valid = min(16, L - p0);
  const long long row0 = (long long)s * seq_stride + (long long)p0 * pos_stride;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int it = 0; it < D / 128; ++it) {
    const int idx = threadIdx.x + it * ROWS_THREADS;
    const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
    float4 v = r < valid ? ld4(qkv + (row0 + (long long)r * pos_stride) * (3 * D) + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x *= qscale; v.y *= qscale; v.z *= qscale; v.w *= qscale;
    *reinterpret_cast<float4*>(xq + r * LD + c) = v;
  }
  __syncthreads();
  const float* kvb = qkv + (long long)s * seq_stride * (3 * D);
  const long long kstride = pos_stride * (3 * D);
  if constexpr (HD == 32) {
    rows_attention_h32(xq, xo, LD, pmb + w * (16 * ROWS_PLD), kvb, kstride, D, L, w, lane);
  } else {
    const int h = w & 3, half = w >> 2;
    float* pm = pmb + w * (16 * ROWS_PLD);
    f32x4 O[8];
    float M[4], l[4];
    rows_attention_core<128>(xq, LD, pm, kvb, kstride, D, L, h, lane, half * 64, 128, O, M, l);
    __syncthreads();
    float* ox = xq + h * (16 * 128);
    if (half == 1) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) ox[(4 * g + i) * 128 + nt * 16 + j] = O[nt][i];
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pm[(4 * g + i) * 2] = M[i];
          pm[(4 * g + i) * 2 + 1] = l[i];
        }
      }
    }
    __syncthreads();
    if (half == 0) {
      const float* p1 = pmb + (w + 4) * (16 * ROWS_PLD);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m1 = p1[(4 * g + i) * 2], l1 = p1[(4 * g + i) * 2 + 1];
        const float mn = fmaxf(M[i], m1);
        const float a0 = __expf(M[i] - mn), a1 = __expf(m1 - mn);
        const float inv = 1.f / (l[i] * a0 + l1 * a1);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          xo[(4 * g + i) * LD + h * 128 + nt * 16 + j] = (O[nt][i] * a0 + ox[(4 * g + i) * 128 + nt * 16 + j] * a1) * inv;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < D / 128; ++it) {
    const int idx = threadIdx.x + it * ROWS_THREADS;
    const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
    if (r < valid) st4(out + (row0 + (long long)r * pos_stride) * D + c, *reinterpret_cast<const float4*>(xo + r * LD + c));
  }
}
template <int D>
constexpr size_t rel_attention_lds() {
  return (size_t)(2 * 16 * (D + 4) + 8 * 16 * ROWS_PLD) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// Conv1d(512, 512, kernel 5, padding 2) + ReLU along the frames of a pair (convolution.py:49-56): y[t] = relu(b + sum_k W_k x[t+k-2])
// ------------------------------------------------------------------------------------------------
constexpr int REL_D2 = 512;
constexpr int REL_LD2 = REL_D2 + 4;

__global__ __launch_bounds__(ROWS_THREADS) void rel_conv5_kernel(const float* __restrict__ wp, const float* __restrict__ bias,
                                                                 const float* __restrict__ x, float* __restrict__ y, int T,
                                                                 int tiles_per_seq) {
  __shared__ __attribute__((aligned(16))) float xt[20 * REL_LD2];          // frames t0-2 .. t0+17, zero outside [0, T)
  const int s = blockIdx.x / tiles_per_seq, tile = blockIdx.x - s * tiles_per_seq;
  const int t0 = tile * 16;
  const int valid = min(16, T - t0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const float* xs = x + (long long)s * T * REL_D2;
  for (int idx = threadIdx.x; idx < 20 * (REL_D2 / 4); idx += ROWS_THREADS) {
    const int r = idx / (REL_D2 / 4), c = (idx % (REL_D2 / 4)) * 4;
    const int t = t0 - 2 + r;
    const float4 v = (t >= 0 && t < T) ? ld4(xs + (long long)t * REL_D2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(xt + r * REL_LD2 + c) = v;
  }
  __syncthreads();
  f32x4 acc[4];
  zero_acc(acc);
#pragma unroll 1
  for (int k = 0; k < 5; ++k)
    rows_gemm<4, REL_D2 / 16, 4>(xt + k * REL_LD2, REL_LD2, wp + (long long)k * REL_D2 * REL_D2, REL_D2 / 16, 0, w * 4, acc, lane);
  float* ys = y + ((long long)s * T + t0) * REL_D2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = (w * 4 + i) * 16 + j;
    const float bv = bias[col];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      if (r < valid) ys[(long long)r * REL_D2 + col] = fmaxf(acc[i][e] + bv, 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tail shared by the four relation models: [filter] [LayerNorm] fc1 relu fc2 relu -> span_head per frame, max_t pred_head
// one workgroup per pair walks the frames in tiles of 16
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROWS_THREADS) void rel_tail_kernel(pvsg_relation_tail Tl, const float* __restrict__ x,
                                                                float* __restrict__ span, float* __restrict__ pred, int T,
                                                                float* __restrict__ part_max, int chunks, int tiles_per_chunk) {
  constexpr int D = REL_D2, LD = REL_LD2, LD1 = 256 + 4, LDq = 128 + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xa = smem;                         // [16][LD] input rows
  float* xb = xa + 16 * LD;                 // [16][LD] normalised rows
  float* h1 = xb + 16 * LD;                 // [16][LD1]
  float* h2 = h1 + 16 * LD1;                // [16][LDq]
  // chunks == 1: one workgroup per pair walks all its frame tiles and writes relation_pred.  chunks > 1 (long videos): workgroup
  // (pair, chunk) walks `tiles_per_chunk` tiles and leaves its column maxima in part_max (P, chunks, 64); rel_tail_max_kernel
  // folds them.  (A rendezvous inside one launch -- last workgroup reduces -- needs a device-scope fence per workgroup, which on
  // the eight-XCD part writes the XCD's L2 back: it made this kernel 1.8 x SLOWER, profiles/r06_rel_tail_rendezvous_ab.txt.)
  const int s = blockIdx.x / chunks, chunk = blockIdx.x - s * chunks;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int R = Tl.num_relations;
  const float* xs = x + (long long)s * T * D;
  float fw[5] = {0.f, 0.f, 1.f, 0.f, 0.f};
  if (Tl.filter) {
#pragma unroll
    for (int k = 0; k < 5; ++k) fw[k] = Tl.filter[k];
  }
  float runmax = -INFINITY;                 // waves 4..7: lane (g, j) follows column 16 (w - 4) + j of pred_head over its rows
  const int t_begin = chunk * tiles_per_chunk * 16, t_end = min(T, t_begin + tiles_per_chunk * 16);
#pragma unroll 1
  for (int t0 = t_begin; t0 < t_end; t0 += 16) {
    const int valid = min(16, T - t0);
#pragma unroll
    for (int it = 0; it < D / 128; ++it) {
      const int idx = threadIdx.x + it * ROWS_THREADS;
      const int r = idx / (D / 4), c = (idx % (D / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < valid) {
        const int t = t0 + r;
        if (Tl.filter) {                    // F.conv1d(x, w, padding=2, groups=C): cross-correlation, zero padding
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const int tt = t + k - 2;
            if (tt >= 0 && tt < T) {
              const float4 u = ld4(xs + (long long)tt * D + c);
              v.x += fw[k] * u.x; v.y += fw[k] * u.y; v.z += fw[k] * u.z; v.w += fw[k] * u.w;
            }
          }
        } else {
          v = ld4(xs + (long long)t * D + c);
        }
      }
      *reinterpret_cast<float4*>(xa + r * LD + c) = v;
    }
    __syncthreads();
    const float* src = xa;
    if (Tl.ln_g) {
      rows_layernorm_t<D>(xa, Tl.ln_g, Tl.ln_b, Tl.eps, xb, nullptr, 0, valid);
      src = xb;
      __syncthreads();
    }
    rows_linear<D, 256>(src, LD, Tl.fc1_w, w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = Tl.fc1_b[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) h1[(4 * g + i) * LD1 + col] = fmaxf(acc[i] + bv, 0.f);
    });
    __syncthreads();
    rows_linear<256, 128>(h1, LD1, Tl.fc2_w, w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = Tl.fc2_b[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) h2[(4 * g + i) * LDq + col] = fmaxf(acc[i] + bv, 0.f);
    });
    __syncthreads();
    // heads: packed rows 0..63 = span_head (R used), 64..127 = pred_head
    rows_linear<128, 128>(h2, LDq, Tl.head_w, w, lane, [&](int t, const f32x4& acc) {
      const int col = t * 16 + j;
      const float bv = Tl.head_b[col];
      if (t < 4) {
        if (col < R) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            if (r < valid) span[((long long)s * T + t0 + r) * R + col] = acc[i] + bv;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * g + i < valid) runmax = fmaxf(runmax, acc[i] + bv);
      }
    });
    __syncthreads();                        // xa / h1 / h2 are rewritten by the next tile
  }
  if (w >= 4) {
    runmax = fmaxf(runmax, __shfl_xor(runmax, 16));
    runmax = fmaxf(runmax, __shfl_xor(runmax, 32));
    const int col = (w - 4) * 16 + j;
    if (g == 0) {
      if (chunks == 1) {
        if (col < R) pred[(long long)s * R + col] = runmax;
      } else {
        part_max[((long long)s * chunks + chunk) * 64 + col] = runmax;     // -inf from a chunk past the last frame
      }
    }
  }
}

__global__ void rel_tail_max_kernel(const float* __restrict__ part_max, float* __restrict__ pred, int P, int chunks, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (pair, column)
  if (i >= P * 64) return;
  const int s = i >> 6, col = i & 63;
  if (col >= R) return;
  float m = -INFINITY;
  for (int c = 0; c < chunks; ++c) m = fmaxf(m, part_max[((long long)s * chunks + c) * 64 + col]);
  pred[(long long)s * R + col] = m;
}

constexpr size_t REL_TAIL_LDS = (size_t)(2 * 16 * REL_LD2 + 16 * (256 + 4) + 16 * (128 + 4)) * sizeof(float);

static int check_encoder_layer(const pvsg_encoder_layer* L, const char* who) {
  PVSG_REQUIRE(L->in_w && L->in_b && L->out_w && L->out_b && L->n1_g && L->n1_b && L->f1_w && L->f1_b && L->f2_w && L->f2_b &&
                   L->n2_g && L->n2_b, "%s: null pointer in pvsg_encoder_layer", who);
  const bool ok = (L->d_model == 256 && L->num_heads == 8) || (L->d_model == 512 && L->num_heads == 4);
  if (!ok || L->ffn_dim != REL_F)
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for (d_model 256, 8 heads) and (d_model 512, 4 heads) with dim_feedforward %d "
                   "(got %d / %d / %d)", who, REL_F, L->d_model, L->num_heads, L->ffn_dim);
  return PVSG_OK;
}

}  // namespace pvsg

extern "C" int pvsg_rel_qkv(const pvsg_encoder_layer* layers, int E, const float* x, const float* gather_sub,
                            const float* gather_obj, const long long* gather_pairs, const float* pe, float* x0_out,
                            float* qkv, long long rows, int L, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(layers && qkv, "rel_qkv: null pointer argument");
  PVSG_REQUIRE(E == 1 || E == 2, "rel_qkv: one or two encoders per launch (got %d)", E);
  PVSG_REQUIRE(rows > 0 && L > 0 && rows % L == 0, "rel_qkv: rows must be a positive multiple of the sequence length");
  PVSG_REQUIRE((x != nullptr) != (gather_pairs != nullptr), "rel_qkv: give either x or the gather form");
  PVSG_REQUIRE(!gather_pairs || (gather_sub && gather_obj), "rel_qkv: the gather form needs both sources");
  for (int e = 0; e < E; ++e) {
    if (int rc = check_encoder_layer(layers + e, "rel_qkv")) return rc;
    PVSG_REQUIRE(layers[e].d_model == layers[0].d_model, "rel_qkv: the encoders of one launch share d_model");
  }
  const pvsg_encoder_layer& A = layers[0];
  const pvsg_encoder_layer& B = layers[E - 1];
  const int nper = (int)((rows + 15) / 16);
  static const bool xcd_split = []() { const char* v = getenv("PVSG_REL_XCD_SPLIT"); return !(v && !strcmp(v, "off")); }();
  const int esplit = (E == 2 && xcd_split) ? nper : 0;
  const dim3 grid(esplit ? (unsigned)(8 * ((nper + 3) / 4)) : (unsigned)nper, esplit ? 1u : (unsigned)E);
  if (A.d_model == 256) {
    hipLaunchKernelGGL(rel_qkv_kernel<256>, grid, dim3(ROWS_THREADS), 0, stream, A.in_w, A.in_b, B.in_w, B.in_b, x, gather_sub,
                       gather_obj, gather_pairs, pe, x0_out, qkv, rows, L, 0.17677669529663687f, esplit);
  } else {
    hipLaunchKernelGGL(rel_qkv_kernel<512>, grid, dim3(ROWS_THREADS), 0, stream, A.in_w, A.in_b, B.in_w, B.in_b, x, gather_sub,
                       gather_obj, gather_pairs, pe, x0_out, qkv, rows, L, 0.08838834764831845f, esplit);
  }
  PVSG_LAUNCH_CHECK("rel_qkv");
  return PVSG_OK;
}

extern "C" int pvsg_rel_encoder_layer(const pvsg_encoder_layer* layers, const pvsg_encoder_layer* next_layers, int E,
                                      const float* x, long long x_encoder_stride, const float* qkv, float* y, float* qkv_next,
                                      int S, int L, long long seq_stride, long long pos_stride, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(layers && x && qkv && y, "rel_encoder_layer: null pointer argument");
  PVSG_REQUIRE(E == 1 || E == 2, "rel_encoder_layer: one or two encoders per launch (got %d)", E);
  PVSG_REQUIRE(S > 0 && L > 0 && seq_stride > 0 && pos_stride > 0, "rel_encoder_layer: non-positive dimension");
  PVSG_REQUIRE((next_layers == nullptr) == (qkv_next == nullptr), "rel_encoder_layer: next_layers and qkv_next go together");
  PVSG_REQUIRE(qkv_next != qkv, "rel_encoder_layer: qkv_next must not alias qkv (other tiles still read their keys)");
  for (int e = 0; e < E; ++e) {
    if (int rc = check_encoder_layer(layers + e, "rel_encoder_layer")) return rc;
    PVSG_REQUIRE(layers[e].d_model == layers[0].d_model, "rel_encoder_layer: the encoders of one launch share d_model");
    if (next_layers) {
      if (int rc = check_encoder_layer(next_layers + e, "rel_encoder_layer")) return rc;
      PVSG_REQUIRE(next_layers[e].d_model == layers[0].d_model, "rel_encoder_layer: next layer of another width");
    }
  }
  const pvsg_encoder_layer& A = layers[0];
  const pvsg_encoder_layer& B = layers[E - 1];
  const float* nw0 = next_layers ? next_layers[0].in_w : nullptr;
  const float* nb0 = next_layers ? next_layers[0].in_b : nullptr;
  const float* nw1 = next_layers ? next_layers[E - 1].in_w : nullptr;
  const float* nb1 = next_layers ? next_layers[E - 1].in_b : nullptr;
  const long long rows = (long long)S * L;
  const int tiles = (L + 15) / 16;
  const int nper = S * tiles;
  static const bool xcd_split = []() { const char* v = getenv("PVSG_REL_XCD_SPLIT"); return !(v && !strcmp(v, "off")); }();
  const int esplit = (E == 2 && xcd_split) ? nper : 0;
  const dim3 grid(esplit ? (unsigned)(8 * ((nper + 3) / 4)) : (unsigned)nper, esplit ? 1u : (unsigned)E);
  if (A.d_model == 256) {
    static std::atomic<unsigned long long> done;
    const hipError_t er = ensure_dynamic_lds(reinterpret_cast<const void*>(&rel_layer256_kernel), (int)rel_layer_lds<256>(), done);
    if (er != hipSuccess) return set_err(PVSG_ERR_HIP, "rel_encoder_layer: LDS attribute: %s", hipGetErrorString(er));
    hipLaunchKernelGGL(rel_layer256_kernel, grid, dim3(ROWS_THREADS), rel_layer_lds<256>(), stream, A, B, nw0, nb0, nw1, nb1, x,
                       x_encoder_stride, qkv, y, qkv_next, rows, L, seq_stride, pos_stride, tiles, 0.17677669529663687f, esplit);
  } else {
    static std::atomic<unsigned long long> done;
    const hipError_t er = ensure_dynamic_lds(reinterpret_cast<const void*>(&rel_layer512_kernel), (int)rel_layer_lds<512>(), done);
    if (er != hipSuccess) return set_err(PVSG_ERR_HIP, "rel_encoder_layer: LDS attribute: %s", hipGetErrorString(er));
    hipLaunchKernelGGL(rel_layer512_kernel, grid, dim3(ROWS_THREADS), rel_layer_lds<512>(), stream, A, B, nw0, nb0, nw1, nb1, x,
                       x_encoder_stride, qkv, y, qkv_next, rows, L, seq_stride, pos_stride, tiles, 0.08838834764831845f, esplit);
  }
  PVSG_LAUNCH_CHECK("rel_encoder_layer");
  return PVSG_OK;
}

extern "C" int pvsg_rel_attention(const float* qkv, float* out, int S, int L, long long seq_stride, long long pos_stride, int d_model,
                                  int num_heads, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(qkv && out, "rel_attention: null pointer argument");
  PVSG_REQUIRE(S > 0 && L > 0 && seq_stride > 0 && pos_stride > 0, "rel_attention: non-positive dimension");
  if (!((d_model == 256 && num_heads == 8) || (d_model == 512 && num_heads == 4)))
    return set_err(PVSG_ERR_UNSUPPORTED, "rel_attention: built for (d_model 256, 8 heads) and (d_model 512, 4 heads) (got %d / %d)",
                   d_model, num_heads);
  const int tiles = (L + 15) / 16;
  const dim3 grid((unsigned)(S * tiles));
  if (d_model == 256) {
    static std::atomic<unsigned long long> done;
    const hipError_t er = ensure_dynamic_lds(reinterpret_cast<const void*>(&rel_attention_kernel<256>), (int)rel_attention_lds<256>(), done);
    if (er != hipSuccess) return set_err(PVSG_ERR_HIP, "rel_attention: LDS attribute: %s", hipGetErrorString(er));
    hipLaunchKernelGGL(rel_attention_kernel<256>, grid, dim3(ROWS_THREADS), rel_attention_lds<256>(), stream, qkv, out, L, seq_stride,
                       pos_stride, tiles, 0.17677669529663687f);
  } else {
    static std::atomic<unsigned long long> done;
    const hipError_t er = ensure_dynamic_lds(reinterpret_cast<const void*>(&rel_attention_kernel<512>), (int)rel_attention_lds<512>(), done);
    if (er != hipSuccess) return set_err(PVSG_ERR_HIP, "rel_attention: LDS attribute: %s", hipGetErrorString(er));
    hipLaunchKernelGGL(rel_attention_kernel<512>, grid, dim3(ROWS_THREADS), rel_attention_lds<512>(), stream, qkv, out, L, seq_stride,
                       pos_stride, tiles, 0.08838834764831845f);
  }
  PVSG_LAUNCH_CHECK("rel_attention");
  return PVSG_OK;
}

extern "C" int pvsg_rel_conv5(const float* w_packed, const float* bias, const float* x, float* y, int P, int T, int C,
                              void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(w_packed && bias && x && y, "rel_conv5: null pointer argument");
  PVSG_REQUIRE(P > 0 && T > 0, "rel_conv5: non-positive dimension");
  PVSG_REQUIRE(x != y, "rel_conv5: in-place is not supported (neighbouring tiles read the halo frames)");
  if (C != REL_D2) return set_err(PVSG_ERR_UNSUPPORTED, "rel_conv5: built for %d channels (got %d)", REL_D2, C);
  const int tiles = (T + 15) / 16;
  hipLaunchKernelGGL(rel_conv5_kernel, dim3((unsigned)(P * tiles)), dim3(ROWS_THREADS), 0, stream, w_packed, bias, x, y, T, tiles);
  PVSG_LAUNCH_CHECK("rel_conv5");
  return PVSG_OK;
}

// scratch of pvsg_rel_tail for videos of more than 64 frames (no initialisation needed): per (pair, chunk) 64 column maxima
static int rel_tail_chunks(int P, int T) {
  const int tiles = (T + 15) / 16;
  if (tiles <= 4) return 1;
  int want = (1024 + P - 1) / P;                                // about four workgroups per CU's worth of (pair, chunk) items
  want = want < 1 ? 1 : want;
  const int by_tiles = (tiles + 1) / 2;                         // at least two tiles per chunk
  return want < by_tiles ? want : by_tiles;
}
extern "C" long long pvsg_rel_tail_workspace_bytes(int P, int T) {
  if (P <= 0 || T <= 0) return 0;
  const int chunks = rel_tail_chunks(P, T);
  return chunks > 1 ? (long long)P * chunks * 64 * 4 : 0;
}

extern "C" int pvsg_rel_tail(const pvsg_relation_tail* tail, const float* x, float* span_pred, float* relation_pred,
                             void* workspace, int P, int T, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(tail && x && span_pred && relation_pred, "rel_tail: null pointer argument");
  PVSG_REQUIRE(P > 0 && T > 0, "rel_tail: non-positive dimension");
  PVSG_REQUIRE(tail->fc1_w && tail->fc1_b && tail->fc2_w && tail->fc2_b && tail->head_w && tail->head_b,
               "rel_tail: null pointer in pvsg_relation_tail");
  PVSG_REQUIRE((tail->ln_g == nullptr) == (tail->ln_b == nullptr), "rel_tail: ln_g and ln_b go together");
  if (tail->dim != REL_D2 || tail->num_relations <= 0 || tail->num_relations > 64)
    return set_err(PVSG_ERR_UNSUPPORTED, "rel_tail: built for input_dim %d and 1..64 relations (got %d / %d)", REL_D2, tail->dim,
                   tail->num_relations);
  static std::atomic<unsigned long long> done;
  const hipError_t er = ensure_dynamic_lds(reinterpret_cast<const void*>(&rel_tail_kernel), (int)REL_TAIL_LDS, done);
  if (er != hipSuccess) return set_err(PVSG_ERR_HIP, "rel_tail: LDS attribute: %s", hipGetErrorString(er));
  const int tiles = (T + 15) / 16;
  const int chunks = workspace ? rel_tail_chunks(P, T) : 1;
  const int tpc = (tiles + chunks - 1) / chunks;
  hipLaunchKernelGGL(rel_tail_kernel, dim3((unsigned)(P * chunks)), dim3(ROWS_THREADS), REL_TAIL_LDS, stream, *tail, x, span_pred,
                     relation_pred, T, static_cast<float*>(workspace), chunks, tpc);
  if (chunks > 1)
    hipLaunchKernelGGL(rel_tail_max_kernel, dim3((unsigned)((P * 64 + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const float*>(workspace), relation_pred, P, chunks, tail->num_relations);
  PVSG_LAUNCH_CHECK("rel_tail");
  return PVSG_OK;
}
