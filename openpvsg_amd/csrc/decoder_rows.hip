// The query-row side of a Mask2Former transformer-decoder layer in two launches.
//
// Replaces, per decoder layer ([3P] mmdet DetrTransformerDecoderLayer with operation_order
// ('cross_attn','norm','self_attn','norm','ffn','norm'), driven from
// models/mask2former/mask2former_head.py:457-470 / models/mask2former_vps/mask2former_video_head.py:435-452):
//   cross-attention out_proj + identity + LayerNorm                      (after pvsg_xattn_combine)
//   self-attention over the Q=100 queries (in_proj, softmax(QK^T)V, out_proj) + identity + LayerNorm
//   FFN 256 -> 2048 -> 256 (+ReLU) + identity + LayerNorm
// and the query-side half of forward_head (mask2former_head.py:375-381 / video_head.py:340-343):
//   post_norm LayerNorm, cls_embed, the 3-layer mask_embed MLP,
// plus the NEXT layer's cross-attention query projection ((q + query_pos) Wq^T + bq) / sqrt(32).
//
// Why: with Q = 100 rows these are ~35 tiny library GEMM / elementwise launches per layer (about 0.6 ms per
// layer, 5.5 ms per clip regardless of the clip length -- the fixed cost that caps frame-sharded scaling).
// Here a workgroup owns 16 query rows of one batch element and walks the whole chain with the activations in
// LDS; every GEMM is  X[16 x K] . W^T  on v_mfma_f32_16x16x4_f32 (exact f32), the weights streamed from L2
// in MFMA-fragment order (pvsg_pack_rows_weight: one contiguous 1 KiB per wave load).  The only cross-row
// step, self-attention, needs the K/V projections of all 100 rows, hence two kernels:
//   decoder_rows_pre   out_proj + LN, self-attention in_proj (q scaled, k, v)            -> x1, qkv
//   decoder_rows_post  self-attention, out_proj + LN, FFN + LN, post_norm, cls / mask embeddings, next q
// `decoder_rows_post` with no layer runs the head part only (the forward_head call on the initial queries).
#include "rows_common.h"

#include "../../include/openpvsg_hip.h"

namespace pvsg {

constexpr int DR_THREADS = ROWS_THREADS;  // 8 waves: 8 heads in the attention step, 8 column groups in the GEMMs (rows_common.h)
constexpr int DR_C = 256;                 // embed dims (8 heads x 32)
constexpr int DR_LD = DR_C + 4;           // LDS row stride: rows 16 B apart mod 256 B -> conflict-free b128 reads
constexpr int DR_FC = 512;                // FFN hidden chunk held in LDS
constexpr int DR_LDH = DR_FC + 4;
constexpr int DR_MAXQ = 128;
constexpr int DR_SPLIT = 8;               // workgroups per row tile in the split form of decoder_rows_post (256 hidden units each)
constexpr int DR_SPLIT_MAX_TILES = 32;    // above this many row tiles the chip is full without the split

// lab builds only (scripts/lab/rows_stamps.sh): wall-clock stamps (s_memrealtime, 10 ns) of the stages of decoder_rows_post for the
// eight workgroups of row tile 0, read back through pvsg_debug_rows_stamps
#ifndef PVSG_ROWS_STAMPS
#define PVSG_ROWS_STAMPS 0
#endif
#if PVSG_ROWS_STAMPS
__device__ unsigned long long g_rows_stamps[8 * 32];
#define DR_STAMP(i)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0 && bt == 0) g_rows_stamps[slice * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DR_STAMP(i) do { } while (0)
#endif

__global__ void pack_rows_weight_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K,
                                        long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long long rest = i >> 8;
  const int nkc = K >> 4;
  const int kc = (int)(rest % nkc), tile = (int)(rest / nkc);
  const int n = tile * 16 + (lane & 15), k = kc * 16 + 4 * (lane >> 4) + j;
  P[i] = n < N ? W[(long long)n * K + k] : 0.f;
}

// f16x2 pack (rows_common.h: rows_gemm_h).  One workgroup reduces max|w| (weights of at most a few MB, once per checkpoint)
// and writes the header [max|w|, 2^-e, 0, 0]; e as in split_common.h: max|w| 2^e in [2^13, 2^14).
__global__ __launch_bounds__(1024) void rows_f16x2_header_kernel(const float* __restrict__ W, float* __restrict__ P, long long n) {
  __shared__ float red[16];
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, __builtin_fabsf(W[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
      e = 13 - ilogbf(m);
      e = e > 126 ? 126 : (e < -126 ? -126 : e);
    }
    P[0] = m;
    P[1] = ldexpf(1.f, -e);
    P[2] = 0.f;
    P[3] = 0.f;
  }
}
// one thread per (column tile, k block, lane, pair of k): the two f16 of each limb it owns
__global__ void pack_rows_weight_f16x2_kernel(const float* __restrict__ W, float* __restrict__ P, int N, int K, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int jp = (int)(i & 3), lane = (int)((i >> 2) & 63);
  const long long rest = i >> 8;
  const int nkb = K >> 5;
  const int kb = (int)(rest % nkb), tile = (int)(rest / nkb);
  const int n = tile * 16 + (lane & 15), k = kb * 32 + 8 * (lane >> 4) + 2 * jp;
  const float un = P[1];                                  // 2^-e, a power of two: dividing by it is exact
  float a0 = 0.f, a1 = 0.f;
  if (n < N) {
    a0 = W[(long long)n * K + k] / un;
    a1 = W[(long long)n * K + k + 1] / un;
  }
  const rows_f16x2 hh = __builtin_convertvector(f32x2{a0, a1}, rows_f16x2);
  const rows_f16x2 ll = __builtin_convertvector(f32x2{a0 - (float)hh[0], a1 - (float)hh[1]}, rows_f16x2);
  unsigned* dst = reinterpret_cast<unsigned*>(P + 4) + ((long long)tile * nkb + kb) * 512 + lane * 4 + jp;   // 2 KiB = 512 words
  dst[0] = __builtin_bit_cast(unsigned, hh);
  dst[256] = __builtin_bit_cast(unsigned, ll);
}

// LayerNorm over the 256 columns of the 16 LDS rows of `x` (two-pass statistics like ATen), 32 threads per row.
// out_a / out_b: LDS destinations (either may be null); add_b: optional (16 x 256, row stride 256... see use)
// row-wise addend for out_b (query_pos); g_out: optional global destination (row pointer of row 0, stride 256).
__device__ __forceinline__ void rows_layernorm(const float* __restrict__ x, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps, float* out_a,
                                               float* out_b, const float* __restrict__ add_b,
                                               float* __restrict__ g_out, int valid_rows) {
  const int r = threadIdx.x >> 5, c0 = (threadIdx.x & 31) * 8;
  const float4 v0 = *reinterpret_cast<const float4*>(x + r * DR_LD + c0);
  const float4 v1 = *reinterpret_cast<const float4*>(x + r * DR_LD + c0 + 4);
  float s = (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s * (1.f / DR_C);
  float d[8] = {v0.x - mean, v0.y - mean, v0.z - mean, v0.w - mean, v1.x - mean, v1.y - mean, v1.z - mean, v1.w - mean};
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) q += d[i] * d[i];
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q * (1.f / DR_C) + eps);
  const float4 g0 = ld4(gamma + c0), g1 = ld4(gamma + c0 + 4), b0 = ld4(beta + c0), b1 = ld4(beta + c0 + 4);
  float4 y0 = make_float4(d[0] * rstd * g0.x + b0.x, d[1] * rstd * g0.y + b0.y, d[2] * rstd * g0.z + b0.z,
                          d[3] * rstd * g0.w + b0.w);
  float4 y1 = make_float4(d[4] * rstd * g1.x + b1.x, d[5] * rstd * g1.y + b1.y, d[6] * rstd * g1.z + b1.z,
                          d[7] * rstd * g1.w + b1.w);
  if (out_a) {
    *reinterpret_cast<float4*>(out_a + r * DR_LD + c0) = y0;
    *reinterpret_cast<float4*>(out_a + r * DR_LD + c0 + 4) = y1;
  }
  if (g_out && r < valid_rows) {
    st4(g_out + (long long)r * DR_C + c0, y0);
    st4(g_out + (long long)r * DR_C + c0 + 4, y1);
  }
  if (out_b) {
    if (add_b && r < valid_rows) {
      const float4 p0 = ld4(add_b + (long long)r * DR_C + c0), p1 = ld4(add_b + (long long)r * DR_C + c0 + 4);
      y0.x += p0.x; y0.y += p0.y; y0.z += p0.z; y0.w += p0.w;
      y1.x += p1.x; y1.y += p1.y; y1.z += p1.z; y1.w += p1.w;
    }
    *reinterpret_cast<float4*>(out_b + r * DR_LD + c0) = y0;
    *reinterpret_cast<float4*>(out_b + r * DR_LD + c0 + 4) = y1;
  }
}

// 16 x 256 tile of a (rows, 256) global tensor -> LDS (zeros past `valid_rows`)
__device__ __forceinline__ void load_tile(float* dst, const float* __restrict__ src, int valid_rows) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = threadIdx.x + it * DR_THREADS;     // float4 index in the 16 x 64 grid
    const int r = idx >> 6, c = (idx & 63) * 4;
    const float4 v = r < valid_rows ? ld4(src + (long long)r * DR_C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dst + r * DR_LD + c) = v;
  }
}

// epilogue of a 256-column GEMM: out[r][c] = act(acc + bias[c] (+ res[r][c])) into LDS
template <int NT>
__device__ __forceinline__ void store_tiles_lds(const f32x4 (&acc)[NT], int t0, const float* __restrict__ bias,
                                                const float* res, int ldres, float* out, int ldo, int col_off,
                                                bool relu, int lane) {
  const int g = lane >> 4, j = lane & 15;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int col = (t0 + i) * 16 + j;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      float v = acc[i][e] + bv;
      if (res) v += res[r * ldres + col - col_off];
      if (relu) v = fmaxf(v, 0.f);
      out[r * ldo + col - col_off] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pre: x1 = LN(core Wo^T + bo + query);  qkv = [ ((x1+pos) Wq^T + bq) * scale | (x1+pos) Wk^T + bk | x1 Wv^T + bv ]
// ------------------------------------------------------------------------------------------------
template <bool H>
__global__ __launch_bounds__(DR_THREADS) void decoder_rows_pre_kernel(
    pvsg_decoder_layer L, const float* __restrict__ core, const float* __restrict__ query,
    const float* __restrict__ qpos, float* __restrict__ x1, float* __restrict__ qkv, int Q, int tiles_per_b,
    float scale, float eps, unsigned* __restrict__ overflow) {
  float amax = 0.f;                         // H: largest |activation| this lane split into f16 limbs
  __shared__ __attribute__((aligned(16))) float xa[16 * DR_LD], xb[16 * DR_LD], xc[16 * DR_LD];
  const int b = blockIdx.x / tiles_per_b, tile = blockIdx.x - b * tiles_per_b;
  const int q0 = tile * 16;
  const int valid = min(16, Q - q0);
  const long long row0 = (long long)b * Q + q0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  load_tile(xa, core + row0 * DR_C, valid);
  load_tile(xc, query + row0 * DR_C, valid);
  __syncthreads();
  {
    f32x4 acc[2];
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xa, DR_LD, L.xo_w, 16, 0, w * 2, acc, lane, amax);
    store_tiles_lds<2>(acc, w * 2, L.xo_b, xc, DR_LD, xb, DR_LD, 0, false, lane);
  }
  __syncthreads();
  // xa <- x1, xc <- x1 + pos, global x1
  rows_layernorm(xb, L.n0_g, L.n0_b, eps, xa, xc, qpos + (long long)q0 * DR_C, x1 + row0 * DR_C, valid);
  __syncthreads();
  const int g = lane >> 4, j = lane & 15;
  {
    f32x4 acc[4];
    zero_acc(acc);
    rows_mm<H, 4, 16, 4>(xc, DR_LD, L.sa_in_w, 16, 0, w * 4, acc, lane, amax);      // q | k columns 0..511
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = (w * 4 + i) * 16 + j;
      const float bv = L.sa_in_b[col];
      const float sc = col < DR_C ? scale : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        if (r < valid) qkv[(row0 + r) * (3 * DR_C) + col] = (acc[i][e] + bv) * sc;
      }
    }
  }
  {
    f32x4 acc[2];
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xa, DR_LD, L.sa_in_w, 16, 0, 32 + w * 2, acc, lane, amax);  // v columns 512..767
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (32 + w * 2 + i) * 16 + j;
      const float bv = L.sa_in_b[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        if (r < valid) qkv[(row0 + r) * (3 * DR_C) + col] = acc[i][e] + bv;
      }
    }
  }
  if constexpr (H) rows_count_overflow(amax, overflow);
}

// ------------------------------------------------------------------------------------------------
// post
// ------------------------------------------------------------------------------------------------
template <bool H>
__global__ __launch_bounds__(DR_THREADS) void decoder_rows_post_kernel(
    pvsg_decoder_layer L, pvsg_decoder_head Hd, int has_layer, const float* __restrict__ next_q_w,
    const float* __restrict__ next_q_b, const float* __restrict__ x1, const float* __restrict__ qkv,
    const float* __restrict__ qpos, float* __restrict__ query_out, float* __restrict__ cls_out,
    float* __restrict__ emb_out, float* __restrict__ next_q_out, int Q, int tiles_per_b, float scale, float eps,
    float* __restrict__ ws_part, int* __restrict__ ws_count, int nspl, unsigned short* __restrict__ emb_pack,
    unsigned* __restrict__ flags_zero, unsigned* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float amax = 0.f;                         // H: largest |activation| this lane split into f16 limbs
  __shared__ int s_last;
  float* xa = smem;                       // [16][DR_LD]
  float* xb = xa + 16 * DR_LD;
  float* xc = xb + 16 * DR_LD;
  float* big = xc + 16 * DR_LD;           // attention probabilities [8][128][16]  /  FFN hidden chunk [16][DR_LDH]
  // nspl > 1: `nspl` workgroups per 16-row tile.  Each repeats the (cheap) self-attention part and computes ONE 256-wide slice
  // of the FFN hidden layer; the partial outputs meet in `ws_part`, and the workgroup that arrives last (ws_count, left at
  // zero again) sums them in slice order and carries on alone.  With Q = 100 rows and one clip the un-split kernel keeps 7
  // of 256 CUs busy for 126 us, 55 of them the FFN's matrix instructions.
  const int bt = nspl > 1 ? blockIdx.x / nspl : blockIdx.x, slice = nspl > 1 ? blockIdx.x - bt * nspl : 0;
  const int b = bt / tiles_per_b, tile = bt - b * tiles_per_b;
  const int q0 = tile * 16;
  const int valid = min(16, Q - q0);
  const long long row0 = (long long)b * Q + q0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;

  if (has_layer) {
    DR_STAMP(0);
    // ---- self-attention: wave = head ---------------------------------------------------------
    load_tile(xc, x1 + row0 * DR_C, valid);                    // x1 tile (identity of the self-attention)
#pragma unroll
    for (int it = 0; it < 2; ++it) {                           // scaled q rows of the tile -> xb
      const int idx = threadIdx.x + it * DR_THREADS;
      const int r = idx >> 6, c = (idx & 63) * 4;
      const float4 v = r < valid ? ld4(qkv + (row0 + r) * (3 * DR_C) + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(xb + r * DR_LD + c) = v;
    }
    __syncthreads();
    DR_STAMP(1);
    // one wave per head, on the matrix cores (rows_common.h: S = Q K^T and O = P V as 16 x 16 x 4 tiles, soft-max on the
    // accumulators); the scalar-FMA form this replaces took ~12 of the kernel's 87 us
    rows_attention_h32(xb, xa, DR_LD, big + w * (DR_MAXQ * 16), qkv + (long long)b * Q * (3 * DR_C), 3 * DR_C, DR_C, Q, w, lane);
    __syncthreads();
    DR_STAMP(2);
    // ---- out_proj + identity + LN -> x2 (xa) -------------------------------------------------
    {
      f32x4 acc[2];
      zero_acc(acc);
      rows_mm<H, 2, 16, 8>(xa, DR_LD, L.sa_out_w, 16, 0, w * 2, acc, lane, amax);
      store_tiles_lds<2>(acc, w * 2, L.sa_out_b, xc, DR_LD, xb, DR_LD, 0, false, lane);
    }
    __syncthreads();
    DR_STAMP(3);
    rows_layernorm(xb, L.n1_g, L.n1_b, eps, xa, nullptr, nullptr, nullptr, valid);
    __syncthreads();
    DR_STAMP(4);
    if (nspl > 1) {
      // ---- FFN, hidden units [256 slice, 256 slice + 256) ------------------------------------------------------
      const int wkc2 = L.ffn_dim >> 4;
      {
        f32x4 acc[2];
        zero_acc(acc);
        rows_mm<H, 2, 16, 8>(xa, DR_LD, L.f1_w, 16, 0, slice * 16 + w * 2, acc, lane, amax);
        store_tiles_lds<2>(acc, slice * 16 + w * 2, L.f1_b, nullptr, 0, big, DR_LD, slice * DR_C, true, lane);
      }
      __syncthreads();
      DR_STAMP(5);
      f32x4 yacc[2];
      zero_acc(yacc);
      rows_mm<H, 2, 16, 8>(big, DR_LD, L.f2_w, wkc2, slice * 16, w * 2, yacc, lane, amax);
      // The partial meets the other seven through memory WITHOUT a release fence: __threadfence() is buffer_wbl2 -- a write-back
      // of everything dirty in this XCD's 4 MB L2 (7.9 us here, and 12 us for the fence + loads on the reading side:
      // profiles/r06_rows_stamps.txt).  Instead the partial itself is stored / loaded at AGENT scope (relaxed atomics = sc1
      // accesses: written through the non-coherent L2, read past it), the stores are complete before the barrier that precedes
      // the arrival count, and the reader's loads are issued after the barrier that follows it.  Layout: the accumulator
      // fragments as the lanes hold them, [wave][tile i][lane][e] -- 8-byte accesses, 512 B per wave instruction.
      unsigned long long* part = reinterpret_cast<unsigned long long*>(ws_part + ((long long)bt * nspl + slice) * (16 * DR_C));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned long long bits = (unsigned long long)__float_as_uint(yacc[i][2 * hh]) |
                                          ((unsigned long long)__float_as_uint(yacc[i][2 * hh + 1]) << 32);
          __hip_atomic_store(part + ((w * 2 + i) * 64 + lane) * 2 + hh, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      DR_STAMP(6);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the write-through stores have been acknowledged
      __syncthreads();
      DR_STAMP(7);
      if (threadIdx.x == 0) {
        const int old = atomicAdd(ws_count + bt, 1);
        s_last = old == nspl - 1;
        if (s_last) ws_count[bt] = 0;                            // ready for the next launch
      }
      __syncthreads();
      if (!s_last) {
        if constexpr (H) rows_count_overflow(amax, overflow);
        return;
      }
      DR_STAMP(8);
      const unsigned long long* pall = reinterpret_cast<const unsigned long long*>(ws_part + (long long)bt * nspl * (16 * DR_C));
      f32x4 sum[2];
      zero_acc(sum);
      unsigned long long bits[DR_SPLIT][2][2];                   // nspl > 1 means nspl == DR_SPLIT: all 32 loads in flight at once
#pragma unroll
      for (int sl = 0; sl < DR_SPLIT; ++sl)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            bits[sl][i][hh] = __hip_atomic_load(pall + (long long)sl * (8 * DR_C) + ((w * 2 + i) * 64 + lane) * 2 + hh,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int sl = 0; sl < DR_SPLIT; ++sl)                      // slice order: the sum does not depend on who arrived when
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            sum[i][2 * hh] += __uint_as_float((unsigned)bits[sl][i][hh]);
            sum[i][2 * hh + 1] += __uint_as_float((unsigned)(bits[sl][i][hh] >> 32));
          }
      store_tiles_lds<2>(sum, w * 2, L.f2_b, xa, DR_LD, xb, DR_LD, 0, false, lane);     // + bias + x2 (identity) -> xb
    } else
    // ---- FFN in hidden chunks of 512 -----------------------------------------------------------
    {
      f32x4 yacc[2], ylo[2];                                     // ylo: the f16x2 form's second accumulator set, finished after the loop
      zero_acc(yacc);
      zero_acc(ylo);
      const int nchunk = L.ffn_dim / DR_FC, wkc2 = L.ffn_dim >> 4;
      for (int c = 0; c < nchunk; ++c) {
        {
          f32x4 acc[4];
          zero_acc(acc);
          rows_mm<H, 4, 16, 4>(xa, DR_LD, L.f1_w, 16, 0, c * 32 + w * 4, acc, lane, amax);
          store_tiles_lds<4>(acc, c * 32 + w * 4, L.f1_b, nullptr, 0, big, DR_LDH, c * DR_FC, true, lane);
        }
        __syncthreads();
        if constexpr (H) rows_gemm_h<2, DR_FC / 32, 6>(big, DR_LDH, L.f2_w, wkc2 >> 1, c * (DR_FC / 32), w * 2, yacc, ylo, lane, amax);
        else rows_gemm<2, DR_FC / 16, 8>(big, DR_LDH, L.f2_w, wkc2, c * (DR_FC / 16), w * 2, yacc, lane);
        __syncthreads();
      }
      if constexpr (H) rows_finish_h<2>(yacc, ylo, L.f2_w);
      store_tiles_lds<2>(yacc, w * 2, L.f2_b, xa, DR_LD, xb, DR_LD, 0, false, lane);
    }
    __syncthreads();
    DR_STAMP(9);
    // x3 -> xc (+ global), x3 + pos -> xb is done below after the head reads
    rows_layernorm(xb, L.n2_g, L.n2_b, eps, xc, nullptr, nullptr, query_out + row0 * DR_C, valid);
    __syncthreads();
  } else {
    load_tile(xc, x1 + row0 * DR_C, valid);                    // head only: x1 carries the queries
    __syncthreads();
  }
  // ---- forward_head, query side: p = post_norm(x3) -> xa; xb <- x3 + pos (for the next q projection) ----
  rows_layernorm(xc, Hd.pn_g, Hd.pn_b, eps, xa, nullptr, nullptr, nullptr, valid);
  {
    const int r = threadIdx.x >> 5, c0 = (threadIdx.x & 31) * 8;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      float4 v = *reinterpret_cast<const float4*>(xc + r * DR_LD + c0 + 4 * hh);
      if (r < valid) {
        const float4 p = ld4(qpos + (long long)(q0 + r) * DR_C + c0 + 4 * hh);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      *reinterpret_cast<float4*>(xb + r * DR_LD + c0 + 4 * hh) = v;
    }
  }
  __syncthreads();
  DR_STAMP(10);
  {  // class logits: one column tile per wave (classes + 1 <= 128)
    f32x4 acc[1];
    zero_acc(acc);
    rows_mm<H, 1, 16, 8>(xa, DR_LD, Hd.cls_w, 16, 0, w, acc, lane, amax);
    const int col = w * 16 + j;
    if (col < Hd.num_cls_out) {
      const float bv = Hd.cls_b[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        if (r < valid) cls_out[(row0 + r) * Hd.num_cls_out + col] = acc[0][e] + bv;
      }
    }
  }
  DR_STAMP(11);
  if (next_q_w) {  // next layer's cross-attention query: ((x3 + pos) Wq^T + bq) * scale
    f32x4 acc[2];
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xb, DR_LD, next_q_w, 16, 0, w * 2, acc, lane, amax);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (w * 2 + i) * 16 + j;
      const float bv = next_q_b[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        if (r < valid) next_q_out[(row0 + r) * DR_C + col] = (acc[i][e] + bv) * scale;
      }
    }
  }
  DR_STAMP(12);
  {  // mask_embed MLP: xa -> (relu) xc -> (relu) xb -> global
    f32x4 acc[2];
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xa, DR_LD, Hd.m0_w, 16, 0, w * 2, acc, lane, amax);
    __syncthreads();                                        // xb (next-q operand) and xc no longer read
    store_tiles_lds<2>(acc, w * 2, Hd.m0_b, nullptr, 0, xc, DR_LD, 0, true, lane);
    __syncthreads();
    DR_STAMP(13);
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xc, DR_LD, Hd.m1_w, 16, 0, w * 2, acc, lane, amax);
    store_tiles_lds<2>(acc, w * 2, Hd.m1_b, nullptr, 0, xb, DR_LD, 0, true, lane);
    __syncthreads();
    DR_STAMP(14);
    zero_acc(acc);
    rows_mm<H, 2, 16, 8>(xb, DR_LD, Hd.m2_w, 16, 0, w * 2, acc, lane, amax);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (w * 2 + i) * 16 + j;
      const float bv = Hd.m2_b[col];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const float v = acc[i][e] + bv;
        if (r < valid) emb_out[(row0 + r) * DR_C + col] = v;
        if (emb_pack) xc[r * DR_LD + col] = r < valid ? v : 0.f;     // xc: last read by the m1 GEMM, free since the barrier above
      }
    }
  }
  DR_STAMP(15);
  if (flags_zero && tile == 0 && threadIdx.x < 4) flags_zero[b * 4 + threadIdx.x] = 0u;   // the flag words the bits kernel ORs into
  if (emb_pack) {
    // The mask embeddings of these 16 queries as the ROW operand of the attention-mask-bits GEMM (csrc/split_conv1x1.h, f16x2
    // form: [k-tile 16][w_h | w_l][k-group 2][128 rows][8] f16), packed here instead of by f16x2_amax_kernel +
    // gemm_f16x2_pack_kernel + two zero_words launches per layer.  The bits only need the SIGN of embedding . feature, so each
    // query row gets its own exact power-of-two scale 2^e with max|row| 2^e in [2^13, 2^14): both limbs stay normal f16
    // numbers for any embedding magnitude, and no un-scaling is needed.  Rows Q..127 of the buffer are zero (zeroed once by
    // the caller; rows of the last tile beyond Q are written as zeros here).
    __syncthreads();
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int r = threadIdx.x >> 5, c0 = (threadIdx.x & 31) * 8;
    const float4 v0 = *reinterpret_cast<const float4*>(xc + r * DR_LD + c0);
    const float4 v1 = *reinterpret_cast<const float4*>(xc + r * DR_LD + c0 + 4);
    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) am = fmaxf(am, __builtin_fabsf(vv[i]));
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));     // the 32 lanes of a row are half a wave
    int ex = 0;
    if (am > 0.f && am < 3.0e38f) {
      ex = 13 - ilogbf(am);
      ex = ex > 126 ? 126 : (ex < -126 ? -126 : ex);
    }
    h8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sv = ldexpf(vv[i], ex);
      const _Float16 hh = (_Float16)sv;
      hi[i] = hh;
      lo[i] = (_Float16)(sv - (float)hh);
    }
    constexpr int NPAD = 128, LIMB = 2 * NPAD * 8;                             // f16 elements
    const int kt = c0 >> 4, kg = (c0 >> 3) & 1;
    unsigned short* dst = emb_pack + (long long)b * (2LL * NPAD * DR_C + 8) + (long long)kt * 2 * LIMB + ((long long)kg * NPAD + q0 + r) * 8;
    *reinterpret_cast<h8*>(dst) = hi;
    *reinterpret_cast<h8*>(dst + LIMB) = lo;
  }
  DR_STAMP(16);
  if constexpr (H) rows_count_overflow(amax, overflow);
}

constexpr size_t DR_POST_LDS = (size_t)(3 * 16 * DR_LD + 8 * DR_MAXQ * 16) * sizeof(float);
static_assert(8 * DR_MAXQ * 16 >= 16 * DR_LDH, "FFN chunk must fit in the attention scratch");

}  // namespace pvsg

// workspace of pvsg_decoder_rows_post: FFN partials (B x tiles x 8 x 16 x 256 floats) + one arrival counter per tile.  The
// CALLER zeroes it once; every launch leaves the counters at zero.  0 when the split is not used for this (B, Q).
extern "C" long long pvsg_decoder_rows_post_workspace_bytes(int B, int Q) {
  using namespace pvsg;
  if (B <= 0 || Q <= 0) return 0;
  const long long tiles = (long long)B * ((Q + 15) / 16);
  return tiles <= DR_SPLIT_MAX_TILES ? tiles * (DR_SPLIT * 16 * DR_C * 4 + 4) : 0;
}

extern "C" int pvsg_pack_rows_weight(const float* W, float* packed, int N, int K, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(W && packed, "pack_rows_weight: null pointer argument");
  PVSG_REQUIRE(N > 0 && K > 0 && K % 16 == 0, "pack_rows_weight: K must be a positive multiple of 16 (N=%d K=%d)", N, K);
  const long long total = (long long)((N + 15) / 16) * 16 * K;
  hipLaunchKernelGGL(pack_rows_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, packed,
                     N, K, total);
  PVSG_LAUNCH_CHECK("pack_rows_weight");
  return PVSG_OK;
}

#if PVSG_ROWS_STAMPS
extern "C" int pvsg_debug_rows_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pvsg::g_rows_stamps), sizeof(unsigned long long) * 8 * 32);
}
#endif

extern "C" long long pvsg_rows_f16x2_packed_floats(int N, int K) {
  return (N > 0 && K > 0 && K % 32 == 0) ? 4 + (long long)((N + 15) / 16) * 16 * K : 0;
}

extern "C" int pvsg_pack_rows_weight_f16x2(const float* W, float* packed, int N, int K, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(W && packed, "pack_rows_weight_f16x2: null pointer argument");
  PVSG_REQUIRE(N > 0 && K > 0 && K % 32 == 0, "pack_rows_weight_f16x2: K must be a positive multiple of 32 (N=%d K=%d)", N, K);
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(packed) & 15u), "pack_rows_weight_f16x2: packed must be 16-byte aligned");
  hipLaunchKernelGGL(rows_f16x2_header_kernel, dim3(1), dim3(1024), 0, stream, W, packed, (long long)N * K);
  const long long total = (long long)((N + 15) / 16) * (K / 32) * 256;
  hipLaunchKernelGGL(pack_rows_weight_f16x2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, W, packed,
                     N, K, total);
  PVSG_LAUNCH_CHECK("pack_rows_weight_f16x2");
  return PVSG_OK;
}

static int check_layer(const pvsg_decoder_layer* L, const char* who) {
  using namespace pvsg;
  PVSG_REQUIRE(L->xo_w && L->xo_b && L->n0_g && L->n0_b && L->sa_in_w && L->sa_in_b && L->sa_out_w && L->sa_out_b &&
                   L->n1_g && L->n1_b && L->f1_w && L->f1_b && L->f2_w && L->f2_b && L->n2_g && L->n2_b,
               "%s: null pointer in pvsg_decoder_layer", who);
  if (L->embed_dims != DR_C || L->num_heads != 8 || L->ffn_dim <= 0 || L->ffn_dim % DR_FC)
    return set_err(PVSG_ERR_UNSUPPORTED, "%s: built for 256 dims, 8 heads, FFN width a multiple of %d (got %d/%d/%d)",
                   who, DR_FC, L->embed_dims, L->num_heads, L->ffn_dim);
  return PVSG_OK;
}

template <bool H>
static int rows_pre_impl(const pvsg_decoder_layer* layer, const float* attn_core, const float* query, const float* query_pos,
                         float* x1, float* qkv, int B, int Q, unsigned* overflow, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(layer && attn_core && query && query_pos && x1 && qkv, "decoder_rows_pre: null pointer argument");
  PVSG_REQUIRE(B > 0 && Q > 0, "decoder_rows_pre: non-positive dimension");
  if (Q > DR_MAXQ) return set_err(PVSG_ERR_UNSUPPORTED, "decoder_rows_pre: at most %d queries (got %d)", DR_MAXQ, Q);
  if (int rc = check_layer(layer, "decoder_rows_pre")) return rc;
  const int tiles = (Q + 15) / 16;
  hipLaunchKernelGGL(decoder_rows_pre_kernel<H>, dim3(B * tiles), dim3(DR_THREADS), 0, stream, *layer, attn_core, query,
                     query_pos, x1, qkv, Q, tiles, 0.17677669529663687f, 1e-5f, overflow);
  PVSG_LAUNCH_CHECK("decoder_rows_pre");
  return PVSG_OK;
}

extern "C" int pvsg_decoder_rows_pre(const pvsg_decoder_layer* layer, const float* attn_core, const float* query,
                                     const float* query_pos, float* x1, float* qkv, int B, int Q,
                                     void* stream_) {
  return rows_pre_impl<false>(layer, attn_core, query, query_pos, x1, qkv, B, Q, nullptr, stream_);
}
extern "C" int pvsg_decoder_rows_pre_f16x2(const pvsg_decoder_layer* layer, const float* attn_core, const float* query,
                                           const float* query_pos, float* x1, float* qkv, int B, int Q,
                                           uint32_t* overflow, void* stream_) {
  return rows_pre_impl<true>(layer, attn_core, query, query_pos, x1, qkv, B, Q, overflow, stream_);
}

template <bool H>
static int rows_post_impl(const pvsg_decoder_layer* layer, const pvsg_decoder_head* head,
                          const float* next_q_w, const float* next_q_b, const float* x1,
                          const float* qkv, const float* query_pos, float* query_out, float* cls_out,
                          float* mask_embed_out, float* next_q_out, void* workspace, void* emb_pack_f16x2,
                          uint32_t* flags_zero, int B, int Q, unsigned* overflow, void* stream_) {
  using namespace pvsg;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PVSG_REQUIRE(head && x1 && query_pos && cls_out && mask_embed_out, "decoder_rows_post: null pointer argument");
  PVSG_REQUIRE(!layer || (qkv && query_out), "decoder_rows_post: a layer needs qkv and query_out");
  PVSG_REQUIRE((next_q_w == nullptr) == (next_q_b == nullptr) && (!next_q_w || next_q_out),
               "decoder_rows_post: next_q_w / next_q_b / next_q_out go together");
  PVSG_REQUIRE(B > 0 && Q > 0, "decoder_rows_post: non-positive dimension");
  PVSG_REQUIRE(!(reinterpret_cast<uintptr_t>(emb_pack_f16x2) & 15u), "decoder_rows_post: emb_pack_f16x2 must be 16-byte aligned");
  PVSG_REQUIRE(head->pn_g && head->pn_b && head->cls_w && head->cls_b && head->m0_w && head->m0_b && head->m1_w &&
                   head->m1_b && head->m2_w && head->m2_b, "decoder_rows_post: null pointer in pvsg_decoder_head");
  if (Q > DR_MAXQ) return set_err(PVSG_ERR_UNSUPPORTED, "decoder_rows_post: at most %d queries (got %d)", DR_MAXQ, Q);
  if (head->num_cls_out <= 0 || head->num_cls_out > 128)
    return set_err(PVSG_ERR_UNSUPPORTED, "decoder_rows_post: 1..128 class outputs (got %d)", head->num_cls_out);
  pvsg_decoder_layer L;
  memset(&L, 0, sizeof(L));
  if (layer) {
    if (int rc = check_layer(layer, "decoder_rows_post")) return rc;
    L = *layer;
  }
  const int tiles = (Q + 15) / 16;
  static std::atomic<unsigned long long> attr_done;
  {
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&decoder_rows_post_kernel<H>), (int)DR_POST_LDS, attr_done);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "decoder_rows_post: LDS attribute: %s", hipGetErrorString(e));
  }
  // few row tiles (a clip, or a handful of frames): split the FFN over DR_SPLIT workgroups per tile (needs the workspace)
  const int nspl = (layer && workspace && B * tiles <= DR_SPLIT_MAX_TILES && L.ffn_dim == DR_SPLIT * DR_C) ? DR_SPLIT : 1;
  float* ws_part = static_cast<float*>(workspace);
  int* ws_count = workspace ? reinterpret_cast<int*>(ws_part + (size_t)B * tiles * DR_SPLIT * 16 * DR_C) : nullptr;
  hipLaunchKernelGGL(decoder_rows_post_kernel<H>, dim3(B * tiles * nspl), dim3(DR_THREADS), DR_POST_LDS, stream, L, *head,
                     layer ? 1 : 0, next_q_w, next_q_b, x1, qkv, query_pos, query_out, cls_out, mask_embed_out,
                     next_q_out, Q, tiles, 0.17677669529663687f, 1e-5f, ws_part, ws_count, nspl,
                     static_cast<unsigned short*>(emb_pack_f16x2), flags_zero, overflow);
  PVSG_LAUNCH_CHECK("decoder_rows_post");
  return PVSG_OK;
}

extern "C" int pvsg_decoder_rows_post(const pvsg_decoder_layer* layer, const pvsg_decoder_head* head,
                                      const float* next_q_w, const float* next_q_b, const float* x1,
                                      const float* qkv, const float* query_pos, float* query_out, float* cls_out,
                                      float* mask_embed_out, float* next_q_out, void* workspace, void* emb_pack_f16x2,
                                      uint32_t* flags_zero, int B, int Q, void* stream_) {
  return rows_post_impl<false>(layer, head, next_q_w, next_q_b, x1, qkv, query_pos, query_out, cls_out, mask_embed_out, next_q_out,
                               workspace, emb_pack_f16x2, flags_zero, B, Q, nullptr, stream_);
}
extern "C" int pvsg_decoder_rows_post_f16x2(const pvsg_decoder_layer* layer, const pvsg_decoder_head* head,
                                            const float* next_q_w, const float* next_q_b, const float* x1,
                                            const float* qkv, const float* query_pos, float* query_out, float* cls_out,
                                            float* mask_embed_out, float* next_q_out, void* workspace, void* emb_pack_f16x2,
                                            uint32_t* flags_zero, int B, int Q, uint32_t* overflow, void* stream_) {
  return rows_post_impl<true>(layer, head, next_q_w, next_q_b, x1, qkv, query_pos, query_out, cls_out, mask_embed_out, next_q_out,
                              workspace, emb_pack_f16x2, flags_zero, B, Q, overflow, stream_);
}
