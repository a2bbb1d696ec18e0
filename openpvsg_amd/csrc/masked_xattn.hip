// Masked cross-attention of the Mask2Former transformer decoder, streaming (flash-style), exact
// f32 on the gfx950 matrix cores.
//
// Replaces: [3P] mmcv MultiheadAttention -> nn.MultiheadAttention(attn_mask=bool (B*8,Q,K)) as called
//   by the decoder loop, models/mask2former/mask2former_head.py:457-468 and
//   models/mask2former_vps/mask2former_video_head.py:435-446 (keys = T*h*w, up to 471 040 at
//   T=32 / stride 8 / 720p), together with the all-masked-row reset at :453-454 / :431-432.
// The reference materialises (B*8, Q, K) logits (1.5 GB) and the bool mask (0.38 GB); here the
// keys are streamed once, the mask is one bit per (query, key) shared by the heads
// (mask_gemm.hip) and the reset is a per-query flag test.
//
// Work split: a workgroup = one (batch element, key range); its 8 waves are the 8 heads, so the
// workgroup consumes whole 1 KiB key/value rows.  Each wave keeps its head's Q (100x32, padded to
// 7 tiles of 16 rows) in registers for the whole range and walks the keys 16 at a time:
//   S^T = K_tile (16x32) . Q^T          56 x v_mfma_f32_16x16x4_f32   (A = K from HBM, B = Q regs)
//   mask bits, running max / sum        per query column, 2 cross-lane steps (rows live in 4 lane groups)
//   O^T += V_tile^T (32x16) . P^T       56 x v_mfma; the S^T accumulator registers ARE the B operand
//                                       (key index of k-step r = 4*(lane>>4)+r on both sides), so P
//                                       never moves between lanes or through LDS
// O^T keeps each lane's values in ONE query column, so the online-softmax rescale is lane-local.
// K fragments are 2 x 16 B and V fragments 4 x 8 B per lane per tile, every 128 B head row is
// consumed entirely by one wave instruction pair (full cache lines, each HBM byte read once).
// Ranges are combined by `xattn_combine_kernel` (log-sum-exp merge); the same partial format is
// what ranks exchange when a clip's frames are sharded over GPUs (openpvsg_amd/parallel.py).
#include "common.h"

namespace pvsg {

constexpr int XQT = 7;  // 7 x 16 = 112 query rows

__device__ __forceinline__ float group_max4(float v) {  // reduce over the 4 lane groups (lane>>4)
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum4(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

__global__ __launch_bounds__(512) void xattn_partial_kernel(
    const float* __restrict__ qp, const float* __restrict__ kp, const float* __restrict__ vp,
    const uint32_t* __restrict__ bits, const uint32_t* __restrict__ flags, float* __restrict__ part_o,
    float* __restrict__ part_ml, int Q, long long K, int NS, long long chunk) {
  constexpr int HD = 256, D = 32, M = 8;
  const int b = blockIdx.x / NS, s = blockIdx.x - b * NS;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const long long k0 = (long long)s * chunk;
  const long long k1 = (k0 + chunk < K) ? k0 + chunk : K;

  // ---- Q fragments (B operand of S^T = K.Q^T): row q = qt*16+j, d = g*8 + step ---------------
  float qf[XQT][8];
  uint32_t honor = 0u;  // bit qt: this lane's query in tile qt honours the mask
  {
    uint32_t fw[4] = {0u, 0u, 0u, 0u};
    if (bits != nullptr) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fw[k] = flags[b * 4 + k];
    }
#pragma unroll
    for (int qt = 0; qt < XQT; ++qt) {
      const int q = qt * 16 + j;
      if (q < Q) {
        const float* p = qp + ((long long)b * Q + q) * HD + h * D + g * 8;
        const float4 a = ld4(p), c = ld4(p + 4);
        qf[qt][0] = a.x; qf[qt][1] = a.y; qf[qt][2] = a.z; qf[qt][3] = a.w;
        qf[qt][4] = c.x; qf[qt][5] = c.y; qf[qt][6] = c.z; qf[qt][7] = c.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[qt][i] = 0.f;
      }
      // a query whose mask blocks EVERY key attends to all keys instead (flag bit = has an
      // allowed key somewhere, over all ranges / ranks)
      if (bits != nullptr && ((fw[qt >> 1] >> ((qt & 1) * 16 + j)) & 1u)) honor |= 1u << qt;
    }
  }

  float mrun[XQT], lrun[XQT];
  f32x4 o[XQT][2];
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    mrun[qt] = -INFINITY;
    lrun[qt] = 0.f;
    o[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    o[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const float* kb = kp + (long long)b * K * HD + h * D;
  const float* vb = vp + (long long)b * K * HD + h * D;
  const uint32_t* mb = bits ? bits + (long long)b * K * 4 : nullptr;

  for (long long kt = k0; kt < k1; kt += 16) {
    // ---- loads for this tile ------------------------------------------------------------------
    const long long ka = kt + j;  // row of the K fragment held by this lane
    float kf[8];
    if (ka < k1) {
      const float* p = kb + ka * HD + g * 8;
      const float4 a = ld4(p), c = ld4(p + 4);
      kf[0] = a.x; kf[1] = a.y; kf[2] = a.z; kf[3] = a.w;
      kf[4] = c.x; kf[5] = c.y; kf[6] = c.z; kf[7] = c.w;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) kf[i] = 0.f;
    }
    float2 vf[4];
    uint4 mw[4];
    bool kvalid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long kr = kt + g * 4 + r;  // key of S^T row (g*4+r) == key of PV k-step r
      kvalid[r] = kr < k1;
      vf[r] = kvalid[r] ? *reinterpret_cast<const float2*>(vb + kr * HD + 2 * j) : make_float2(0.f, 0.f);
      mw[r] = (mb != nullptr && kvalid[r]) ? *reinterpret_cast<const uint4*>(mb + kr * 4)
                                           : make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- S^T = K . Q^T ------------------------------------------------------------------------
    f32x4 st[XQT];
#pragma unroll
    for (int qt = 0; qt < XQT; ++qt) st[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int qt = 0; qt < XQT; ++qt)
        st[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[qt][i], st[qt], 0, 0, 0);

    // ---- mask + online softmax (per query column) ---------------------------------------------
#pragma unroll
    for (int qt = 0; qt < XQT; ++qt) {
      const bool hq = (honor >> qt) & 1u;
      const int sh = (qt & 1) * 16 + j;
      float sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t w = (qt >> 1) == 0 ? mw[r].x : (qt >> 1) == 1 ? mw[r].y : (qt >> 1) == 2 ? mw[r].z : mw[r].w;
        const bool masked = !kvalid[r] || (hq && ((w >> sh) & 1u));
        sv[r] = masked ? -INFINITY : st[qt][r];
      }
      const float tmax = group_max4(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
      const float mnew = fmaxf(mrun[qt], tmax);
      float alpha = 1.f, psum = 0.f;
      if (mnew == -INFINITY) {  // nothing allowed so far for this query
#pragma unroll
        for (int r = 0; r < 4; ++r) st[qt][r] = 0.f;
      } else {
        alpha = __expf(mrun[qt] - mnew);  // exp(-inf) = 0 on first hit
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __expf(sv[r] - mnew);
          st[qt][r] = p;
          psum += p;
        }
      }
      psum = group_sum4(psum);
      lrun[qt] = lrun[qt] * alpha + psum;
      mrun[qt] = mnew;
      o[qt][0] *= alpha;
      o[qt][1] *= alpha;
    }

    // ---- O^T += V^T . P^T ---------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int qt = 0; qt < XQT; ++qt) {
        o[qt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r].x, st[qt][r], o[qt][0], 0, 0, 0);
        o[qt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r].y, st[qt][r], o[qt][1], 0, 0, 0);
      }
  }

  // ---- write the un-normalised partial (o, m, l) ------------------------------------------------
  // lane holds O[q = qt*16+j][d = 8g + 2r' + dt]  -> 8 consecutive floats
  const long long slot = ((long long)b * NS + s) * M + h;
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    const int q = qt * 16 + j;
    if (q < Q) {
      float* op = part_o + (slot * Q + q) * D + g * 8;
      st4(op, make_float4(o[qt][0][0], o[qt][1][0], o[qt][0][1], o[qt][1][1]));
      st4(op + 4, make_float4(o[qt][0][2], o[qt][1][2], o[qt][0][3], o[qt][1][3]));
      if (g == 0) {
        float* mp = part_ml + (slot * Q + q) * 2;
        *reinterpret_cast<float2*>(mp) = make_float2(mrun[qt], lrun[qt]);
      }
    }
  }
}

// out[b, q, h*32+d] = sum_s e^{m_s - m*} o_s / sum_s e^{m_s - m*} l_s     (m* = max_s m_s)
// One block per (q, b); thread = (range lane sl = tid>>8 ... ) see below: 8 heads x 32 dims x 4
// range lanes; every range lane walks s = sl, sl+4, ... with an online (m, num, den) triple and the
// 4 triples are merged through LDS.
__global__ __launch_bounds__(1024) void xattn_combine_kernel(const float* __restrict__ part_o,
                                                            const float* __restrict__ part_ml,
                                                            float* __restrict__ out, int Q, int NS) {
  constexpr int D = 32, M = 8, SL = 4;
  __shared__ float sm[SL][M * D], sn[SL][M * D], sd[SL][M * D];
  const int b = blockIdx.y, q = blockIdx.x;
  const int hd = threadIdx.x & 255, sl = threadIdx.x >> 8;
  const int h = hd >> 5, d = hd & 31;
  float m = -INFINITY, num = 0.f, den = 0.f;
  for (int s = sl; s < NS; s += SL) {
    const long long slot = ((long long)b * NS + s) * M + h;
    const float2 ml = *reinterpret_cast<const float2*>(part_ml + (slot * Q + q) * 2);
    const float ov = part_o[(slot * Q + q) * D + d];
    if (ml.x == -INFINITY) continue;
    const float mn = fmaxf(m, ml.x);
    const float a = __expf(m - mn), w = __expf(ml.x - mn);
    num = num * a + w * ov;
    den = den * a + w * ml.y;
    m = mn;
  }
  sm[sl][hd] = m; sn[sl][hd] = num; sd[sl][hd] = den;
  __syncthreads();
  if (sl == 0) {
    float mstar = sm[0][hd];
#pragma unroll
    for (int i = 1; i < SL; ++i) mstar = fmaxf(mstar, sm[i][hd]);
    float n2 = 0.f, d2 = 0.f;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      if (sm[i][hd] == -INFINITY) continue;
      const float w = __expf(sm[i][hd] - mstar);
      n2 += w * sn[i][hd];
      d2 += w * sd[i][hd];
    }
    out[((long long)b * Q + q) * (M * D) + hd] = n2 / d2;
  }
}

}  // namespace pvsg

extern "C" int pvsg_xattn_num_splits(int B, long long K) {
  long long ns = (256 + B - 1) / B;
  const long long maxs = (K + 255) / 256;
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  return (int)ns;
}

extern "C" int pvsg_masked_xattn_partial(const float* q_proj, const float* k_proj, const float* v_proj,
                                         const uint32_t* mask_bits, const uint32_t* mask_flags,
                                         float* part_o, float* part_ml, int B, int Q, long long K,
                                         int M, int D, int NS, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(q_proj && k_proj && v_proj && part_o && part_ml, "masked_xattn_partial: null pointer argument");
  PVSG_REQUIRE((mask_bits == nullptr) == (mask_flags == nullptr),
               "masked_xattn_partial: mask_bits and mask_flags must be given together");
  PVSG_REQUIRE(B > 0 && Q > 0 && K > 0 && NS > 0, "masked_xattn_partial: non-positive dimension");
  if (M != 8 || D != 32 || Q > XQT * 16)
    return set_err(PVSG_ERR_UNSUPPORTED, "masked_xattn_partial: built for 8 heads x 32 dims, Q<=112 (got M=%d D=%d Q=%d)", M, D, Q);
  PVSG_REQUIRE(((reinterpret_cast<uintptr_t>(q_proj) | reinterpret_cast<uintptr_t>(k_proj) |
                 reinterpret_cast<uintptr_t>(v_proj) | reinterpret_cast<uintptr_t>(part_o) |
                 reinterpret_cast<uintptr_t>(mask_bits)) & 15u) == 0,
               "masked_xattn_partial: pointers must be 16-byte aligned");
  long long chunk = (K + NS - 1) / NS;
  chunk = (chunk + 15) / 16 * 16;
  // ranges past the end are legal: they publish (m=-inf, l=0, o=0) and the merge skips them
  hipLaunchKernelGGL(xattn_partial_kernel, dim3(B * NS), dim3(512), 0, stream, q_proj, k_proj, v_proj,
                     mask_bits, mask_flags, part_o, part_ml, Q, K, NS, chunk);
  PVSG_LAUNCH_CHECK("masked_xattn_partial");
  return PVSG_OK;
}

extern "C" int pvsg_xattn_combine(const float* part_o, const float* part_ml, float* out, int B, int Q,
                                  int M, int D, int NS, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(part_o && part_ml && out, "xattn_combine: null pointer argument");
  PVSG_REQUIRE(B > 0 && Q > 0 && NS > 0, "xattn_combine: non-positive dimension");
  if (M != 8 || D != 32)
    return set_err(PVSG_ERR_UNSUPPORTED, "xattn_combine: built for 8 heads x 32 dims");
  hipLaunchKernelGGL(xattn_combine_kernel, dim3(Q, B), dim3(1024), 0, stream, part_o, part_ml, out, Q, NS);
  PVSG_LAUNCH_CHECK("xattn_combine");
  return PVSG_OK;
}
