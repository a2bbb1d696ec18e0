// Multi-scale deformable attention forward (sampling core) for gfx950.
//
// Replaces: mmcv-full 1.4.0 `ext_module.ms_deform_attn_forward`
//   (mmcv/ops/multi_scale_deform_attn.py -> ms_deformable_im2col_gpu_kernel; third-party,
//    not under /root/reference) as selected by
//    configs/mask2former/..._custom_single_video_test.py:46-56 and reached from
//    models/mask2former/mask2former_head.py:417 (pixel decoder, 6x per frame).
//
// Semantics, for every (b, q, m):
//   out[b,q,m,:] = sum_l sum_p w[b,q,m,l,p] * bilinear(value_l[b,:,:,m,:],
//                      x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5)   (zeros outside)
//
// MI355X mapping.  The CUDA original runs one THREAD per output scalar, so a
// 32-wide warp re-derives the same 4 tap addresses 32 times and every tap is a
// 4-byte read.  Here one 64-lane WAVE owns one query and all 8 heads:
//   lane = (head m = lane>>3, channel quad c4 = lane&7)
// so each bilinear tap is one 16-byte load per lane and 8 lanes cover a head's
// whole 128-byte channel row (one full cache line per tap, 8 lines per
// instruction).  The wave's result is one contiguous 1 KiB row of `out`.
// Locations/weights are read as float4 broadcast loads (8 lanes share an
// address).  Blocks are remapped so each XCD walks a contiguous query range:
// neighbouring queries sample neighbouring rows, which keeps the per-XCD L2
// working set to a thin band of each level.
#include "common.h"

namespace pvsg {

template <int L, int P>
__global__ __launch_bounds__(256) void msda_fwd_m8d32(
    const float* __restrict__ value, const long long* __restrict__ shapes,
    const long long* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, float* __restrict__ out, int S, int Lq,
    long long nq_total, unsigned nblk) {
  constexpr int M = 8, D = 32;
  const unsigned lb = xcd_contiguous_block(blockIdx.x, nblk);
  const long long gq = (long long)lb * 4 + (threadIdx.x >> 6);  // global query id = b*Lq + q
  if (gq >= nq_total) return;
  const int lane = threadIdx.x & 63;
  const int m = lane >> 3, c4 = lane & 7;
  const int b = (int)(gq / Lq);

  const float* locp = loc + ((gq * M + m) * (long long)(L * P * 2));
  const float* wp = attw + ((gq * M + m) * (long long)(L * P));
  const float* vbase = value + (long long)b * S * (M * D) + m * D + c4 * 4;

  float lx[L * P], ly[L * P], aw[L * P];
  if constexpr ((P * 2) % 4 == 0) {
#pragma unroll
    for (int i = 0; i < L * P * 2 / 4; ++i) {
      const float4 t = ld4(locp + 4 * i);
      lx[2 * i] = t.x; ly[2 * i] = t.y; lx[2 * i + 1] = t.z; ly[2 * i + 1] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < L * P; ++i) { lx[i] = locp[2 * i]; ly[i] = locp[2 * i + 1]; }
  }
  if constexpr ((L * P) % 4 == 0) {
#pragma unroll
    for (int i = 0; i < L * P / 4; ++i) {
      const float4 t = ld4(wp + 4 * i);
      aw[4 * i] = t.x; aw[4 * i + 1] = t.y; aw[4 * i + 2] = t.z; aw[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < L * P; ++i) aw[i] = wp[i];
  }

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vbase + lsi[l] * (long long)(M * D);
    float4 v[P][4];
    float cw[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float him = ly[l * P + p] * (float)H - 0.5f;
      const float wim = lx[l * P + p] * (float)W - 0.5f;
      const float hf = floorf(him), wf = floorf(wim);
      const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
      // int conversion of an out-of-range float is clamped first so NaN/inf
      // locations cannot index outside the level (they get weight 0 below).
      const float hfc = fminf(fmaxf(hf, -2.f), (float)H + 1.f);
      const float wfc = fminf(fmaxf(wf, -2.f), (float)W + 1.f);
      const int h0 = (int)hfc, w0 = (int)wfc, h1 = h0 + 1, w1 = w0 + 1;
      const bool inside = (him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W);
      const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
      const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
      const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h1, 0), H - 1);
      const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w1, 0), W - 1);
      cw[p][0] = (vh0 && vw0) ? hh * hw : 0.f;
      cw[p][1] = (vh0 && vw1) ? hh * lw : 0.f;
      cw[p][2] = (vh1 && vw0) ? lh * hw : 0.f;
      cw[p][3] = (vh1 && vw1) ? lh * lw : 0.f;
      v[p][0] = ld4(vl + (long long)(h0c * W + w0c) * (M * D));
      v[p][1] = ld4(vl + (long long)(h0c * W + w1c) * (M * D));
      v[p][2] = ld4(vl + (long long)(h1c * W + w0c) * (M * D));
      v[p][3] = ld4(vl + (long long)(h1c * W + w1c) * (M * D));
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      // attention weight applied after the 4-tap sum (col += bilinear * weight)
      const float a = aw[l * P + p];
      float4 s;
      s.x = cw[p][0] * v[p][0].x + cw[p][1] * v[p][1].x + cw[p][2] * v[p][2].x + cw[p][3] * v[p][3].x;
      s.y = cw[p][0] * v[p][0].y + cw[p][1] * v[p][1].y + cw[p][2] * v[p][2].y + cw[p][3] * v[p][3].y;
      s.z = cw[p][0] * v[p][0].z + cw[p][1] * v[p][1].z + cw[p][2] * v[p][2].z + cw[p][3] * v[p][3].z;
      s.w = cw[p][0] * v[p][0].w + cw[p][1] * v[p][1].w + cw[p][2] * v[p][2].w + cw[p][3] * v[p][3].w;
      acc.x += a * s.x; acc.y += a * s.y; acc.z += a * s.z; acc.w += a * s.w;
    }
  }
  st4(out + gq * (M * D) + m * D + c4 * 4, acc);
}

// The same op with the per-tap arithmetic shared by the eight lanes of a head (see msda_sample_query_coop below: lane c4 of a
// head works out taps c4 and 8 + c4 -- location, bilinear weights, 32-bit byte offsets of the four corners -- and the gather
// loop fetches each tap's nine numbers from its owner lane with ds_bpermute).  848 -> ~500 VALU instructions per wave; same
// per-tap formulas and summation order.  Needs one image's value rows within 2 GB (checked by the caller).
template <int L, int P>
__global__ __launch_bounds__(256) void msda_fwd_m8d32_coop(
    const float* __restrict__ value, const long long* __restrict__ shapes,
    const long long* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, float* __restrict__ out, int S, int Lq,
    long long nq_total, unsigned nblk) {
  constexpr int M = 8, D = 32, LP = L * P, NT = (LP + 7) / 8;     // NT taps per lane
  static_assert(LP <= 16, "at most two taps per lane");
  const unsigned lb = xcd_contiguous_block(blockIdx.x, nblk);
  const long long gq = (long long)lb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (gq >= nq_total) return;
  const int lane = threadIdx.x & 63;
  const int m = lane >> 3, c4 = lane & 7;
  const int b = (int)(gq / Lq);
  const float* locp = loc + ((gq * M + m) * (long long)(LP * 2));
  const float* wp = attw + ((gq * M + m) * (long long)LP);
  int own_off[NT][4];
  float own_cw[NT][4], own_a[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int p = min(k * 8 + c4, LP - 1);                         // lanes without this tap repeat the last one (never read)
    const int l = p / P;
    const float2 xy = *reinterpret_cast<const float2*>(locp + 2 * p);
    own_a[k] = wp[p];
    int H = (int)shapes[0], W = (int)shapes[1];
    long long base = lsi[0];
#pragma unroll
    for (int j = 1; j < L; ++j)
      if (l == j) { H = (int)shapes[2 * j]; W = (int)shapes[2 * j + 1]; base = lsi[j]; }
    const float him = xy.y * (float)H - 0.5f, wim = xy.x * (float)W - 0.5f;
    const float hf = floorf(him), wf = floorf(wim);
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const float hfc = fminf(fmaxf(hf, -2.f), (float)H + 1.f);
    const float wfc = fminf(fmaxf(wf, -2.f), (float)W + 1.f);
    const int h0 = (int)hfc, w0 = (int)wfc, h1 = h0 + 1, w1 = w0 + 1;
    const bool inside = (him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W);
    const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
    const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
    const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h1, 0), H - 1);
    const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w1, 0), W - 1);
    own_cw[k][0] = (vh0 && vw0) ? hh * hw : 0.f;
    own_cw[k][1] = (vh0 && vw1) ? hh * lw : 0.f;
    own_cw[k][2] = (vh1 && vw0) ? lh * hw : 0.f;
    own_cw[k][3] = (vh1 && vw1) ? lh * lw : 0.f;
    const int b0 = (int)base;
    own_off[k][0] = (b0 + h0c * W + w0c) * (M * D * 4);
    own_off[k][1] = (b0 + h0c * W + w1c) * (M * D * 4);
    own_off[k][2] = (b0 + h1c * W + w0c) * (M * D * 4);
    own_off[k][3] = (b0 + h1c * W + w1c) * (M * D * 4);
  }
  const char* vbase = reinterpret_cast<const char*>(value + (long long)b * S * (M * D));
  const int lane_off = (m * D + c4 * 4) * 4;
  const int grp = (lane & ~7) << 2;                                  // ds_bpermute takes byte addresses (lane * 4)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int k = p >> 3, src = grp + ((p & 7) << 2);
    int off[4];
    float cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      off[c] = __builtin_amdgcn_ds_bpermute(src, own_off[k][c]);
      cw[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_cw[k][c])));
    }
    const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_a[k])));
    float4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(vbase + (unsigned)(off[c] + lane_off));
    float4 s4;
    s4.x = cw[0] * v[0].x + cw[1] * v[1].x + cw[2] * v[2].x + cw[3] * v[3].x;
    s4.y = cw[0] * v[0].y + cw[1] * v[1].y + cw[2] * v[2].y + cw[3] * v[3].y;
    s4.z = cw[0] * v[0].z + cw[1] * v[1].z + cw[2] * v[2].z + cw[3] * v[3].z;
    s4.w = cw[0] * v[0].w + cw[1] * v[1].w + cw[2] * v[2].w + cw[3] * v[3].w;
    acc.x += a * s4.x; acc.y += a * s4.y; acc.z += a * s4.z; acc.w += a * s4.w;
  }
  st4(out + gq * (M * D) + m * D + c4 * 4, acc);
}

// Shape-generic path (any M, D, L, P): one lane per output scalar.  Only the
// R50 configs' (M=8, D=32) shape is tuned; this keeps the op total.
__global__ __launch_bounds__(256) void msda_fwd_generic(
    const float* __restrict__ value, const long long* __restrict__ shapes,
    const long long* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attw, float* __restrict__ out, int S, int Lq, int M, int D,
    int L, int P, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int m = (int)((i / D) % M);
    const long long gq = i / ((long long)D * M);
    const int b = (int)(gq / Lq);
    const float* locp = loc + (gq * M + m) * (long long)(L * P * 2);
    const float* wp = attw + (gq * M + m) * (long long)(L * P);
    const float* vb = value + (long long)b * S * M * D + m * D + d;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float* vl = vb + lsi[l] * (long long)(M * D);
      for (int p = 0; p < P; ++p) {
        const float him = locp[(l * P + p) * 2 + 1] * (float)H - 0.5f;
        const float wim = locp[(l * P + p) * 2] * (float)W - 0.5f;
        if (!((him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W))) continue;
        const float hf = floorf(him), wf = floorf(wim);
        const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
        const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (h0 >= 0 && w0 >= 0) v1 = vl[(long long)(h0 * W + w0) * M * D];
        if (h0 >= 0 && w1 <= W - 1) v2 = vl[(long long)(h0 * W + w1) * M * D];
        if (h1 <= H - 1 && w0 >= 0) v3 = vl[(long long)(h1 * W + w0) * M * D];
        if (h1 <= H - 1 && w1 <= W - 1) v4 = vl[(long long)(h1 * W + w1) * M * D];
        acc += wp[l * P + p] * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
      }
    }
    out[i] = acc;
  }
}

}  // namespace pvsg

extern "C" int pvsg_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes,
                                           const int64_t* level_start_index,
                                           const float* sampling_loc, const float* attn_weight,
                                           float* out, int B, int S, int M, int D, int Lq, int L,
                                           int P, int im2col_step, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
               "ms_deform_attn_forward: null pointer argument");
  PVSG_REQUIRE(B > 0 && S > 0 && M > 0 && D > 0 && Lq > 0 && L > 0 && P > 0,
               "ms_deform_attn_forward: non-positive dimension (B=%d S=%d M=%d D=%d Lq=%d L=%d P=%d)",
               B, S, M, D, Lq, L, P);
  PVSG_REQUIRE(im2col_step > 0, "ms_deform_attn_forward: im2col_step must be positive");
  {
    // same divisibility contract as the mmcv op: batch % min(batch, im2col_step) == 0
    const int step = B < im2col_step ? B : im2col_step;
    PVSG_REQUIRE(B % step == 0, "ms_deform_attn_forward: batch(%d) must divide im2col_step(%d)", B,
                 step);
  }
  const long long nq = (long long)B * Lq;
  const long long* sh = reinterpret_cast<const long long*>(spatial_shapes);
  const long long* ls = reinterpret_cast<const long long*>(level_start_index);
  const bool aligned = ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(sampling_loc) |
                         reinterpret_cast<uintptr_t>(attn_weight) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  if (M == 8 && D == 32 && aligned && ((L == 3 && P == 4) || (L == 4 && P == 4) || (L == 1 && P == 4))) {
    const long long nblk_ll = (nq + 3) / 4;
    PVSG_REQUIRE(nblk_ll < (1ll << 31), "ms_deform_attn_forward: too many queries");
    const unsigned nblk = (unsigned)nblk_ll;
    const char* cp = getenv("PVSG_MSDA_COOP");                       // =0: every lane of a head computes every tap (A/B)
    const bool coop = (long long)S * M * D * 4 < (1ll << 31) && !(cp && cp[0] == '0');
    if (coop && L == 3)
      hipLaunchKernelGGL((msda_fwd_m8d32_coop<3, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
    else if (coop && L == 4)
      hipLaunchKernelGGL((msda_fwd_m8d32_coop<4, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
    else if (coop)
      hipLaunchKernelGGL((msda_fwd_m8d32_coop<1, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
    else if (L == 3)
      hipLaunchKernelGGL((msda_fwd_m8d32<3, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
    else if (L == 4)
      hipLaunchKernelGGL((msda_fwd_m8d32<4, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
    else
      hipLaunchKernelGGL((msda_fwd_m8d32<1, 4>), dim3(nblk), dim3(256), 0, stream, value, sh, ls,
                         sampling_loc, attn_weight, out, S, Lq, nq, nblk);
  } else {
    const long long total = nq * M * D;
    long long nb = (total + 255) / 256;
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(msda_fwd_generic, dim3((unsigned)nb), dim3(256), 0, stream, value, sh, ls,
                       sampling_loc, attn_weight, out, S, Lq, M, D, L, P, total);
  }
  PVSG_LAUNCH_CHECK("ms_deform_attn_forward");
  return PVSG_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused form used by the pixel decoder (a1 + the elementwise work around it):
//   inputs are the RAW outputs of ONE projection GEMM per layer,  y = x [Wv | Woff | Watt]^T :
//     value      = y[..., 0:256]            (row stride = 544 floats: no copy)
//     oa         = y[..., 256:544] + pos_oa (pos_oa = pos [Woff | Watt]^T + b, cached per layer/shape)
//   the kernel does the softmax over the 12 (level, point) logits of its head, forms
//   loc = ref + off / (W_l, H_l) and samples -- i.e. mmcv MultiScaleDeformableAttention.forward
//   steps 4-6 without materialising offsets, weights or locations
//   (1.2 KB/query written + read by the un-fused path) and without the `query + query_pos` pass.
// ------------------------------------------------------------------------------------------------
namespace pvsg {

template <int L, int P>
__device__ __forceinline__ float4 msda_sample_query(const float* __restrict__ value, long long value_stride,
                                                    const float* __restrict__ oa, long long oa_stride,
                                                    const float* __restrict__ pos_oa, const float* __restrict__ ref,
                                                    const long long* __restrict__ shapes,
                                                    const long long* __restrict__ lsi, int S, int Lq, long long gq,
                                                    int lane) {
  constexpr int M = 8, D = 32, LP = L * P;
  const int m = lane >> 3, c4 = lane & 7;
  const int b = (int)(gq / Lq);
  const int q = (int)(gq - (long long)b * Lq);
  const float* offp = oa + gq * oa_stride + m * (LP * 2);
  const float* logp = oa + gq * oa_stride + M * LP * 2 + m * LP;
  const float* poff = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + m * (LP * 2) : nullptr;
  const float* plog = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + M * LP * 2 + m * LP : nullptr;
  float ox[LP], oy[LP], aw[LP];
#pragma unroll
  for (int i = 0; i < LP / 2; ++i) {
    float4 t = ld4_stream(offp + 4 * i);
    if (poff) { const float4 u = ld4(poff + 4 * i); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    ox[2 * i] = t.x; oy[2 * i] = t.y; ox[2 * i + 1] = t.z; oy[2 * i + 1] = t.w;
  }
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    float4 t = ld4_stream(logp + 4 * i);
    if (plog) { const float4 u = ld4(plog + 4 * i); t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    aw[4 * i] = t.x; aw[4 * i + 1] = t.y; aw[4 * i + 2] = t.z; aw[4 * i + 3] = t.w;
  }
  float mx = aw[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, aw[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) { aw[i] = __expf(aw[i] - mx); sum += aw[i]; }
  const float inv = 1.f / sum;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  const float* vbase = value + (long long)b * S * value_stride + m * D + c4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float* vl = vbase + lsi[l] * value_stride;
    const float invW = 1.f / (float)W, invH = 1.f / (float)H;
    float4 v[P][4];
    float cw[P][4];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float locx = rx + ox[l * P + p] * invW, locy = ry + oy[l * P + p] * invH;
      const float him = locy * (float)H - 0.5f, wim = locx * (float)W - 0.5f;
      const float hf = floorf(him), wf = floorf(wim);
      const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
      const float hfc = fminf(fmaxf(hf, -2.f), (float)H + 1.f);
      const float wfc = fminf(fmaxf(wf, -2.f), (float)W + 1.f);
      const int h0 = (int)hfc, w0 = (int)wfc, h1 = h0 + 1, w1 = w0 + 1;
      const bool inside = (him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W);
      const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
      const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
      const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h1, 0), H - 1);
      const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w1, 0), W - 1);
      cw[p][0] = (vh0 && vw0) ? hh * hw : 0.f;
      cw[p][1] = (vh0 && vw1) ? hh * lw : 0.f;
      cw[p][2] = (vh1 && vw0) ? lh * hw : 0.f;
      cw[p][3] = (vh1 && vw1) ? lh * lw : 0.f;
      v[p][0] = ld4(vl + (long long)(h0c * W + w0c) * value_stride);
      v[p][1] = ld4(vl + (long long)(h0c * W + w1c) * value_stride);
      v[p][2] = ld4(vl + (long long)(h1c * W + w0c) * value_stride);
      v[p][3] = ld4(vl + (long long)(h1c * W + w1c) * value_stride);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float a = aw[l * P + p] * inv;
      float4 s;
      s.x = cw[p][0] * v[p][0].x + cw[p][1] * v[p][1].x + cw[p][2] * v[p][2].x + cw[p][3] * v[p][3].x;
      s.y = cw[p][0] * v[p][0].y + cw[p][1] * v[p][1].y + cw[p][2] * v[p][2].y + cw[p][3] * v[p][3].y;
      s.z = cw[p][0] * v[p][0].z + cw[p][1] * v[p][1].z + cw[p][2] * v[p][2].z + cw[p][3] * v[p][3].z;
      s.w = cw[p][0] * v[p][0].w + cw[p][1] * v[p][1].w + cw[p][2] * v[p][2].w + cw[p][3] * v[p][3].w;
      acc.x += a * s.x; acc.y += a * s.y; acc.z += a * s.z; acc.w += a * s.w;
    }
  }
  return acc;
}

// The same query with the per-point work SHARED by the eight lanes of a head.  In msda_sample_query every lane of a head repeats
// the location / bilinear-weight / address arithmetic of all 12 (level, point) taps of its head -- 1 219 VALU instructions per
// wave, i.e. 1.4 ms of pure VALU issue per launch at 32 x 720p (the kernel runs 1.9 ms).  Here lane c4 of a head works out taps
// c4 and 8 + c4 only (token byte offsets of the four corners, their bilinear weights, the attention weight), and the gather
// loop fetches each tap's nine numbers from its owner lane with ds_bpermute (the LDS crossbar, otherwise idle in this kernel).
// Same formulas, same order of the floating-point operations per tap and over the taps.  32-bit byte offsets: the caller checks
// that one image's value rows fit 2 GB.
template <int L, int P>
__device__ __forceinline__ float4 msda_sample_query_coop(const float* __restrict__ value, long long value_stride,
                                                         const float* __restrict__ oa, long long oa_stride,
                                                         const float* __restrict__ pos_oa, const float* __restrict__ ref,
                                                         const long long* __restrict__ shapes,
                                                         const long long* __restrict__ lsi, int S, int Lq, long long gq,
                                                         int lane) {
  constexpr int M = 8, D = 32, LP = L * P;
  static_assert(LP > 8 && LP <= 16 && P == 4, "two taps per lane");
  const int m = lane >> 3, c4 = lane & 7;
  const int b = (int)(gq / Lq);
  const int q = (int)(gq - (long long)b * Lq);
  const float* offp = oa + gq * oa_stride + m * (LP * 2);
  const float* logp = oa + gq * oa_stride + M * LP * 2 + m * LP;
  const float* poff = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + m * (LP * 2) : nullptr;
  const float* plog = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + M * LP * 2 + m * LP : nullptr;
  const float rx = ref[2 * q], ry = ref[2 * q + 1];
  // logits of the lane's own taps; their maximum over the head by three butterfly steps inside the 8-lane group
  float own_lg[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (k == 0) ? c4 : min(8 + c4, LP - 1);
    own_lg[k] = logp[p] + (plog ? plog[p] : 0.f);
  }
  float mx = fmaxf(own_lg[0], own_lg[1]);                           // lanes without a second tap hold the last tap twice: harmless
  mx = fmaxf(mx, __shfl_xor(mx, 1));
  mx = fmaxf(mx, __shfl_xor(mx, 2));
  mx = fmaxf(mx, __shfl_xor(mx, 4));
  // ---- the lane's own taps ----------------------------------------------------------------------------------
  int own_off[2][4];
  float own_cw[2][4], own_a[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (k == 0) ? c4 : min(8 + c4, LP - 1);            // lanes without a second tap repeat the last one (never read)
    const int l = p >> 2;
    float2 oxy = *reinterpret_cast<const float2*>(offp + 2 * p);
    if (poff) { const float2 u = *reinterpret_cast<const float2*>(poff + 2 * p); oxy.x += u.x; oxy.y += u.y; }
    const float ox = oxy.x, oy = oxy.y;
    own_a[k] = __expf(own_lg[k] - mx);                              // normalised below, once the head's sum is known
    int H = (int)shapes[0], W = (int)shapes[1];
    long long base = lsi[0];
#pragma unroll
    for (int j = 1; j < L; ++j)
      if (l == j) { H = (int)shapes[2 * j]; W = (int)shapes[2 * j + 1]; base = lsi[j]; }
    const float invW = 1.f / (float)W, invH = 1.f / (float)H;
    const float locx = rx + ox * invW, locy = ry + oy * invH;
    const float him = locy * (float)H - 0.5f, wim = locx * (float)W - 0.5f;
    const float hf = floorf(him), wf = floorf(wim);
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const float hfc = fminf(fmaxf(hf, -2.f), (float)H + 1.f);
    const float wfc = fminf(fmaxf(wf, -2.f), (float)W + 1.f);
    const int h0 = (int)hfc, w0 = (int)wfc, h1 = h0 + 1, w1 = w0 + 1;
    const bool inside = (him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W);
    const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
    const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
    const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h1, 0), H - 1);
    const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w1, 0), W - 1);
    own_cw[k][0] = (vh0 && vw0) ? hh * hw : 0.f;
    own_cw[k][1] = (vh0 && vw1) ? hh * lw : 0.f;
    own_cw[k][2] = (vh1 && vw0) ? lh * hw : 0.f;
    own_cw[k][3] = (vh1 && vw1) ? lh * lw : 0.f;
    const int vs4 = (int)value_stride * 4, b0 = (int)base;
    own_off[k][0] = (b0 + h0c * W + w0c) * vs4;
    own_off[k][1] = (b0 + h0c * W + w1c) * vs4;
    own_off[k][2] = (b0 + h1c * W + w0c) * vs4;
    own_off[k][3] = (b0 + h1c * W + w1c) * vs4;
  }
  {   // the soft-max denominator of the head: the 12 exponentials live one or two per lane
    float e = own_a[0] + (8 + c4 < LP ? own_a[1] : 0.f);
    e += __shfl_xor(e, 1);
    e += __shfl_xor(e, 2);
    e += __shfl_xor(e, 4);
    const float inv = 1.f / e;
    own_a[0] *= inv;
    own_a[1] *= inv;
  }
  // ---- gather: every tap of the head from its owner lane -----------------------------------------------------------
  const char* vbase = reinterpret_cast<const char*>(value + (long long)b * S * value_stride);
  const int lane_off = (m * D + c4 * 4) * 4;
  const int grp = (lane & ~7) << 2;                                  // ds_bpermute takes byte addresses (lane * 4)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < LP; ++p) {
    const int k = p >> 3, src = grp + ((p & 7) << 2);
    int off[4];
    float cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      off[c] = __builtin_amdgcn_ds_bpermute(src, own_off[k][c]);
      cw[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_cw[k][c])));
    }
    const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_a[k])));
    float4 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(vbase + (unsigned)(off[c] + lane_off));
    float4 s4;
    s4.x = cw[0] * v[0].x + cw[1] * v[1].x + cw[2] * v[2].x + cw[3] * v[3].x;
    s4.y = cw[0] * v[0].y + cw[1] * v[1].y + cw[2] * v[2].y + cw[3] * v[3].y;
    s4.z = cw[0] * v[0].z + cw[1] * v[1].z + cw[2] * v[2].z + cw[3] * v[3].z;
    s4.w = cw[0] * v[0].w + cw[1] * v[1].w + cw[2] * v[2].w + cw[3] * v[3].w;
    acc.x += a * s4.x; acc.y += a * s4.y; acc.z += a * s4.z; acc.w += a * s4.w;
  }
  return acc;
}

// Order in which the queries of one image are WORKED ON when they are the cells of the value pyramid themselves (encoder
// self-attention: Lq == S, level-major).  Level-major order sweeps an image's value maps once per level of queries (3 x 19.8 MB
// per 720p frame against 4 MB of L2 per XCD: rocprofv3 counted 4.6 GB of HBM traffic per launch for 2.0 GB of algorithmic bytes);
// band order takes, for every row band of the coarsest level, the rows of ALL levels that cover it, so the cells a band samples
// are fetched once.  position (0 .. Lq) -> query index; a bijection for any shapes (band b of level l = rows
// [b H_l / nb, (b+1) H_l / nb), nb = rows of the last level).
template <int L>
__device__ __forceinline__ int msda_band_query(int pos, const long long* __restrict__ shapes, const long long* __restrict__ lsi) {
  const int nb = (int)shapes[2 * (L - 1)];
  const float inv_nb = 1.f / (float)nb;
  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) { Hs[l] = (int)shapes[2 * l]; Ws[l] = (int)shapes[2 * l + 1]; }
  // floor(b H / nb) without an integer division: for integers x, nb the fraction of x / nb is a multiple of 1 / nb, so
  // (x + 0.5) / nb has the same floor and sits 0.5 / nb away from every integer -- far more than the float rounding (x < 2^22)
  auto row0 = [&](int b, int l) { return (int)(((float)(b * Hs[l]) + 0.5f) * inv_nb); };
  auto start = [&](int b) {
    int o = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) o += Ws[l] * row0(b, l);
    return o;
  };
  int lo = 0, hi = nb - 1;                       // largest b with start(b) <= pos
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start(mid) <= pos) lo = mid; else hi = mid - 1;
  }
  int r = pos - start(lo);
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int r0 = row0(lo, l), r1 = row0(lo + 1, l);
    const int n = (r1 - r0) * Ws[l];
    if (r < n) return (int)lsi[l] + r0 * Ws[l] + r;
    r -= n;
  }
  return pos;                                     // not reached for consistent shapes
}

// (round 6 lab, profiles/r06_msda_occupancy.txt: the kernel is latency-bound -- fewer resident workgroups are monotonically slower --
// but asking the register allocator for 8 waves per SIMD instead of the 7 its 70 VGPRs allow, `__launch_bounds__(256, 8)`: 64 VGPRs,
// no scratch, is a tie, 1.492 vs 1.490 ms)
template <int L, int P, bool BANDS, bool COOP>
__global__ __launch_bounds__(256) void msda_fused_m8d32(
    const float* __restrict__ value, long long value_stride, const float* __restrict__ oa,
    long long oa_stride, const float* __restrict__ pos_oa, const float* __restrict__ ref,
    const long long* __restrict__ shapes, const long long* __restrict__ lsi, float* __restrict__ out,
    int S, int Lq, long long nq_total, unsigned nblk) {
  const unsigned lb = xcd_contiguous_block(blockIdx.x, nblk);
  long long gq = (long long)lb * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (gq >= nq_total) return;
  if (BANDS) {
    const int b = (int)((unsigned long long)gq / (unsigned)Lq);
    gq = (long long)b * Lq + msda_band_query<L>((int)(gq - (long long)b * Lq), shapes, lsi);
  }
  const int lane = threadIdx.x & 63;
  const float4 acc = COOP ? msda_sample_query_coop<L, P>(value, value_stride, oa, oa_stride, pos_oa, ref, shapes, lsi, S, Lq, gq, lane)
                          : msda_sample_query<L, P>(value, value_stride, oa, oa_stride, pos_oa, ref, shapes, lsi, S, Lq, gq, lane);
  st4_stream(out + gq * 256 + lane * 4, acc);            // lane = (head, 4-channel quad): column 4*lane
}

// ------------------------------------------------------------------------------------------------
// Round 5: the COARSEST level served from LDS.  msda_fused_m8d32 is bound by the texture addresser (48 gathers per query,
// TA busy 84 %, profiles/r04_msda_pmc.txt).  One head's slice of the coarsest level of a 720p frame is 23 x 40 cells x 128 B =
// 118 KB: it fits the CU's 160 KB of LDS.  Here a workgroup is (frame, HEAD, chunk of the frame's queries in band order): it
// stages that slice once (rows padded to 144 B: the 16-byte reads of eight random rows then spread over the banks) and walks its
// queries with an 8-lane group per (query, head) -- the lane is the 4-channel quad, as before, so every corner is still one
// 16-byte access per lane and 128 B per group -- taking the 4 x 4 coarsest-level corners by ds_read_b128 and only the other
// 32 through the texture path: a third of the gathers gone.  The per-tap arithmetic is msda_sample_query_coop's, bit for bit
// (lane c4 works out taps c4 and 8 + c4, the group shares them by ds_bpermute; same order of the floating-point operations).
// Differences in kind to the three earlier LDS attempts (DESIGN history): the slice is per HEAD (not all heads of a tile), the
// workgroup keeps 16 waves resident (1 024 threads) and never waits on a barrier after the fill, and the fill is amortised over
// thousands of queries (two chunks per (frame, head) at 32 frames; more chunks on short clips so that >= 512 workgroups exist).
template <int L, int P, bool BANDS, bool USE_LDS = true>
__global__ __launch_bounds__(1024) void msda_fused_hlds(
    const float* __restrict__ value, long long value_stride, const float* __restrict__ oa, long long oa_stride,
    const float* __restrict__ pos_oa, const float* __restrict__ ref, const long long* __restrict__ shapes,
    const long long* __restrict__ lsi, float* __restrict__ out, int S, int Lq, int nch, int per_chunk, unsigned nblk,
    int lds_cells) {
  constexpr int M = 8, D = 32, LP = L * P, ROW = 36;               // LDS row stride in floats (144 B)
  static_assert(LP > 8 && LP <= 16 && P == 4, "two taps per lane");
  extern __shared__ __attribute__((aligned(16))) float lds0[];
  const unsigned logical = xcd_contiguous_block(blockIdx.x, nblk);
  const int m = (int)(logical & 7u);
  const int chunk = (int)((logical >> 3) % (unsigned)nch), b = (int)((logical >> 3) / (unsigned)nch);
  const int H0 = (int)shapes[0], W0 = (int)shapes[1], n0 = H0 * W0;
  if (n0 > lds_cells) __builtin_trap();                              // the host sized the LDS for S / 21 cells
  if (USE_LDS) {
    const float* vb = value + ((long long)b * S + lsi[0]) * value_stride + m * D;
    for (int i = threadIdx.x; i < n0 * 8; i += 1024) {
      const int r = i >> 3, c = i & 7;
      *reinterpret_cast<float4*>(lds0 + r * ROW + c * 4) = ld4(vb + (long long)r * value_stride + c * 4);
    }
  }
  __syncthreads();
  const int c4 = threadIdx.x & 7, slot = threadIdx.x >> 3;
  const int lane = threadIdx.x & 63;
  const int grp = (lane & ~7) << 2;                                  // ds_bpermute byte address of the group's lane 0
  const int pend = min((chunk + 1) * per_chunk, Lq);
  const char* vbase = reinterpret_cast<const char*>(value + (long long)b * S * value_stride);
  const int lane_off = (m * D + c4 * 4) * 4;
  const int vs4 = (int)value_stride * 4;
  for (int pos = chunk * per_chunk + slot; pos < pend; pos += 128) {
    const int q = BANDS ? msda_band_query<L>(pos, shapes, lsi) : pos;
    const long long gq = (long long)b * Lq + q;
    const float* offp = oa + gq * oa_stride + m * (LP * 2);
    const float* logp = oa + gq * oa_stride + M * LP * 2 + m * LP;
    const float* poff = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + m * (LP * 2) : nullptr;
    const float* plog = pos_oa ? pos_oa + (long long)q * (M * LP * 3) + M * LP * 2 + m * LP : nullptr;
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    float own_lg[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = (k == 0) ? c4 : min(8 + c4, LP - 1);
      own_lg[k] = logp[p] + (plog ? plog[p] : 0.f);
    }
    float mx = fmaxf(own_lg[0], own_lg[1]);
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    int own_off[2][4];
    float own_cw[2][4], own_a[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int p = (k == 0) ? c4 : min(8 + c4, LP - 1);
      const int l = p >> 2;
      float2 oxy = *reinterpret_cast<const float2*>(offp + 2 * p);
      if (poff) { const float2 u = *reinterpret_cast<const float2*>(poff + 2 * p); oxy.x += u.x; oxy.y += u.y; }
      const float ox = oxy.x, oy = oxy.y;
      own_a[k] = __expf(own_lg[k] - mx);
      int H = H0, W = W0;
      long long base = lsi[0];
#pragma unroll
      for (int j = 1; j < L; ++j)
        if (l == j) { H = (int)shapes[2 * j]; W = (int)shapes[2 * j + 1]; base = lsi[j]; }
      const float invW = 1.f / (float)W, invH = 1.f / (float)H;
      const float locx = rx + ox * invW, locy = ry + oy * invH;
      const float him = locy * (float)H - 0.5f, wim = locx * (float)W - 0.5f;
      const float hf = floorf(him), wf = floorf(wim);
      const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
      const float hfc = fminf(fmaxf(hf, -2.f), (float)H + 1.f);
      const float wfc = fminf(fmaxf(wf, -2.f), (float)W + 1.f);
      const int h0 = (int)hfc, w0 = (int)wfc, h1 = h0 + 1, w1 = w0 + 1;
      const bool inside = (him > -1.f) && (wim > -1.f) && (him < (float)H) && (wim < (float)W);
      const bool vh0 = inside && h0 >= 0, vh1 = inside && h1 <= H - 1;
      const bool vw0 = w0 >= 0, vw1 = w1 <= W - 1;
      const int h0c = min(max(h0, 0), H - 1), h1c = min(max(h1, 0), H - 1);
      const int w0c = min(max(w0, 0), W - 1), w1c = min(max(w1, 0), W - 1);
      own_cw[k][0] = (vh0 && vw0) ? hh * hw : 0.f;
      own_cw[k][1] = (vh0 && vw1) ? hh * lw : 0.f;
      own_cw[k][2] = (vh1 && vw0) ? lh * hw : 0.f;
      own_cw[k][3] = (vh1 && vw1) ? lh * lw : 0.f;
      // level 0: byte offsets into the LDS slice; other levels: byte offsets of the token rows in the image's value tensor
      const int mul = (USE_LDS && l == 0) ? ROW * 4 : vs4, b0 = (USE_LDS && l == 0) ? 0 : (int)base;
      own_off[k][0] = (b0 + h0c * W + w0c) * mul;
      own_off[k][1] = (b0 + h0c * W + w1c) * mul;
      own_off[k][2] = (b0 + h1c * W + w0c) * mul;
      own_off[k][3] = (b0 + h1c * W + w1c) * mul;
    }
    {
      float e = own_a[0] + (8 + c4 < LP ? own_a[1] : 0.f);
      e += __shfl_xor(e, 1);
      e += __shfl_xor(e, 2);
      e += __shfl_xor(e, 4);
      const float inv = 1.f / e;
      own_a[0] *= inv;
      own_a[1] *= inv;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < LP; ++p) {
      const int k = p >> 3, src = grp + ((p & 7) << 2);
      int off[4];
      float cw[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        off[c] = __builtin_amdgcn_ds_bpermute(src, own_off[k][c]);
        cw[c] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_cw[k][c])));
      }
      const float a = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(own_a[k])));
      float4 v[4];
      if (USE_LDS && p < P) {                                        // the coarsest level: from the staged slice
#pragma unroll
        for (int c = 0; c < 4; ++c)
          v[c] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(lds0) + off[c] + c4 * 16);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(vbase + (unsigned)(off[c] + lane_off));
      }
      float4 s4;
      s4.x = cw[0] * v[0].x + cw[1] * v[1].x + cw[2] * v[2].x + cw[3] * v[3].x;
      s4.y = cw[0] * v[0].y + cw[1] * v[1].y + cw[2] * v[2].y + cw[3] * v[3].y;
      s4.z = cw[0] * v[0].z + cw[1] * v[1].z + cw[2] * v[2].z + cw[3] * v[3].z;
      s4.w = cw[0] * v[0].w + cw[1] * v[1].w + cw[2] * v[2].w + cw[3] * v[3].w;
      acc.x += a * s4.x; acc.y += a * s4.y; acc.z += a * s4.z; acc.w += a * s4.w;
    }
    st4_stream(out + gq * 256 + m * D + c4 * 4, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// msda_proj_ln: the whole attention half of an encoder layer after the projection GEMM,
//   out = LayerNorm( identity + ( MSDA(value, offsets, logits) Wo^T + bo ) )
// i.e. mmcv MultiScaleDeformableAttention.forward steps 4-8 + BaseTransformerLayer's first norm in ONE launch.
// The sampling is bound by the texture path (profiles/r01_msda_pmc.txt: TA busy 98 %), the 256x256 output
// projection by the matrix pipe: a workgroup gathers 64 queries into LDS (8 waves x 8 queries, lane = (head,
// 4-channel quad) as in msda_fused_m8d32), multiplies the 64x256 tile by Wo^T on v_mfma_f32_16x16x4_f32 with the
// weight fragments streamed from L2 (packed by pvsg_pack_rows_weight, 4 KB per query), adds bias + identity and
// normalises the rows.  Two workgroups share a CU (66.5 KB LDS each), so one's matrix phase runs under the
// other's gather.  Removes per layer: the 633 MB sampled-output round trip, the library GEMM launch and the
// add_layernorm pass (another 1.9 GB).
// ------------------------------------------------------------------------------------------------
constexpr int MP_ROWS = 64, MP_LD = 260, MP_THREADS = 512;

template <bool MAX>
__device__ __forceinline__ float msda_wave_allreduce(float v) {
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false)));
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false)));
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false)));
  v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xf, 0xf, false)));
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(v);
  auto b2 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return op(__uint_as_float(b2[0]), __uint_as_float(b2[1]));
}

template <int L, int P, bool COOP>
__global__ __launch_bounds__(MP_THREADS, 4) void msda_proj_ln_kernel(
    const float* __restrict__ value, long long value_stride, const float* __restrict__ oa, long long oa_stride,
    const float* __restrict__ pos_oa, const float* __restrict__ ref, const long long* __restrict__ shapes,
    const long long* __restrict__ lsi, const float* __restrict__ wo, const float* __restrict__ wo_bias,
    const float* __restrict__ identity, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out, int S, int Lq, long long nq_total, unsigned ntiles, float eps) {
  __shared__ __attribute__((aligned(16))) float xs[MP_ROWS * MP_LD];
  const unsigned tile = xcd_contiguous_block(blockIdx.x, ntiles);
  const long long q0 = (long long)tile * MP_ROWS;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // ---- phase 1: gather (wave w -> rows 8w .. 8w+7) ----------------------------------------------------
#pragma unroll 1
  for (int i = 0; i < 8; ++i) {
    const int r = w * 8 + i;
    const long long gq = q0 + r;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gq < nq_total)
      acc = COOP ? msda_sample_query_coop<L, P>(value, value_stride, oa, oa_stride, pos_oa, ref, shapes, lsi, S, Lq, gq, lane)
                 : msda_sample_query<L, P>(value, value_stride, oa, oa_stride, pos_oa, ref, shapes, lsi, S, Lq, gq, lane);
    *reinterpret_cast<float4*>(xs + r * MP_LD + lane * 4) = acc;       // lane = (head, quad) -> column 4*lane
  }
  __syncthreads();
  // ---- phase 2: Y = X Wo^T, wave w -> column tiles 2w, 2w+1, all four row tiles ----------------------------
  f32x4 acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) { acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  {
    constexpr int DEPTH = 4;
    const float* xa = xs + (lane & 15) * MP_LD + 4 * (lane >> 4);
    const float* wb = wo + (long long)(2 * w) * 16 * 256 + lane * 4;     // [tile][kc][lane][4], 16 k chunks per tile
    float4 ring[DEPTH][2];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { ring[d][0] = ld4(wb + d * 256); ring[d][1] = ld4(wb + 16 * 256 + d * 256); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      float4 a[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) a[rt] = *reinterpret_cast<const float4*>(xa + rt * 16 * MP_LD + kc * 16);
      const float4 b0 = ring[kc % DEPTH][0], b1 = ring[kc % DEPTH][1];
      if (kc + DEPTH < 16) {
        ring[kc % DEPTH][0] = ld4(wb + (kc + DEPTH) * 256);
        ring[kc % DEPTH][1] = ld4(wb + 16 * 256 + (kc + DEPTH) * 256);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b0.x, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b1.x, acc[rt][1], 0, 0, 0);
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b0.y, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b1.y, acc[rt][1], 0, 0, 0);
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, b0.z, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, b1.z, acc[rt][1], 0, 0, 0);
        acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, b0.w, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, b1.w, acc[rt][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                   // every wave is done reading X
  {
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int col = (2 * w + c) * 16 + j;
      const float bv = wo_bias ? wo_bias[col] : 0.f;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[(rt * 16 + 4 * g + e) * MP_LD + col] = acc[rt][c][e] + bv;
    }
  }
  __syncthreads();
  // ---- phase 3: + identity, LayerNorm, coalesced row stores (wave w -> rows 8w .. 8w+7) ------------------------
  const float4 gm = ld4(gamma + lane * 4), bt = ld4(beta + lane * 4);
#pragma unroll 2
  for (int i = 0; i < 8; ++i) {
    const int r = w * 8 + i;
    const long long gq = q0 + r;
    if (gq >= nq_total) break;
    float4 x = *reinterpret_cast<const float4*>(xs + r * MP_LD + lane * 4);
    const float4 idv = ld4_stream(identity + gq * 256 + lane * 4);
    x.x += idv.x; x.y += idv.y; x.z += idv.z; x.w += idv.w;
    const float mean = msda_wave_allreduce<false>(x.x + x.y + x.z + x.w) * (1.f / 256.f);
    const float dx = x.x - mean, dy = x.y - mean, dz = x.z - mean, dw = x.w - mean;
    const float var = msda_wave_allreduce<false>(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / 256.f);
    const float rstd = rsqrtf(var + eps);
    st4_stream(out + gq * 256 + lane * 4,
               make_float4(dx * rstd * gm.x + bt.x, dy * rstd * gm.y + bt.y, dz * rstd * gm.z + bt.z, dw * rstd * gm.w + bt.w));
  }
}

// out = LayerNorm(a + b + bias) * gamma + beta over the last dim C = 256 * NV (NV = 1: the encoder / decoder width; NV = 2: the
// relation head's TemporalTransformer, d_model 512); one wave per row, lane owns the float4 groups at 4 (64 v + lane).
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
    long long rows, float eps) {
  constexpr int C = 256 * NV;
  const int lane = threadIdx.x & 63;
  float4 g[NV], be[NV], bi[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    g[v] = ld4(gamma + 256 * v + lane * 4);
    be[v] = ld4(beta + 256 * v + lane * 4);
    bi[v] = bias ? ld4(bias + 256 * v + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // two rows per wave and iteration (independent load / reduce chains); the row reductions run in the VALU (DPP +
  // permlane swaps) instead of twelve ds_bpermute round trips per row; streaming (touched-once) loads and stores
  const long long stride = (long long)gridDim.x * 8;
  for (long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 6) * 2; row < rows; row += stride) {
    const bool two = row + 1 < rows;
    float4 x0[NV], x1[NV];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const long long o0 = row * C + 256 * v + lane * 4, o1 = o0 + C;
      x0[v] = ld4_stream(a + o0);
      x1[v] = two ? ld4_stream(a + o1) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (b) {
        const float4 y0 = ld4_stream(b + o0);
        const float4 y1 = two ? ld4_stream(b + o1) : make_float4(0.f, 0.f, 0.f, 0.f);
        x0[v].x += y0.x; x0[v].y += y0.y; x0[v].z += y0.z; x0[v].w += y0.w;
        x1[v].x += y1.x; x1[v].y += y1.y; x1[v].z += y1.z; x1[v].w += y1.w;
      }
      x0[v].x += bi[v].x; x0[v].y += bi[v].y; x0[v].z += bi[v].z; x0[v].w += bi[v].w;
      x1[v].x += bi[v].x; x1[v].y += bi[v].y; x1[v].z += bi[v].z; x1[v].w += bi[v].w;
      s0 += (x0[v].x + x0[v].y) + (x0[v].z + x0[v].w);
      s1 += (x1[v].x + x1[v].y) + (x1[v].z + x1[v].w);
    }
    const float m0 = msda_wave_allreduce<false>(s0) * (1.f / C);
    const float m1 = msda_wave_allreduce<false>(s1) * (1.f / C);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      x0[v].x -= m0; x0[v].y -= m0; x0[v].z -= m0; x0[v].w -= m0;
      x1[v].x -= m1; x1[v].y -= m1; x1[v].z -= m1; x1[v].w -= m1;
      q0 += (x0[v].x * x0[v].x + x0[v].y * x0[v].y) + (x0[v].z * x0[v].z + x0[v].w * x0[v].w);
      q1 += (x1[v].x * x1[v].x + x1[v].y * x1[v].y) + (x1[v].z * x1[v].z + x1[v].w * x1[v].w);
    }
    const float r0 = rsqrtf(msda_wave_allreduce<false>(q0) * (1.f / C) + eps);
    const float r1 = rsqrtf(msda_wave_allreduce<false>(q1) * (1.f / C) + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const long long o0 = row * C + 256 * v + lane * 4;
      st4_stream(out + o0, make_float4(x0[v].x * r0 * g[v].x + be[v].x, x0[v].y * r0 * g[v].y + be[v].y,
                                       x0[v].z * r0 * g[v].z + be[v].z, x0[v].w * r0 * g[v].w + be[v].w));
      if (two)
        st4_stream(out + o0 + C, make_float4(x1[v].x * r1 * g[v].x + be[v].x, x1[v].y * r1 * g[v].y + be[v].y,
                                             x1[v].z * r1 * g[v].z + be[v].z, x1[v].w * r1 * g[v].w + be[v].w));
    }
  }
}

}  // namespace pvsg

extern "C" int pvsg_msda_fused_forward(const float* value, long long value_row_stride, const float* oa,
                                       long long oa_row_stride, const float* pos_oa, const float* ref_points,
                                       const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       float* out, int B, int S, int M, int D, int Lq, int L, int P,
                                       hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(value && oa && ref_points && spatial_shapes && level_start_index && out,
               "msda_fused_forward: null pointer argument");
  PVSG_REQUIRE(B > 0 && S > 0 && Lq > 0, "msda_fused_forward: non-positive dimension");
  if (M != 8 || D != 32 || L != 3 || P != 4)
    return set_err(PVSG_ERR_UNSUPPORTED, "msda_fused_forward: built for M=8 D=32 L=3 P=4 (got %d %d %d %d)", M, D, L, P);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(oa) |
                  reinterpret_cast<uintptr_t>(pos_oa) | reinterpret_cast<uintptr_t>(out)) & 15u) &&
               !(value_row_stride & 3) && !(oa_row_stride & 3) && value_row_stride >= M * D,
               "msda_fused_forward: 16-byte alignment / row strides multiple of 4 floats required");
  const long long nq = (long long)B * Lq;
  const long long nblk_ll = (nq + 3) / 4;
  PVSG_REQUIRE(nblk_ll < (1ll << 31), "msda_fused_forward: too many queries");
  const unsigned nblk = (unsigned)nblk_ll;
  // queries = the cells of the value pyramid (encoder self-attention): band order (msda_band_query); PVSG_MSDA_ORDER=level keeps
  // the level-major order for the A/B.  The results do not depend on the order.
  const char* ord = getenv("PVSG_MSDA_ORDER");
  const bool bands = Lq == S && !(ord && ord[0] == 'l');
  // the per-tap arithmetic shared by the lanes of a head (msda_sample_query_coop) where 32-bit byte offsets reach every value row
  // of an image; PVSG_MSDA_COOP=0 keeps every lane computing every tap (A/B)
  const char* cp = getenv("PVSG_MSDA_COOP");
  const bool coop = (long long)S * value_row_stride * 4 < (1ll << 31) && !(cp && cp[0] == '0');
  // PVSG_MSDA_LDS=on: the coarsest level of a head from LDS (msda_fused_hlds; lab A/B: scripts/kbench.py msda)
  {
    const char* hl = getenv("PVSG_MSDA_LDS");
    const long long* shp = reinterpret_cast<const long long*>(spatial_shapes);
    // (the host cannot read the device-side shape table: the caller passes the coarsest level's cell count through the
    // environment-free rule below -- Lq == S and the first level is the smallest by construction of the pixel decoder; the LDS
    // size is taken from S: levels are x4 apart, so the coarsest holds S / 21 cells)
    const long long n0 = S / 21;
    if (hl && hl[0] == 'm' && coop && Lq == S && S % 21 == 0 && bands) {      // lab: the (frame, head, chunk) mapping WITHOUT the LDS slice
      int nch = (int)((512 + (long long)B * 8 - 1) / ((long long)B * 8));
      if (nch < 1) nch = 1;
      const unsigned nb = (unsigned)(B * 8 * nch);
      hipLaunchKernelGGL((msda_fused_hlds<3, 4, true, false>), dim3(nb), dim3(1024), 16, stream, value, value_row_stride, oa,
                         oa_row_stride, pos_oa, ref_points, reinterpret_cast<const long long*>(spatial_shapes),
                         reinterpret_cast<const long long*>(level_start_index), out, S, Lq, nch, (Lq + nch - 1) / nch, nb, (int)n0);
      PVSG_LAUNCH_CHECK("msda_fused_forward");
      return PVSG_OK;
    }
    if (hl && hl[0] == 'o' && hl[1] == 'n' && coop && Lq == S && S % 21 == 0 && n0 * 144 <= 160 * 1024 - 1024) {
      static std::atomic<unsigned long long> done_b{0}, done_l{0};
      const int lds_bytes = (int)(n0 * 144);
      hipError_t e = bands ? ensure_dynamic_lds(reinterpret_cast<const void*>(msda_fused_hlds<3, 4, true>), lds_bytes, done_b)
                           : ensure_dynamic_lds(reinterpret_cast<const void*>(msda_fused_hlds<3, 4, false>), lds_bytes, done_l);
      if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "msda_fused_forward: dynamic LDS: %s", hipGetErrorString(e));
      int nch = (int)((512 + (long long)B * 8 - 1) / ((long long)B * 8));
      if (nch < 1) nch = 1;
      const int per_chunk = (Lq + nch - 1) / nch;
      const unsigned nb = (unsigned)(B * 8 * nch);
      (void)shp;
      if (bands)
        hipLaunchKernelGGL((msda_fused_hlds<3, 4, true>), dim3(nb), dim3(1024), lds_bytes, stream, value, value_row_stride, oa,
                           oa_row_stride, pos_oa, ref_points, reinterpret_cast<const long long*>(spatial_shapes),
                           reinterpret_cast<const long long*>(level_start_index), out, S, Lq, nch, per_chunk, nb, (int)n0);
      else
        hipLaunchKernelGGL((msda_fused_hlds<3, 4, false>), dim3(nb), dim3(1024), lds_bytes, stream, value, value_row_stride, oa,
                           oa_row_stride, pos_oa, ref_points, reinterpret_cast<const long long*>(spatial_shapes),
                           reinterpret_cast<const long long*>(level_start_index), out, S, Lq, nch, per_chunk, nb, (int)n0);
      PVSG_LAUNCH_CHECK("msda_fused_forward");
      return PVSG_OK;
    }
  }
  // lab knob (profiles/r06_msda_occupancy.txt): unused dynamic LDS per workgroup caps the workgroups resident on a CU (160 KB / n)
  static const unsigned occ_lds = [] { const char* e = getenv("PVSG_MSDA_OCC_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
#define PVSG_MSDA_LAUNCH(BANDS, COOP)                                                                                        \
  hipLaunchKernelGGL((msda_fused_m8d32<3, 4, BANDS, COOP>), dim3(nblk), dim3(256), occ_lds, stream, value, value_row_stride, oa,  \
                     oa_row_stride, pos_oa, ref_points, reinterpret_cast<const long long*>(spatial_shapes),                \
                     reinterpret_cast<const long long*>(level_start_index), out, S, Lq, nq, nblk)
  if (bands && coop) PVSG_MSDA_LAUNCH(true, true);
  else if (bands) PVSG_MSDA_LAUNCH(true, false);
  else if (coop) PVSG_MSDA_LAUNCH(false, true);
  else PVSG_MSDA_LAUNCH(false, false);
#undef PVSG_MSDA_LAUNCH
  PVSG_LAUNCH_CHECK("msda_fused_forward");
  return PVSG_OK;
}

extern "C" int pvsg_msda_proj_ln_forward(const float* value, long long value_row_stride, const float* oa,
                                         long long oa_row_stride, const float* pos_oa, const float* ref_points,
                                         const int64_t* spatial_shapes, const int64_t* level_start_index,
                                         const float* wo_packed, const float* wo_bias, const float* identity,
                                         const float* gamma, const float* beta, float* out, int B, int S, int M,
                                         int D, int Lq, int L, int P, float eps, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(value && oa && ref_points && spatial_shapes && level_start_index && wo_packed && identity && gamma &&
                   beta && out, "msda_proj_ln_forward: null pointer argument");
  PVSG_REQUIRE(B > 0 && S > 0 && Lq > 0, "msda_proj_ln_forward: non-positive dimension");
  if (M != 8 || D != 32 || L != 3 || P != 4)
    return set_err(PVSG_ERR_UNSUPPORTED, "msda_proj_ln_forward: built for M=8 D=32 L=3 P=4 (got %d %d %d %d)", M, D, L, P);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(oa) |
                  reinterpret_cast<uintptr_t>(pos_oa) | reinterpret_cast<uintptr_t>(out) |
                  reinterpret_cast<uintptr_t>(wo_packed) | reinterpret_cast<uintptr_t>(identity) |
                  reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15u) &&
               !(value_row_stride & 3) && !(oa_row_stride & 3) && value_row_stride >= M * D,
               "msda_proj_ln_forward: 16-byte alignment / row strides multiple of 4 floats required");
  const long long nq = (long long)B * Lq;
  const long long nt = (nq + MP_ROWS - 1) / MP_ROWS;
  PVSG_REQUIRE(nt < (1ll << 31), "msda_proj_ln_forward: too many queries");
  const char* cp = getenv("PVSG_MSDA_COOP");
  const bool coop = (long long)S * value_row_stride * 4 < (1ll << 31) && !(cp && cp[0] == '0');
#define PVSG_MPL_LAUNCH(C)                                                                                          \
  hipLaunchKernelGGL((msda_proj_ln_kernel<3, 4, C>), dim3((unsigned)nt), dim3(MP_THREADS), 0, stream, value,        \
                     value_row_stride, oa, oa_row_stride, pos_oa, ref_points,                                        \
                     reinterpret_cast<const long long*>(spatial_shapes),                                             \
                     reinterpret_cast<const long long*>(level_start_index), wo_packed, wo_bias, identity, gamma, beta, \
                     out, S, Lq, nq, (unsigned)nt, eps)
  if (coop) PVSG_MPL_LAUNCH(true); else PVSG_MPL_LAUNCH(false);
#undef PVSG_MPL_LAUNCH
  PVSG_LAUNCH_CHECK("msda_proj_ln_forward");
  return PVSG_OK;
}

extern "C" int pvsg_add_layernorm(const float* a, const float* b, const float* bias, const float* gamma,
                                  const float* beta, float* out, long long rows, int C, float eps,
                                  hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(a && gamma && beta && out, "add_layernorm: null pointer argument");
  PVSG_REQUIRE(rows > 0, "add_layernorm: no rows");
  if (C != 256 && C != 512) return set_err(PVSG_ERR_UNSUPPORTED, "add_layernorm: built for 256 or 512 channels (got %d)", C);
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) |
                  reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(gamma) |
                  reinterpret_cast<uintptr_t>(beta)) & 15u), "add_layernorm: 16-byte alignment required");
  long long nb = (rows + 7) / 8;
  if (nb > 256 * 16) nb = 256 * 16;
  if (C == 256)
    hipLaunchKernelGGL(add_layernorm_kernel<1>, dim3((unsigned)nb), dim3(256), 0, stream, a, b, bias, gamma, beta, out, rows, eps);
  else
    hipLaunchKernelGGL(add_layernorm_kernel<2>, dim3((unsigned)nb), dim3(256), 0, stream, a, b, bias, gamma, beta, out, rows, eps);
  PVSG_LAUNCH_CHECK("add_layernorm");
  return PVSG_OK;
}
