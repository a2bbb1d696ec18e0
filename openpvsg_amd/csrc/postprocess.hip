// Fused panoptic post-processing: bilinear up-sampling of the stride-4 mask logits to the input
// size + sigmoid + score*mask argmax over the kept queries + area / low-score filters + segment-id
// write, without ever materialising the (Q, H, W) up-sampled tensor.
//
// Replaces (reference):
//   F.interpolate(mask_pred, batch_input_shape, bilinear)   models/mask2former/mask2former_head.py:675-679,
//                                                           models/mask2former_vps/mask2former_video_head.py:660-667
//   crop to img_shape                                        models/mask2former/mask2former_fusion_head.py:372-374
//   panoptic_postprocess_with_query                          mask2former_fusion_head.py:96-171
// (SURVEY.md section 8f row 1).  The reference writes 100*736*1280*4 B = 377 MB per 720p frame and
// re-reads it >= 4 times, with 3-5 device syncs per kept query; here a frame costs one read of the
// kept queries' stride-4 logits (<= 23.5 MB, L2 resident) and 5 B per output pixel.
//
// Three launches for all T frames of a clip (the kept set is shared by the frames in clip mode):
//   1 owner      per output pixel: p_k = sigmoid(up4(L_k)), owner = argmax_k s_k p_k (first max wins,
//                like torch.argmax), conf = p_owner >= 0.5; per-query histograms area[k] (pixels owned),
//                orig[k] (pixels with p_k >= 0.5), region[k] (owned [and confident if filter_low_score])
//                via wave ballots + LDS counters + one global atomic per query and block.
//   2 decide     per frame: ok_k = area>0 && orig>0 && !(area/orig < iou_thr) && region>0 (float64 ratio
//                as in the Python original); instance ids = running count of accepted THING queries in
//                query order; seg_id_k = class (+ instance*1000 for things) or -1.
//   3 paint      panoptic[y,x] = ok[owner] && (conf || !filter) ? seg_id[owner] : num_classes.
// The up-sampling arithmetic is ATen's upsample_bilinear2d (align_corners=False): src = (dst+0.5)*in/out
// - 0.5 clamped at 0, i1 = min(i0+1, in-1), and the same association of the four products.
// With rescale=True and ori_shape != img_shape the reference resizes the cropped maps a second time
// (mask2former_fusion_head.py:376-383); pan_owner_2stage_kernel composes both resizes per output pixel
// (4 stage-2 taps, each a 4-tap stage-1 interpolation of the stride-4 logits: 16 source taps), so that
// case never materialises (Q,H,W) either.  pvsg_instance_masks does the same for instance_postprocess
// (mask2former_fusion_head.py:192-242): binary masks, mask-quality sums and boxes of the selected queries.
#include "common.h"

namespace pvsg {

constexpr int MAXK = 128;
#define PVSG_SEL_MAXK 128       // include/openpvsg_hip.h (not included here: this file declares its streams as hipStream_t); tubes.hip checks the layout

__global__ __launch_bounds__(256) void pan_owner_kernel(
    const float* __restrict__ logits, const int* __restrict__ kept_idx, const float* __restrict__ kept_score,
    unsigned char* __restrict__ owner_out, int* __restrict__ counters, int Q, int K, int h, int w,
    int H, int W, int ih, int iw, const int* __restrict__ kdev) {
  if (kdev) K = *kdev;                       // kept count decided on the device (pvsg_panoptic_select)
  __shared__ int s_area[MAXK], s_orig[MAXK], s_region_conf[MAXK];
  __shared__ int s_idx[MAXK];
  __shared__ float s_score[MAXK];
  const int t = blockIdx.z;
  for (int k = threadIdx.x; k < MAXK; k += blockDim.x) {
    s_area[k] = 0; s_orig[k] = 0; s_region_conf[k] = 0;
    if (k < K) { s_idx[k] = kept_idx[k]; s_score[k] = kept_score[k]; }
  }
  __syncthreads();
  // 64 x 4 pixel tile per block: a wave = one 64-pixel row segment (coalesced owner stores)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const bool inside = x < iw && y < ih;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 > h - 1 ? h - 1 : y0;
  x0 = x0 > w - 1 ? w - 1 : x0;
  const int yp = (y0 < h - 1) ? 1 : 0, xp = (x0 < w - 1) ? 1 : 0;
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const float* base = logits + (long long)t * Q * h * w + (long long)y0 * w + x0;
  float best = -1.f;
  int own = 0;
  float pown = 0.f;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < K; ++k) {
    const float* p = base + (long long)s_idx[k] * h * w;
    float v = 0.f;
    if (inside)
      v = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[yp * w] + lx1 * p[yp * w + xp]);
    const float prob = 1.f / (1.f + expf(-v));
    const float sc = s_score[k] * prob;
    if (sc > best) { best = sc; own = k; pown = prob; }
    const unsigned long long conf = __ballot(inside && prob >= 0.5f);
    if (lane == 0 && conf) atomicAdd(&s_orig[k], __popcll(conf));
  }
  if (inside) {
    const bool c = pown >= 0.5f;
    atomicAdd(&s_area[own], 1);
    if (c) atomicAdd(&s_region_conf[own], 1);
    owner_out[((long long)t * ih + y) * iw + x] = (unsigned char)(own | (c ? 0x80 : 0));
  }
  __syncthreads();
  int* ct = counters + (long long)t * 3 * MAXK;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    if (s_area[k]) atomicAdd(ct + k, s_area[k]);
    if (s_orig[k]) atomicAdd(ct + MAXK + k, s_orig[k]);
    if (s_region_conf[k]) atomicAdd(ct + 2 * MAXK + k, s_region_conf[k]);
  }
}

// Exact x4 case (H = 4h, W = 4w -- every /32-padded input): the 16 output pixels y in [4i+2, 4i+5],
// x in [4j+2, 4j+5] interpolate the SAME four source taps (i,j),(i,j+1),(i+1,j),(i+1,j+1) with the fixed
// weights {1/8, 3/8, 5/8, 7/8}.  One lane therefore owns such a 4x4 block (shifted by -2 so that the
// clamped border rows/columns fall out of the same formula): 4 tap loads per kept query instead of 64,
// 16 running arg-max states in registers.  Same arithmetic per pixel as pan_owner_kernel (ATen's
// formula specialised to scale 1/4), so the two kernels agree bit for bit.
// SKIP (default): a kept query whose logits are far below zero on a lane's four taps cannot own any of its 16 pixels once every
// one of them has a better candidate: sigmoid(v) < e^v, v <= max of the taps (bilinear weights are a convex combination), so
// s_k e^{vmax} (with a 1e-4 margin for the roundings) < min over the pixels of best[] means `sc > best[e]` is false sixteen times
// and prob >= 0.5 is false too (vmax < -1e-3).  The wave then skips the 16 exact sigmoids (exp + IEEE division: 4/5 of the
// kernel's instructions) for that query unless one of its lanes needs them -- most (query, region) pairs of a real frame, where
// an object's mask is strongly negative away from the object.  The decisions are the unskipped kernel's, bit for bit
// (tests/test_postprocess.py::test_pan_owner_skip_equals_full_evaluation).  Lanes of a wave form an 8 x 8 patch of 4 x 4
// blocks (32 x 32 pixels) so that a wave's lanes tend to agree.
template <bool SKIP>
__global__ __launch_bounds__(256) void pan_owner_x4_kernel(
    const float* __restrict__ logits, const int* __restrict__ kept_idx, const float* __restrict__ kept_score,
    unsigned char* __restrict__ owner_out, int* __restrict__ counters, int Q, int K, int h, int w, int ih, int iw, const int* __restrict__ kdev) {
  if (kdev) K = *kdev;                       // kept count decided on the device (pvsg_panoptic_select)
  __shared__ int s_area[MAXK], s_orig[MAXK], s_region_conf[MAXK];
  __shared__ int s_idx[MAXK];
  __shared__ float s_score[MAXK];
  const int t = blockIdx.z;
  for (int k = threadIdx.x; k < MAXK; k += blockDim.x) {
    s_area[k] = 0; s_orig[k] = 0; s_region_conf[k] = 0;
    if (k < K) { s_idx[k] = kept_idx[k]; s_score[k] = kept_score[k]; }
  }
  __syncthreads();
  // block (bx, by) of the (w+1) x (h+1) grid of 4x4 blocks; block (bj, bi) covers output rows 4*bi-2 .. 4*bi+1
  const int bj = SKIP ? blockIdx.x * 32 + (int)(threadIdx.x >> 6) * 8 + (int)(threadIdx.x & 7) : blockIdx.x * 32 + (threadIdx.x & 31);
  const int bi = SKIP ? blockIdx.y * 8 + (int)((threadIdx.x & 63) >> 3) : blockIdx.y * 8 + (threadIdx.x >> 5);
  const bool active = bj <= w && bi <= h;
  const int i0 = max(bi - 1, 0), i1 = min(bi, h - 1), j0 = max(bj - 1, 0), j1 = min(bj, w - 1);
  // output row 4*bi-2+a: src = (y+0.5)/4-0.5 -> weight of tap i1 is (2a+1)/8 inside, clamped rows give i0==i1
  float best[16], pown[16];
  int own[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    // SKIP: pixels outside the image never take a candidate (`ok` below): they must not hold the minimum of best[] down
    const int y = 4 * bi - 2 + (e >> 2), x = 4 * bj - 2 + (e & 3);
    best[e] = (SKIP && !(active && y >= 0 && y < ih && x >= 0 && x < iw)) ? INFINITY : -1.f;
    pown[e] = 0.f; own[e] = 0;
  }
  float minbest = active ? -1.f : INFINITY;                 // min over the lane's pixels of best[] (SKIP)
  const float* base = logits + (long long)t * Q * h * w;
  const int lane = threadIdx.x & 63;
  // SKIP, first pass: a floor under the FINAL best[] of all 16 pixels -- the largest, over the kept queries, of s_k sigmoid(min
  // of the four taps) (1 - 1e-4).  A query whose bound stays below the floor is strictly beaten at every pixel by the query that
  // set the floor, whichever comes first in the list, so passing it over changes neither the final owner (arg-max, first maximum)
  // nor any count.  Without the floor only the queries BEHIND a block's owner in list order could be skipped.
  float floor_best = -1.f;
  if (SKIP && active) {
    for (int k = 0; k < K; ++k) {
      const float* p = base + (long long)s_idx[k] * h * w;
      const float vmin = fminf(fminf(p[i0 * w + j0], p[i0 * w + j1]), fminf(p[i1 * w + j0], p[i1 * w + j1]));
      const float vlo = vmin - 1e-5f * __builtin_fabsf(vmin) - 1e-6f;
      const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * vlo));
      floor_best = fmaxf(floor_best, s_score[k] * sg * 0.9999f);             // NaN taps: fmaxf keeps the old floor
    }
  }
  for (int k = 0; k < K; ++k) {
    const float* p = base + (long long)s_idx[k] * h * w;
    float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
    if (active) { v00 = p[i0 * w + j0]; v01 = p[i0 * w + j1]; v10 = p[i1 * w + j0]; v11 = p[i1 * w + j1]; }
    const float sc_k = s_score[k];
    if (SKIP) {
      const float vmax = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
      // e^{vmax (1 - 2^-18)} (1 + 1e-4) s_k: above s_k sigmoid(v) of every pixel of the block for vmax < -1e-3
      const float ub = sc_k * 1.0001f * __builtin_amdgcn_exp2f(vmax * (1.4426950408889634f * (1.f - 3.8e-6f)));
      const bool need = active && !(vmax < -1e-3f && vmax > -80.f && ub < fmaxf(minbest, floor_best));     // below -80: e^v leaves the normal range
      if (!__builtin_amdgcn_ballot_w64(need)) continue;       // wave-uniform: nobody's owner or counts can change
    }
    int conf_cnt = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 4 * bi - 2 + a;
      // ATen: src = 0.25*(y+0.5)-0.5 clamped at 0; lambda1 = src - floor(src)
      float fy = 0.25f * ((float)y + 0.5f) - 0.5f;
      fy = fy < 0.f ? 0.f : fy;
      const int yb = min((int)fy, h - 1);
      const float ly1 = fy - (float)yb, ly0 = 1.f - ly1;
      const bool rowok = active && y >= 0 && y < ih;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = 4 * bj - 2 + c;
        float fx = 0.25f * ((float)x + 0.5f) - 0.5f;
        fx = fx < 0.f ? 0.f : fx;
        const int xb = min((int)fx, w - 1);
        const float lx1 = fx - (float)xb, lx0 = 1.f - lx1;
        const bool ok = rowok && x >= 0 && x < iw;
        // rows i0/i1 and columns j0/j1 are this block's taps for every one of its 16 pixels (clamped
        // borders: the two taps coincide or the far weight is exactly 0), so only the weights vary
        const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        const float prob = 1.f / (1.f + expf(-v));
        const float sc = sc_k * prob;
        const int e = a * 4 + c;
        if (ok && sc > best[e]) { best[e] = sc; own[e] = k; pown[e] = prob; }
        conf_cnt += (ok && prob >= 0.5f) ? 1 : 0;
      }
    }
    if (SKIP) {
      float m = best[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) m = fminf(m, best[e]);
      minbest = m;
    }
    // per-query "original area": wave-sum of the per-lane counts
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) conf_cnt += __shfl_xor(conf_cnt, off);
    if (lane == 0 && conf_cnt) atomicAdd(&s_orig[k], conf_cnt);
  }
  if (active) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 4 * bi - 2 + a;
      if (y < 0 || y >= ih) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int x = 4 * bj - 2 + c;
        if (x < 0 || x >= iw) continue;
        const int e = a * 4 + c;
        const bool cf = pown[e] >= 0.5f;
        atomicAdd(&s_area[own[e]], 1);
        if (cf) atomicAdd(&s_region_conf[own[e]], 1);
        owner_out[((long long)t * ih + y) * iw + x] = (unsigned char)(own[e] | (cf ? 0x80 : 0));
      }
    }
  }
  __syncthreads();
  int* ct = counters + (long long)t * 3 * MAXK;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    if (s_area[k]) atomicAdd(ct + k, s_area[k]);
    if (s_orig[k]) atomicAdd(ct + MAXK + k, s_orig[k]);
    if (s_region_conf[k]) atomicAdd(ct + 2 * MAXK + k, s_region_conf[k]);
  }
}

// ATen's source index for one output coordinate of a bilinear resize in -> out (align_corners=False,
// no scale_factor: scale = in/out in float).
struct Tap { int i0; int ip; float l0, l1; };
__device__ __forceinline__ Tap bilinear_tap(int dst, int in, int out) {
  const float scale = (float)in / (float)out;
  float f = scale * ((float)dst + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  int i0 = (int)f;
  i0 = i0 > in - 1 ? in - 1 : i0;
  Tap t;
  t.i0 = i0;
  t.ip = (i0 < in - 1) ? 1 : 0;
  t.l1 = f - (float)i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// Both resizes composed: output pixel (oy,ox) of the (oh,ow) map <- bilinear over the (ih,iw) crop of the
// (H,W) map <- bilinear over the (h,w) stride-4 logits.
struct Tap2 {
  int r[4], c[4];          // source rows / columns: (stage-2 tap a) x (stage-1 tap)
  float ly[4], lx[4];      // stage-1 weights per (a, tap)
  float Ly0, Ly1, Lx0, Lx1;
};
__device__ __forceinline__ Tap2 make_tap2(int oy, int ox, int h, int w, int H, int W, int ih, int iw, int oh, int ow) {
  Tap2 t;
  const Tap ty = bilinear_tap(oy, ih, oh), tx = bilinear_tap(ox, iw, ow);
  t.Ly0 = ty.l0; t.Ly1 = ty.l1; t.Lx0 = tx.l0; t.Lx1 = tx.l1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const Tap sy = bilinear_tap(ty.i0 + a * ty.ip, h, H), sx = bilinear_tap(tx.i0 + a * tx.ip, w, W);
    t.r[2 * a] = sy.i0; t.r[2 * a + 1] = sy.i0 + sy.ip; t.ly[2 * a] = sy.l0; t.ly[2 * a + 1] = sy.l1;
    t.c[2 * a] = sx.i0; t.c[2 * a + 1] = sx.i0 + sx.ip; t.lx[2 * a] = sx.l0; t.lx[2 * a + 1] = sx.l1;
  }
  return t;
}
__device__ __forceinline__ float sample2(const float* __restrict__ p, int w, const Tap2& t) {
  float v[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float* r0 = p + (long long)t.r[2 * a] * w;
      const float* r1 = p + (long long)t.r[2 * a + 1] * w;
      v[a][b] = t.ly[2 * a] * (t.lx[2 * b] * r0[t.c[2 * b]] + t.lx[2 * b + 1] * r0[t.c[2 * b + 1]]) +
                t.ly[2 * a + 1] * (t.lx[2 * b] * r1[t.c[2 * b]] + t.lx[2 * b + 1] * r1[t.c[2 * b + 1]]);
    }
  return t.Ly0 * (t.Lx0 * v[0][0] + t.Lx1 * v[0][1]) + t.Ly1 * (t.Lx0 * v[1][0] + t.Lx1 * v[1][1]);
}

__global__ __launch_bounds__(256) void pan_owner_2stage_kernel(
    const float* __restrict__ logits, const int* __restrict__ kept_idx, const float* __restrict__ kept_score,
    unsigned char* __restrict__ owner_out, int* __restrict__ counters, int Q, int K, int h, int w,
    int H, int W, int ih, int iw, int oh, int ow, const int* __restrict__ kdev) {
  if (kdev) K = *kdev;                       // kept count decided on the device (pvsg_panoptic_select)
  __shared__ int s_area[MAXK], s_orig[MAXK], s_region_conf[MAXK];
  __shared__ int s_idx[MAXK];
  __shared__ float s_score[MAXK];
  const int t = blockIdx.z;
  for (int k = threadIdx.x; k < MAXK; k += blockDim.x) {
    s_area[k] = 0; s_orig[k] = 0; s_region_conf[k] = 0;
    if (k < K) { s_idx[k] = kept_idx[k]; s_score[k] = kept_score[k]; }
  }
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const bool inside = x < ow && y < oh;
  const Tap2 tp = make_tap2(inside ? y : 0, inside ? x : 0, h, w, H, W, ih, iw, oh, ow);
  const float* base = logits + (long long)t * Q * h * w;
  float best = -1.f, pown = 0.f;
  int own = 0;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < K; ++k) {
    const float v = inside ? sample2(base + (long long)s_idx[k] * h * w, w, tp) : 0.f;
    const float prob = 1.f / (1.f + expf(-v));
    const float sc = s_score[k] * prob;
    if (sc > best) { best = sc; own = k; pown = prob; }
    const unsigned long long conf = __ballot(inside && prob >= 0.5f);
    if (lane == 0 && conf) atomicAdd(&s_orig[k], __popcll(conf));
  }
  if (inside) {
    const bool c = pown >= 0.5f;
    atomicAdd(&s_area[own], 1);
    if (c) atomicAdd(&s_region_conf[own], 1);
    owner_out[((long long)t * oh + y) * ow + x] = (unsigned char)(own | (c ? 0x80 : 0));
  }
  __syncthreads();
  int* ct = counters + (long long)t * 3 * MAXK;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    if (s_area[k]) atomicAdd(ct + k, s_area[k]);
    if (s_orig[k]) atomicAdd(ct + MAXK + k, s_orig[k]);
    if (s_region_conf[k]) atomicAdd(ct + 2 * MAXK + k, s_region_conf[k]);
  }
}

// instance_postprocess (mask2former_fusion_head.py:192-242) for n selected queries of T frames: the binary
// mask `logit > 0` at output resolution, sum(sigmoid * binary), count(binary) and the box of the binary mask.
//   stat_sum (T,n) float64, stat_box (T,n,5) int32 = {count, min x, min y, max x, max y}
__global__ void inst_init_kernel(double* __restrict__ stat_sum, int* __restrict__ stat_box, int n, int big) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stat_sum[i] = 0.0;
  int* b = stat_box + 5ll * i;
  b[0] = 0; b[1] = big; b[2] = big; b[3] = -1; b[4] = -1;
}

// A thread owns a column of IM_ROWS output pixels (its column taps are computed once; the five statistics are reduced over
// the wave once per IM_ROWS pixels instead of per pixel -- the per-pixel form spent most of its 44 ms at 32 x 100 x 720p in
// the 42 cross-lane shuffles).  ONE_STAGE: output size == crop size, the second resize is the identity and the composed
// tap (sample2) degenerates to one bilinear tap with bit-identical arithmetic (its stage-2 weights are exactly 1 and 0).
constexpr int IM_ROWS = 8;
template <bool ONE_STAGE>
__global__ __launch_bounds__(256) void inst_masks_kernel(
    const float* __restrict__ logits, const int* __restrict__ sel_idx, unsigned char* __restrict__ masks,
    double* __restrict__ stat_sum, int* __restrict__ stat_box, int Q, int n, int sel_per_frame, int h, int w,
    int H, int W, int ih, int iw, int oh, int ow) {
  __shared__ double s_sum[4];
  __shared__ int s_box[4][5];
  const int e = blockIdx.z % n, t = blockIdx.z / n;
  const int wv = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ybase = (blockIdx.y * 4 + wv) * IM_ROWS;
  const float* p = logits + ((long long)t * Q + sel_idx[sel_per_frame ? t * n + e : e]) * h * w;
  const bool xin = x < ow;
  const Tap tx = bilinear_tap(xin ? x : 0, w, W);                       // ONE_STAGE only
  unsigned char* mrow = masks ? masks + (((long long)t * n + e) * oh) * ow + x : nullptr;
  double sg = 0.0;
  int cnt = 0, x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
#pragma unroll
  for (int i = 0; i < IM_ROWS; ++i) {
    const int y = ybase + i;
    const bool inside = xin && y < oh;
    float v;
    if (ONE_STAGE) {
      const Tap ty = bilinear_tap(inside ? y : 0, h, H);
      const float* r0 = p + (long long)ty.i0 * w;
      const float* r1 = r0 + (long long)ty.ip * w;
      v = ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i0 + tx.ip]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i0 + tx.ip]);
    } else {
      const Tap2 tp = make_tap2(inside ? y : 0, xin ? x : 0, h, w, H, W, ih, iw, oh, ow);
      v = sample2(p, w, tp);
    }
    const bool on = inside && v > 0.f;
    if (mrow && inside) mrow[(long long)y * ow] = on ? 1 : 0;
    if (on) {
      sg += (double)(1.f / (1.f + expf(-v)));
      ++cnt;
      y0 = y0 < y ? y0 : y;           // rows ascend: first / last 'on' row of the column
      y1 = y;
    }
  }
  if (cnt) { x0 = x; x1 = x; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sg += __shfl_xor(sg, off);
    cnt += __shfl_xor(cnt, off);
    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[wv] = sg; s_box[wv][0] = cnt; s_box[wv][1] = x0; s_box[wv][2] = y0; s_box[wv][3] = x1; s_box[wv][4] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      s_sum[0] += s_sum[i]; s_box[0][0] += s_box[i][0];
      s_box[0][1] = min(s_box[0][1], s_box[i][1]); s_box[0][2] = min(s_box[0][2], s_box[i][2]);
      s_box[0][3] = max(s_box[0][3], s_box[i][3]); s_box[0][4] = max(s_box[0][4], s_box[i][4]);
    }
    if (s_box[0][0]) {
      const long long o = (long long)t * n + e;
      atomicAdd(stat_sum + o, s_sum[0]);
      int* b = stat_box + 5 * o;
      atomicAdd(b, s_box[0][0]);
      atomicMin(b + 1, s_box[0][1]); atomicMin(b + 2, s_box[0][2]);
      atomicMax(b + 3, s_box[0][3]); atomicMax(b + 4, s_box[0][4]);
    }
  }
}

// Exact x4 up-sampling (H = 4h, W = 4w, no second resize, ow % 4 == 0 -- every /32-padded input at its own resolution): a thread
// owns 4 columns x 8 rows of output pixels.  Their taps are the 3 source columns j-1 .. j+1 and the 4 source rows 2b-1 .. 2b+2
// (clamped), loaded ONCE (12 loads per 32 pixels; the per-pixel form issues 4 per pixel), the weights still come from
// bilinear_tap and the expression is that of inst_masks_kernel<true>: masks, counts and boxes are identical
// (tests/test_postprocess.py::test_instance_masks_x4_kernel_equals_per_pixel_kernel), the sigmoid sums agree to f32 rounding; a
// row of four mask bytes leaves as one 4-byte store.  One 720p image x 100 queries: 373 -> ~110 us (post-process span of the
// shipped IPS flow 1.49 -> 1.23 ms per image).
__global__ __launch_bounds__(256) void inst_masks_x4_kernel(
    const float* __restrict__ logits, const int* __restrict__ sel_idx, unsigned char* __restrict__ masks,
    double* __restrict__ stat_sum, int* __restrict__ stat_box, int Q, int n, int sel_per_frame, int h, int w,
    int H, int W, int oh, int ow) {
  __shared__ double s_sum[4];
  __shared__ int s_box[4][5];
  const int e = blockIdx.z % n, t = blockIdx.z / n;
  const int wv = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);                  // source column: output columns 4j .. 4j+3
  const int b = blockIdx.y * 4 + wv;                                   // output rows 8b .. 8b+7
  const float* p = logits + ((long long)t * Q + sel_idx[sel_per_frame ? t * n + e : e]) * h * w;
  const bool xin = 4 * j < ow;
  const int jc = xin ? j : 0;
  const int cj[3] = {max(jc - 1, 0), jc, min(jc + 1, w - 1)};
  const int rbase = 2 * b - 1;
  float v[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = min(max(rbase + k, 0), h - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[k][c] = p[(long long)r * w + cj[c]];
  }
  Tap tx[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) tx[c] = bilinear_tap(min(4 * jc + c, ow - 1), w, W);
  double sg = 0.0;
  int cnt = 0, x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int y = 8 * b + i;
    const bool yin = y < oh;
    const Tap ty = bilinear_tap(yin ? y : 0, h, H);
    // rows ty.i0, ty.i0 + ty.ip of the four loaded ones (index - rbase in 0..3, see above)
    const int k0 = yin ? ty.i0 - rbase : 1, k1 = k0 + ty.ip;
    float r0[3], r1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r0[c] = k0 == 0 ? v[0][c] : k0 == 1 ? v[1][c] : k0 == 2 ? v[2][c] : v[3][c];
      r1[c] = k1 == 0 ? v[0][c] : k1 == 1 ? v[1][c] : k1 == 2 ? v[2][c] : v[3][c];
    }
    unsigned bytes = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // columns tx.i0, tx.i0 + tx.ip of the three loaded ones
      const int a0 = tx[c].i0, a1 = tx[c].i0 + tx[c].ip;
      const float p00 = a0 == cj[1] ? r0[1] : (a0 == cj[0] ? r0[0] : r0[2]), p01 = a1 == cj[1] ? r0[1] : (a1 == cj[2] ? r0[2] : r0[0]);
      const float p10 = a0 == cj[1] ? r1[1] : (a0 == cj[0] ? r1[0] : r1[2]), p11 = a1 == cj[1] ? r1[1] : (a1 == cj[2] ? r1[2] : r1[0]);
      const float val = ty.l0 * (tx[c].l0 * p00 + tx[c].l1 * p01) + ty.l1 * (tx[c].l0 * p10 + tx[c].l1 * p11);
      const int x = 4 * j + c;
      const bool on = xin && yin && x < ow && val > 0.f;
      if (on) {
        bytes |= 1u << (8 * c);
        sg += (double)(1.f / (1.f + expf(-val)));
        ++cnt;
        x0 = min(x0, x); x1 = max(x1, x);
        y0 = min(y0, y); y1 = y;
      }
    }
    if (masks && xin && yin) *reinterpret_cast<unsigned*>(masks + (((long long)t * n + e) * oh + y) * ow + 4 * j) = bytes;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sg += __shfl_xor(sg, off);
    cnt += __shfl_xor(cnt, off);
    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
  }
  if ((threadIdx.x & 63) == 0) {
    s_sum[wv] = sg; s_box[wv][0] = cnt; s_box[wv][1] = x0; s_box[wv][2] = y0; s_box[wv][3] = x1; s_box[wv][4] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      s_sum[0] += s_sum[i]; s_box[0][0] += s_box[i][0];
      s_box[0][1] = min(s_box[0][1], s_box[i][1]); s_box[0][2] = min(s_box[0][2], s_box[i][2]);
      s_box[0][3] = max(s_box[0][3], s_box[i][3]); s_box[0][4] = max(s_box[0][4], s_box[i][4]);
    }
    if (s_box[0][0]) {
      const long long o = (long long)t * n + e;
      atomicAdd(stat_sum + o, s_sum[0]);
      int* bx = stat_box + 5 * o;
      atomicAdd(bx, s_box[0][0]);
      atomicMin(bx + 1, s_box[0][1]); atomicMin(bx + 2, s_box[0][2]);
      atomicMax(bx + 3, s_box[0][3]); atomicMax(bx + 4, s_box[0][4]);
    }
  }
}

__global__ void pan_decide_kernel(const int* __restrict__ counters, const int* __restrict__ kept_class,
                                  int* __restrict__ seg_id, int K, int num_things, double iou_thr,
                                  int filter_low, const int* __restrict__ kdev, int seg_stride) {
  const int t = blockIdx.x;
  if (threadIdx.x != 0) return;
  if (kdev) {                                  // device-side kept count: rows of `seg_stride` ids, unused slots = -1
    K = *kdev;
    for (int k = K; k < seg_stride; ++k) seg_id[(long long)t * seg_stride + k] = -1;
  }
  const int* ct = counters + (long long)t * 3 * MAXK;
  int inst = 0;
  for (int k = 0; k < K; ++k) {
    const int area = ct[k], orig = ct[MAXK + k];
    const int region = filter_low ? ct[2 * MAXK + k] : area;
    bool ok = area > 0 && orig > 0;
    if (ok && ((double)area / (double)orig < iou_thr)) ok = false;
    if (ok && region <= 0) ok = false;
    const int cls = kept_class[k];
    int id = -1;
    if (ok) {
      if (cls < num_things) { ++inst; id = cls + inst * 1000; }
      else id = cls;
    }
    seg_id[(long long)t * seg_stride + k] = id;
  }
}

__global__ __launch_bounds__(256) void pan_paint_kernel(const unsigned char* __restrict__ owner,
                                                       const int* __restrict__ seg_id,
                                                       int* __restrict__ panoptic, int K, long long npix,
                                                       int num_classes, int filter_low, const int* __restrict__ kdev,
                                                       int seg_stride) {
  const int t = blockIdx.y;
  if (kdev) K = *kdev;
  __shared__ int s_id[MAXK];
  for (int k = threadIdx.x; k < MAXK; k += blockDim.x) s_id[k] = k < K ? seg_id[(long long)t * seg_stride + k] : -1;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned char o = owner[(long long)t * npix + i];
    const int id = s_id[o & 0x7f];
    const bool paint = id >= 0 && (!filter_low || (o & 0x80));
    panoptic[(long long)t * npix + i] = paint ? id : num_classes;
  }
}

}  // namespace pvsg

static int panoptic_fuse_run(const char* nm, const float* mask_logits, const int* kept_idx, const float* kept_score,
                            const int* kept_class, const int* kdev, int seg_stride, int* panoptic, int* seg_id,
                            unsigned char* owner_ws, int* counter_ws, int T, int Q, int K, int h, int w, int H, int W, int ih,
                            int iw, int oh, int ow, int num_things, int num_classes, double iou_thr, int filter_low_score,
                            hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(mask_logits && panoptic && owner_ws && counter_ws, "%s: null pointer argument", nm);
  PVSG_REQUIRE(T > 0 && Q > 0 && h > 0 && w > 0 && H > 0 && W > 0 && ih > 0 && iw > 0 && ih <= H && iw <= W &&
                   oh > 0 && ow > 0,
               "%s: bad geometry (h=%d w=%d H=%d W=%d ih=%d iw=%d oh=%d ow=%d)", nm, h, w, H, W, ih, iw, oh, ow);
  PVSG_REQUIRE(K >= 0 && K <= MAXK - 1, "%s: at most %d kept queries (got %d)", nm, MAXK - 1, K);
  PVSG_REQUIRE(K == 0 || (kept_idx && kept_score && kept_class && seg_id), "%s: null kept-query tables", nm);
  const long long npix = (long long)oh * ow;
  hipError_t e = zero_words_async(counter_ws, (size_t)T * 3 * MAXK * sizeof(int), stream);
  if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "%s: memset: %s", nm, hipGetErrorString(e));
  if (K > 0) {
    if (oh != ih || ow != iw)
      hipLaunchKernelGGL(pan_owner_2stage_kernel, dim3((ow + 63) / 64, (oh + 3) / 4, T), dim3(256), 0, stream,
                         mask_logits, kept_idx, kept_score, owner_ws, counter_ws, Q, K, h, w, H, W, ih, iw, oh, ow, kdev);
    else if (H == 4 * h && W == 4 * w) {
      const char* sk = getenv("PVSG_PAN_SKIP");                  // =0: every sigmoid evaluated (A/B tests)
      if (sk && sk[0] == '0')
        hipLaunchKernelGGL(pan_owner_x4_kernel<false>, dim3((w + 1 + 31) / 32, (h + 1 + 7) / 8, T), dim3(256), 0, stream,
                           mask_logits, kept_idx, kept_score, owner_ws, counter_ws, Q, K, h, w, ih, iw, kdev);
      else
        hipLaunchKernelGGL(pan_owner_x4_kernel<true>, dim3((w + 1 + 31) / 32, (h + 1 + 7) / 8, T), dim3(256), 0, stream,
                           mask_logits, kept_idx, kept_score, owner_ws, counter_ws, Q, K, h, w, ih, iw, kdev);
    }
    else
      hipLaunchKernelGGL(pan_owner_kernel, dim3((iw + 63) / 64, (ih + 3) / 4, T), dim3(256), 0, stream,
                         mask_logits, kept_idx, kept_score, owner_ws, counter_ws, Q, K, h, w, H, W, ih, iw, kdev);
    PVSG_LAUNCH_CHECK("panoptic_fuse(owner)");
    hipLaunchKernelGGL(pan_decide_kernel, dim3(T), dim3(64), 0, stream, counter_ws, kept_class, seg_id, K,
                       num_things, iou_thr, filter_low_score, kdev, seg_stride);
    PVSG_LAUNCH_CHECK("panoptic_fuse(decide)");
  }
  long long nb = (npix + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(pan_paint_kernel, dim3((unsigned)nb, T), dim3(256), 0, stream, owner_ws, seg_id,
                     panoptic, K, npix, num_classes, filter_low_score, kdev, seg_stride);
  PVSG_LAUNCH_CHECK("panoptic_fuse(paint)");
  return PVSG_OK;
}

extern "C" int pvsg_panoptic_fuse(const float* mask_logits, const int* kept_idx, const float* kept_score,
                                  const int* kept_class, int* panoptic, int* seg_id,
                                  unsigned char* owner_ws, int* counter_ws, int T, int Q, int K, int h,
                                  int w, int H, int W, int ih, int iw, int oh, int ow, int num_things,
                                  int num_classes, double iou_thr, int filter_low_score, hipStream_t stream) {
  return panoptic_fuse_run("panoptic_fuse", mask_logits, kept_idx, kept_score, kept_class, nullptr, K > 0 ? K : 1, panoptic,
                           seg_id, owner_ws, counter_ws, T, Q, K, h, w, H, W, ih, iw, oh, ow, num_things, num_classes, iou_thr,
                           filter_low_score, stream);
}

// The same with the kept set decided on the device: `sel` = the record pvsg_panoptic_select wrote (kept count in sel[0], the
// kept-query tables behind it); seg_id rows have PVSG_SEL_MAXK entries, unused ones -1.  No host round trip between the
// class decision and the fusion.
extern "C" int pvsg_panoptic_fuse_sel(const float* mask_logits, const int* sel, int* panoptic, int* seg_id,
                                      unsigned char* owner_ws, int* counter_ws, int T, int Q, int h, int w, int H, int W,
                                      int ih, int iw, int oh, int ow, int num_things, int num_classes, double iou_thr,
                                      int filter_low_score, hipStream_t stream) {
  PVSG_REQUIRE(sel && seg_id, "panoptic_fuse_sel: null pointer argument");
  return panoptic_fuse_run("panoptic_fuse_sel", mask_logits, sel + 4, reinterpret_cast<const float*>(sel + 4 + 2 * PVSG_SEL_MAXK),
                           sel + 4 + PVSG_SEL_MAXK, sel, PVSG_SEL_MAXK, panoptic, seg_id, owner_ws, counter_ws, T, Q,
                           pvsg::MAXK - 1, h, w, H, W, ih, iw, oh, ow, num_things, num_classes, iou_thr, filter_low_score, stream);
}

extern "C" int pvsg_instance_masks(const float* mask_logits, const int* sel_idx, unsigned char* masks,
                                   double* stat_sum, int* stat_box, int T, int Q, int n, int sel_per_frame,
                                   int h, int w, int H, int W, int ih, int iw, int oh, int ow,
                                   hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(mask_logits && stat_sum && stat_box, "instance_masks: null pointer argument");
  PVSG_REQUIRE(T > 0 && Q > 0 && n >= 0 && h > 0 && w > 0 && H > 0 && W > 0 && ih > 0 && iw > 0 && ih <= H &&
                   iw <= W && oh > 0 && ow > 0,
               "instance_masks: bad geometry (h=%d w=%d H=%d W=%d ih=%d iw=%d oh=%d ow=%d)", h, w, H, W, ih, iw, oh, ow);
  PVSG_REQUIRE((long long)T * n <= 65535, "instance_masks: T*n = %lld exceeds the grid limit 65535", (long long)T * n);
  if (n == 0) return PVSG_OK;
  PVSG_REQUIRE(sel_idx, "instance_masks: null selection table");
  hipLaunchKernelGGL(inst_init_kernel, dim3((T * n + 255) / 256), dim3(256), 0, stream, stat_sum, stat_box, T * n,
                     0x7fffffff);
  PVSG_LAUNCH_CHECK("instance_masks(init)");
  const dim3 grid((ow + 63) / 64, (oh + 4 * IM_ROWS - 1) / (4 * IM_ROWS), T * n);
  {
    const char* sel = getenv("PVSG_INST_X4");                    // =0: the per-pixel kernel (A/B tests)
    if (oh == ih && ow == iw && H == 4 * h && W == 4 * w && ow % 4 == 0 && h >= 2 && w >= 2 && !(sel && sel[0] == '0') &&
        (!masks || (reinterpret_cast<uintptr_t>(masks) & 3u) == 0)) {
      const dim3 g4((ow / 4 + 63) / 64, (oh + 31) / 32, T * n);
      hipLaunchKernelGGL(inst_masks_x4_kernel, g4, dim3(256), 0, stream, mask_logits, sel_idx, masks, stat_sum, stat_box, Q, n,
                         sel_per_frame, h, w, H, W, oh, ow);
      PVSG_LAUNCH_CHECK("instance_masks");
      return PVSG_OK;
    }
  }
  if (oh == ih && ow == iw)
    hipLaunchKernelGGL(inst_masks_kernel<true>, grid, dim3(256), 0, stream, mask_logits, sel_idx, masks, stat_sum, stat_box,
                       Q, n, sel_per_frame, h, w, H, W, ih, iw, oh, ow);
  else
    hipLaunchKernelGGL(inst_masks_kernel<false>, grid, dim3(256), 0, stream, mask_logits, sel_idx, masks, stat_sum, stat_box,
                       Q, n, sel_per_frame, h, w, H, W, ih, iw, oh, ow);
  PVSG_LAUNCH_CHECK("instance_masks");
  return PVSG_OK;
}
