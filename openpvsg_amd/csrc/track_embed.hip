// IPS tube association: per-object appearance embeddings for the tracker.
//
// Replaces MaskAssociationTracker.extract_emb (models/unitrack/mask.py:21-47): for every object the
// reference multiplies the whole (1,d,h,w) feature map by the object's mask, resizes the WHOLE product
// bilinearly by sqrt(max_mask_area / area) when the object covers more than max_mask_area cells, resizes
// the mask (nearest) the same way and keeps the columns under it.  Only those <= 300 columns are ever
// used, so this kernel evaluates exactly them: one wave per kept output cell, the four bilinear taps of
// torch's upsample_bilinear2d(align_corners=False, scale given) with the mask applied per tap, all d
// channels from an NHWC copy of the features (one contiguous 4 KB row per tap), plus the L2-normalised
// row the reconstruction distance needs (F.normalize over channels, matching.py:193-194).
#include "common.h"

namespace pvsg {

__global__ __launch_bounds__(256) void mask_embed_kernel(
    const float* __restrict__ feat,      // (h, w, d)
    const int* __restrict__ pan_low,     // (h, w) object id per feature cell
    const int* __restrict__ entries,     // (k, 3): object slot, out row, out col (row-major per object)
    const int* __restrict__ obj_id,      // (n) id looked up in pan_low
    const float* __restrict__ obj_scale, // (n) float(1 / scale_factor); 1 = no rescale
    float* __restrict__ out, float* __restrict__ out_n, int h, int w, int d, int k) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= k) return;
  const int lane = threadIdx.x & 63;
  const int o = entries[3 * e], oy = entries[3 * e + 1], ox = entries[3 * e + 2];
  const int id = obj_id[o];
  const float sc = obj_scale[o];
  // area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
  const float sy = fmaxf(sc * ((float)oy + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(sc * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
  const int y1 = y0 + ((y0 < h - 1) ? 1 : 0), x1 = x0 + ((x0 < w - 1) ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float m00 = pan_low[y0 * w + x0] == id ? 1.f : 0.f, m01 = pan_low[y0 * w + x1] == id ? 1.f : 0.f;
  const float m10 = pan_low[y1 * w + x0] == id ? 1.f : 0.f, m11 = pan_low[y1 * w + x1] == id ? 1.f : 0.f;
  const float* r00 = feat + ((long long)y0 * w + x0) * d;
  const float* r01 = feat + ((long long)y0 * w + x1) * d;
  const float* r10 = feat + ((long long)y1 * w + x0) * d;
  const float* r11 = feat + ((long long)y1 * w + x1) * d;
  float* dst = out + (long long)e * d;
  float ss = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 a = ld4(r00 + c), b = ld4(r01 + c), g = ld4(r10 + c), q = ld4(r11 + c);
    float4 v;
    v.x = ly0 * (lx0 * (a.x * m00) + lx1 * (b.x * m01)) + ly1 * (lx0 * (g.x * m10) + lx1 * (q.x * m11));
    v.y = ly0 * (lx0 * (a.y * m00) + lx1 * (b.y * m01)) + ly1 * (lx0 * (g.y * m10) + lx1 * (q.y * m11));
    v.z = ly0 * (lx0 * (a.z * m00) + lx1 * (b.z * m01)) + ly1 * (lx0 * (g.z * m10) + lx1 * (q.z * m11));
    v.w = ly0 * (lx0 * (a.w * m00) + lx1 * (b.w * m01)) + ly1 * (lx0 * (g.w * m10) + lx1 * (q.w * m11));
    st4(dst + c, v);
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (out_n) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    float* dn = out_n + (long long)e * d;
    for (int c = lane * 4; c < d; c += 256) {
      float4 v = ld4(dst + c);   // written by this lane above
      v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
      st4(dn + c, v);
    }
  }
}

}  // namespace pvsg

extern "C" int pvsg_mask_embed_forward(const float* feat_hwd, const int* pan_low, const int* entries,
                                       const int* obj_id, const float* obj_inv_scale, float* out,
                                       float* out_normalised, int h, int w, int d, int k, int n_obj,
                                       hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(feat_hwd && pan_low && out, "mask_embed_forward: null pointer argument");
  PVSG_REQUIRE(h > 0 && w > 0 && d > 0 && k >= 0 && n_obj >= 0, "mask_embed_forward: negative dimension");
  if (k == 0) return PVSG_OK;
  PVSG_REQUIRE(entries && obj_id && obj_inv_scale && n_obj > 0, "mask_embed_forward: entries without objects");
  PVSG_REQUIRE(!(d & 3), "mask_embed_forward: channel count must be a multiple of 4");
  PVSG_REQUIRE(!((reinterpret_cast<uintptr_t>(feat_hwd) | reinterpret_cast<uintptr_t>(out) |
                  reinterpret_cast<uintptr_t>(out_normalised)) & 15u),
               "mask_embed_forward: 16-byte alignment required");
  hipLaunchKernelGGL(mask_embed_kernel, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, stream, feat_hwd, pan_low,
                     entries, obj_id, obj_inv_scale, out, out_normalised, h, w, d, k);
  PVSG_LAUNCH_CHECK("mask_embed_forward");
  return PVSG_OK;
}
