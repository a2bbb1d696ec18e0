// Fused per-channel affine (+ residual) (+ ReLU) over NCHW planes, in place or into `out`.
//
// Replaces, in the frozen-BN ResNet-50 backbone ([3P] mmdet ResNet, norm_eval=True; selected by
// configs/mask2former/..._custom_single_video_test.py:14-24): BatchNorm2d(eval) -> ReLU (2 passes) and
// BatchNorm2d(eval) -> + identity -> ReLU (3 passes) by ONE pass:  y = relu(x * scale[c] + shift[c] (+ r)),
// scale = gamma / sqrt(var + eps), shift = beta - mean * scale.  Pure HBM streaming: 8 B/element
// (12 with a residual), float4 accesses, one plane slice per block so scale/shift are block-uniform.
#include "common.h"

namespace pvsg {

template <bool VEC>
__global__ __launch_bounds__(256) void affine_act_nchw_kernel(const float* x, float* out, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ residual, int C, long long HW,
                                                             int relu) {
  const long long plane = blockIdx.x;   // planes on grid.x (up to 2^31-1), chunks of a plane on grid.y
  const int c = (int)(plane % C);
  const float sc = scale[c], sh = shift[c];
  const float* xp = x + plane * HW;
  float* op = out + plane * HW;
  const float* rp = residual ? residual + plane * HW : nullptr;
  if constexpr (VEC) {
    const long long n4 = HW >> 2;
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.y * blockDim.x) {
      float4 v = ld4(xp + 4 * i);
      v.x = v.x * sc + sh; v.y = v.y * sc + sh; v.z = v.z * sc + sh; v.w = v.w * sc + sh;
      if (rp) { const float4 r = ld4(rp + 4 * i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      st4(op + 4 * i, v);
    }
  } else {
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.y * blockDim.x) {
      float v = xp[i] * sc + sh;
      if (rp) v += rp[i];
      op[i] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

}  // namespace pvsg

extern "C" int pvsg_affine_act_nchw(float* x, const float* scale, const float* shift, const float* residual,
                                    float* out, long long planes, int C, long long HW, int relu, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(x && scale && shift, "affine_act_nchw: null pointer argument");
  PVSG_REQUIRE(planes > 0 && C > 0 && HW > 0 && planes % C == 0, "affine_act_nchw: bad shape (planes=%lld C=%d HW=%lld)", planes, C, HW);
  PVSG_REQUIRE(planes < (1LL << 31), "affine_act_nchw: too many planes");
  if (!out) out = x;
  const bool vec = !(HW & 3) && !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual) |
                                   reinterpret_cast<uintptr_t>(out)) & 15u);
  long long per = vec ? (HW >> 2) : HW;
  int bx = (int)((per + 255) / 256);
  if (bx > 64) bx = 64;
  dim3 grid((unsigned)planes, bx);
  if (vec)
    hipLaunchKernelGGL((affine_act_nchw_kernel<true>), grid, dim3(256), 0, stream, x, out, scale, shift, residual, C, HW, relu);
  else
    hipLaunchKernelGGL((affine_act_nchw_kernel<false>), grid, dim3(256), 0, stream, x, out, scale, shift, residual, C, HW, relu);
  PVSG_LAUNCH_CHECK("affine_act_nchw");
  return PVSG_OK;
}
