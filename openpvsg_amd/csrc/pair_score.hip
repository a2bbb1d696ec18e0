// Pairwise relation proposal scorer: pair[i,j] = w2 . relu(W1 [s_i ; o_j] + b1) + b2, i != j.
//
// Replaces: models/relation_head/base.py:43-62 PairProposalNetwork.forward -- N(N-1) Python-level
// module calls on 512-vectors, each writing one scalar into a CPU matrix (0.45 s at N=100).
//
// Closed form: W1 = [W1s | W1o] (column blocks) so  W1 [s;o] = W1s s + W1o o.
//   prepare   (once per checkpoint) W1 (1024,512) -> W1T (2,256,1024): hidden index fastest, so a
//             wave reads 256 contiguous bytes per input channel.
//   kernel 1  tokens = max over frames (base.py:50-51); U[i,k] = b1[k] + sum_c W1s[k,c] s_i[c],
//             VT[k,j] = sum_c W1o[k,c] o_j[c].  A lane owns hidden unit k for 4 objects (the
//             weight stream is shared by 4 accumulators); tokens sit in LDS (broadcast reads).
//   kernel 2  block = (subject i, 64 objects); lane = object j, the 4 waves split the 1024 hidden
//             units, so U[i,k] and w2[k] are wave-uniform (scalar loads) and VT[k, j0..j0+63] is one
//             coalesced 256-byte read.  The four partial sums meet in LDS; the block writes 64
//             contiguous scores.  Diagonal = 0 exactly like torch.zeros(N, N) in the reference.
// Everything is KB-scale: the op is launch/latency bound (reported in microseconds, not as a
// roofline fraction -- SURVEY.md section 8d).
#include "common.h"

namespace pvsg {

constexpr int CF = 256;    // feature_dim
constexpr int HDN = 1024;  // hidden_dim
constexpr int OB = 4;      // objects per block in the projection kernel

__global__ __launch_bounds__(256) void pair_prepare_kernel(const float* __restrict__ W1,
                                                          float* __restrict__ W1T) {
  // W1T[which][c][k] = W1[k][which*256 + c]
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over 2*256*1024
  const int k = idx & (HDN - 1), c = (idx >> 10) & (CF - 1), which = idx >> 18;
  W1T[idx] = W1[(long long)k * (2 * CF) + which * CF + c];
}

__global__ __launch_bounds__(256) void pair_token_proj_kernel(
    const float* __restrict__ sub, const float* __restrict__ obj, const float* __restrict__ W1T,
    const float* __restrict__ b1, float* __restrict__ U, float* __restrict__ VT,
    float* __restrict__ tok_out, int N, int T) {
  __shared__ float tok[OB][CF];
  const int which = blockIdx.z;  // 0 = subject half, 1 = object half
  const int i0 = blockIdx.y * OB;
  const int k = blockIdx.x * 256 + threadIdx.x;
  const float* src = which == 0 ? sub : obj;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    const int i = i0 + ob;
    float m = 0.f;
    if (i < N) {
      const float* p = src + (long long)i * T * CF + threadIdx.x;
      m = -INFINITY;
      for (int t = 0; t < T; ++t) m = fmaxf(m, p[(long long)t * CF]);
      if (tok_out && blockIdx.x == 0) tok_out[((long long)which * N + i) * CF + threadIdx.x] = m;
    }
    tok[ob][threadIdx.x] = m;
  }
  __syncthreads();
  const float* wp = W1T + (long long)which * CF * HDN + k;
  float acc[OB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int c = 0; c < CF; ++c) {
    const float w = wp[(long long)c * HDN];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) acc[ob] += w * tok[ob][c];
  }
  if (which == 0) {
    const float bias = b1[k];
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
      if (i0 + ob < N) U[(long long)(i0 + ob) * HDN + k] = acc[ob] + bias;
  } else {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
      if (i0 + ob < N) VT[(long long)k * N + i0 + ob] = acc[ob];
  }
}

__global__ __launch_bounds__(256) void pair_score_kernel(const float* __restrict__ U,
                                                        const float* __restrict__ VT,
                                                        const float* __restrict__ w2,
                                                        const float* __restrict__ b2,
                                                        float* __restrict__ out, int N) {
  __shared__ float part[4][64];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int j = blockIdx.y * 64 + lane;
  const int jc = j < N ? j : N - 1;
  const float* up = U + (long long)i * HDN + ks * 256;   // wave-uniform -> scalar loads
  const float* wp = w2 + ks * 256;
  const float* vp = VT + (long long)(ks * 256) * N + jc;
  float acc = 0.f;
#pragma unroll 16
  for (int k = 0; k < 256; ++k) acc += wp[k] * fmaxf(up[k] + vp[(long long)k * N], 0.f);
  part[ks][lane] = acc;
  __syncthreads();
  if (ks == 0 && j < N) {
    const float s = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] + b2[0];
    out[(long long)i * N + j] = (j == i) ? 0.f : s;
  }
}

}  // namespace pvsg

extern "C" int pvsg_pair_prepare_weights(const float* W1, float* W1T, int C, int Hd, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(W1 && W1T, "pair_prepare_weights: null pointer argument");
  if (C != CF || Hd != HDN)
    return set_err(PVSG_ERR_UNSUPPORTED, "pair_prepare_weights: built for feature_dim 256 / hidden 1024 (got %d / %d)", C, Hd);
  hipLaunchKernelGGL(pair_prepare_kernel, dim3(2 * CF * HDN / 256), dim3(256), 0, stream, W1, W1T);
  PVSG_LAUNCH_CHECK("pair_prepare_weights");
  return PVSG_OK;
}

extern "C" int pvsg_pair_score_forward(const float* sub_feats, const float* obj_feats, const float* W1T,
                                       const float* b1, const float* w2, const float* b2,
                                       float* work_uv, float* tokens_out, float* pair_matrix, int N,
                                       int T, int C, int Hd, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(sub_feats && obj_feats && W1T && b1 && w2 && b2 && work_uv && pair_matrix,
               "pair_score_forward: null pointer argument");
  PVSG_REQUIRE(N > 0 && T > 0, "pair_score_forward: need at least one object and one frame (N=%d T=%d)", N, T);
  if (C != CF || Hd != HDN)
    return set_err(PVSG_ERR_UNSUPPORTED, "pair_score_forward: built for feature_dim 256 / hidden 1024 (got %d / %d)", C, Hd);
  float* U = work_uv;
  float* VT = work_uv + (long long)N * Hd;
  hipLaunchKernelGGL(pair_token_proj_kernel, dim3(HDN / 256, (N + OB - 1) / OB, 2), dim3(256), 0, stream,
                     sub_feats, obj_feats, W1T, b1, U, VT, tokens_out, N, T);
  PVSG_LAUNCH_CHECK("pair_score_forward(proj)");
  hipLaunchKernelGGL(pair_score_kernel, dim3(N, (N + 63) / 64), dim3(256), 0, stream, U, VT, w2, b2,
                     pair_matrix, N);
  PVSG_LAUNCH_CHECK("pair_score_forward(score)");
  return PVSG_OK;
}
