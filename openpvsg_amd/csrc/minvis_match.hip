// On-device MinVIS query matching across the frames of a video.
//
// Replaces: Mask2FormerVideoCustomMinVIS.match_from_embds (models/mask2former_vps/mask2former_min_vis.py:
// 244-258: L2-normalise, cost = 1 - cur.tgt^T, `C.cpu()`, scipy.optimize.linear_sum_assignment on C^T) and
// the frame chaining loop of models/mask2former_vps/mask2former.py:146-158 -- one host sync + one CPU LAP
// per frame in the reference; here ONE launch walks all T frames of a video:
//   for t = 1..T-1:  tgt_j = embds[t-1][perm[t-1][j]],  cur_i = embds[t][i]
//                    cost[j][i] = 1 - <cur_i, tgt_j> / (|cur_i| |tgt_j|)          (whole workgroup, LDS)
//                    perm[t] = argmin assignment (rows = target slots j, columns = current queries i)
// The assignment is the exact shortest-augmenting-path (Hungarian / JV) optimum, float64 potentials like
// scipy's solver, run by ONE wave: a lane owns columns lane and lane+64 (Q <= 128), the row potentials live
// in LDS, the per-step arg-min is a wave reduction.  Ties are measure-zero on real embeddings.
#include "common.h"
#include <stdlib.h>

namespace pvsg {

constexpr int MQ = 128;   // max queries

__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(idx, off);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

// The cosine table of a frame pair does not depend on the chain: cost[j][i] = 1 - G_t[perm[t-1][j]][i] with
// G_t[a][i] = <prev_a, cur_i> / (|prev_a| |cur_i|).  All T-1 tables are therefore computed up front by the whole GPU
// (minvis_gram_kernel: one workgroup per (video, t, row a)), and the single-workgroup chain kernel only gathers permuted rows
// and runs the sequential assignments: 38 -> ~9 ms for 32 frames (the in-chain table took 2/3 of a step on one CU).
__global__ __launch_bounds__(128) void minvis_gram_kernel(const float* __restrict__ embds, float* __restrict__ gram, int T,
                                                         int Q, int C) {
  // blockIdx.x = (vid * (T-1) + (t-1)) * Q + a
  const int a = blockIdx.x % Q;
  const long long vt = blockIdx.x / Q;
  const int t = (int)(vt % (T - 1)) + 1;
  const long long vid = vt / (T - 1);
  const float* E = embds + vid * T * Q * C;
  const float* pa = E + ((long long)(t - 1) * Q + a) * C;
  __shared__ float na;
  if (threadIdx.x == 0) {                    // same summation order as the per-row norm of the chain kernel
    float s = 0.f;
    for (int c = 0; c < C; c += 4) { const float4 x = ld4(pa + c); s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
    na = 1.f / sqrtf(s);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Q; i += blockDim.x) {
    const float* ci = E + ((long long)t * Q + i) * C;
    float s = 0.f, n2 = 0.f;
    for (int c = 0; c < C; c += 4) {
      const float4 x = ld4(ci + c), y = ld4(pa + c);
      s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      n2 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    gram[(vt * Q + a) * Q + i] = 1.f - s * (1.f / sqrtf(n2)) * na;
  }
}

// One wave: assignment of the Q x Q cost matrix in LDS (rows = target slots, columns = current queries) by shortest
// augmenting paths in float64, rows taken in order like scipy's linear_sum_assignment ([3P], called from
// mask2former_min_vis.py:257).  pnew[row] = column.  `u` = Q+1 row potentials in LDS.
__device__ __forceinline__ void lap_wave(const float* __restrict__ cost, double* __restrict__ u, int* __restrict__ pnew,
                                         int Q, int lane) {
    const int c0 = lane + 1, c1 = lane + 65;          // 1-based columns owned by this lane
    const bool has0 = c0 <= Q, has1 = c1 <= Q;
    double v0 = 0.0, v1 = 0.0;                         // column potentials
    int p0 = 0, p1 = 0;                                // row matched to the column (0 = free)
    for (int r = lane; r <= Q; r += 64) u[r] = 0.0;
    for (int i = 1; i <= Q; ++i) {
      double minv0 = 1e300, minv1 = 1e300;
      int way0 = 0, way1 = 0;
      bool used0 = false, used1 = false;
      int j0 = 0;                                      // current column (0 = virtual column holding row i)
      int i0 = i;
      for (;;) {
        if (j0 == c0) used0 = true;
        if (j0 == c1) used1 = true;
        const double ui0 = u[i0];
        double best = 1e300;
        int bj = 0x7fffffff;
        if (has0 && !used0) {
          const double curv = (double)cost[(i0 - 1) * Q + (c0 - 1)] - ui0 - v0;
          if (curv < minv0) { minv0 = curv; way0 = j0; }
          if (minv0 < best) { best = minv0; bj = c0; }
        }
        if (has1 && !used1) {
          const double curv = (double)cost[(i0 - 1) * Q + (c1 - 1)] - ui0 - v1;
          if (curv < minv1) { minv1 = curv; way1 = j0; }
          if (minv1 < best || (minv1 == best && c1 < bj)) { best = minv1; bj = c1; }
        }
        wave_argmin(best, bj);
        const double delta = best;
        // update potentials: used columns (and the virtual column 0 = row i)
        if (used0) { u[p0] += delta; v0 -= delta; } else if (has0) minv0 -= delta;
        if (used1) { u[p1] += delta; v1 -= delta; } else if (has1) minv1 -= delta;
        if (lane == 0) u[i] += delta;                   // p[0] = i
        __builtin_amdgcn_wave_barrier();
        j0 = bj;
        // row matched to column j0 (owned by one lane) -> broadcast
        int pj = (j0 == c0) ? p0 : (j0 == c1 ? p1 : 0);
        const int owner = (j0 - 1) & 63;
        pj = __shfl(pj, owner);
        if (pj == 0) break;
        i0 = pj;
      }
      // augment along `way`: walk back from j0 to the virtual column
      while (j0 != 0) {
        const int owner = (j0 - 1) & 63;
        int w = (j0 == c0) ? way0 : (j0 == c1 ? way1 : 0);
        w = __shfl(w, owner);                          // previous column on the path
        int pw = 0;                                    // row currently at column w (or i if w is virtual)
        if (w == 0) pw = i;
        else {
          const int wo = (w - 1) & 63;
          int t2 = (w == c0) ? p0 : (w == c1 ? p1 : 0);
          pw = __shfl(t2, wo);
        }
        if (j0 == c0) p0 = pw;
        if (j0 == c1) p1 = pw;
        j0 = w;
      }
    }
    // p[col] = row: perm[j = row-1] = i = col-1
    if (has0) pnew[p0 - 1] = c0 - 1;
    if (has1) pnew[p1 - 1] = c1 - 1;
}

__global__ __launch_bounds__(256) void minvis_chain_kernel(const float* __restrict__ embds, const float* __restrict__ gram,
                                                          int* __restrict__ perm_out, int T, int Q, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cost = sm;                         // [Q][Q]  rows = target slot j, cols = current query i
  float* ncur = cost + ((Q * Q + 1) & ~1);  // [Q] 1/|cur_i|  (cost region rounded to 8 bytes: `u` below is float64)
  float* ntgt = ncur + MQ;                  // [Q] 1/|tgt_j|
  double* u = reinterpret_cast<double*>(ntgt + MQ);   // [Q+1] row potentials (1-based)
  int* pprev = reinterpret_cast<int*>(u + MQ + 1);    // [Q] previous frame's permutation
  int* pnew = pprev + MQ;                             // [Q]
  const int vid = blockIdx.x, tid = threadIdx.x;
  int* P = perm_out + (long long)vid * T * Q;
  for (int j = tid; j < Q; j += blockDim.x) { pprev[j] = j; P[j] = j; }
  __syncthreads();

  for (int t = 1; t < T; ++t) {
    // ---- cost rows of this step: row j = table row of the previous frame's query that sits in slot j -------------
    const float* G = gram + ((long long)vid * (T - 1) + (t - 1)) * Q * Q;
    for (int e = tid; e < Q * Q; e += blockDim.x) {
      const int j = e / Q, i = e - j * Q;
      cost[e] = G[pprev[j] * Q + i];
    }
    __syncthreads();
    // ---- assignment: one wave, shortest augmenting paths -------------------------------------------------
    if (tid < 64) lap_wave(cost, u, pnew, Q, tid);
    __syncthreads();
    for (int j = tid; j < Q; j += blockDim.x) { pprev[j] = pnew[j]; P[(long long)t * Q + j] = pnew[j]; }
    __syncthreads();
  }
}


// The chain without its serial dependency.  Step t of the reference solves the assignment of
// cost[j][i] = G_t[pprev[j]][i] -- the raw frame-pair matrix G_t with its ROWS permuted by the previous step's result.  An
// optimal assignment of a row-permuted matrix is the permuted optimal assignment of the matrix, so when the optimum is
// unique (always, short of exactly tied float64 totals) the chain is  perm_t[j] = sigma_t[perm_{t-1}[j]]  with
// sigma_t = assignment of the raw G_t: the T-1 solves are independent (one wave each, all in flight together) and the
// chain is a composition of permutations.  26 ms -> 1 ms for 32 frames x 100 queries.  PVSG_MINVIS=chain selects the
// literal serial form (it also fixes the order in which exact ties would break).
__global__ __launch_bounds__(64) void minvis_pair_kernel(const float* __restrict__ gram, int* __restrict__ sigma, int Q) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* cost = sm;
  double* u = reinterpret_cast<double*>(cost + ((Q * Q + 1) & ~1));
  int* pnew = reinterpret_cast<int*>(u + MQ + 1);
  const long long vt = blockIdx.x;
  const float* G = gram + vt * Q * Q;
  for (int e = threadIdx.x; e < Q * Q; e += 64) cost[e] = G[e];
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  lap_wave(cost, u, pnew, Q, threadIdx.x);
  __syncthreads();
  for (int j = threadIdx.x; j < Q; j += 64) sigma[vt * Q + j] = pnew[j];
}

__global__ void minvis_compose_kernel(const int* __restrict__ sigma, int* __restrict__ perm_out, int T, int Q) {
  const int vid = blockIdx.x, j = threadIdx.x;
  if (j >= Q) return;
  int* P = perm_out + (long long)vid * T * Q;
  int cur = j;
  P[j] = j;
  for (int t = 1; t < T; ++t) {
    cur = sigma[((long long)vid * (T - 1) + (t - 1)) * Q + cur];
    P[(long long)t * Q + j] = cur;
  }
}

}  // namespace pvsg

extern "C" long long pvsg_minvis_chain_workspace_bytes(int V, int T, int Q) {
  return T > 1 ? (long long)V * (T - 1) * Q * (Q + 1) * 4 : 4;      // frame-pair cost matrices + their assignments
}

extern "C" int pvsg_minvis_chain(const float* embds, int* perm, float* workspace, int V, int T, int Q, int C, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(embds && perm && workspace, "minvis_chain: null pointer argument");
  PVSG_REQUIRE(V > 0 && T > 0 && Q > 0 && C > 0, "minvis_chain: non-positive dimension");
  if (Q > MQ || (C & 3) || (reinterpret_cast<uintptr_t>(embds) & 15u))
    return set_err(PVSG_ERR_UNSUPPORTED, "minvis_chain: needs Q <= 128, C %% 4 == 0, 16-byte aligned embeddings (Q=%d C=%d)", Q, C);
  const size_t lds = (size_t)((Q * Q + 1) & ~1) * 4 + 2 * MQ * 4 + (MQ + 1) * 8 + 2 * MQ * 4 + 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&minvis_chain_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (T > 1)
    hipLaunchKernelGGL(minvis_gram_kernel, dim3((unsigned)((long long)V * (T - 1) * Q)), dim3(128), 0, stream, embds, workspace, T,
                       Q, C);
  const char* mode = getenv("PVSG_MINVIS");
  if (T == 1 || (mode && mode[0] == 'c')) {
    hipLaunchKernelGGL(minvis_chain_kernel, dim3(V), dim3(256), lds, stream, embds, workspace, perm, T, Q, C);
  } else {
    int* sigma = reinterpret_cast<int*>(workspace + (size_t)V * (T - 1) * Q * Q);
    const size_t lds2 = (size_t)((Q * Q + 1) & ~1) * 4 + (MQ + 1) * 8 + MQ * 4 + 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&minvis_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds2);
    hipLaunchKernelGGL(minvis_pair_kernel, dim3((unsigned)(V * (T - 1))), dim3(64), lds2, stream, workspace, sigma, Q);
    hipLaunchKernelGGL(minvis_compose_kernel, dim3(V), dim3(MQ), 0, stream, sigma, perm, T, Q);
  }
  PVSG_LAUNCH_CHECK("minvis_chain");
  return PVSG_OK;
}
