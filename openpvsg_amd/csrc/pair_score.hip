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
#pragma unroll 8
      for (int t = 0; t < T; ++t) m = fmaxf(m, p[(long long)t * CF]);      // 8 independent loads in flight
      if (tok_out && blockIdx.x == 0) tok_out[((long long)which * N + i) * CF + threadIdx.x] = m;
    }
    tok[ob][threadIdx.x] = m;
  }
  __syncthreads();
  const float* wp = W1T + (long long)which * CF * HDN + k;
  float acc[OB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 32
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

// ------------------------------------------------------------------------------------------------
// pick_top_pairs_eval (models/relation_head/test_utils.py:4-22): the P best off-diagonal entries of the pair matrix, best first,
// as (subject, object) index pairs.  One workgroup: radix select of the P-th largest key (4 passes over 8 bits), unordered
// collection of the larger entries, index-ordered collection of the entries equal to the threshold, rank sort of the P
// candidates.  Replaces clone + fill_diagonal_ + topk (radix sort + merge) + div + remainder + stack (12 launches, 70 us).
// Ties go to the lower flat index (torch.topk leaves their order unspecified).
// ------------------------------------------------------------------------------------------------
constexpr int TP_THREADS = 1024;
constexpr int TP_MAXP = 1024;
constexpr int TP_SLOTS = 16;                                // keys per thread, held in registers: N^2 <= 16 384

__device__ __forceinline__ unsigned tp_key(float v) {      // order-preserving: larger float <-> larger unsigned; NaN on top
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// count over the block of a per-thread number (every thread gets the total); two barriers
__device__ __forceinline__ int tp_block_sum(int v, int* wave_tot, int lane, int w) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) wave_tot[w] = v;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int ww = 0; ww < TP_THREADS / 64; ++ww) tot += wave_tot[ww];
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(TP_THREADS) void top_pairs_kernel(const float* __restrict__ m, long long* __restrict__ pairs, int N,
                                                               int P) {
  __shared__ unsigned cand_key[TP_MAXP];
  __shared__ int cand_idx[TP_MAXP];
  __shared__ int s_ngt;
  __shared__ int wave_tot[TP_THREADS / 64];
  const int n = N * N, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // slot s of thread t = flat element s * 1024 + t (consecutive threads = consecutive elements); key 0 = not a candidate
  unsigned key[TP_SLOTS];
#pragma unroll
  for (int s = 0; s < TP_SLOTS; ++s) {
    const int i = s * TP_THREADS + tid;
    unsigned k = 0u;
    if (i < n) {
      const int r = i / N;
      if (i - r * N != r) k = tp_key(m[i]);                // the diagonal sorts below everything
    }
    key[s] = k;
  }
  if (tid == 0) s_ngt = 0;
  // the P-th largest key, bit by bit from the top: the candidate prefix with the next bit set is kept if at least `need` keys
  // carry it (a histogram over float keys would pile its LDS atomics onto the two or three exponent bins the values live in)
  unsigned prefix = 0u;
  int need = P;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned cand = prefix | (1u << bit), himask = 0xFFFFFFFFu << bit;
    int c = 0;
#pragma unroll
    for (int s = 0; s < TP_SLOTS; ++s) c += (key[s] & himask) == cand ? 1 : 0;
    const int tot = tp_block_sum(c, wave_tot, lane, w);
    if (tot >= need) prefix = cand;                        // the P-th largest has this bit set
    else need -= tot;                                      // all `tot` keys with the bit set are larger than it
  }
  // prefix = key of the P-th best entry; `need` entries equal to it are taken (lowest index first), everything above it is
#pragma unroll
  for (int s = 0; s < TP_SLOTS; ++s) {
    if (key[s] > prefix) {
      const int pos = atomicAdd(&s_ngt, 1);
      cand_key[pos] = key[s];
      cand_idx[pos] = s * TP_THREADS + tid;
    }
  }
  __syncthreads();
  const int ngt = s_ngt;                                   // == P - need
  int taken = 0;
#pragma unroll
  for (int s = 0; s < TP_SLOTS; ++s) {
    if (taken < need && s * TP_THREADS < n) {              // uniform
      const bool flag = key[s] == prefix && prefix != 0u;
      const unsigned long long bal = __ballot(flag);
      const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[w] = __popcll(bal);
      __syncthreads();
      int off = 0, tot = 0;
#pragma unroll
      for (int ww = 0; ww < TP_THREADS / 64; ++ww) {
        const int c = wave_tot[ww];
        off += ww < w ? c : 0;
        tot += c;
      }
      const int pos = taken + off + wpre;
      if (flag && pos < need) {
        cand_key[ngt + pos] = prefix;
        cand_idx[ngt + pos] = s * TP_THREADS + tid;
      }
      taken += tot;
      __syncthreads();
    }
  }
  __syncthreads();
  if (tid < P) {
    const unsigned k = cand_key[tid];
    const int ix = cand_idx[tid];
    int rank = 0;
    for (int j = 0; j < P; ++j) {
      const unsigned kj = cand_key[j];
      rank += (kj > k || (kj == k && cand_idx[j] < ix)) ? 1 : 0;
    }
    pairs[2 * rank] = ix / N;
    pairs[2 * rank + 1] = ix % N;
  }
}

}  // namespace pvsg

extern "C" int pvsg_top_pairs(const float* pair_matrix, long long* pairs, int N, int P, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(pair_matrix && pairs, "top_pairs: null pointer argument");
  PVSG_REQUIRE(N > 1 && P > 0 && (long long)P <= (long long)N * N - N, "top_pairs: need 1 <= P <= N^2 - N (N=%d P=%d)", N, P);
  if (P > TP_MAXP || N > 128)
    return set_err(PVSG_ERR_UNSUPPORTED, "top_pairs: at most %d pairs of at most 128 objects (got %d / %d)", TP_MAXP, P, N);
  hipLaunchKernelGGL(top_pairs_kernel, dim3(1), dim3(TP_THREADS), 0, stream, pair_matrix, pairs, N, P);
  PVSG_LAUNCH_CHECK("top_pairs");
  return PVSG_OK;
}

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
