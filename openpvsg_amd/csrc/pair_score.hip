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

// 1024 threads: thread (q = tid & 255, part = tid >> 8).  Token phase: part = object of the block's four, q = channel, all
// frames of the object in flight at once (the serial form -- 4 objects x T dependent-latency batches per thread -- was 2/3 of
// this kernel's 34 us).  Projection: hidden unit k = 256 blockIdx.x + q, part = quarter of the 256 input channels, the four
// partial sums meet in LDS.
__global__ __launch_bounds__(1024) void pair_token_proj_kernel(
    const float* __restrict__ sub, const float* __restrict__ obj, const float* __restrict__ W1T,
    const float* __restrict__ b1, float* __restrict__ U, float* __restrict__ VT,
    float* __restrict__ tok_out, int N, int T) {
  __shared__ float tok[OB][CF];
  __shared__ float part_sum[3][OB][256];
  const int which = blockIdx.z;  // 0 = subject half, 1 = object half
  const int i0 = blockIdx.y * OB;
  const int q = threadIdx.x & 255, part = threadIdx.x >> 8;
  const int k = blockIdx.x * 256 + q;
  const float* src = which == 0 ? sub : obj;
  {
    const int i = i0 + part;
    float m = 0.f;
    if (i < N) {
      const float* p = src + (long long)i * T * CF + q;
      float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int t = 0;
      for (; t + 16 <= T; t += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(long long)(t + u) * CF];
#pragma unroll
        for (int u = 0; u < 16; ++u) m4[u & 3] = fmaxf(m4[u & 3], v[u]);
      }
      for (; t < T; ++t) m4[0] = fmaxf(m4[0], p[(long long)t * CF]);
      m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
      if (tok_out && blockIdx.x == 0) tok_out[((long long)which * N + i) * CF + q] = m;
    }
    tok[part][q] = m;
  }
  __syncthreads();
  const float* wp = W1T + ((long long)which * CF + part * 64) * HDN + k;
  float acc[OB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int c0 = 0; c0 < 64; c0 += 32) {
    float wv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wv[u] = wp[(long long)(c0 + u) * HDN];
#pragma unroll
    for (int u = 0; u < 32; ++u)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) acc[ob] += wv[u] * tok[ob][part * 64 + c0 + u];
  }
  if (part > 0) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) part_sum[part - 1][ob][q] = acc[ob];
  }
  __syncthreads();
  if (part > 0) return;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) acc[ob] = ((acc[ob] + part_sum[0][ob][q]) + part_sum[1][ob][q]) + part_sum[2][ob][q];
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

__global__ __launch_bounds__(TP_THREADS) void top_pairs_kernel(const float* __restrict__ m, long long* __restrict__ pairs, int N,
                                                               int P) {
  __shared__ unsigned long long cand[TP_MAXP];             // (key << 32) | ~flat index: larger = better, ties to the lower index
  __shared__ int rank_of[TP_MAXP];
  __shared__ int digit_cnt[16][4];                         // per select step: keys with digit >= 3 / 2 / 1 / 0 under the prefix
  __shared__ int s_ngt, s_eq;
  __shared__ int wave_tot[TP_THREADS / 64];
  const int n = N * N, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // slot s of thread t = flat element s * 1024 + t (consecutive threads = consecutive elements); key 0 = not a candidate
  unsigned key[TP_SLOTS];
  {
    float v[TP_SLOTS];                                     // every load issued before the first use (a load inside each
#pragma unroll                                             // `if (i < n)` made sixteen dependent memory latencies)
    for (int s = 0; s < TP_SLOTS; ++s) {
      const int i = s * TP_THREADS + tid;
      v[s] = m[i < n ? i : 0];
    }
    const int r0 = tid / N, c0 = tid - r0 * N;              // element s * 1024 + tid: (row, col) advance by 1024 = qN * N + rN
    const int qN = TP_THREADS / N, rN = TP_THREADS - qN * N;
    int r = r0, c = c0;
#pragma unroll
    for (int s = 0; s < TP_SLOTS; ++s) {
      const int i = s * TP_THREADS + tid;
      key[s] = (i < n && r != c) ? tp_key(v[s]) : 0u;      // the diagonal sorts below everything
      r += qN;
      c += rN;
      if (c >= N) { c -= N; ++r; }
    }
  }
  if (tid < 64) (&digit_cnt[0][0])[tid] = 0;
  if (tid == 0) { s_ngt = 0; s_eq = 0; }
  if (tid < TP_MAXP) rank_of[tid] = 0;
  __syncthreads();
  // The P-th largest key, two bits at a time from the top: per digit value d = 3, 2, 1 the number of keys that share the prefix
  // found so far and whose digit is >= d -- wave ballots + scalar pop-counts, three LDS atomics per wave and ONE barrier per
  // step (a histogram over float keys piles its atomics onto the two or three exponent bins the values live in: 66 us for 10^4
  // keys; sixteen waves each summing the sixteen per-wave counts made the LDS the limit: 15 us for an empty matrix).
  unsigned prefix = 0u;
  int need = P;
  bool take_all = false;                                   // every key >= prefix is selected (the step loop ended early)
  const int nslots = (n + TP_THREADS - 1) / TP_THREADS;
#pragma unroll 1
  for (int step = 0; step < 16; ++step) {
    const int bit = 30 - 2 * step;
    const unsigned himask = bit == 30 ? 0u : (0xFFFFFFFFu << (bit + 2));
    int c3 = 0, c2 = 0, c1 = 0, c0 = 0;
#pragma unroll
    for (int s = 0; s < TP_SLOTS; ++s) {
      if (s < nslots) {                                    // uniform
        const bool match = (key[s] & himask) == prefix && key[s] != 0u;
        const unsigned d = (key[s] >> bit) & 3u;
        c3 += __popcll(__ballot(match && d == 3u));
        c2 += __popcll(__ballot(match && d >= 2u));
        c1 += __popcll(__ballot(match && d >= 1u));
        c0 += __popcll(__ballot(match));
      }
    }
    if (lane == 0) {
      if (c3) atomicAdd(&digit_cnt[step][0], c3);
      if (c2) atomicAdd(&digit_cnt[step][1], c2);
      if (c1) atomicAdd(&digit_cnt[step][2], c1);
      if (c0) atomicAdd(&digit_cnt[step][3], c0);
    }
    __syncthreads();
    const int t3 = digit_cnt[step][0], t2 = digit_cnt[step][1], t1 = digit_cnt[step][2], t0 = digit_cnt[step][3];
    unsigned digit;
    int left;                                              // keys that still share the prefix after this digit
    if (t3 >= need) { digit = 3u; left = t3; }
    else if (t2 >= need) { digit = 2u; need -= t3; left = t2 - t3; }
    else if (t1 >= need) { digit = 1u; need -= t2; left = t1 - t2; }
    else { digit = 0u; need -= t1; left = t0 - t1; }
    prefix |= digit << bit;
    if (left == need) {                                    // all of them are wanted: the lower bits do not matter (on
      take_all = true;                                     // random scores this ends the search after 8 - 10 steps)
      break;
    }
  }
  // full search: prefix = key of the P-th best entry; `need` entries equal to it are taken, everything above it is
  if (!take_all) {
    int ceq = 0;
#pragma unroll
    for (int s = 0; s < TP_SLOTS; ++s)
      if (s < nslots) ceq += __popcll(__ballot(key[s] == prefix && prefix != 0u));
    if (lane == 0 && ceq) atomicAdd(&s_eq, ceq);
  }
  __syncthreads();
  const bool all_equal_taken = take_all || s_eq == need;   // the usual case (no ties at the threshold): order does not matter
#pragma unroll
  for (int s = 0; s < TP_SLOTS; ++s) {
    if (s < nslots && (key[s] > prefix || (all_equal_taken && key[s] >= prefix && key[s] != 0u))) {
      const int pos = atomicAdd(&s_ngt, 1);
      cand[pos] = ((unsigned long long)key[s] << 32) | (unsigned)(~(s * TP_THREADS + tid));
    }
  }
  __syncthreads();
  if (!all_equal_taken) {                                  // ties at the threshold: the lowest flat indices win
    const int ngt = s_ngt;                                 // == P - need
    int taken = 0;
#pragma unroll
    for (int s = 0; s < TP_SLOTS; ++s) {
      if (taken < need && s < nslots) {                    // uniform
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
        if (flag && pos < need) cand[ngt + pos] = ((unsigned long long)prefix << 32) | (unsigned)(~(s * TP_THREADS + tid));
        taken += tot;
        __syncthreads();
      }
    }
    __syncthreads();
  }
  // rank sort of the P candidates: candidate c = tid % Ppad is compared against a 1 / G share of the list by each of G threads
  int ppad = 1;
  while (ppad < P) ppad <<= 1;
  const int G = TP_THREADS / ppad, c = tid & (ppad - 1), part = tid / ppad;
  if (c < P) {
    const unsigned long long mine = cand[c];
    int rank = 0;
    const int per = (P + G - 1) / G, j0 = part * per, j1 = min(P, j0 + per);
#pragma unroll 4
    for (int jj = j0; jj < j1; ++jj) rank += cand[jj] > mine ? 1 : 0;
    if (rank) atomicAdd(&rank_of[c], rank);
  }
  __syncthreads();
  if (tid < P) {
    const int ix = (int)(~(unsigned)cand[tid]);
    const int rank = rank_of[tid];
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
  hipLaunchKernelGGL(pair_token_proj_kernel, dim3(HDN / 256, (N + OB - 1) / OB, 2), dim3(1024), 0, stream,
                     sub_feats, obj_feats, W1T, b1, U, VT, tokens_out, N, T);
  PVSG_LAUNCH_CHECK("pair_score_forward(proj)");
  hipLaunchKernelGGL(pair_score_kernel, dim3(N, (N + 63) / 64), dim3(256), 0, stream, U, VT, w2, b2,
                     pair_matrix, N);
  PVSG_LAUNCH_CHECK("pair_score_forward(score)");
  return PVSG_OK;
}
