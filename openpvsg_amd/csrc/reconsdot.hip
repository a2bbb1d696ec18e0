// IPS tube association: the reconstruction distance between the tracks and the observations of a frame.
//
// Replaces the tail of reconsdot_distance (models/unitrack/core/association/matching.py:194-225).  With the track cells
// (t, p) as rows and the observation cells (d, s) as columns of the affinity A = F_trk F_det^T (both sides zero-padded to the
// longest object, as get_track_feat :174-191 pads them; the padded cells take part in the soft-max with logit 0, as there):
//     P  = softmax over columns of (tmp A)          Pc = softmax over rows of (tmp A)
//     recons_trk[t, d] = P[(t,.), (d,.)] F_det[d]    recons_det[d, t] = Pc[(t,.), (d,.)]^T F_trk[t]
//     cost[t, d] = 1 - ( <normalize(recons_trk[t,d]), F_trk[t]> / ||F_trk[t]|| + <normalize(recons_det[d,t]), F_det[d]> / ||F_det[d]|| ) / 2
// Neither the soft-max matrices nor the (cells x objects x channels) reconstructions are formed in memory:
//     <recons_trk[t,d], F_trk[t]> = sum over the (t,d) block of P o A
//     ||recons_trk[t,d]||^2       = sum_p  P[p,:] G_d P[p,:]^T ,   G_d = F_det[d] F_det[d]^T   (Gram matrix of the object's cells)
// and symmetrically with Pc and G_t (y = tmp a is rounded to f32 before the maximum is subtracted, as torch's softmax(tmp * aff)
// sees it).  Passes over A (272 MB for 28 x 27 objects of 300 cells): row statistics, column statistics,
// then one pass per direction in which a workgroup takes a 32-row (32-column) strip of one block, builds its soft-max values in
// LDS, multiplies them with the Gram matrix on the f32 matrix cores (v_mfma_f32_32x32x2_f32: the arithmetic stays that of the
// f32 reference; 16-bit limbs would need a scale per block, the soft-max mass of a non-matching block being ~1e-20 of a row)
// and reduces  sum (X G) o X  and  sum X o A  to one pair of numbers.  Every reduction has a fixed order: results are
// reproducible bit for bit.
#include "common.h"

namespace pvsg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int RD_STRIP = 32;       // rows (columns) of a block one workgroup takes
constexpr int RD_COL_CHUNK = 64;   // rows per partial of the column statistics
constexpr int RD_LDS_PAD = 4;

__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}

// one workgroup per track cell (t, p < Pt): max and sum of exp over the observation cells (d, s < Pd)
__global__ __launch_bounds__(256) void reconsdot_row_stats(const float* __restrict__ A, float* __restrict__ rmax, float* __restrict__ rsum,
                                                            int Pt, int Ptp, int Nd, int Pd, int Pdp, float tmp) {
  __shared__ float sh[4];
  const int t = blockIdx.x / Pt, p = blockIdx.x - t * Pt;
  const long long row = (long long)t * Ptp + p;
  const int Np = Nd * Pdp;
  const float* a = A + row * Np;
  float m = -INFINITY;
  for (int c = threadIdx.x * 4; c < Np; c += 1024) {
    const int s = c % Pdp;
    const float4 v = ld4(a + c);
    if (s < Pd) m = fmaxf(m, __fmul_rn(v.x, tmp));
    if (s + 1 < Pd) m = fmaxf(m, __fmul_rn(v.y, tmp));
    if (s + 2 < Pd) m = fmaxf(m, __fmul_rn(v.z, tmp));
    if (s + 3 < Pd) m = fmaxf(m, __fmul_rn(v.w, tmp));
  }
  m = block_reduce_max(m, sh);
  float sum = 0.f;
  for (int c = threadIdx.x * 4; c < Np; c += 1024) {
    const int s = c % Pdp;
    const float4 v = ld4(a + c);
    if (s < Pd) sum += expf(__fmul_rn(v.x, tmp) - m);
    if (s + 1 < Pd) sum += expf(__fmul_rn(v.y, tmp) - m);
    if (s + 2 < Pd) sum += expf(__fmul_rn(v.z, tmp) - m);
    if (s + 3 < Pd) sum += expf(__fmul_rn(v.w, tmp) - m);
  }
  sum = block_reduce_sum(sum, sh);
  if (threadIdx.x == 0) {
    rmax[row] = m;
    rsum[row] = sum;
  }
}

// column statistics, first half: thread = observation cell (column), blockIdx.y = chunk of RD_COL_CHUNK rows of A
__global__ __launch_bounds__(256) void reconsdot_col_partial(const float* __restrict__ A, float* __restrict__ pm, float* __restrict__ ps,
                                                              int Mp, int Pt, int Ptp, int Np, int Pd, int Pdp, float tmp) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= Np) return;
  const int r0 = blockIdx.y * RD_COL_CHUNK, r1 = min(r0 + RD_COL_CHUNK, Mp);
  float m = -INFINITY, sum = 0.f;
  if ((c % Pdp) < Pd) {
    for (int r = r0; r < r1; ++r)
      if ((r % Ptp) < Pt) m = fmaxf(m, __fmul_rn(A[(long long)r * Np + c], tmp));
    if (m > -INFINITY)
      for (int r = r0; r < r1; ++r)
        if ((r % Ptp) < Pt) sum += expf(__fmul_rn(A[(long long)r * Np + c], tmp) - m);
  }
  pm[(long long)blockIdx.y * Np + c] = m;
  ps[(long long)blockIdx.y * Np + c] = sum;
}

// second half: 64 columns per workgroup, the chunks dealt to its four waves, merged through LDS in a fixed order
__global__ __launch_bounds__(256) void reconsdot_col_finish(const float* __restrict__ pm, const float* __restrict__ ps, float* __restrict__ cmax,
                                                             float* __restrict__ csum, int Np, int nchunk) {
  __shared__ float shm[4][64], shs[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float m = -INFINITY, sum = 0.f;
  if (c < Np) {
    for (int k = w; k < nchunk; k += 4) m = fmaxf(m, pm[(long long)k * Np + c]);
    if (m > -INFINITY)
      for (int k = w; k < nchunk; k += 4) {
        const float q = pm[(long long)k * Np + c];
        if (q > -INFINITY) sum += ps[(long long)k * Np + c] * expf(q - m);
      }
  }
  shm[w][lane] = m;
  shs[w][lane] = sum;
  __syncthreads();
  if (w == 0 && c < Np) {
    float mm = fmaxf(fmaxf(shm[0][lane], shm[1][lane]), fmaxf(shm[2][lane], shm[3][lane]));
    float tot = 0.f;
    if (mm > -INFINITY)
      for (int i = 0; i < 4; ++i)
        if (shm[i][lane] > -INFINITY) tot += shs[i][lane] * expf(shm[i][lane] - mm);
    cmax[c] = mm;
    csum[c] = tot;
  }
}

// One strip of one (track, observation) block.  COLS = false: 32 track cells x all observation cells of d, X = P (row soft-max),
// G = G_d;  COLS = true: 32 observation cells x all track cells of t, X = Pc^T (column soft-max), G = G_t.
// part[((t Nd + d) nstrip + strip) 2 + {0,1}] = { sum X o A , sum (X G) o X }.
template <bool COLS>
__global__ __launch_bounds__(256) void reconsdot_strip(const float* __restrict__ A, const float* __restrict__ G, const float* __restrict__ smax,
                                                        const float* __restrict__ ssum, float* __restrict__ part, int Pt, int Ptp, int Nd, int Pd,
                                                        int Pdp, float tmp, const unsigned char* __restrict__ needed) {
  extern __shared__ float X[];                       // [32][Kp + pad]
  __shared__ float sh[4];
  const int strip = blockIdx.x, d = blockIdx.y, t = blockIdx.z;
  if (needed && !needed[t * Nd + d]) return;        // a pair the caller gates away (another class): its cost is written as +inf
  const int Kp = COLS ? Ptp : Pdp, Kv = COLS ? Pt : Pd;          // length of a strip line, its valid part
  const int Sv = COLS ? Pd : Pt;                                   // valid strip lines of the block
  const int ld = Kp + RD_LDS_PAD;
  const int Np = Nd * Pdp;
  const int l0 = strip * RD_STRIP;
  const float* Ab = A + (long long)t * Ptp * Np + (long long)d * Pdp;
  float num = 0.f;
  if (!COLS) {
    for (int e = threadIdx.x; e < RD_STRIP * Kp; e += 256) {
      const int r = e / Kp, k = e - r * Kp;
      float x = 0.f;
      if (l0 + r < Sv && k < Kv) {
        const long long row = (long long)t * Ptp + l0 + r;
        const float a = Ab[(long long)(l0 + r) * Np + k];
        x = expf(__fmul_rn(a, tmp) - smax[row]) / ssum[row];
        num += x * a;
      }
      X[r * ld + k] = x;
    }
  } else {
    for (int e = threadIdx.x; e < RD_STRIP * Kp; e += 256) {
      const int k = e >> 5, r = e & 31;                            // 32 consecutive columns of one row of A: one 128-byte line
      float x = 0.f;
      if (l0 + r < Sv && k < Kv) {
        const int col = d * Pdp + l0 + r;
        const float a = Ab[(long long)k * Np + l0 + r];
        x = expf(__fmul_rn(a, tmp) - smax[col]) / ssum[col];
        num += x * a;
      }
      X[r * ld + k] = x;
    }
  }
  __syncthreads();
  const float* Gb = G + (long long)(COLS ? t : d) * Kp * Kp;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, h = lane >> 5;
  float q = 0.f;
  for (int j = w; j < Kp / 32; j += 4) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* xr = X + li * ld + 4 * h;                         // X[line li][k0 + 4 h ...]
    const float* gr = Gb + (long long)(j * 32 + li) * Kp + 4 * h;  // G symmetric: G[k][col] read as G[col][k], 16 contiguous bytes
    float4 gn = ld4(gr);
    for (int k0 = 0; k0 < Kp; k0 += 8) {
      const float4 g = gn;
      if (k0 + 8 < Kp) gn = ld4(gr + k0 + 8);
      const float4 x = ld4(xr + k0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, g.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, g.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, g.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, g.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i >> 2) * 8 + h * 4 + (i & 3);
      q += acc[i] * X[row * ld + j * 32 + li];
    }
  }
  num = block_reduce_sum(num, sh);
  q = block_reduce_sum(q, sh);
  if (threadIdx.x == 0) {
    const long long o = (((long long)t * Nd + d) * gridDim.x + strip) * 2;
    part[o] = num;
    part[o + 1] = q;
  }
}

// one wave per (track, observation) pair: strip partials and Gram diagonals summed in a fixed lane order
__global__ __launch_bounds__(256) void reconsdot_finish(const float* __restrict__ part_td, const float* __restrict__ part_dt, const float* __restrict__ Gt,
                                                         const float* __restrict__ Gd, float* __restrict__ cost, int Nt, int Pt, int Ptp, int Nd,
                                                         int Pd, int Pdp, int nst, int nsd, const unsigned char* __restrict__ needed) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Nt * Nd) return;
  const int lane = threadIdx.x & 63;
  if (needed && !needed[i]) {
    if (lane == 0) cost[i] = INFINITY;
    return;
  }
  const int t = i / Nd, d = i - t * Nd;
  const float eps = 1e-12f;
  float nt = 0.f, nd = 0.f, num_td = 0.f, q_td = 0.f, num_dt = 0.f, q_dt = 0.f;
  for (int p = lane; p < Pt; p += 64) nt += Gt[((long long)t * Ptp + p) * Ptp + p];
  for (int s = lane; s < Pd; s += 64) nd += Gd[((long long)d * Pdp + s) * Pdp + s];
  for (int k = lane; k < nst; k += 64) {
    num_td += part_td[((long long)i * nst + k) * 2];
    q_td += part_td[((long long)i * nst + k) * 2 + 1];
  }
  for (int k = lane; k < nsd; k += 64) {
    num_dt += part_dt[((long long)i * nsd + k) * 2];
    q_dt += part_dt[((long long)i * nsd + k) * 2 + 1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    nt += __shfl_xor(nt, off);
    nd += __shfl_xor(nd, off);
    num_td += __shfl_xor(num_td, off);
    q_td += __shfl_xor(q_td, off);
    num_dt += __shfl_xor(num_dt, off);
    q_dt += __shfl_xor(q_dt, off);
  }
  if (lane == 0) {
    nt = fmaxf(sqrtf(fmaxf(nt, 0.f)), eps);
    nd = fmaxf(sqrtf(fmaxf(nd, 0.f)), eps);
    const float dot_td = num_td / (fmaxf(sqrtf(fmaxf(q_td, 0.f)), eps) * nt);
    const float dot_dt = num_dt / (fmaxf(sqrtf(fmaxf(q_dt, 0.f)), eps) * nd);
    cost[i] = 1.f - 0.5f * (dot_td + dot_dt);
  }
}

struct RdLayout {
  long long rmax, rsum, cmax, csum, pm, ps, ptd, pdt, total;
  int nchunk, nst, nsd;
};

RdLayout rd_layout(int Nt, int Pt, int Ptp, int Nd, int Pd, int Pdp) {
  RdLayout L;
  const long long Mp = (long long)Nt * Ptp, Np = (long long)Nd * Pdp;
  L.nchunk = (int)((Mp + RD_COL_CHUNK - 1) / RD_COL_CHUNK);
  L.nst = (Pt + RD_STRIP - 1) / RD_STRIP;
  L.nsd = (Pd + RD_STRIP - 1) / RD_STRIP;
  long long o = 0;
  L.rmax = o; o += Mp;
  L.rsum = o; o += Mp;
  L.cmax = o; o += Np;
  L.csum = o; o += Np;
  L.pm = o; o += (long long)L.nchunk * Np;
  L.ps = o; o += (long long)L.nchunk * Np;
  L.ptd = o; o += (long long)Nt * Nd * L.nst * 2;
  L.pdt = o; o += (long long)Nt * Nd * L.nsd * 2;
  L.total = o;
  return L;
}

std::atomic<unsigned long long> g_lds_rows{0}, g_lds_cols{0};

}  // namespace
}  // namespace pvsg

extern "C" long long pvsg_reconsdot_workspace_bytes(int Nt, int Pt, int Nd, int Pd) {
  if (Nt <= 0 || Nd <= 0 || Pt <= 0 || Pd <= 0) return 0;
  const int Ptp = (Pt + 31) / 32 * 32, Pdp = (Pd + 31) / 32 * 32;
  return pvsg::rd_layout(Nt, Pt, Ptp, Nd, Pd, Pdp).total * 4;
}

extern "C" int pvsg_reconsdot_cost(const float* A, const float* Gt, const float* Gd, int Nt, int Pt, int Nd, int Pd, float tmp,
                                   const unsigned char* needed, float* workspace, float* cost, void* stream) {
  using namespace pvsg;
  PVSG_REQUIRE(A && Gt && Gd && workspace && cost, "pvsg_reconsdot_cost: null pointer");
  PVSG_REQUIRE(Nt > 0 && Nd > 0 && Pt > 0 && Pd > 0, "pvsg_reconsdot_cost: sizes must be positive (Nt %d, Pt %d, Nd %d, Pd %d)", Nt, Pt, Nd, Pd);
  const int Ptp = (Pt + 31) / 32 * 32, Pdp = (Pd + 31) / 32 * 32;
  PVSG_REQUIRE(Ptp <= 1024 && Pdp <= 1024, "pvsg_reconsdot_cost: at most 1024 cells per object (Pt %d, Pd %d)", Pt, Pd);
  PVSG_REQUIRE((long long)Nt * Ptp < (1ll << 24) && (long long)Nd * Pdp < (1ll << 24) && Nt < 65536 && Nd < 65536,
               "pvsg_reconsdot_cost: too many cells (Nt %d x %d, Nd %d x %d)", Nt, Ptp, Nd, Pdp);
  hipStream_t st = (hipStream_t)stream;
  const RdLayout L = rd_layout(Nt, Pt, Ptp, Nd, Pd, Pdp);
  const int Mp = Nt * Ptp, Np = Nd * Pdp;
  float* ws = workspace;
  reconsdot_row_stats<<<Nt * Pt, 256, 0, st>>>(A, ws + L.rmax, ws + L.rsum, Pt, Ptp, Nd, Pd, Pdp, tmp);
  PVSG_LAUNCH_CHECK("reconsdot_row_stats");
  reconsdot_col_partial<<<dim3((Np + 255) / 256, L.nchunk), 256, 0, st>>>(A, ws + L.pm, ws + L.ps, Mp, Pt, Ptp, Np, Pd, Pdp, tmp);
  PVSG_LAUNCH_CHECK("reconsdot_col_partial");
  reconsdot_col_finish<<<(Np + 63) / 64, 256, 0, st>>>(ws + L.pm, ws + L.ps, ws + L.cmax, ws + L.csum, Np, L.nchunk);
  PVSG_LAUNCH_CHECK("reconsdot_col_finish");
  const int lds_rows = RD_STRIP * (Pdp + RD_LDS_PAD) * 4, lds_cols = RD_STRIP * (Ptp + RD_LDS_PAD) * 4;
  if (lds_rows > 64 * 1024) {
    const hipError_t e = ensure_dynamic_lds((const void*)reconsdot_strip<false>, lds_rows, g_lds_rows);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "pvsg_reconsdot_cost: %d bytes of LDS: %s", lds_rows, hipGetErrorString(e));
  }
  if (lds_cols > 64 * 1024) {
    const hipError_t e = ensure_dynamic_lds((const void*)reconsdot_strip<true>, lds_cols, g_lds_cols);
    if (e != hipSuccess) return set_err(PVSG_ERR_HIP, "pvsg_reconsdot_cost: %d bytes of LDS: %s", lds_cols, hipGetErrorString(e));
  }
  reconsdot_strip<false><<<dim3(L.nst, Nd, Nt), 256, lds_rows, st>>>(A, Gd, ws + L.rmax, ws + L.rsum, ws + L.ptd, Pt, Ptp, Nd, Pd, Pdp, tmp, needed);
  PVSG_LAUNCH_CHECK("reconsdot_strip<rows>");
  reconsdot_strip<true><<<dim3(L.nsd, Nd, Nt), 256, lds_cols, st>>>(A, Gt, ws + L.cmax, ws + L.csum, ws + L.pdt, Pt, Ptp, Nd, Pd, Pdp, tmp, needed);
  PVSG_LAUNCH_CHECK("reconsdot_strip<cols>");
  reconsdot_finish<<<(Nt * Nd + 3) / 4, 256, 0, st>>>(ws + L.ptd, ws + L.pdt, Gt, Gd, cost, Nt, Pt, Ptp, Nd, Pd, Pdp, L.nst, L.nsd, needed);
  PVSG_LAUNCH_CHECK("reconsdot_finish");
  return PVSG_OK;
}
