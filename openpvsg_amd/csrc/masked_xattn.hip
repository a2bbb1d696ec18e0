// Masked cross-attention of the Mask2Former transformer decoder, streaming (flash-style), f32-exact on the gfx950 matrix
// cores.  Two kernels behind one entry point: `xattn_partial_bf16x3_kernel` (default since round 3: every f32 operand as
// three bf16 limbs on the bf16 matrix pipe, 0.516 ms = 93 TF/s of model arithmetic at 471 040 keys) and
// `xattn_partial_lds_kernel` (f32 MFMA, 0.705 ms = 68 TF/s; PVSG_XATTN=f32).  Both are described at their definitions;
// the common design is below.
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
// workgroup consumes whole 1 KiB key/value rows: they are staged in LDS by LDS-DMA, 32 keys per tile,
// double buffered (details at the kernel).  Each wave keeps its head's Q (100x32, padded to 7 tiles of 16
// rows) in registers for the whole range and walks the tile 16 keys at a time:
//   S^T = K_tile (16x32) . Q^T          56 x v_mfma_f32_16x16x4_f32   (A = K fragment from LDS, B = Q regs)
//   mask bits, running max / sum        per query column; the 16 key rows live in 4 lane groups, reduced
//                                       with v_permlane16_swap / v_permlane32_swap (VALU, no LDS crossbar)
//   O^T += V_tile^T (32x16) . P^T       56 x v_mfma; the S^T accumulator registers ARE the B operand
//                                       (key index of k-step r = 4*(lane>>4)+r on both sides), so P
//                                       never moves between lanes or through LDS
// O^T keeps each lane's values in ONE query column, so the online-softmax rescale is lane-local.
// Every HBM byte of K/V is read once, in full 1 KiB rows.  Measured (round 1, 471 040 keys): 0.75 ms =
// 64 TF (41 % of the f32 matrix peak); a variant loading fragments straight from HBM into registers
// measured 0.72 ms -- both are limited by the QK -> softmax -> PV dependency inside a wave at 2 waves
// per SIMD, not by the loads.
// Tried and measured slower (round 2, 471 040 keys, 0.755 ms baseline): (a) 16 waves per workgroup, wave = (head,
// half of the query tiles), 128 VGPRs -> 4 waves/SIMD: 1.006 ms (31 spilled VGPRs, K/V fragments read twice, an
// eighth padding tile); (b) rescaling O only when a wave vote says the running maximum moved + b128 mask reads:
// 1.03 ms; (c) one wave per SIMD (4-wave workgroups, 256 VGPRs + 170 AccVGPRs), S^T double / triple buffered so that
// P.V(i-1) and S^T(i+1) are issued between the soft-max instructions of sub-tile i (sched_group_barrier pipeline,
// verified in the ISA): 0.97-1.10 ms.  Reading of (a)-(c): the f32-input MFMA runs at the f32 VECTOR rate on the same
// SIMD units (cdna_hip_programming.md, "FP32-input MFMA"), so soft-max VALU work does not hide under it whatever the
// interleave or the wave count -- only fewer non-MFMA instructions per key would help (457 VALU + 35 exp + 183 SALU
// per 112 MFMA today).
// (d, adopted = LEAN) stale reference maximum + per-lane sums + log2-domain logits: half the VALU instructions, 0.733 ms;
// (e) with the soft-max removed the kernel takes 0.46 ms: the rest is the latency of the QK -> soft-max -> PV chain.
// Ranges are combined by `xattn_combine_kernel` (log-sum-exp merge); the same partial format is
// what ranks exchange when a clip's frames are sharded over GPUs (openpvsg_amd/parallel.py).
#include "common.h"
#include <stdlib.h>

namespace pvsg {

constexpr int XQT = 7;  // 7 x 16 = 112 query rows

// Reductions over the 4 lane groups (lane>>4) that hold the 16 key rows of one S^T tile.  gfx950's
// v_permlane16_swap / v_permlane32_swap exchange 16- / 32-lane rows between two registers in the VALU
// (no LDS crossbar trip like ds_bpermute): with both operands = v the results are (r0,r0,r2,r2) and
// (r1,r1,r3,r3), so one max/add gives the xor-16 butterfly step; the 32-lane swap gives the xor-32 step.
__device__ __forceinline__ float group_max4(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum4(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ------------------------------------------------------------------------------------------------
// K/V staging: the workgroup's 8 waves (= heads) share the key/value ROWS (1 KiB = all heads),
// so rows are brought in whole by LDS-DMA (global_load_lds_dwordx4: one row per wave instruction, fully
// coalesced, no VGPR staging), 32 keys per tile, double buffered: the DMA of tile t+1 flies while tile t
// is consumed.  A row r is stored with its 16-byte chunks XOR-swizzled by (r & 15) -- applied on the
// SOURCE address, the LDS side of the DMA is lane-linear -- which makes the per-head fragment reads
// (16 keys x 32 B for K, 4 keys x 128 B for V) bank-conflict free.
// ------------------------------------------------------------------------------------------------
constexpr int TK = 32;                                   // keys per LDS tile
constexpr int XLDS_TILE_FLOATS = TK * 256 * 2 + TK * 4;  // K rows + V rows + mask words (as floats)

template <bool LEAN>
__global__ __launch_bounds__(512) void xattn_partial_lds_kernel(
    const float* __restrict__ qp, const float* __restrict__ kp, const float* __restrict__ vp,
    const uint32_t* __restrict__ bits, const uint32_t* __restrict__ flags, float* __restrict__ part_o,
    float* __restrict__ part_ml, int Q, long long K, int NS, long long chunk, long long kvs) {
  constexpr int HD = 256, D = 32, M = 8;          // kvs: floats between consecutive key / value rows (>= HD)
  extern __shared__ __attribute__((aligned(16))) float xl[];   // [2][XLDS_TILE_FLOATS]
  const int b = blockIdx.x / NS, s = blockIdx.x - b * NS;
  const int h = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, g = lane >> 4;
  const long long k0 = (long long)s * chunk;
  const long long k1 = (k0 + chunk < K) ? k0 + chunk : K;
  const bool use_mask = bits != nullptr;

  float qf[XQT][8];
  uint32_t honor = 0u;
  {
    uint32_t fw[4] = {0u, 0u, 0u, 0u};
    if (use_mask) {
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
        if (LEAN) {        // logits in the log2 domain: exp2 without the per-element multiply
#pragma unroll
          for (int i = 0; i < 8; ++i) qf[qt][i] *= 1.4426950408889634f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[qt][i] = 0.f;
      }
      if (use_mask && ((fw[qt >> 1] >> ((qt & 1) * 16 + j)) & 1u)) honor |= 1u << qt;
    }
  }
  float mrun[XQT], lrun[XQT];
  f32x4 o[XQT][2];
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    mrun[qt] = -INFINITY; lrun[qt] = 0.f;
    o[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* kb = kp + (long long)b * K * kvs;
  const float* vb = vp + (long long)b * K * kvs;
  const uint32_t* mb = use_mask ? bits + (long long)b * K * 4 : nullptr;
  const long long klast = k1 - 1;

  // wave w brings rows 4w..4w+3 of K and of V (+ 16 mask dwords) of a tile: 9 DMA instructions per wave
  auto issue = [&](long long kt, int buf) {
    float* base = xl + buf * XLDS_TILE_FLOATS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = h * 4 + i;
      long long key = kt + r;
      key = key < klast ? key : klast;
      const int src_chunk = lane ^ (r & 15);
      __builtin_amdgcn_global_load_lds(kb + key * kvs + src_chunk * 4,
                                       (__attribute__((address_space(3))) void*)(base + r * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(vb + key * kvs + src_chunk * 4,
                                       (__attribute__((address_space(3))) void*)(base + TK * 256 + r * 256), 16, 0, 0);
    }
    if (use_mask && lane < 16) {
      long long word = (kt + (h * 16 + lane) / 4) ;
      word = word < klast ? word : klast;
      __builtin_amdgcn_global_load_lds(mb + word * 4 + (lane & 3),
                                       (__attribute__((address_space(3))) void*)(base + TK * 512 + h * 16), 4, 0, 0);
    }
  };

  const int ntile = (int)((k1 - k0 + TK - 1) / TK);
  if (ntile > 0) issue(k0, 0);
  for (int t = 0; t < ntile; ++t) {
    const long long kt = k0 + (long long)t * TK;
    if (t + 1 < ntile) {
      issue(kt + TK, (t + 1) & 1);
      if (use_mask) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // tile t landed, tile t+1 in flight
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                      // every wave's rows of tile t are in LDS
    const float* tb = xl + (t & 1) * XLDS_TILE_FLOATS;
    const uint32_t* tm = reinterpret_cast<const uint32_t*>(tb + TK * 512);

#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      // ---- S^T = K . Q^T for 16 keys: K fragment = two swizzled 16-byte chunks of row sub*16 + j ----------
      const int rk = sub * 16 + j;
      const float4 a = *reinterpret_cast<const float4*>(tb + rk * 256 + (((h * 8 + g * 2) ^ (rk & 15)) << 2));
      const float4 c = *reinterpret_cast<const float4*>(tb + rk * 256 + (((h * 8 + g * 2 + 1) ^ (rk & 15)) << 2));
      const float kf[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      f32x4 st[XQT];
#pragma unroll
      for (int qt = 0; qt < XQT; ++qt) st[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int qt = 0; qt < XQT; ++qt)
          st[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[i], qf[qt][i], st[qt], 0, 0, 0);
      bool kvalid[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) kvalid[r] = kt + sub * 16 + g * 4 + r < k1;
      if constexpr (LEAN) {
        // ---- lean online soft-max.  The f32 MFMA runs at the vector rate on the same SIMD, so every VALU instruction
        // here adds to the kernel's time.  The reference maximum `mrun` is allowed to go STALE: p = 2^(s - mrun) with
        // s - mrun <= 10 cannot overflow, so as long as no lane sees a logit more than 10 above its reference the
        // sub-tile needs no cross-group max, no exp for alpha and no rescale of O; the partial sums stay per lane and
        // are reduced over the four lane groups once, at the end.  One wave vote per 16 keys picks the path.
        float tl[XQT];
        bool need = false;
#pragma unroll
        for (int qt = 0; qt < XQT; ++qt) {
          const bool hq = (honor >> qt) & 1u;
          const int sh = (qt & 1) * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t w = hq ? tm[(sub * 16 + g * 4 + r) * 4 + (qt >> 1)] : 0u;
            const bool masked = !kvalid[r] || ((w >> sh) & 1u);
            st[qt][r] = masked ? -INFINITY : st[qt][r];
          }
          tl[qt] = fmaxf(fmaxf(st[qt][0], st[qt][1]), fmaxf(st[qt][2], st[qt][3]));
          need = need || (tl[qt] > mrun[qt] + 10.f);
        }
        if (__ballot(need) != 0ull) {
          // rare path: new common reference per query column (the same in all four lane groups), rescale O and l
#pragma unroll
          for (int qt = 0; qt < XQT; ++qt) {
            const float mnew = fmaxf(mrun[qt], group_max4(tl[qt]));
            const float alpha = (mnew == -INFINITY) ? 1.f : exp2f(mrun[qt] - mnew);
            lrun[qt] *= alpha;
            o[qt][0] *= alpha;
            o[qt][1] *= alpha;
            mrun[qt] = mnew;
          }
        }
#pragma unroll
        for (int qt = 0; qt < XQT; ++qt) {
          const float mref = (mrun[qt] == -INFINITY) ? 0.f : mrun[qt];      // nothing unblocked yet: every p = 2^-inf = 0
          float psum = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = exp2f(st[qt][r] - mref);
            st[qt][r] = p;
            psum += p;
          }
          lrun[qt] += psum;                                                  // per-lane partial: groups merged at the end
        }
      } else {
        // ---- mask + online softmax (per query column) ---------------------------------------------------------
  #pragma unroll
        for (int qt = 0; qt < XQT; ++qt) {
          const bool hq = (honor >> qt) & 1u;
          const int sh = (qt & 1) * 16 + j;
          float sv[4];
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            // mask word of this key for query tile qt, read on demand from the LDS copy
            const uint32_t w = hq ? tm[(sub * 16 + g * 4 + r) * 4 + (qt >> 1)] : 0u;
            const bool masked = !kvalid[r] || ((w >> sh) & 1u);
            sv[r] = masked ? -INFINITY : st[qt][r];
          }
          const float tmax = group_max4(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
          const float mnew = fmaxf(mrun[qt], tmax);
          float alpha = 1.f, psum = 0.f;
          if (mnew == -INFINITY) {
  #pragma unroll
            for (int r = 0; r < 4; ++r) st[qt][r] = 0.f;
          } else {
            alpha = __expf(mrun[qt] - mnew);
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
      }
      // ---- O^T += V^T . P^T (V fragments from LDS right before use) ---------------------------------------------
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = sub * 16 + g * 4 + r;
        const int ch = (h * 8 + (j >> 1)) ^ (row & 15);
        const float2 vf = *reinterpret_cast<const float2*>(tb + TK * 256 + row * 256 + ch * 4 + (j & 1) * 2);
#pragma unroll
        for (int qt = 0; qt < XQT; ++qt) {
          o[qt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, st[qt][r], o[qt][0], 0, 0, 0);
          o[qt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, st[qt][r], o[qt][1], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everyone is done with this buffer before tile t+2 overwrites it
  }

  const long long slot = ((long long)b * NS + s) * M + h;
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    const int q = qt * 16 + j;
    if (q < Q) {
      float* op = part_o + (slot * Q + q) * D + g * 8;
      st4(op, make_float4(o[qt][0][0], o[qt][1][0], o[qt][0][1], o[qt][1][1]));
      st4(op + 4, make_float4(o[qt][0][2], o[qt][1][2], o[qt][0][3], o[qt][1][3]));
      float mo = mrun[qt], lo = lrun[qt];
      if (LEAN) { lo = group_sum4(lo); mo *= 0.6931471805599453f; }        // log2 domain -> natural, -inf stays -inf
      if (g == 0) *reinterpret_cast<float2*>(part_ml + (slot * Q + q) * 2) = make_float2(mo, lo);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix pipe, exact: every f32 operand is split into three bf16 limbs (hi + mid + lo == x,
// the scheme of split_common.h) and a product is the six limb products of weight >= 2^-16 accumulated in f32 --
// f32-class error (tests/test_xattn.py runs both kernels against the same f64 statement at the same tolerance).
// ------------------------------------------------------------------------------------------------
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 xbf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned xu32x2 __attribute__((ext_vector_type(2)));
typedef short xs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void xsplit2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(xf32x2{a0, a1}, xbf16x2));
  const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(xf32x2{r0, r1}, xbf16x2));
  const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(xf32x2{s0, s1}, xbf16x2));
}
__device__ __forceinline__ void xsplit2v(float a0, float a1, xu32x4& h, xu32x4& m, xu32x4& l, int i) {
  unsigned x, y, z;
  xsplit2(a0, a1, x, y, z);
  h[i] = x; m[i] = y; l[i] = z;
}
__device__ __forceinline__ void xsplit2w(float a0, float a1, xu32x2& h, xu32x2& m, xu32x2& l, int i) {
  unsigned x, y, z;
  xsplit2(a0, a1, x, y, z);
  h[i] = x; m[i] = y; l[i] = z;
}
// 32-deep: lane (row/col j, k-group g) holds k = 8g..8g+7 -- the eight head dims a lane of the f32 kernel keeps
__device__ __forceinline__ f32x4 xmfma_bf16(xu32x4 a, xu32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xbf16x8, a), __builtin_bit_cast(xbf16x8, b), c, 0, 0, 0);
}
// 16-deep (two VGPRs per operand): lane (row/col j, k-group g) holds k = 4g..4g+3 -- the four keys a lane group owns
__device__ __forceinline__ f32x4 xmfma_bf16_k16(xu32x2 a, xu32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(xs16x4, a), __builtin_bit_cast(xs16x4, b), c, 0, 0, 0);
}

constexpr int TKB = 16;                                   // keys per LDS tile = one S^T tile
constexpr int XBX_RING = 3;                               // tiles in the ring: one consumed, two in flight
constexpr int xbx_tile_floats(int hw) { return TKB * hw * 32 * 2 + TKB * 4; }     // K rows + V rows of hw heads + mask words
constexpr int xbx_lds_floats(int hw) { return XBX_RING * xbx_tile_floats(hw) + hw * XQT * 64 * 4; }   // + low limbs of Q

// Workgroup = (batch, key range) [HW = 8 heads, whole 1 KiB rows] or (batch, key range, half of the heads) [HW = 4, 512 B
// runs of every row]; wave = head.  Registers are the scarce resource: per lane, Q as three limbs is 84 VGPRs, O^T 56,
// S^T 28, the running maximum / sum 14 -- so the LOW limb of Q lives in LDS (read a query tile ahead, 4 VGPRs), and the K/V
// tiles are 16 keys in a ring of three to make room for it.  Nothing may spill inside the key loop: a scratch reload waits
// on vmcnt, which also counts the LDS-DMA loads in flight, and the staging pipeline would drain on every tile
// (measured: the 24-spill first version ran at the speed of the f32 kernel).
// The bf16 MFMA has its own pipe (the f32 MFMA of the kernel above runs at the vector rate on the VALU), but a wave issues
// in order: measured, the matrix and vector phases of the straightforward schedule add up.  So the loop is software
// pipelined over the query tiles, each stage pairing matrix instructions of tile s with vector work of tile s-1:
//   stage A(s):  S^T[s] = K . Q^T[s], 6 limb products (one dependent chain)      |  mask + row maximum of tile s-1
//   stage B(s):  P[s] = 2^(S - m), partial sums, 3-limb split                    |  O^T[s-1] += V^T . P^T[s-1] (12 MFMAs)
// sched_barrier fences keep the pairing, and an empty asm pins each vector result where it is computed (plain arithmetic
// is otherwise free to sink past a fence).
// Measured on the way (471 040 keys; scripts/lab/issue_lab.hip, valu_lab.hip, pmc_xattn.sh, profiles/r03_xattn_*):
//   straightforward schedule (all MFMAs of a phase, then its vector work), 24 spilled VGPRs      0.705 ms (= the f32 kernel)
//   + stages paired as above, P.V on the 16-deep MFMA (2 VGPRs per operand)                       0.590 ms
//   + low Q limb in LDS, 16-key ring: no spill in the loop (expected to be the big one: it was not)  0.592 ms
//   + mask select without the scalar detour (v_and_or / v_cmp / v_cndmask), vote flag as integer  0.564 ms
//   + v_max3 by asm (no re-quieting), row-validity words only in the last tile                    0.537 ms
//   + half-head workgroups (HW = 4)                                                               0.516 ms
// One barrier per tile placed between the stages with the LDS reads of the next tile behind it: 0.575 ms (slower: one
// tile in flight instead of two).  Counters at 0.59 ms: matrix pipe 38 % busy, a wave issues 39 % of its cycles
// (VALU 28 %, scalar 9.5 %), waits to issue 21 %, waits on a dependency 40 %; the lab puts an MFMA at ~6.5 issue cycles
// next to 4.2 per VALU instruction and shows a dependent MFMA chain hiding NO vector work of the same wave, two
// independent chains most of it.  What is left is instruction count (about 460 VALU + 126 MFMA per wave and 16 keys,
// 154 of the VALU in the three-limb split of P): every instruction removed showed up in the time.  P.V contracts over the tile's 16 keys with v_mfma_f32_16x16x16_bf16, whose k
// index 4g+i is the key row 4g+i of the S^T accumulator: P never leaves its lane.
template <int HW>
__global__ __launch_bounds__(HW * 64, HW == 8 ? 1 : 2) void xattn_partial_bf16x3_kernel(
    const float* __restrict__ qp, const float* __restrict__ kp, const float* __restrict__ vp,
    const uint32_t* __restrict__ bits, const uint32_t* __restrict__ flags, float* __restrict__ part_o,
    float* __restrict__ part_ml, int Q, long long K, int NS, long long chunk, long long kvs) {
  constexpr int HD = 256, D = 32, M = 8;          // kvs: floats between consecutive key / value rows (>= HD)
  constexpr int TILE = xbx_tile_floats(HW), RS = HW * 32;            // RS: floats per staged row
  extern __shared__ __attribute__((aligned(16))) float xl[];          // [XBX_RING][TILE] then ql[HW][XQT][64] (16 B each)
  const int half = HW == 8 ? 0 : (int)(blockIdx.x & 1u), bs = HW == 8 ? (int)blockIdx.x : (int)(blockIdx.x >> 1);
  const int b = bs / NS, s = bs - b * NS;
  const int hl = threadIdx.x >> 6, lane = threadIdx.x & 63;           // hl: head within the workgroup
  const int h = half * HW + hl;
  const int j = lane & 15, g = lane >> 4;
  const long long k0 = (long long)s * chunk;
  const long long k1 = (k0 + chunk < K) ? k0 + chunk : K;
  const bool use_mask = bits != nullptr;
  xu32x4* qlow = reinterpret_cast<xu32x4*>(xl + XBX_RING * TILE) + hl * XQT * 64 + lane;   // [qt * 64]

  xu32x4 qh[XQT], qm[XQT];
  uint32_t honor = 0u;
  {
    uint32_t fw[4] = {0u, 0u, 0u, 0u};
    if (use_mask) {
#pragma unroll
      for (int k = 0; k < 4; ++k) fw[k] = flags[b * 4 + k];
    }
#pragma unroll
    for (int qt = 0; qt < XQT; ++qt) {
      const int q = qt * 16 + j;
      float qf[8];
      if (q < Q) {
        const float* p = qp + ((long long)b * Q + q) * HD + h * D + g * 8;
        const float4 a = ld4(p), c = ld4(p + 4);
        qf[0] = a.x; qf[1] = a.y; qf[2] = a.z; qf[3] = a.w; qf[4] = c.x; qf[5] = c.y; qf[6] = c.z; qf[7] = c.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] *= 1.4426950408889634f;     // logits in the log2 domain
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) qf[i] = 0.f;
      }
      xu32x4 lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) xsplit2v(qf[2 * i], qf[2 * i + 1], qh[qt], qm[qt], lo, i);
      qlow[qt * 64] = lo;                                              // written and read by this lane only
      if (use_mask && ((fw[qt >> 1] >> ((qt & 1) * 16 + j)) & 1u)) honor |= 1u << qt;
    }
  }
  // running reference maximum (log2 domain); "nothing unblocked yet" is the finite XNONE instead of -inf, so that
  // 2^(s - m) needs no special case: s = -inf gives 0, and a finite s against XNONE trips the renewal vote first
  constexpr float XNONE = -1.0e30f;
  float mrun[XQT], lrun[XQT];
  f32x4 o[XQT][2];
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    mrun[qt] = XNONE; lrun[qt] = 0.f;
    o[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* kb = kp + (long long)b * K * kvs + half * RS;
  const float* vb = vp + (long long)b * K * kvs + half * RS;
  const uint32_t* mb = use_mask ? bits + (long long)b * K * 4 : nullptr;
  const long long klast = k1 - 1;

  // HW = 8: wave w brings rows 2w, 2w+1 of K and of V (+ 8 mask dwords) of a tile; HW = 4: the half rows 4w..4w+3, two per
  // instruction (lanes 0-31 / 32-63), + 16 mask dwords.  Rows are stored with their 16-byte chunks XOR-swizzled by the row
  // index (applied on the source address; the LDS side of the DMA is lane-linear), see the f32 kernel.
  auto issue = [&](long long kt, int slot) {
    float* base = xl + slot * TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = HW == 8 ? hl * 2 + i : 2 * (hl * 2 + i) + (lane >> 5);
      const int ln = HW == 8 ? lane : (lane & 31);
      const int r0 = HW == 8 ? r : 2 * (hl * 2 + i);
      long long key = kt + r;
      key = key < klast ? key : klast;
      const int src_chunk = ln ^ r;
      __builtin_amdgcn_global_load_lds(kb + key * kvs + src_chunk * 4,
                                       (__attribute__((address_space(3))) void*)(base + r0 * RS), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(vb + key * kvs + src_chunk * 4,
                                       (__attribute__((address_space(3))) void*)(base + TKB * RS + r0 * RS), 16, 0, 0);
    }
    constexpr int MW = TKB * 4 / HW;                                   // mask dwords per wave
    if (use_mask && lane < MW) {
      long long word = kt + (hl * MW + lane) / 4;
      word = word < klast ? word : klast;
      __builtin_amdgcn_global_load_lds(mb + word * 4 + (lane & 3),
                                       (__attribute__((address_space(3))) void*)(base + TKB * RS * 2 + hl * MW), 4, 0, 0);
    }
  };

#define XFENCE __builtin_amdgcn_sched_barrier(0)
#define XPIN(x) asm volatile("" : "+v"(x))
  const int ntile = (int)((k1 - k0 + TKB - 1) / TKB);
  if (ntile > 0) issue(k0, 0);
  if (ntile > 1) issue(k0 + TKB, 1);
  xu32x4 qlr = qlow[0];                                                // low limb of query tile 0
  uint32_t inv[4] = {0u, 0u, 0u, 0u};     // all ones for key rows past the end of the range (last tile only)
  int slot = 0;
  for (int t = 0; t < ntile; ++t) {
    const long long kt = k0 + (long long)t * TKB;
    if (t + 1 < ntile) {
      if (use_mask) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // tile t landed, tile t+1 may be in flight
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();            // every wave's rows of tile t are in LDS, and everyone is done with tile t-1,
    if (t + 2 < ntile) issue(kt + 2 * TKB, slot == 0 ? 2 : slot - 1);     // whose slot tile t+2 now overwrites
    const float* tb = xl + slot * TILE;
    const uint32_t* tm = reinterpret_cast<const uint32_t*>(tb + TKB * RS * 2);
    slot = slot == 2 ? 0 : slot + 1;

    // K fragment = two swizzled 16-byte chunks of row j
    const float4 ka = *reinterpret_cast<const float4*>(tb + j * RS + (((hl * 8 + g * 2) ^ j) << 2));
    const float4 kc = *reinterpret_cast<const float4*>(tb + j * RS + (((hl * 8 + g * 2 + 1) ^ j) << 2));
    xu32x4 kh, km, kl;
    xsplit2v(ka.x, ka.y, kh, km, kl, 0); xsplit2v(ka.z, ka.w, kh, km, kl, 1);
    xsplit2v(kc.x, kc.y, kh, km, kl, 2); xsplit2v(kc.z, kc.w, kh, km, kl, 3);
    if (t == ntile - 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) inv[r] = kt + g * 4 + r < k1 ? 0u : 0xffffffffu;
    }
    f32x4 st[XQT];
    uint32_t need = 0u;
    uint32_t mw[4];                          // mask words of the key rows 4g..4g+3 for a PAIR of query tiles (16 bits each)
#pragma unroll
    for (int r = 0; r < 4; ++r) mw[r] = tm[(g * 4 + r) * 4];
    XFENCE;
#pragma unroll
    for (int s2 = 0; s2 <= XQT; ++s2) {
      // chunk = one MFMA of S^T[s2] + one piece of the mask / maximum of tile s2-1
      const int qc = s2 < XQT ? s2 : XQT - 1, qv = s2 > 0 ? s2 - 1 : 0;
      const uint32_t hm = ((honor >> qv) & 1u) << ((qv & 1) * 16 + j);   // this lane's bit of the pair's word, 0 if not honoured
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s2 < XQT) acc = xmfma_bf16(kh, qlr, acc);                   // the three small terms first
      XFENCE;
      if (s2 < XQT) acc = xmfma_bf16(km, qm[qc], acc);
      qlr = qlow[(s2 + 1 < XQT ? s2 + 1 : 0) * 64];                   // next tile's low limb (tile 0 again for the next keys)
      XFENCE;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (s2 < XQT) acc = c == 0 ? xmfma_bf16(kl, qh[qc], acc) : c == 1 ? xmfma_bf16(kh, qm[qc], acc)
                          : c == 2 ? xmfma_bf16(km, qh[qc], acc) : xmfma_bf16(kh, qh[qc], acc);
        if (s2 > 0) {
          float v = ((mw[c] & hm) | inv[c]) ? -INFINITY : st[qv][c];   // v_and_or, v_cmp, v_cndmask: no scalar detour
          XPIN(v);
          st[qv][c] = v;
        }
        XFENCE;
      }
      if (s2 < XQT) st[qc] = acc;
      if (s2 > 0) {
        // (asm: the values are already canonical, fmaxf would re-quiet every input; it also pins the result here)
        float tl;
        asm volatile("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %3, %4, %0"
                     : "=&v"(tl) : "v"(st[qv][2]), "v"(st[qv][3]), "v"(st[qv][0]), "v"(st[qv][1]));
        need |= (uint32_t)(tl > mrun[qv] + 10.f);
        if ((qv & 1) && qv + 1 < XQT) {                               // last use of this pair's words: fetch the next pair's
#pragma unroll
          for (int r = 0; r < 4; ++r) mw[r] = tm[(g * 4 + r) * 4 + ((qv + 1) >> 1)];
        }
      }
      XFENCE;
    }
    // lean online soft-max (see the f32 kernel): the reference maximum may go stale by 10 (log2 domain) before a wave vote
    // takes the rare path that renews it and rescales O and l
    if (__ballot(need != 0u) != 0ull) {
#pragma unroll
      for (int qt = 0; qt < XQT; ++qt) {
        const float tl = fmaxf(fmaxf(st[qt][0], st[qt][1]), fmaxf(st[qt][2], st[qt][3]));
        const float mnew = fmaxf(mrun[qt], group_max4(tl));
        const float alpha = exp2f(mrun[qt] - mnew);                     // XNONE -> finite: 0 (O and l are still 0); equal: 1
        lrun[qt] *= alpha;
        o[qt][0] *= alpha;
        o[qt][1] *= alpha;
        mrun[qt] = mnew;
      }
    }
    float2 vf[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = g * 4 + r;
      const int ch = (hl * 8 + (j >> 1)) ^ row;
      vf[r] = *reinterpret_cast<const float2*>(tb + TKB * RS + row * RS + ch * 4 + (j & 1) * 2);
    }
    xu32x2 vh[2], vm[2], vl[2];
    xsplit2w(vf[0].x, vf[1].x, vh[0], vm[0], vl[0], 0);
    xsplit2w(vf[2].x, vf[3].x, vh[0], vm[0], vl[0], 1);
    xsplit2w(vf[0].y, vf[1].y, vh[1], vm[1], vl[1], 0);
    xsplit2w(vf[2].y, vf[3].y, vh[1], vm[1], vl[1], 1);
    xu32x2 ph[2], pm[2], pl[2];
    XFENCE;
#pragma unroll
    for (int s2 = 0; s2 <= XQT; ++s2) {
      // chunk = the two MFMAs (one per 16-row half of O^T) of one limb product of tile s2-1 + a piece of P[s2]
      const int qc = s2 < XQT ? s2 : XQT - 1, bc = s2 & 1, qv = s2 > 0 ? s2 - 1 : 0, bv = qv & 1;
      f32x4 a0 = o[qv][0], a1 = o[qv][1];
      xf32x2 p01, p23;
      if (s2 > 0) { a0 = xmfma_bf16_k16(vm[0], pm[bv], a0); a1 = xmfma_bf16_k16(vm[1], pm[bv], a1); }
      const xf32x2 mref = {mrun[qc], mrun[qc]};
      if (s2 < XQT) {
        const xf32x2 d01 = xf32x2{st[qc][0], st[qc][1]} - mref;           // one v_pk_add_f32
        p01[0] = __builtin_amdgcn_exp2f(d01[0]);                          // raw v_exp_f32: arguments <= 10, and a result
        p01[1] = __builtin_amdgcn_exp2f(d01[1]);                          // below 2^-126 may flush to zero
        XPIN(p01);
      }
      XFENCE;
      if (s2 > 0) { a0 = xmfma_bf16_k16(vh[0], pl[bv], a0); a1 = xmfma_bf16_k16(vh[1], pl[bv], a1); }
      if (s2 < XQT) {
        const xf32x2 d23 = xf32x2{st[qc][2], st[qc][3]} - mref;
        p23[0] = __builtin_amdgcn_exp2f(d23[0]);
        p23[1] = __builtin_amdgcn_exp2f(d23[1]);
        XPIN(p23);
        const xf32x2 ps = p01 + p23;
        lrun[qc] += ps[0] + ps[1];                                         // per-lane partial: groups merged at the end
      }
      XFENCE;
      // P limbs of tile s2 go to buffer bc, the MFMAs read buffer bv = the other one
      if (s2 > 0) { a0 = xmfma_bf16_k16(vl[0], ph[bv], a0); a1 = xmfma_bf16_k16(vl[1], ph[bv], a1); }
      if (s2 < XQT) { xsplit2w(p01[0], p01[1], ph[bc], pm[bc], pl[bc], 0); XPIN(pl[bc]); }
      XFENCE;
      if (s2 > 0) { a0 = xmfma_bf16_k16(vh[0], pm[bv], a0); a1 = xmfma_bf16_k16(vh[1], pm[bv], a1); }
      if (s2 < XQT) { xsplit2w(p23[0], p23[1], ph[bc], pm[bc], pl[bc], 1); XPIN(pl[bc]); }
      XFENCE;
      if (s2 > 0) {
        a0 = xmfma_bf16_k16(vm[0], ph[bv], a0); a1 = xmfma_bf16_k16(vm[1], ph[bv], a1);
        a0 = xmfma_bf16_k16(vh[0], ph[bv], a0); a1 = xmfma_bf16_k16(vh[1], ph[bv], a1);
        o[qv][0] = a0; o[qv][1] = a1;
      }
      XFENCE;
    }
  }
#undef XFENCE
#undef XPIN

  const long long slot_o = ((long long)b * NS + s) * M + h;
#pragma unroll
  for (int qt = 0; qt < XQT; ++qt) {
    const int q = qt * 16 + j;
    if (q < Q) {
      float* op = part_o + (slot_o * Q + q) * D + g * 8;
      st4(op, make_float4(o[qt][0][0], o[qt][1][0], o[qt][0][1], o[qt][1][1]));
      st4(op + 4, make_float4(o[qt][0][2], o[qt][1][2], o[qt][0][3], o[qt][1][3]));
      // log2 domain -> natural; a range with nothing unblocked publishes m = -inf (the merge skips it)
      const float mo = mrun[qt] == XNONE ? -INFINITY : mrun[qt] * 0.6931471805599453f, lo = group_sum4(lrun[qt]);
      if (g == 0) *reinterpret_cast<float2*>(part_ml + (slot_o * Q + q) * 2) = make_float2(mo, lo);
    }
  }
}

// out[b, q, h*32+d] = sum_s e^{m_s - m*} o_s / sum_s e^{m_s - m*} l_s     (m* = max_s m_s)
// Range merge shared by `xattn_combine_kernel` and `xattn_merge_local_kernel`: one 1024-thread block per (q, b), two passes
// with no dependent chain across ranges.  Pass 1: thread (head = tid >> 7, lane = tid & 127) takes the maximum of its ranges'
// m_s; two shuffles trees + 16 floats of LDS give m*[head].  Pass 2: thread (head, range lane sl = (tid & 127) >> 3, dim quad
// d4 = tid & 7) walks s = sl, sl + 16, ... with INDEPENDENT 16-byte loads of o_s and accumulates w_s o_s, w_s l_s with
// w_s = e^{m_s - m*}; the 16 range lanes meet through shuffles and LDS.  The round-5 form kept an online (m, num, den) triple per
// lane -- a chain of NS / 4 dependent exponentials (combine, 30-39 us at NS = 256) or NS of them (merge_local, 97 us): the
// step of a frame shard paid 0.87 ms for nine merges.
struct RangeMerge { float4 num; float den, mstar; bool writer; int h, d4; };
__device__ __forceinline__ RangeMerge merge_ranges(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                   int b, int q, int Q, int NS) {
  constexpr int D = 32, M = 8;
  __shared__ float s_m[M][2];
  __shared__ float s_acc[M][8][5];
  const int tid = threadIdx.x, h = tid >> 7, r = tid & 127, wave2 = (tid >> 6) & 1;
  const long long base = (long long)b * NS * M + h;                 // slot(s) = base + s * M
  float m = -INFINITY;
  for (int s = r; s < NS; s += 128) m = fmaxf(m, part_ml[((base + (long long)s * M) * Q + q) * 2]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) s_m[h][wave2] = m;
  __syncthreads();
  const float mstar = fmaxf(s_m[h][0], s_m[h][1]);
  const int d4 = r & 7, sl = r >> 3;
  float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
  float den = 0.f;
#pragma unroll 4
  for (int s = sl; s < NS; s += 16) {
    const long long row = (base + (long long)s * M) * Q + q;
    const float2 ml = *reinterpret_cast<const float2*>(part_ml + row * 2);
    const float4 ov = *reinterpret_cast<const float4*>(part_o + row * D + d4 * 4);
    const float w = ml.x == -INFINITY ? 0.f : __expf(ml.x - mstar);   // an empty range publishes (-inf, 0, 0)
    num.x = fmaf(w, ov.x, num.x); num.y = fmaf(w, ov.y, num.y); num.z = fmaf(w, ov.z, num.z); num.w = fmaf(w, ov.w, num.w);
    den = fmaf(w, ml.y, den);
  }
  // range lanes: bits 3..5 of the lane within a wave, then the two waves of a head
#pragma unroll
  for (int off = 8; off <= 32; off <<= 1) {
    num.x += __shfl_xor(num.x, off); num.y += __shfl_xor(num.y, off); num.z += __shfl_xor(num.z, off);
    num.w += __shfl_xor(num.w, off); den += __shfl_xor(den, off);
  }
  if (wave2 == 1 && (tid & 63) < 8) {
    float* a = s_acc[h][d4];
    a[0] = num.x; a[1] = num.y; a[2] = num.z; a[3] = num.w; a[4] = den;
  }
  __syncthreads();
  RangeMerge o;
  o.writer = wave2 == 0 && (tid & 63) < 8;
  if (o.writer) {
    const float* a = s_acc[h][d4];
    num.x += a[0]; num.y += a[1]; num.z += a[2]; num.w += a[3]; den += a[4];
  }
  o.num = num; o.den = den; o.mstar = mstar; o.h = h; o.d4 = d4;
  return o;
}

__global__ __launch_bounds__(1024) void xattn_combine_kernel(const float* __restrict__ part_o,
                                                            const float* __restrict__ part_ml,
                                                            float* __restrict__ out, int Q, int NS) {
  constexpr int D = 32, M = 8;
  const int b = blockIdx.y, q = blockIdx.x;
  const RangeMerge o = merge_ranges(part_o, part_ml, b, q, Q, NS);
  if (o.writer)
    *reinterpret_cast<float4*>(out + ((long long)b * Q + q) * (M * D) + o.h * D + o.d4 * 4) =
        make_float4(o.num.x / o.den, o.num.y / o.den, o.num.z / o.den, o.num.w / o.den);
}

// ---- frame-sharded clips: one message per decoder layer and rank -------------------------------------------------------
// A rank's key ranges are merged locally into ONE un-normalised partial per (batch, head, query) and packed with the
// rank's 128-bit "this query has an unblocked key among MY keys" flags into a record of M*Q*(D+2) + 4 floats
// (108.8 KB at M=8, Q=100, D=32); the records of all ranks are all-gathered and merged by xattn_combine_packed.
// The all-blocked reset of mask2former_head.py:453-454 is a property of the WHOLE clip: a rank whose keys are all
// blocked for a query attends unmasked (its kernel cannot know the other ranks' bits); the merge keeps that
// contribution only if EVERY rank reported the query blocked -- so no flag exchange is needed before the attention.
__global__ __launch_bounds__(1024) void xattn_merge_local_kernel(const float* __restrict__ part_o,
                                                                const float* __restrict__ part_ml,
                                                                const uint32_t* __restrict__ flags,
                                                                float* __restrict__ packed, int Q, int NS, int rec) {
  constexpr int D = 32, M = 8;
  const int b = blockIdx.y, q = blockIdx.x;
  const RangeMerge o = merge_ranges(part_o, part_ml, b, q, Q, NS);
  float* r = packed + (long long)b * rec;
  if (o.writer) {
    *reinterpret_cast<float4*>(r + ((long long)o.h * Q + q) * D + o.d4 * 4) = o.num;
    if (o.d4 == 0) *reinterpret_cast<float2*>(r + (long long)M * Q * D + ((long long)o.h * Q + q) * 2) = make_float2(o.mstar, o.den);
  }
  if (q == 0 && threadIdx.x < 4)
    reinterpret_cast<uint32_t*>(r + (long long)M * Q * (D + 2))[threadIdx.x] = flags ? flags[b * 4 + threadIdx.x] : 0xffffffffu;
}

__global__ __launch_bounds__(256) void xattn_combine_packed_kernel(const float* __restrict__ packed,
                                                                  float* __restrict__ out, int R, int B, int Q, int rec) {
  constexpr int D = 32, M = 8;
  const int b = blockIdx.y, q = blockIdx.x;
  const int hd = threadIdx.x, h = hd >> 5, d = hd & 31;
  // does any rank see an unblocked key for this query?
  bool any = false;
  for (int r = 0; r < R; ++r) {
    const uint32_t* fl = reinterpret_cast<const uint32_t*>(packed + ((long long)r * B + b) * rec + (long long)M * Q * (D + 2));
    any = any || ((fl[q >> 5] >> (q & 31)) & 1u);
  }
  float m = -INFINITY, num = 0.f, den = 0.f;
  for (int r = 0; r < R; ++r) {
    const float* rp = packed + ((long long)r * B + b) * rec;
    const uint32_t* fl = reinterpret_cast<const uint32_t*>(rp + (long long)M * Q * (D + 2));
    const bool mine = (fl[q >> 5] >> (q & 31)) & 1u;
    if (any && !mine) continue;                     // this rank attended unmasked for a query that is not reset
    const float2 ml = *reinterpret_cast<const float2*>(rp + (long long)M * Q * D + ((long long)h * Q + q) * 2);
    if (ml.x == -INFINITY) continue;
    const float ov = rp[((long long)h * Q + q) * D + d];
    const float mn = fmaxf(m, ml.x);
    const float a = __expf(m - mn), w = __expf(ml.x - mn);
    num = num * a + w * ov;
    den = den * a + w * ml.y;
    m = mn;
  }
  out[((long long)b * Q + q) * (M * D) + hd] = num / den;
}

}  // namespace pvsg

extern "C" int pvsg_xattn_num_splits(int B, long long K) {
  long long ns = (256 + B - 1) / B;
  const long long maxs = (K + 63) / 64;     // down to two 32-key tiles per workgroup: short key ranges (few frames
                                            // per GPU, self-attention) are latency-bound, spread them over the CUs
  if (ns > maxs) ns = maxs;
  if (ns < 1) ns = 1;
  return (int)ns;
}

extern "C" int pvsg_masked_xattn_partial_strided(const float* q_proj, const float* k_proj, const float* v_proj,
                                                 const uint32_t* mask_bits, const uint32_t* mask_flags,
                                                 float* part_o, float* part_ml, int B, int Q, long long K,
                                                 int M, int D, int NS, long long kv_row_stride, hipStream_t stream);
extern "C" int pvsg_masked_xattn_partial(const float* q_proj, const float* k_proj, const float* v_proj,
                                         const uint32_t* mask_bits, const uint32_t* mask_flags,
                                         float* part_o, float* part_ml, int B, int Q, long long K,
                                         int M, int D, int NS, hipStream_t stream) {
  return pvsg_masked_xattn_partial_strided(q_proj, k_proj, v_proj, mask_bits, mask_flags, part_o, part_ml, B, Q, K, M, D, NS,
                                           (long long)M * D, stream);
}

extern "C" int pvsg_masked_xattn_partial_strided(const float* q_proj, const float* k_proj, const float* v_proj,
                                                 const uint32_t* mask_bits, const uint32_t* mask_flags,
                                                 float* part_o, float* part_ml, int B, int Q, long long K,
                                                 int M, int D, int NS, long long kv_row_stride, hipStream_t stream) {
  using namespace pvsg;
  const long long kvs = kv_row_stride;
  PVSG_REQUIRE(kvs >= (long long)M * D && kvs % 4 == 0, "masked_xattn_partial: key / value row stride must be >= M*D floats and a multiple of 4");
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
  chunk = (chunk + TK - 1) / TK * TK;
  const size_t lds = (size_t)2 * XLDS_TILE_FLOATS * sizeof(float);
  static std::atomic<unsigned long long> attr_done;
  {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&xattn_partial_lds_kernel<true>), (int)lds, attr_done);
    static std::atomic<unsigned long long> attr_done2;
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void*>(&xattn_partial_lds_kernel<false>), (int)lds, attr_done2);
    if (e != hipSuccess)
      return set_err(PVSG_ERR_HIP, "masked_xattn_partial: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
  }
  // default: the bf16-split kernel (f32-exact on the bf16 matrix pipe), half-head workgroups.  PVSG_XATTN=f32 selects the
  // f32-MFMA kernel, PVSG_XATTN_LEAN=0 its textbook online soft-max, PVSG_XATTN_HW=8 whole-row workgroups for the split
  // kernel.  Read per call (a getenv is nanoseconds against a launch) so that one process can test them all.
  const auto env_is = [](const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; };
  const bool lean = !env_is("PVSG_XATTN_LEAN", '0');
  const bool bx = !env_is("PVSG_XATTN", 'f') && lean;
  const bool hw4 = !env_is("PVSG_XATTN_HW", '8');
  if (bx) {
    const long long cb = (chunk + TKB - 1) / TKB * TKB;
    const int hw = hw4 ? 4 : 8;
    const size_t ldsb = (size_t)xbx_lds_floats(hw) * sizeof(float);
    const void* kern = hw4 ? reinterpret_cast<const void*>(&xattn_partial_bf16x3_kernel<4>)
                           : reinterpret_cast<const void*>(&xattn_partial_bf16x3_kernel<8>);
    static std::atomic<unsigned long long> attr_done3, attr_done4;
    const hipError_t e = ensure_dynamic_lds(kern, (int)ldsb, hw4 ? attr_done4 : attr_done3);
    if (e != hipSuccess)
      return set_err(PVSG_ERR_HIP, "masked_xattn_partial: cannot reserve %zu bytes of LDS: %s", ldsb, hipGetErrorString(e));
    if (hw4)
      hipLaunchKernelGGL(xattn_partial_bf16x3_kernel<4>, dim3(B * NS * 2), dim3(256), ldsb, stream, q_proj, k_proj, v_proj,
                         mask_bits, mask_flags, part_o, part_ml, Q, K, NS, cb, kvs);
    else
      hipLaunchKernelGGL(xattn_partial_bf16x3_kernel<8>, dim3(B * NS), dim3(512), ldsb, stream, q_proj, k_proj, v_proj,
                         mask_bits, mask_flags, part_o, part_ml, Q, K, NS, cb, kvs);
  } else if (lean) {
    hipLaunchKernelGGL(xattn_partial_lds_kernel<true>, dim3(B * NS), dim3(512), lds, stream, q_proj, k_proj, v_proj,
                       mask_bits, mask_flags, part_o, part_ml, Q, K, NS, chunk, kvs);
  } else {
    hipLaunchKernelGGL(xattn_partial_lds_kernel<false>, dim3(B * NS), dim3(512), lds, stream, q_proj, k_proj, v_proj,
                       mask_bits, mask_flags, part_o, part_ml, Q, K, NS, chunk, kvs);
  }
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

extern "C" int pvsg_xattn_merge_local(const float* part_o, const float* part_ml, const uint32_t* mask_flags,
                                      float* packed, int B, int Q, int M, int D, int NS, hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(part_o && part_ml && packed, "xattn_merge_local: null pointer argument");
  PVSG_REQUIRE(B > 0 && Q > 0 && NS > 0, "xattn_merge_local: non-positive dimension");
  if (M != 8 || D != 32 || Q > 128)
    return set_err(PVSG_ERR_UNSUPPORTED, "xattn_merge_local: built for 8 heads x 32 dims, Q<=128");
  const int rec = M * Q * (D + 2) + 4;
  hipLaunchKernelGGL(xattn_merge_local_kernel, dim3(Q, B), dim3(1024), 0, stream, part_o, part_ml, mask_flags, packed, Q,
                     NS, rec);
  PVSG_LAUNCH_CHECK("xattn_merge_local");
  return PVSG_OK;
}

extern "C" int pvsg_xattn_combine_packed(const float* packed, float* out, int R, int B, int Q, int M, int D,
                                         hipStream_t stream) {
  using namespace pvsg;
  PVSG_REQUIRE(packed && out, "xattn_combine_packed: null pointer argument");
  PVSG_REQUIRE(R > 0 && B > 0 && Q > 0, "xattn_combine_packed: non-positive dimension");
  if (M != 8 || D != 32 || Q > 128)
    return set_err(PVSG_ERR_UNSUPPORTED, "xattn_combine_packed: built for 8 heads x 32 dims, Q<=128");
  const int rec = M * Q * (D + 2) + 4;
  hipLaunchKernelGGL(xattn_combine_packed_kernel, dim3(Q, B), dim3(256), 0, stream, packed, out, R, B, Q, rec);
  PVSG_LAUNCH_CHECK("xattn_combine_packed");
  return PVSG_OK;
}
