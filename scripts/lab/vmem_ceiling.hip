// What one CU's vector-memory path delivers for `global_load_dwordx4` by all 64 lanes (1 KiB per wave instruction), the access
// shape of msda_fused's gathers (8 heads x 128-byte rows) and of the row kernels' weight stream.  Every workgroup (one per CU,
// `waves` waves) walks a private or shared buffer of `span` bytes `reps` times:
//   span small  (16 KB)         -> L1 (TCP) hits
//   span medium (1 MB / XCD)    -> L2 hits, L1 misses
//   span large  (64 MB)         -> MALL / HBM
// patterns: 0 contiguous 1 KiB per wave instruction; 1 eight 128-byte rows 1 KiB apart; 2 eight 128-byte rows at random places
// (the MSDA gather: lane (h, q) reads 16 bytes of head h's row).  Prints GB/s per CU (HIP events) and bytes per s_memtime tick.
//   hipcc --offload-arch=gfx950 -O3 scripts/lab/vmem_ceiling.hip -o /tmp/vmem_ceiling && /tmp/vmem_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(1024) void walk(const char* __restrict__ buf, long long span, long long wg_stride, int reps,
                                             float* __restrict__ sink, unsigned long long* __restrict__ clocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = buf + (long long)blockIdx.x * wg_stride;
  // Eight independent wave loads per step in every pattern (8 KiB requested per wave and step):
  //  0: eight contiguous 1 KiB slabs (lane l: bytes 16 l .. 16 l + 15 of each)
  //  1: an 8 KiB slab as eight 128-byte rows 1 KiB apart, segment k of every row per instruction (lane (h = l >> 3, q = l & 7))
  //  2: GATHER -- per instruction eight 128-byte rows at pseudo-random 128-byte-aligned places of the span, one per lane group
  //     (the shape of msda_fused's corner loads: a head's 32 channels of one token)
  const long long nstep = span / 8192;
  const unsigned nrow = (unsigned)(span / 128);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned seed = (blockIdx.x * 131u + wave * 17u + (lane >> 3)) * 2654435761u + 12345u;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    const long long steps = nstep > nw ? nstep : nw;            // small spans: every wave still walks (step -> slab modulo the span)
    for (long long s0 = wave; s0 < steps; s0 += nw) {
      const long long s = s0 % nstep;
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (PATTERN == 0) {
          v[k] = *reinterpret_cast<const f32x4*>(base + s * 8192 + k * 1024 + lane * 16);
        } else if (PATTERN == 1) {
          v[k] = *reinterpret_cast<const f32x4*>(base + s * 8192 + (lane >> 3) * 1024 + k * 128 + (lane & 7) * 16);
        } else {
          seed = seed * 1664525u + 1013904223u;
          const unsigned row = (seed >> 8) % nrow;
          v[k] = *reinterpret_cast<const f32x4*>(base + (long long)row * 128 + (lane & 7) * 16);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) sink[0] = acc[0];
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t total = (size_t)cus * (64u << 20) / 8;             // up to 8 MB per CU
  char* buf;
  float* sink;
  unsigned long long* clocks;
  hipMalloc(&buf, total);
  hipMemset(buf, 0, total);
  hipMalloc(&sink, 16);
  hipMalloc(&clocks, sizeof(unsigned long long) * cus);
  std::vector<unsigned long long> h(cus);
  printf("# %s, %d CUs, shader clock %d MHz nominal\n", prop.gcnArchName, cus, prop.clockRate / 1000);
  printf("# pattern  waves/CU  span/CU                bytes per s_memtime tick per CU   GB/s/CU (HIP events)   aggregate TB/s\n");
  struct Cfg { long long span, stride; const char* what; };
  const Cfg cfgs[] = {{16 << 10, 8 << 20, "16 KB private (L1)"}, {256 << 10, 8 << 20, "256 KB private (L2)"},
                      {1 << 20, 0, "1 MB shared (L2)"}, {8 << 20, 8 << 20, "8 MB private (HBM)"}};
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int waves : {8, 16})
      for (const Cfg& c : cfgs) {
        const long long bytes_target = 64ll << 20;               // bytes each CU pulls
        const long long per_rep = c.span > (long long)waves * 8192 ? c.span : (long long)waves * 8192;
        const int reps = (int)(bytes_target / per_rep > 0 ? bytes_target / per_rep : 1);
        for (int it = 0; it < 2; ++it) {
          hipEvent_t e0, e1;
          hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          if (pattern == 2) hipLaunchKernelGGL(walk<2>, dim3(cus), dim3(waves * 64), 0, 0, buf, c.span, c.stride, reps, sink, clocks);
          else if (pattern == 0) hipLaunchKernelGGL(walk<0>, dim3(cus), dim3(waves * 64), 0, 0, buf, c.span, c.stride, reps, sink, clocks);
          else hipLaunchKernelGGL(walk<1>, dim3(cus), dim3(waves * 64), 0, 0, buf, c.span, c.stride, reps, sink, clocks);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms = 0.f;
          hipEventElapsedTime(&ms, e0, e1);
          if (it == 0) continue;
          hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost);
          double clk = 0;
          for (int i = 0; i < cus; ++i) clk += (double)h[i];
          clk /= cus;
          const double bytes = (double)per_rep * reps;
          printf("  %d        %2d        %-22s %8.1f                        %8.1f             %8.2f\n", pattern, waves, c.what, bytes / clk, bytes / (ms * 1e6),
                 bytes * cus / (ms * 1e9));
        }
      }
  return 0;
}
