// What one CU's vector-memory path delivers for `global_load_dwordx4` by all 64 lanes (1 KiB per wave instruction), the access
// shape of msda_fused's gathers (8 heads x 128-byte rows) and of the row kernels' weight stream.  Every workgroup (one per CU,
// `waves` waves) walks a private or shared buffer of `span` bytes `reps` times:
//   span small  (16 KB)         -> L1 (TCP) hits
//   span medium (1 MB / XCD)    -> L2 hits, L1 misses
//   span large  (64 MB)         -> MALL / HBM
// pattern 0: one contiguous 1 KiB per wave instruction; pattern 1: eight 128-byte rows 1 KiB apart (the MSDA shape: lane (h, q)
// reads 16 bytes of row h).  Prints bytes / clock / CU (s_memtime clocks of the shader) and GB/s per CU.
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
  // PATTERN 0: lane l reads bytes [16 l, 16 l + 16) of a 1 KiB slab; PATTERN 1: lane (h = l >> 3, q = l & 7) reads 16 bytes at
  // h * 1 KiB + 16 q of an 8 KiB slab (eight 128-byte rows): 8 slabs' worth of row 0..7 segments are visited by 8 instructions
  const long long slab = PATTERN == 0 ? 1024 : 8192;
  const long long nslab = span / slab;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    for (long long s = wave; s < nslab; s += nw) {
      if (PATTERN == 0) {
        acc += *reinterpret_cast<const f32x4*>(base + s * 1024 + lane * 16);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)       // segment k of each of the 8 rows of this slab
          acc += *reinterpret_cast<const f32x4*>(base + s * 8192 + (lane >> 3) * 1024 + k * 128 + (lane & 7) * 16);
      }
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
  const Cfg cfgs[] = {{16 << 10, 8 << 20, "16 KB private (L1)"}, {64 << 10, 8 << 20, "64 KB private (L2)"},
                      {1 << 20, 0, "1 MB shared (L2)"}, {8 << 20, 8 << 20, "8 MB private (HBM)"}};
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int waves : {4, 8, 16})
      for (const Cfg& c : cfgs) {
        const long long bytes_target = 64ll << 20;               // bytes each CU pulls
        const int reps = (int)(bytes_target / c.span > 0 ? bytes_target / c.span : 1);
        for (int it = 0; it < 2; ++it) {
          hipEvent_t e0, e1;
          hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0);
          if (pattern == 0) hipLaunchKernelGGL(walk<0>, dim3(cus), dim3(waves * 64), 0, 0, buf, c.span, c.stride, reps, sink, clocks);
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
          const double bytes = (double)c.span * reps;
          printf("  %d        %2d        %-22s %8.1f                        %8.1f             %8.2f\n", pattern, waves, c.what, bytes / clk, bytes / (ms * 1e6),
                 bytes * cus / (ms * 1e9));
        }
      }
  return 0;
}
