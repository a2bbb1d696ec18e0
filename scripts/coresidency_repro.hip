// Single-file reproducer (no torch, no library) of the co-residency finding of DESIGN.md section 3.12: a kernel that keeps the
// bf16 matrix pipe busy in ONE process corrupts results of ANOTHER process's kernels when waves of both share compute units.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o coresidency_repro scripts/coresidency_repro.hip
//   ./coresidency_repro [co-runner] [launches] [victim]
//       co-runner: none | bf16_16x16x32 | f16_16x16x32 | bf16_32x32x16 | f16_32x32x16 | f32_32x32x2 | valu   (default bf16_16x16x32)
//       victim:    msda | msdav<flags> | trans | imul | misc | gather | regs | both (= gather + regs) | all                (default both)
//   HSA_CU_MASK is honoured per process through the environment variables REPRO_VICTIM_CU_MASK / REPRO_SPIN_CU_MASK
//   (e.g. REPRO_VICTIM_CU_MASK=0:0-127 REPRO_SPIN_CU_MASK=0:128-255 -> disjoint CU sets -> no corruption expected).
//
// The program forks BEFORE touching HIP: the child runs the co-runner in a loop until told to stop, the parent launches the
// victim kernels `launches` times on static inputs and compares every output word with the first launch (which ran alone,
// before the child started).  Two victims:
//   gather  out[i] = sum of 8 16-byte rows table[idx[i][k]] (the access pattern of the deformable-attention gather the
//           finding was made on: per-lane computed addresses, rows of a 32 MB table)
//   regs    no memory traffic while it runs: every lane fills 64 VGPRs with a lane-dependent pattern, idles ~20 us in a
//           dependent integer chain, then checks the registers and writes ONE word (0 = intact) -- separates "register file /
//           wave state corrupted" from "memory path returned wrong data"
// Exit code 0, one JSON line on stdout: {"corunner":..., "launches":N, "gather_bad_launches":..., "gather_bad_words":...,
// "gather_bad_lanes_histogram":[64 counters], "regs_bad_launches":..., "runtime":..., "device":...}
#include <hip/hip_runtime.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e__ = (x);                                                                       \
    if (e__ != hipSuccess) {                                                                    \
      fprintf(stderr, "[%d] %s:%d %s -> %s\n", (int)getpid(), __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

// the product's deformable-attention kernels, compiled into this file (victim "msda": the kernel the finding was made on)
#include "../openpvsg_amd/csrc/msda.hip"
namespace pvsg {
static thread_local char g_err_[512];
char* err_buf() { return g_err_; }
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err_, sizeof(g_err_), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace pvsg

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- co-runners: nothing but matrix (or vector) instructions on registers, ~iters x 16 instructions per wave ----------
template <int KIND>
__global__ __launch_bounds__(256) void spin_kernel(float* sink, int iters) {
  const int t = threadIdx.x;
  float total = 0.f;
  if (KIND == 0) {                                   // v_mfma_f32_16x16x32_bf16
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.01f * (t + i)); b[i] = (__bf16)(1.0f - 0.02f * (t - i)); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
    for (int u = 0; u < 4; ++u) total += acc[u][0];
  } else if (KIND == 5) {                            // v_mfma_f32_16x16x32_f16 (round 4: the instruction of the f16x2 split kernels)
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.01f * (t + i)); b[i] = (_Float16)(1.0f - 0.02f * (t - i)); }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
    for (int u = 0; u < 4; ++u) total += acc[u][0];
  } else if (KIND == 1) {                            // v_mfma_f32_32x32x16_bf16
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.01f * (t + i)); b[i] = (__bf16)(1.0f - 0.02f * (t - i)); }
    f32x16 acc[2];
    for (int u = 0; u < 2; ++u)
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 1], 0, 0, 0);
    total = acc[0][0] + acc[1][0];
  } else if (KIND == 2) {                            // v_mfma_f32_32x32x16_f16
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.01f * (t + i)); b[i] = (_Float16)(1.0f - 0.02f * (t - i)); }
    f32x16 acc[2];
    for (int u = 0; u < 2; ++u)
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
    total = acc[0][0] + acc[1][0];
  } else if (KIND == 3) {                            // v_mfma_f32_32x32x2_f32
    const float a = 0.5f + 0.01f * t, b = 1.0f - 0.02f * t;
    f32x16 acc[2];
    for (int u = 0; u < 2; ++u)
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
    total = acc[0][0] + acc[1][0];
  } else {                                           // vector ALU only
    float x = 0.5f + t, y = 1.f;
    for (int it = 0; it < iters * 16; ++it) { x = fmaf(x, 1.0000001f, 0.25f); y = fmaf(y, 0.9999999f, x); }
    total = x + y;
  }
  if (total == 123.456f) sink[t] = total;
}

// ---- victims ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_kernel(const f32x4* __restrict__ table, const int* __restrict__ idx,
                                                    f32x4* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) s += table[idx[i * 8 + k]];
  out[i] = s;
}


// victim "msdav<flags>": the gather kernel above with parts switched off, to find what the co-runner disturbs
//   bit 0: plain loads / stores instead of nontemporal ones   bit 1: no soft-max (weights = raw logits)
//   bit 2: no sampling (output = per-head sums of offsets and weights only)   bit 3: one wave per workgroup
template <int FLAGS>
__global__ __launch_bounds__(256) void msda_variant_kernel(const float* __restrict__ value, long long value_stride,
                                                          const float* __restrict__ oa, long long oa_stride,
                                                          const float* __restrict__ pos_oa, const float* __restrict__ ref,
                                                          const long long* __restrict__ shapes, const long long* __restrict__ lsi,
                                                          float* __restrict__ out, int S, int Lq, long long nq_total) {
  constexpr int L = 3, P = 4, M = 8, D = 32, LP = L * P;
  const long long gq = (FLAGS & 8) ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gq >= nq_total) return;
  const int lane = threadIdx.x & 63;
  const int m = lane >> 3, c4 = lane & 7;
  const int b = (int)(gq / Lq);
  const int q = (int)(gq - (long long)b * Lq);
  const float* offp = oa + gq * oa_stride + m * (LP * 2);
  const float* logp = oa + gq * oa_stride + M * LP * 2 + m * LP;
  const float* poff = pos_oa + (long long)q * (M * LP * 3) + m * (LP * 2);
  const float* plog = pos_oa + (long long)q * (M * LP * 3) + M * LP * 2 + m * LP;
  float ox[LP], oy[LP], aw[LP];
#pragma unroll
  for (int i = 0; i < LP / 2; ++i) {
    float4 t = (FLAGS & 1) ? pvsg::ld4(offp + 4 * i) : pvsg::ld4_stream(offp + 4 * i);
    const float4 u = pvsg::ld4(poff + 4 * i);
    ox[2 * i] = t.x + u.x; oy[2 * i] = t.y + u.y; ox[2 * i + 1] = t.z + u.z; oy[2 * i + 1] = t.w + u.w;
  }
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    float4 t = (FLAGS & 1) ? pvsg::ld4(logp + 4 * i) : pvsg::ld4_stream(logp + 4 * i);
    const float4 u = pvsg::ld4(plog + 4 * i);
    aw[4 * i] = t.x + u.x; aw[4 * i + 1] = t.y + u.y; aw[4 * i + 2] = t.z + u.z; aw[4 * i + 3] = t.w + u.w;
  }
  float inv = 1.f;
  if (!(FLAGS & 2)) {
    float mx = aw[0];
#pragma unroll
    for (int i = 1; i < LP; ++i) mx = fmaxf(mx, aw[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { aw[i] = __expf(aw[i] - mx); sum += aw[i]; }
    inv = 1.f / sum;
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FLAGS & 4) {
#pragma unroll
    for (int i = 0; i < LP; ++i) { acc.x += ox[i]; acc.y += oy[i]; acc.z += aw[i] * inv; acc.w += ox[i] * aw[i]; }
  } else {
    const float rx = ref[2 * q], ry = ref[2 * q + 1];
    const float* vbase = value + (long long)b * S * value_stride + m * D + c4 * 4;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float* vl = vbase + lsi[l] * value_stride;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float locx = rx + ox[l * P + p] / (float)W, locy = ry + oy[l * P + p] / (float)H;
        const float him = locy * (float)H - 0.5f, wim = locx * (float)W - 0.5f;
        const int h0 = min(max((int)floorf(him), 0), H - 1), w0 = min(max((int)floorf(wim), 0), W - 1);
        const float4 v = pvsg::ld4(vl + (long long)(h0 * W + w0) * value_stride);
        const float a = aw[l * P + p] * inv;
        acc.x += a * v.x; acc.y += a * v.y; acc.z += a * v.z; acc.w += a * v.w;
      }
    }
  }
  if (FLAGS & 1) pvsg::st4(out + gq * 256 + lane * 4, acc);
  else pvsg::st4_stream(out + gq * 256 + lane * 4, acc);
}

// victim "trans": no memory traffic while it runs; each lane evaluates chains of ONE transcendental instruction each
// (v_exp_f32, v_rcp_f32, v_sqrt_f32, v_log_f32) plus a plain FMA chain as the control, one output word per chain
__global__ __launch_bounds__(256) void trans_kernel(float* __restrict__ out, int reps) {
  const unsigned gid = threadIdx.x + blockIdx.x * 256u;
  const float x0 = 0.001f * (float)(gid & 1023u);
  float e = 0.f, r = 0.f, q = 0.f, l = 0.f, f = 0.f;
  for (int it = 0; it < reps; ++it) {
    const float x = x0 + 0.01f * (float)(it & 63);
    e += __builtin_amdgcn_exp2f(x - 3.f);
    r += __builtin_amdgcn_rcpf(x + 1.5f);
    q += __builtin_amdgcn_sqrtf(x + 0.25f);
    l += __builtin_amdgcn_logf(x + 2.f);
    f = fmaf(f, 0.999f, x);
  }
  out[gid * 5 + 0] = e;
  out[gid * 5 + 1] = r;
  out[gid * 5 + 2] = q;
  out[gid * 5 + 3] = l;
  out[gid * 5 + 4] = f;
}

// victim "misc": chains of the remaining instruction groups of the gather kernel: packed f32 math (v_pk_fma_f32), IEEE division
// (v_div_scale / v_div_fmas / v_div_fixup), 16-bit boolean logic, compare + select through VCC, divergent branches
__global__ __launch_bounds__(256) void misc_kernel(float* __restrict__ out, int reps) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const unsigned gid = threadIdx.x + blockIdx.x * 256u;
  const float x0 = 0.37f + 0.001f * (float)(gid & 1023u);
  f2 pk = {x0, 1.f - x0}, pk2 = {0.5f, 0.25f};
  float dv = 0.f, sel = 0.f, br = 0.f;
  unsigned short flags = 0;
  for (int it = 0; it < reps; ++it) {
    const float x = x0 + 0.01f * (float)(it & 63);
    pk = pk * f2{0.999f, 1.001f} + pk2;                                  // v_pk_fma_f32 / v_pk_mul_f32
    pk2 = pk2 + f2{x, -x} * 1e-3f;                                       // v_pk_add_f32
    dv += (x + 1.f) / (x * x + 0.37f);                                   // IEEE division sequence
    const bool a = x > 0.5f, b = (gid + it) & 1u, c = dv > (float)it * 0.5f;
    flags = (unsigned short)(flags * 3u + ((a && b) ? 1u : 0u) + ((b != c) ? 2u : 0u));   // 16-bit logic
    sel += a ? (c ? x : -x) : (b ? 0.5f : 0.25f);                        // v_cmp + v_cndmask through VCC
    if ((gid ^ it) & 4u) br = br * 0.5f + x; else if ((gid + it) % 3u == 0u) br -= 0.125f;   // divergent branches
  }
  out[gid * 5 + 0] = pk[0] + pk[1];
  out[gid * 5 + 1] = dv;
  out[gid * 5 + 2] = (float)flags;
  out[gid * 5 + 3] = sel;
  out[gid * 5 + 4] = br + pk2[0];
}

// victim "imul": chains of the quarter-rate integer multiplies (v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32) and of the
// float -> int conversions / floor the gather's address arithmetic is made of; one output word per chain
__global__ __launch_bounds__(256) void imul_kernel(unsigned* __restrict__ out, int reps) {
  const unsigned gid = threadIdx.x + blockIdx.x * 256u;
  unsigned a = gid * 2654435761u + 12345u, b = gid ^ 0x9e3779b9u, c = gid + 77u;
  unsigned long long d = gid * 1000003ull + 1ull;
  float x = 0.37f * (float)(gid & 255u);
  unsigned cv = 0;
  for (int it = 0; it < reps; ++it) {
    a = a * (b | 1u) + 17u;                                   // v_mul_lo_u32
    b = __umulhi(b + 3u, a | 0x10000u) + b;                   // v_mul_hi_u32
    d = d * (unsigned long long)(c | 1u) + a;                 // v_mad_u64_u32
    c += (unsigned)(d >> 32);
    x = x * 1.0009765625f + 0.25f;
    cv += (unsigned)(int)floorf(x) + (unsigned)__float2int_rz(x * 3.f);      // v_floor_f32, v_cvt_i32_f32
    if (x > 60000.f) x -= 60000.f;
  }
  out[gid * 5 + 0] = a;
  out[gid * 5 + 1] = b;
  out[gid * 5 + 2] = (unsigned)d;
  out[gid * 5 + 3] = (unsigned)(d >> 32) ^ c;
  out[gid * 5 + 4] = cv;
}

__global__ __launch_bounds__(256) void regs_kernel(unsigned* __restrict__ bad, int spin) {
  const unsigned lane = threadIdx.x + blockIdx.x * 256u;
  unsigned r[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) r[k] = lane * 2654435761u + (unsigned)k * 0x9e3779b9u;
#pragma unroll
  for (int k = 0; k < 64; ++k) asm volatile("" : "+v"(r[k]));          // keep all 64 values live in VGPRs across the idle loop
  unsigned chain = lane;
  for (int it = 0; it < spin; ++it) chain = chain * 1664525u + 1013904223u;   // dependent chain: time passes, registers idle
  unsigned wrong = 0;
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    asm volatile("" : "+v"(r[k]));
    wrong += r[k] != lane * 2654435761u + (unsigned)k * 0x9e3779b9u;
  }
  bad[lane] = wrong + (chain == 0x12345u ? 1u : 0u) * 0u;
}

static std::atomic<int> g_stop{0};

// REPRO_SAME_PROCESS=1: the co-runner runs in THIS process on its own stream (a helper thread keeps it busy)
static void spin_thread(const char* kind) {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float* sink;
  CK(hipMalloc(&sink, 4096));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const unsigned grid = (unsigned)p.multiProcessorCount * 3u;
  while (!g_stop.load()) {
    for (int i = 0; i < 4; ++i) {
      if (!strcmp(kind, "bf16_16x16x32")) spin_kernel<0><<<grid, 256, 0, st>>>(sink, 4000);
      else if (!strcmp(kind, "f16_16x16x32")) spin_kernel<5><<<grid, 256, 0, st>>>(sink, 4000);
      else if (!strcmp(kind, "bf16_32x32x16")) spin_kernel<1><<<grid, 256, 0, st>>>(sink, 2000);
      else if (!strcmp(kind, "f16_32x32x16")) spin_kernel<2><<<grid, 256, 0, st>>>(sink, 2000);
      else if (!strcmp(kind, "f32_32x32x2")) spin_kernel<3><<<grid, 256, 0, st>>>(sink, 1000);
      else spin_kernel<4><<<grid, 256, 0, st>>>(sink, 4000);
    }
    CK(hipStreamSynchronize(st));
  }
}

static void run_spinner(const char* kind, int readyfd) {
  const char* m = getenv("REPRO_SPIN_CU_MASK");
  if (m) setenv("HSA_CU_MASK", m, 1);
  float* sink;
  CK(hipMalloc(&sink, 4096));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const unsigned grid = (unsigned)p.multiProcessorCount * 3u;
  auto launch = [&]() {
    if (!strcmp(kind, "bf16_16x16x32")) spin_kernel<0><<<grid, 256>>>(sink, 4000);
    else if (!strcmp(kind, "f16_16x16x32")) spin_kernel<5><<<grid, 256>>>(sink, 4000);
    else if (!strcmp(kind, "bf16_32x32x16")) spin_kernel<1><<<grid, 256>>>(sink, 2000);
    else if (!strcmp(kind, "f16_32x32x16")) spin_kernel<2><<<grid, 256>>>(sink, 2000);
    else if (!strcmp(kind, "f32_32x32x2")) spin_kernel<3><<<grid, 256>>>(sink, 1000);
    else spin_kernel<4><<<grid, 256>>>(sink, 4000);
  };
  launch();
  CK(hipDeviceSynchronize());
  char c = 'r';
  if (write(readyfd, &c, 1) != 1) exit(3);
  for (;;) {                                           // until the parent kills us
    for (int i = 0; i < 8; ++i) launch();
    CK(hipDeviceSynchronize());
  }
}

int main(int argc, char** argv) {
  const char* kind = argc > 1 ? argv[1] : "bf16_16x16x32";
  const int launches = argc > 2 ? atoi(argv[2]) : 2000;
  const char* victim = argc > 3 ? argv[3] : "both";
  int pfd[2];
  if (pipe(pfd)) return 3;
  pid_t child = -1;
  const bool same_process = getenv("REPRO_SAME_PROCESS") && getenv("REPRO_SAME_PROCESS")[0] == '1';
  const bool spin = strcmp(kind, "none") != 0 && !same_process;
  // first the reference outputs, alone on the GPU -- but HIP must not be initialised before fork(): the parent forks a
  // helper that WAITS for a go signal before it starts spinning
  int gofd[2];
  if (pipe(gofd)) return 3;
  if (spin) {
    child = fork();
    if (child == 0) {
      char c;
      alarm(300);                                      // never outlive the experiment
      if (read(gofd[0], &c, 1) != 1) exit(3);
      run_spinner(kind, pfd[1]);
      exit(0);
    }
  }
  const char* vm = getenv("REPRO_VICTIM_CU_MASK");
  if (vm) setenv("HSA_CU_MASK", vm, 1);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int rt = 0, drv = 0;
  CK(hipRuntimeGetVersion(&rt));
  CK(hipDriverGetVersion(&drv));
  const int n = 618240 / 4;                           // threads of one gather launch
  const int rows = 2 * 1024 * 1024;                    // 32 MB table of 16-byte rows
  std::vector<f32x4> h_table(rows);
  std::vector<int> h_idx((size_t)n * 8);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  for (int i = 0; i < rows; ++i) h_table[i] = f32x4{(float)(rnd() & 1023), (float)(rnd() & 1023), (float)(rnd() & 1023), (float)i};
  for (size_t i = 0; i < h_idx.size(); ++i) h_idx[i] = (int)((i / 8 * 37 + rnd() % 4096) % rows);   // neighbours share rows
  f32x4 *d_table, *d_out;
  int* d_idx;
  unsigned* d_bad;
  const int regs_blocks = prop.multiProcessorCount * 4;
  CK(hipMalloc(&d_table, (size_t)rows * 16));
  CK(hipMalloc(&d_idx, h_idx.size() * 4));
  CK(hipMalloc(&d_out, (size_t)n * 16));
  CK(hipMalloc(&d_bad, (size_t)regs_blocks * 256 * 4));
  CK(hipMemcpy(d_table, h_table.data(), (size_t)rows * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, h_idx.data(), h_idx.size() * 4, hipMemcpyHostToDevice));
  std::vector<f32x4> ref(n), got(n);
  std::vector<unsigned> bad((size_t)regs_blocks * 256);
  gather_kernel<<<(n + 255) / 256, 256>>>(d_table, d_idx, d_out, n);
  CK(hipMemcpy(ref.data(), d_out, (size_t)n * 16, hipMemcpyDeviceToHost));
  // victim "msda": pvsg_msda_fused_forward on the probe's tiny shape (4 frames, levels 8x12 / 4x6 / 2x3, static inputs)
  const int mB = 4, mS = 8 * 12 + 4 * 6 + 2 * 3, mW = 544;
  std::vector<float> h_y((size_t)mB * mS * mW), h_pos((size_t)mS * 288), h_ref((size_t)mS * 2);
  auto frand = [&]() { return ((rnd() & 0xffff) / 32768.f - 1.f) * 1.7f; };
  for (auto& v : h_y) v = frand();
  for (auto& v : h_pos) v = frand();
  for (auto& v : h_ref) v = (rnd() & 0xffff) / 65536.f;
  const int64_t h_shapes[6] = {8, 12, 4, 6, 2, 3}, h_lsi[3] = {0, 96, 120};
  float *d_y, *d_pos, *d_ref, *d_mo;
  int64_t *d_shapes, *d_lsi;
  CK(hipMalloc(&d_y, h_y.size() * 4)); CK(hipMalloc(&d_pos, h_pos.size() * 4)); CK(hipMalloc(&d_ref, h_ref.size() * 4));
  CK(hipMalloc(&d_mo, (size_t)mB * mS * 256 * 4)); CK(hipMalloc(&d_shapes, 48)); CK(hipMalloc(&d_lsi, 24));
  CK(hipMemcpy(d_y, h_y.data(), h_y.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_pos, h_pos.data(), h_pos.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ref, h_ref.data(), h_ref.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_shapes, h_shapes, 48, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lsi, h_lsi, 24, hipMemcpyHostToDevice));
  std::vector<float> m_ref((size_t)mB * mS * 256), m_got(m_ref.size());
  const int vflags = !strncmp(victim, "msdav", 5) ? atoi(victim + 5) : -1;
  auto msda = [&]() {
    if (vflags >= 0) {
      const long long nq = (long long)mB * mS;
      const unsigned g1 = (unsigned)((nq + 3) / 4), g8 = (unsigned)nq;
#define VLAUNCH(F) msda_variant_kernel<F><<<((F) & 8) ? g8 : g1, ((F) & 8) ? 64 : 256>>>(d_y, mW, d_y + 256, mW, d_pos, d_ref, (const long long*)d_shapes, (const long long*)d_lsi, d_mo, mS, mS, nq)
      switch (vflags) {
        case 0: VLAUNCH(0); break; case 1: VLAUNCH(1); break; case 2: VLAUNCH(2); break; case 3: VLAUNCH(3); break;
        case 4: VLAUNCH(4); break; case 5: VLAUNCH(5); break; case 6: VLAUNCH(6); break; case 7: VLAUNCH(7); break;
        case 8: VLAUNCH(8); break; case 15: VLAUNCH(15); break; default: fprintf(stderr, "no such variant\n"); exit(5);
      }
#undef VLAUNCH
      return;
    }
    const int rc = pvsg_msda_fused_forward(d_y, mW, d_y + 256, mW, d_pos, d_ref, d_shapes, d_lsi, d_mo, mB, mS, 8, 32, mS, 3, 4, nullptr);
    if (rc) { fprintf(stderr, "msda: %s\n", pvsg::err_buf()); exit(4); }
  };
  const bool do_msda = !strcmp(victim, "msda") || !strcmp(victim, "all") || vflags >= 0;
  long long m_bad_launches = 0, m_bad_words = 0;
  long long m_lane_hist[64] = {0};
  msda();
  CK(hipMemcpy(m_ref.data(), d_mo, m_ref.size() * 4, hipMemcpyDeviceToHost));
  float* d_tr0;
  CK(hipMalloc(&d_tr0, (size_t)prop.multiProcessorCount * 4 * 256 * 5 * 4));
  if (!strcmp(victim, "imul")) imul_kernel<<<prop.multiProcessorCount * 4, 256>>>(reinterpret_cast<unsigned*>(d_tr0), 400);
  else if (!strcmp(victim, "misc")) misc_kernel<<<prop.multiProcessorCount * 4, 256>>>(d_tr0, 400);
  else trans_kernel<<<prop.multiProcessorCount * 4, 256>>>(d_tr0, 400);
  std::vector<float> t_ref0((size_t)prop.multiProcessorCount * 4 * 256 * 5);
  CK(hipMemcpy(t_ref0.data(), d_tr0, t_ref0.size() * 4, hipMemcpyDeviceToHost));
  std::thread helper;
  if (same_process && strcmp(kind, "none") != 0) {
    helper = std::thread(spin_thread, kind);
    usleep(200000);
  }
  if (spin) {
    char c = 'g';
    if (write(gofd[1], &c, 1) != 1) return 3;
    if (read(pfd[0], &c, 1) != 1) return 3;            // the co-runner is up and has completed a launch
  }
  const bool do_trans = !strcmp(victim, "trans") || !strcmp(victim, "imul") || !strcmp(victim, "misc") || !strcmp(victim, "all");
  const bool imul = !strcmp(victim, "imul"), misc = !strcmp(victim, "misc");
  const int t_blocks = prop.multiProcessorCount * 4;
  float* d_tr;
  CK(hipMalloc(&d_tr, (size_t)t_blocks * 256 * 5 * 4));
  std::vector<float> t_ref((size_t)t_blocks * 256 * 5), t_got(t_ref.size());
  long long t_bad_launches = 0, t_bad[5] = {0, 0, 0, 0, 0}, t_lane_hist[64] = {0};
  const bool do_gather = !strcmp(victim, "gather") || !strcmp(victim, "both") || !strcmp(victim, "all");
  const bool do_regs = !strcmp(victim, "regs") || !strcmp(victim, "both") || !strcmp(victim, "all");
  long long g_bad_launches = 0, g_bad_words = 0, r_bad_launches = 0, r_bad_lanes = 0;
  long long lane_hist[64] = {0};
  for (int it = 0; it < launches; ++it) {
    if (do_gather) {
      gather_kernel<<<(n + 255) / 256, 256>>>(d_table, d_idx, d_out, n);
      CK(hipMemcpy(got.data(), d_out, (size_t)n * 16, hipMemcpyDeviceToHost));
      long long w = 0;
      for (int i = 0; i < n; ++i)
        for (int e = 0; e < 4; ++e)
          if (got[i][e] != ref[i][e]) { ++w; ++lane_hist[i & 63]; }
      g_bad_words += w;
      g_bad_launches += w != 0;
    }
    if (do_msda) {
      msda();
      CK(hipMemcpy(m_got.data(), d_mo, m_got.size() * 4, hipMemcpyDeviceToHost));
      long long w = 0;
      for (size_t i = 0; i < m_got.size(); ++i)
        if (m_got[i] != m_ref[i]) { ++w; ++m_lane_hist[(i / 4) & 63]; }
      m_bad_words += w;
      m_bad_launches += w != 0;
    }
    if (do_trans) {
      if (imul) imul_kernel<<<t_blocks, 256>>>(reinterpret_cast<unsigned*>(d_tr), 400);
      else if (misc) misc_kernel<<<t_blocks, 256>>>(d_tr, 400);
      else trans_kernel<<<t_blocks, 256>>>(d_tr, 400);
      CK(hipMemcpy(t_got.data(), d_tr, t_got.size() * 4, hipMemcpyDeviceToHost));
      long long w = 0;
      for (size_t i = 0; i < t_got.size(); ++i)
        if (memcmp(&t_got[i], &t_ref0[i], 4)) { ++w; ++t_bad[i % 5]; ++t_lane_hist[(i / 5) & 63]; }
      t_bad_launches += w != 0;
    }
    if (do_regs) {
      regs_kernel<<<regs_blocks, 256>>>(d_bad, 20000);
      CK(hipMemcpy(bad.data(), d_bad, bad.size() * 4, hipMemcpyDeviceToHost));
      long long w = 0;
      for (unsigned v : bad) w += v != 0;
      r_bad_lanes += w;
      r_bad_launches += w != 0;
    }
  }
  if (helper.joinable()) {
    g_stop.store(1);
    helper.join();
  }
  if (child > 0) {
    kill(child, SIGKILL);
    int st;
    waitpid(child, &st, 0);
  }
  printf("{\"corunner\": \"%s\", \"corunner_in\": \"%s\", \"launches\": %d, \"victim_cu_mask\": \"%s\", \"corunner_cu_mask\": \"%s\", "
         "\"msda_bad_launches\": %lld, \"msda_bad_words\": %lld, "
         "\"gather_bad_launches\": %lld, \"gather_bad_words\": %lld, \"regs_bad_launches\": %lld, \"regs_bad_lanes\": %lld, "
         "\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"hip_runtime\": %d, \"hip_driver\": %d, \"xnack_env\": \"%s\", "
         "\"gather_bad_lanes_histogram\": [",
         kind, same_process ? "same process, second stream" : "other process", launches, vm ? vm : "", getenv("REPRO_SPIN_CU_MASK") ? getenv("REPRO_SPIN_CU_MASK") : "", m_bad_launches, m_bad_words, g_bad_launches, g_bad_words,
         r_bad_launches, r_bad_lanes, prop.name, prop.gcnArchName, prop.multiProcessorCount, rt, drv,
         getenv("HSA_XNACK") ? getenv("HSA_XNACK") : "");
  for (int i = 0; i < 64; ++i) printf("%lld%s", lane_hist[i], i < 63 ? ", " : "");
  printf("], \"trans_bad_launches\": %lld, \"trans_bad_words_exp_rcp_sqrt_log_fma\": [%lld, %lld, %lld, %lld, %lld], \"trans_bad_lanes_histogram\": [",
         t_bad_launches, t_bad[0], t_bad[1], t_bad[2], t_bad[3], t_bad[4]);
  for (int i = 0; i < 64; ++i) printf("%lld%s", t_lane_hist[i], i < 63 ? ", " : "");
  printf("], \"msda_bad_lanes_histogram\": [");
  for (int i = 0; i < 64; ++i) printf("%lld%s", m_lane_hist[i], i < 63 ? ", " : "");
  printf("]}\n");
  return 0;
}
