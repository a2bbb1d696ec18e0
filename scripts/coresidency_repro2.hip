// Round 4 companion of scripts/coresidency_repro.hip: the two kernels that WILL share a GPU with the split GEMMs if a rank of
// the 8-GPU layout ever overlaps its per-layer exchange with compute -- as victims next to a register-only 16-bit MFMA loop:
//   allgather   RCCL ncclAllGather of one 108.8 KB attention record (one-rank communicator: the copy kernel RCCL launches)
//   combine     pvsg_xattn_combine_packed (the product's merge of the gathered records; linked from libopenpvsg_hip.so)
// co-runner (other process, or REPRO_SAME_PROCESS=1: a second stream of this process): bf16_16x16x32 | f16_16x16x32 | none
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/coresidency_repro2 scripts/coresidency_repro2.hip \
//         -I/opt/rocm/include -L/opt/rocm/lib -lrccl -Lopenpvsg_amd/lib -lopenpvsg_hip -Wl,-rpath,$PWD/openpvsg_amd/lib -lpthread
//   ./scripts/coresidency_repro2 <co-runner> <launches> <victim>        -> one JSON line
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)
extern "C" int pvsg_xattn_combine_packed(const float* packed, float* out, int R, int B, int Q, int M, int D, hipStream_t stream);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void spin_kernel(float* sink, int iters) {
  const int t = threadIdx.x;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (KIND == 0) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + 0.01f * (t + i)); b[i] = (__bf16)(1.0f - 0.02f * (t - i)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u & 3], 0, 0, 0);
  } else {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.01f * (t + i)); b[i] = (_Float16)(1.0f - 0.02f * (t - i)); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
  }
  const float total = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
  if (total == 12345.678f) sink[0] = total;
}

static std::atomic<bool> g_stop{false};
static void spin_forever(const char* kind, hipStream_t st, int readyfd) {
  float* sink;
  CK(hipMalloc(&sink, 4096));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const unsigned grid = (unsigned)p.multiProcessorCount * 3u;
  bool first = true;
  while (!g_stop.load()) {
    for (int i = 0; i < 4; ++i) {
      if (!strcmp(kind, "bf16_16x16x32")) spin_kernel<0><<<grid, 256, 0, st>>>(sink, 4000);
      else spin_kernel<1><<<grid, 256, 0, st>>>(sink, 4000);
    }
    CK(hipStreamSynchronize(st));
    if (first && readyfd >= 0) { char c = 'r'; if (write(readyfd, &c, 1) != 1) exit(3); first = false; }
  }
}

int main(int argc, char** argv) {
  const char* kind = argc > 1 ? argv[1] : "bf16_16x16x32";
  const int launches = argc > 2 ? atoi(argv[2]) : 400;
  const char* victim = argc > 3 ? argv[3] : "combine";
  const bool same = getenv("REPRO_SAME_PROCESS") && getenv("REPRO_SAME_PROCESS")[0] == '1';
  const bool spin = strcmp(kind, "none") != 0;
  int pfd[2], gofd[2];
  if (pipe(pfd) || pipe(gofd)) return 3;
  pid_t child = -1;
  if (spin && !same) {                                 // HIP must not be initialised before fork()
    child = fork();
    if (child == 0) {
      char c;
      alarm(300);
      if (read(gofd[0], &c, 1) != 1) exit(3);
      const char* m = getenv("REPRO_SPIN_CU_MASK");
      if (m) setenv("HSA_CU_MASK", m, 1);
      spin_forever(kind, nullptr, pfd[1]);
      exit(0);
    }
  }
  const char* vm = getenv("REPRO_VICTIM_CU_MASK");
  if (vm) setenv("HSA_CU_MASK", vm, 1);
  const int REC = 8 * 100 * 34 + 4, R = 8;
  std::vector<float> h((size_t)R * REC);
  uint32_t s = 777u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 9) / 8388608.f + 0.01f; }
  for (int r = 0; r < R; ++r) for (int k = 0; k < 4; ++k) h[(size_t)r * REC + REC - 4 + k] = 0.f;     // flag words: nothing blocked everywhere
  float *d_in, *d_out;
  CK(hipMalloc(&d_in, h.size() * 4));
  CK(hipMalloc(&d_out, (size_t)R * REC * 4));
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipStream_t vs;
  CK(hipStreamCreate(&vs));
  ncclComm_t comm = nullptr;
  const bool ag = !strcmp(victim, "allgather");
  if (ag) {
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { fprintf(stderr, "rccl init failed\n"); return 4; }
  }
  const size_t out_n = ag ? (size_t)REC : (size_t)100 * 256;
  auto run = [&]() {
    if (ag) { if (ncclAllGather(d_in, d_out, REC, ncclFloat, comm, vs) != ncclSuccess) { fprintf(stderr, "allgather failed\n"); exit(4); } }
    else if (pvsg_xattn_combine_packed(d_in, d_out, R, 1, 100, 8, 32, vs)) { fprintf(stderr, "combine failed\n"); exit(4); }
    CK(hipStreamSynchronize(vs));
  };
  std::vector<float> ref(out_n), got(out_n);
  run();
  CK(hipMemcpy(ref.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
  std::thread helper;
  hipStream_t ss = nullptr;
  if (spin && same) {
    CK(hipStreamCreate(&ss));
    helper = std::thread(spin_forever, kind, ss, -1);
    usleep(200000);
  } else if (spin) {
    char c = 'g';
    if (write(gofd[1], &c, 1) != 1 || read(pfd[0], &c, 1) != 1) return 3;
  }
  long long bad_launches = 0, bad_words = 0;
  for (int i = 0; i < launches; ++i) {
    CK(hipMemsetAsync(d_out, 0xff, out_n * 4, vs));
    run();
    CK(hipMemcpy(got.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
    long long b = 0;
    for (size_t k = 0; k < out_n; ++k) b += memcmp(&got[k], &ref[k], 4) != 0;
    bad_launches += b != 0;
    bad_words += b;
  }
  g_stop.store(true);
  if (helper.joinable()) helper.join();
  if (child > 0) { kill(child, SIGKILL); waitpid(child, nullptr, 0); }
  printf("{\"victim\": \"%s\", \"corunner\": \"%s\", \"corunner_in\": \"%s\", \"launches\": %d, \"bad_launches\": %lld, \"bad_words\": %lld, "
         "\"victim_cu_mask\": \"%s\", \"corunner_cu_mask\": \"%s\"}\n", ag ? "rccl all_gather (1 rank, 108.8 KB record)" : "pvsg_xattn_combine_packed (8 records)",
         kind, !spin ? "-" : same ? "same process, second stream" : "other process", launches, bad_launches, bad_words,
         vm ? vm : "", getenv("REPRO_SPIN_CU_MASK") ? getenv("REPRO_SPIN_CU_MASK") : "");
  return 0;
}
