// graph_gap_bench.hip — what a dependent kernel costs on a stream and inside a HIP graph on gfx950:
// chains of three tiny dependent kernels (each reads what its predecessor wrote), 300 rounds, as
// plain launches, as a three-node graph launched 300 times, and as one graph of 900 nodes.
// build: hipcc -O2 --offload-arch=gfx950 tools/graph_gap_bench.hip -o gpurun_out/graph_gap_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(e)                                                                    \
  do {                                                                              \
    hipError_t err__ = (e);                                                         \
    if (err__ != hipSuccess) {                                                      \
      std::printf("%s failed: %s\n", #e, hipGetErrorString(err__));                 \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

__global__ void step_kernel(const int* __restrict__ in, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + 1;
}

int main() {
  const int n = 64 * 256, rounds = 300;
  int *a = nullptr, *b = nullptr;
  CHECK(hipMalloc((void**)&a, sizeof(int) * n));
  CHECK(hipMalloc((void**)&b, sizeof(int) * n));
  CHECK(hipMemset(a, 0, sizeof(int) * n));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto chain = [&](hipStream_t s) {
    hipLaunchKernelGGL(step_kernel, dim3(n / 256), dim3(256), 0, s, a, b);
    hipLaunchKernelGGL(step_kernel, dim3(n / 256), dim3(256), 0, s, b, a);
    hipLaunchKernelGGL(step_kernel, dim3(n / 256), dim3(256), 0, s, a, b);
  };
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    // (a) plain launches
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < rounds; ++r) chain(st);
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("plain launches:        %.2f us per kernel\n", 1e3 * ms / (3 * rounds));
    // (b) a three-node graph, launched once per round
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    chain(st);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < rounds; ++r) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("three-node graph x %d: %.2f us per kernel\n", rounds, 1e3 * ms / (3 * rounds));
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
    // (c) one graph of all rounds
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int r = 0; r < rounds; ++r) chain(st);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("one graph of %d nodes:  %.2f us per kernel\n", 3 * rounds, 1e3 * ms / (3 * rounds));
    CHECK(hipGraphExecDestroy(ge));
    CHECK(hipGraphDestroy(g));
  }
  return 0;
}
