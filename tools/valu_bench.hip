// valu_bench.hip — issue-rate micro-benchmark for the f32 VALU forms the 1-NN kernel uses
// (plain vs packed add/mul/fma, v_min3_f32) on gfx950.  Prints lane-operations per second for
// the whole chip.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define CHAINS 8

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x[CHAINS];
  v2f y[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    x[i] = a + threadIdx.x * 1e-3f + i;
    y[i] = {a + i, b + threadIdx.x * 1e-3f};
  }
  const v2f bb = {b, b * 1.0001f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (MODE == 0) x[i] = x[i] * b;                       // v_mul_f32
      if (MODE == 1) y[i] = y[i] * bb;                      // v_pk_mul_f32
      if (MODE == 2) x[i] = __builtin_fmaf(x[i], b, a);     // v_fma_f32
      if (MODE == 3) y[i] = __builtin_elementwise_fma(y[i], bb, bb);  // v_pk_fma_f32
      if (MODE == 4) x[i] = x[i] + b;                       // v_add_f32
      if (MODE == 5) y[i] = y[i] + bb;                      // v_pk_add_f32
      if (MODE == 6) x[i] = __builtin_fminf(__builtin_fminf(x[i], a + it), b);  // v_min3_f32
    }
  }
  float s = 0;
  for (int i = 0; i < CHAINS; ++i) s += x[i] + y[i][0] + y[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(float* d, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = 5.0 * blocks * 256.0 * ITERS * CHAINS;  // lane-instructions
  return insts / (ms * 1e-3);
}

int main() {
  const int blocks = 256 * 8;
  float* d;
  hipMalloc(&d, sizeof(float) * blocks * 256);
  const char* names[] = {"v_mul_f32", "v_pk_mul_f32", "v_fma_f32", "v_pk_fma_f32",
                         "v_add_f32", "v_pk_add_f32", "v_min3_f32"};
  double r[7];
  r[0] = run<0>(d, blocks);
  r[1] = run<1>(d, blocks);
  r[2] = run<2>(d, blocks);
  r[3] = run<3>(d, blocks);
  r[4] = run<4>(d, blocks);
  r[5] = run<5>(d, blocks);
  r[6] = run<6>(d, blocks);
  for (int i = 0; i < 7; ++i) {
    const double lanes_ops = r[i] * ((i == 1 || i == 3 || i == 5) ? 2.0 : 1.0);
    printf("%-14s %8.2f T lane-instr/s  %8.2f T elem-ops/s\n", names[i], r[i] * 1e-12,
           lanes_ops * 1e-12);
  }
  hipFree(d);
  return 0;
}
