// How many workgroups with 47 KB of LDS does a CU hold at once?  Prints the device's LDS sizes, the
// runtime's occupancy answer, and measures it: every workgroup records when it started and ended.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int KB>
__global__ __launch_bounds__(256) void hold(long long* t, long long ticks) {
  __shared__ double s[KB * 128];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) t[2 * blockIdx.x] = t0, t[2 * blockIdx.x + 1] = wall_clock64() + (long long)s[5];
}
template <int KB>
void run(const char* name) {
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hold<KB>, 256, 0);
  const int blocks = 256 * 8;
  long long* d;
  hipMalloc(&d, sizeof(long long) * 2 * blocks);
  hipLaunchKernelGGL(hold<KB>, dim3(blocks), dim3(256), 0, 0, d, 2000);  // 20 us each
  hipDeviceSynchronize();
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), d, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  long long first = h[0], last = 0;
  for (int b = 0; b < blocks; ++b) first = std::min(first, h[2 * b]), last = std::max(last, h[2 * b + 1]);
  // concurrent at the moment the first block ends
  long long t_probe = first + 1000;
  int conc = 0;
  for (int b = 0; b < blocks; ++b) conc += h[2 * b] <= t_probe && h[2 * b + 1] > t_probe;
  printf("%s: %d KB LDS per workgroup: runtime says %d blocks/CU; measured %d blocks in flight (%.2f per CU), %d blocks took %.1f us\n",
         name, KB, nb, conc, conc / 256.0, blocks, (last - first) / 100.0);
  hipFree(d);
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.gcnArchName,
         p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  run<47>("a"); run<31>("b"); run<20>("c"); run<15>("d"); run<1>("e");
  return 0;
}
