// teaser_graph.hip — the graph half of the TEASERPP initialisation on the device
// (IcpLoopComputation::GetTeaserInitialAlignment, loop_closure/src/IcpLoopComputation.cc:996-1054 ->
// teaser::RobustRegistrationSolver with PMC_HEU inlier selection): core numbers of the
// scale-consistency graph and the greedy clique heuristic of teaser.h, on the bit matrix the
// consistency_graph_kernel left in device memory (it never travels to the host).
//
// Core numbers: the H-operator iteration (Lu, Zhou, Zhang, Stanley 2016: starting from the degrees,
// c(v) <- H-index of the neighbours' values converges to the core numbers).  Every round is one
// fully parallel launch (one wave per vertex, the previous round's values in LDS) and a few tens of
// rounds suffice on these graphs, where level-by-level peeling pays a grid-wide step for each of
// the thousands of distinct core values and bucket peeling is sequential.  Core numbers are a
// property of the graph, so the result equals the host's bucket peeling (teaser::core_bounds)
// entry by entry.
//
// clique_starts_kernel: one wave per start of the greedy heuristic (teaser::greedy_clique's
// start_from): candidates = neighbours of the start vertex whose bound exceeds the best size so
// far, kept as a bit set in LDS; repeatedly the first candidate in preference order (bound, then
// id, descending) joins the clique and the set is intersected with its row; the start gives up
// as soon as clique + candidates cannot beat the best size.  Starts are independent given that
// size, so a window of them runs side by side; the host applies the heuristic's serial rule to
// the window's results (a start that improves the size ends the window, the starts after it are
// evaluated again against the new size) and obtains the serial algorithm's clique bit for bit.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include "lgicp_internal.h"

namespace lgicp {

// degree of every vertex: one wave per row of the bit matrix
__global__ __launch_bounds__(256) void graph_degree_kernel(const unsigned long long* __restrict__ bits, int n,
                                                           int words, int* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n) return;  // wave-uniform
  const unsigned long long* r = bits + (size_t)v * words;
  int d = 0;
  for (int w = lane; w < words; w += 64) d += __builtin_popcountll(r[w]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
  if (lane == 0) deg[v] = d;
}

// One round of the H-operator: c_new(v) = the largest h such that at least h neighbours u of v
// have c_old(u) >= h (never more than c_old(v)).  The values of a round come from the round
// before alone, so the rounds are deterministic; `*changed` is raised when any value moved.
// dynamic LDS: c_old as n unsigned shorts (n <= 65536, so every value fits).
__global__ __launch_bounds__(256) void core_hindex_kernel(const unsigned long long* __restrict__ bits, int n,
                                                          int words, const int* __restrict__ c_old,
                                                          int* __restrict__ c_new, int* __restrict__ changed) {
  extern __shared__ unsigned short cs[];
  for (int i = threadIdx.x; i < n; i += 256) cs[i] = (unsigned short)c_old[i];
  if (threadIdx.x == 0) cs[n] = 0;  // sentinel for missing neighbours
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int v = blockIdx.x * 4 + wave; v < n; v += gridDim.x * 4) {  // wave-uniform
    const int cv = (int)cs[v];
    const unsigned long long* r = bits + (size_t)v * words;
    // neighbours whose value is at least h.  Each lane walks the bits of its words rotated by
    // its lane number: in a dense row the lanes would otherwise all sit at the same bit of 64
    // consecutive words, 64 values that share one LDS bank.
    auto count_ge = [&](int h) {  // h >= 1
      int c = 0;
      for (int w = lane; w < words; w += 64) {
        unsigned long long m = r[w];
        m = (m >> lane) | (m << ((64 - lane) & 63));
        const int base = w << 6;
        while (m) {
          // four neighbours per trip, so that four LDS reads are in flight at a time; a missing
          // one reads the sentinel cs[n] = 0, which never reaches h
          int u[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            u[q] = m ? base + ((__builtin_ctzll(m) + lane) & 63) : n;
            m &= m - 1;
          }
          const int a0 = cs[u[0]], a1 = cs[u[1]], a2 = cs[u[2]], a3 = cs[u[3]];
          c += (a0 >= h) + (a1 >= h) + (a2 >= h) + (a3 >= h);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
      return c;
    };
    int h = cv;
    if (cv > 0 && count_ge(cv) < cv) {
      // the new value lies below cv, usually just below: gallop down from the top, then bisect
      int hi = cv - 1, lo = 0, step = 1;  // count_ge(lo) >= lo holds (lo = 0 trivially)
      for (;;) {
        const int probe = hi - step + 1;
        if (probe <= 0) break;
        if (count_ge(probe) >= probe) {
          lo = probe;
          break;
        }
        hi = probe - 1;
        step <<= 1;
      }
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (count_ge(mid) >= mid) lo = mid; else hi = mid - 1;
      }
      h = lo;
    }
    if (lane == 0) {
      c_new[v] = h;
      if (h != cv) *changed = 1;
    }
  }
}

// K = core number + 1: the bound on the size of a clique through a vertex
__global__ __launch_bounds__(256) void core_bound_kernel(const int* __restrict__ c, int n, int* __restrict__ K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) K[i] = c[i] + 1;
}

// elig bit u = K[u] > mc
__global__ __launch_bounds__(256) void clique_eligible_kernel(const int* __restrict__ K, int n, int words,
                                                              int mc, unsigned long long* __restrict__ elig) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= words) return;
  unsigned long long m = 0;
  for (int b = 0; b < 64; ++b) {
    const int u = w * 64 + b;
    if (u < n && K[u] > mc) m |= 1ull << b;
  }
  elig[w] = m;
}

constexpr int CLQ_MAX_WORDS = 1024;  // 65536 matches

// one wave per start; out_size[s] = members found besides the start vertex (0: cannot beat mc);
// out_list (optional, single-start launches only) receives those members in the order they joined
__global__ __launch_bounds__(64) void clique_starts_kernel(
    const unsigned long long* __restrict__ bits, int n, int words, const int* __restrict__ pref,
    const unsigned long long* __restrict__ elig, const int* __restrict__ starts, int mc,
    int* __restrict__ out_size, int* __restrict__ out_list) {
  __shared__ unsigned long long cand[CLQ_MAX_WORDS];
  const int s = blockIdx.x, lane = threadIdx.x;
  const int v = starts[s];
  const unsigned long long* rv = bits + (size_t)v * words;
  int np = 0;
  for (int w = lane; w < words; w += 64) {
    const unsigned long long x = rv[w] & elig[w];
    cand[w] = x;
    np += __builtin_popcountll(x);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) np += __shfl_xor(np, off);
  __syncthreads();
  int size = 0;
  if (np > mc) {
    int* const list = out_list;
    int at = 0;
    while (at < n) {  // every pass advances `at`: at most n / 64 + clique size passes
      // the first candidate in preference order from `at` on
      const int idx = at + lane;
      const int u = idx < n ? pref[idx] : -1;
      const bool hit = u >= 0 && ((cand[u >> 6] >> (u & 63)) & 1ull);
      const unsigned long long m = __ballot(hit);
      if (m == 0) {
        at += 64;
        continue;
      }
      const int first = __builtin_ctzll(m);
      const int pick = __builtin_amdgcn_readlane(u, first);
      at += first + 1;
      if (list && lane == 0) list[size] = pick;
      ++size;
      const unsigned long long* r = bits + (size_t)pick * words;
      int left = 0;
      __syncthreads();  // all lanes have read cand for the ballot above
      for (int w = lane; w < words; w += 64) {
        const unsigned long long x = cand[w] & r[w];
        cand[w] = x;
        left += __builtin_popcountll(x);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) left += __shfl_xor(left, off);
      __syncthreads();
      // the start vertex, the members so far and every remaining candidate together cannot beat
      // the best size: whatever this start ends with would be discarded
      if (1 + size + left <= mc) {
        size = 0;
        break;
      }
      if (left == 0) break;
    }
  }
  if (lane == 0) out_size[s] = size;
}

}  // namespace lgicp

namespace lgicp_detail {

namespace {
// buffers of the clique search, one set per host thread and device, reused from pair to pair
struct CliqueSearchDev {
  int device = -1;
  DevBuf<int> K, pref, starts, sizes, lists, cbuf, flags;
  DevBuf<unsigned long long> elig;
  void release() {
    if (device >= 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != device) (void)hipSetDevice(device);
      K.release(), pref.release(), starts.release(), sizes.release(), lists.release(), cbuf.release();
      flags.release();
      elig.release();
      if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    device = -1;
  }
  ~CliqueSearchDev() { release(); }
};
constexpr int kMaxWindow = 65536;
}  // namespace

// largest graph the device path takes: the H-operator kernel keeps one unsigned short per vertex
// in LDS (128 KiB at 65536 vertices, the size the consistency graph is capped at anyway)
int clique_device_max_vertices() { return 65536; }

// teaser::greedy_clique on a bit matrix in device memory.  Returns false when the device
// refuses; *clique comes back sorted ascending, equal to the host function's result.
bool greedy_clique_device(int device, hipStream_t st, const unsigned long long* d_bits, int n, int words,
                          std::vector<int>* clique) {
  clique->clear();
  if (n <= 0) return true;
  if (n > clique_device_max_vertices() || words > lgicp::CLQ_MAX_WORDS) return false;
  thread_local CliqueSearchDev B;
  if (B.device != device) {
    B.release();
    if (hipSetDevice(device) != hipSuccess) return false;
    B.device = device;
  }
  if (B.K.reserve((size_t)n) != hipSuccess || B.pref.reserve((size_t)n) != hipSuccess ||
      B.starts.reserve(kMaxWindow) != hipSuccess ||
      B.sizes.reserve(kMaxWindow) != hipSuccess || B.elig.reserve((size_t)words) != hipSuccess ||
      B.lists.reserve((size_t)n) != hipSuccess)
    return false;
  const bool trace = std::getenv("LGICP_TRACE") != nullptr;
  const auto t_0 = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
  };
  // ---- core numbers ---------------------------------------------------------------------------
  const size_t lds_bytes = sizeof(unsigned short) * ((size_t)n + 2);
  static std::once_flag attr_once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(attr_once, [] {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(lgicp::core_hindex_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 64);
  });
  if (attr_rc != hipSuccess) return false;
  constexpr int kRoundsPerCheck = 8;
  if (B.cbuf.reserve((size_t)2 * n) != hipSuccess || B.flags.reserve(kRoundsPerCheck) != hipSuccess) return false;
  int* c_a = B.cbuf.p;
  int* c_b = B.cbuf.p + n;
  hipLaunchKernelGGL(lgicp::graph_degree_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d_bits, n, words, c_a);
  const int hx_blocks = std::max(1, std::min((n + 3) / 4, 2048));
  int n_rounds = 0;
  for (;;) {  // every round lowers at least one value or ends the loop: at most n * n rounds
    if (hipMemsetAsync(B.flags.p, 0, sizeof(int) * kRoundsPerCheck, st) != hipSuccess) return false;
    for (int r = 0; r < kRoundsPerCheck; ++r) {
      hipLaunchKernelGGL(lgicp::core_hindex_kernel, dim3(hx_blocks), dim3(256), lds_bytes, st, d_bits, n, words,
                         c_a, c_b, B.flags.p + r);
      std::swap(c_a, c_b);
    }
    n_rounds += kRoundsPerCheck;
    int flags[kRoundsPerCheck];
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(flags, B.flags.p, sizeof(flags), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    if (flags[kRoundsPerCheck - 1] == 0) break;  // the last round moved nothing: fixed point
    if (n_rounds > 4 * n + 64) return false;
  }
  hipLaunchKernelGGL(lgicp::core_bound_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c_a, n, B.K.p);
  if (hipGetLastError() != hipSuccess) return false;
  std::vector<int> K((size_t)n);
  if (hipMemcpyAsync(K.data(), B.K.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return false;
  const double ms_core = ms_since(t_0);
  const auto t_1 = std::chrono::steady_clock::now();
  // ---- orders (teaser::greedy_clique): starts by decreasing bound, ascending id among equals;
  // candidates by decreasing bound, then decreasing id
  std::vector<int> order((size_t)n), pref((size_t)n);
  for (int i = 0; i < n; ++i) order[i] = pref[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return K[a] > K[b]; });
  std::sort(pref.begin(), pref.end(), [&](int a, int b) { return K[a] != K[b] ? K[a] > K[b] : a > b; });
  if (hipMemcpyAsync(B.pref.p, pref.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)
    return false;
  // ---- windows of starts ------------------------------------------------------------------------
  const double ms_sort = ms_since(t_1);
  const auto t_2 = std::chrono::steady_clock::now();
  int n_windows = 0, n_starts = 0;
  std::vector<int> best, window, window_pos, sizes;
  int mc = 0, at = 0, win = 1;
  while (at < n) {
    window.clear(), window_pos.clear();
    int scan = at;
    while (scan < n && (int)window.size() < win) {
      const int v = order[scan++];
      if (K[v] > mc) window.push_back(v), window_pos.push_back(scan);
    }
    if (window.empty()) break;
    const int S = (int)window.size();
    ++n_windows, n_starts += S;
    sizes.assign((size_t)S, 0);
    if (hipMemcpyAsync(B.starts.p, window.data(), sizeof(int) * (size_t)S, hipMemcpyHostToDevice, st) != hipSuccess)
      return false;
    hipLaunchKernelGGL(lgicp::clique_eligible_kernel, dim3((words + 255) / 256), dim3(256), 0, st, B.K.p, n,
                       words, mc, B.elig.p);
    hipLaunchKernelGGL(lgicp::clique_starts_kernel, dim3(S), dim3(64), 0, st, d_bits, n, words, B.pref.p,
                       B.elig.p, B.starts.p, mc, B.sizes.p, (int*)nullptr);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(sizes.data(), B.sizes.p, sizeof(int) * (size_t)S, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    int improved = -1;
    for (int k = 0; k < S; ++k)
      if (sizes[k] > 0 && sizes[k] + 1 > mc) {
        improved = k;
        break;
      }
    if (improved < 0) {
      at = scan;
      win = std::min(win * 16, kMaxWindow);
      continue;
    }
    // the serial rule: this start raises the size; the starts after it see the new size.  Its
    // members come from running it once more on its own (improvements are rare, and the windows
    // need not keep a member list per start this way).
    best.resize((size_t)sizes[improved]);
    hipLaunchKernelGGL(lgicp::clique_starts_kernel, dim3(1), dim3(64), 0, st, d_bits, n, words, B.pref.p,
                       B.elig.p, B.starts.p + improved, mc, B.sizes.p, B.lists.p);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(best.data(), B.lists.p, sizeof(int) * best.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    best.push_back(window[improved]);
    mc = (int)best.size();
    at = window_pos[improved];
    win = std::min(std::max(win, 4) * 16, kMaxWindow);
  }
  if (trace)
    std::fprintf(stderr, "[lgicp] clique search on the device: %d vertices, core numbers %.2f ms (%d rounds), "
                 "orders %.2f ms, %d windows / %d starts %.2f ms, clique %zu\n", n, ms_core, n_rounds, ms_sort, n_windows,
                 n_starts, ms_since(t_2), best.size());
  std::sort(best.begin(), best.end());
  *clique = std::move(best);
  return true;
}

}  // namespace lgicp_detail
