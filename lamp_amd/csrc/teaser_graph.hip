// teaser_graph.hip — the graph half of the TEASERPP initialisation on the device
// (IcpLoopComputation::GetTeaserInitialAlignment, loop_closure/src/IcpLoopComputation.cc:996-1054 ->
// teaser::RobustRegistrationSolver with PMC_HEU inlier selection): core numbers of the
// scale-consistency graph and the greedy clique heuristic of teaser.h, on the bit matrices the
// consistency_graph_kernel left in device memory (they never travel to the host), for a whole
// group of candidate pairs at a time.
//
// Both halves are chains of small dependent steps (a peeling step, a window of clique starts, each
// lasting as long as its slowest wave), so one pair alone leaves the GPU idle; every launch here
// therefore carries all pairs of the group.
//
// Core numbers: peeling, one workgroup per graph with the remaining degrees in registers as bit
// planes (core_peel_group_kernel).  Core numbers are a property of the graph, so the result equals
// the host's bucket peeling (teaser::core_bounds) entry by entry.
//
// clique_starts_group_kernel: one wave per start of the greedy heuristic (teaser::greedy_clique's
// start_from): candidates = neighbours of the start vertex whose bound exceeds the best size so
// far, kept as a bit set in LDS; repeatedly the first candidate in preference order (bound, then
// id, descending) joins the clique and the set is intersected with its row; the start gives up
// as soon as clique + candidates cannot beat the best size.  The heuristic is defined serially (a
// start sees the best size its predecessors left), but a start's result against an older, smaller
// size decides what it would have found against the current one (see the host loop), so windows
// of starts run side by side against the size at the window's beginning, every start exactly
// once, and the host replays the serial rule over the window's results: the serial algorithm's
// clique, bit for bit.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "lgicp_internal.h"

namespace lgicp {

// One pair's graph as the kernels see it.  Per-vertex arrays of all pairs of a group share one
// allocation each; a pair's part starts at c_off (a multiple of 8 elements, padding zeroed).
struct GraphDev {
  const unsigned long long* bits;  // [n][words]
  int n, words;
  int c_off;     // first element of the pair's per-vertex arrays
  int w_off;     // first word of its eligibility set
  int mc;        // best clique size so far
  int list_off;  // first word of the member-set slots this round's starts may fill
  int n_pad;     // elements of the per-vertex arrays (n rounded up to a multiple of 8, plus 8)
  int nbits;     // bits of a per-vertex value (values < n <= 2^nbits)
  int stride;    // words of a member-set slot (words rounded up to even)
};

constexpr int CLQ_MAX_WORDS = 1024;  // 65536 matches
constexpr int CLQ_SLOTS = 16;        // member sets a pair can keep per round (+1 for a start run alone)

// sum over the 64 lanes of a wave, in every lane: butterfly within rows of 16 by DPP, the four row
// sums chained by row broadcasts, total read from lane 63 (no LDS crossbar round trips)
__device__ __forceinline__ int wave_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

// degree of every vertex: one wave per row of the bit matrix
__global__ __launch_bounds__(256) void graph_degree_kernel(const unsigned long long* __restrict__ bits, int n,
                                                           int words, unsigned short* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n) return;  // wave-uniform
  const unsigned long long* r = bits + (size_t)v * words;
  int d = 0;
  for (int w = lane; w < words; w += 64) d += __builtin_popcountll(r[w]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
  if (lane == 0) deg[v] = (unsigned short)d;  // d < n <= 65536
}

// Core numbers by peeling, one workgroup per graph, one lane per 64-bit word of the alive set.
// Level k takes out every alive vertex whose remaining degree is <= k (its core number is k) and
// lowers the degree of each alive neighbour by one, until no such vertex is left; then k moves on to
// the smallest remaining degree.  Which vertices of a level leave first does not matter for the
// result: core numbers are a property of the graph, equal to the host's bucket peeling entry by entry.
// A lane keeps the remaining degrees of its 64 vertices in registers as 16 bit planes: "degree <= k"
// for all 64 is a bit-sliced comparison, and a leaving vertex v lowers the 64 by its row word in one
// bit-sliced decrement (a borrow rippling up the planes, gone after a few) - no counter per vertex in
// memory, no atomics per edge, and the row of v is read once, coalesced over the lanes.  The vertices
// that leave in a step are handed over through a list in LDS (two lists taking turns: one barrier per
// step).  A graph is a chain of some thousand dependent steps of a microsecond or two, so one graph
// keeps a few waves busy and the graphs of a group run side by side, one launch for all of them.
// c: in the degrees, out the core numbers.  steps_out[graph] = steps taken (diagnostics).
// blockDim.x = 64 * ceil(widest graph's words / 64).
constexpr int PEEL_LIST = 2048;  // vertices that leave per step at most; the rest of a level waits a step
constexpr int PEEL_AHEAD = 4;    // rows requested together

__device__ __forceinline__ void peel_decrement(unsigned long long (&pl)[16], unsigned long long m) {
  unsigned long long borrow = m;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    if (p > 0 && (p & 3) == 0 && __ballot(borrow != 0) == 0) break;  // wave-uniform
    const unsigned long long t = pl[p];
    pl[p] = t ^ borrow;
    borrow &= ~t;
  }
}

__global__ __launch_bounds__(1024) void core_peel_group_kernel(const GraphDev* __restrict__ graphs,
                                                               unsigned short* __restrict__ c,
                                                               int* __restrict__ steps_out) {
  __shared__ unsigned short list[2][PEEL_LIST];
  __shared__ int leaving_s[3], alive_s[3], min_s[3];
  const GraphDev G = graphs[blockIdx.x];
  const int n = G.n, words = G.words, nbits = G.nbits;
  if (n <= 0) return;  // block-uniform
  const int w = threadIdx.x;  // this lane's word of the alive set
  unsigned short* const cg = c + G.c_off;
  unsigned long long pl[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) pl[p] = 0;
  unsigned long long alive = 0;
  if (w < words) {
    const int left = n - 64 * w;  // >= 1: words = ceil(n / 64)
    alive = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
    for (int b = 0; b < 64 && b < left; ++b) {
      const unsigned long long d = cg[64 * w + b];
#pragma unroll
      for (int p = 0; p < 16; ++p) pl[p] |= ((d >> p) & 1ull) << b;
    }
  }
  if (threadIdx.x < 3) leaving_s[threadIdx.x] = 0, alive_s[threadIdx.x] = 0, min_s[threadIdx.x] = INT_MAX;
  __syncthreads();
  int k = 0, steps = 0;
  for (int it = 0;; ++it) {
    // the counters rotate through three slots: slot s is filled before this step's barrier, read
    // after it, and cleared two steps later, when every thread is past the barrier in between
    const int s = it % 3;
    unsigned short* const mine = list[it & 1];
    if (threadIdx.x == 0) {
      const int z = (it + 1) % 3;
      leaving_s[z] = 0, alive_s[z] = 0, min_s[z] = INT_MAX;
    }
    // alive vertices of this word with a remaining degree <= k
    unsigned long long lt = 0, eq = alive;
#pragma unroll
    for (int p = 15; p >= 0; --p)
      if (p < nbits) {
        if ((k >> p) & 1) lt |= eq & ~pl[p], eq &= pl[p];
        else eq &= ~pl[p];
      }
    unsigned long long go = lt | eq;
    if (go) {
      const int want = __builtin_popcountll(go);
      const int base = atomicAdd(&leaving_s[s], want);
      int take = min(want, max(0, PEEL_LIST - base));  // a full list: the others stay alive for now
      int pos = base;
      unsigned long long taken = 0;
      for (unsigned long long m = go; m && take > 0; m &= m - 1, --take) {
        const int b = __builtin_ctzll(m);
        taken |= 1ull << b;
        mine[pos++] = (unsigned short)(64 * w + b);
        cg[64 * w + b] = (unsigned short)k;
      }
      alive &= ~taken;
    }
    if (alive) alive_s[s] = 1;
    __syncthreads();
    if (alive_s[s] == 0) {  // nothing is left: the vertices that have just left need no processing
      if (leaving_s[s] > 0) ++steps;
      break;
    }
    const int leaving = min(leaving_s[s], PEEL_LIST);
    if (leaving == 0) {
      // no vertex at this level: on to the smallest remaining degree (per lane: the smallest of its
      // alive vertices, bit by bit from the top plane down)
      int val = INT_MAX;
      if (alive) {
        unsigned long long cand = alive;
        val = 0;
#pragma unroll
        for (int p = 15; p >= 0; --p) {
          const unsigned long long t = cand & ~pl[p];
          cand = t ? t : cand;
          val |= t ? 0 : (1 << p);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) val = min(val, __shfl_xor(val, off));
      if ((threadIdx.x & 63) == 0 && val != INT_MAX) atomicMin(&min_s[s], val);
      __syncthreads();
      k = max(k + 1, min_s[s]);
      continue;
    }
    ++steps;
    // every alive neighbour of a leaving vertex loses one
    for (int j0 = 0; j0 < leaving; j0 += PEEL_AHEAD) {
      unsigned long long row[PEEL_AHEAD];
#pragma unroll
      for (int i = 0; i < PEEL_AHEAD; ++i)
        row[i] = (w < words && j0 + i < leaving) ? G.bits[(size_t)mine[j0 + i] * words + w] : 0ull;
#pragma unroll
      for (int i = 0; i < PEEL_AHEAD; ++i) {
        const unsigned long long m = row[i] & alive;
        if (__ballot(m != 0)) peel_decrement(pl, m);
      }
    }
  }
  if (threadIdx.x == 0 && steps_out) steps_out[blockIdx.x] = steps;
}

// elig bit u = (core number of u) + 1 > mc, for every pair (blockIdx.y); also hands the pair's
// set slots out afresh (tickets[p] = slots taken, best_key[p] = best (size, earliest start) so far)
__global__ __launch_bounds__(256) void clique_eligible_group_kernel(const GraphDev* __restrict__ graphs,
                                                                    const unsigned short* __restrict__ c,
                                                                    unsigned long long* __restrict__ elig,
                                                                    int* __restrict__ tickets,
                                                                    unsigned long long* __restrict__ best_key) {
  const GraphDev G = graphs[blockIdx.y];
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w == 0) tickets[blockIdx.y] = 0, best_key[blockIdx.y] = 0ull;
  if (w >= G.words) return;
  const unsigned short* cp = c + G.c_off + w * 64;
  unsigned long long m = 0;
  const int left = G.n - w * 64;  // vertices of this word
  for (int b = 0; b < 64 && b < left; ++b)
    if ((int)cp[b] + 1 > G.mc) m |= 1ull << b;
  elig[G.w_off + w] = m;
}

// One wave per start.  starts[s] = (pair, start vertex); out[3 s] = members found besides the start
// vertex (0: cannot beat the pair's mc), out[3 s + 1] = the slot that holds them as a bit set
// (-1: none; the host then runs the start once more on its own with kAlone, which always writes
// its members, into the pair's extra slot), out[3 s + 2] = for a start that beats mc, the number
// of its neighbours whose bound exceeds its own result (what the heuristic's entry test would
// count if the best size were that result: the host needs it to apply the serial rule).
// A start is a chain of dependent steps (find the next member, fetch its row, intersect, count),
// and a launch lasts as long as its deepest start, so the chain is kept short: a window of 64
// entries of the preference order sits in registers (the next one is requested a window ahead),
// and while a member's row is being intersected the row of the candidate that follows it in the
// order is already on its way - it is the next member whenever it survives the intersection, which
// in these dense graphs it mostly does.  WPL = 64-bit words of a row per lane.
template <bool kAlone, int WPL>
__global__ __launch_bounds__(64) void clique_starts_group_kernel(
    const GraphDev* __restrict__ graphs, const unsigned short* __restrict__ pref_all,
    const unsigned long long* __restrict__ elig_all, const unsigned short* __restrict__ c_all,
    const int2* __restrict__ starts, int* __restrict__ out, unsigned long long* __restrict__ sets,
    int* __restrict__ tickets,
    unsigned long long* __restrict__ best_key, int lds_words, unsigned long long* __restrict__ rows_read) {
  // dynamic LDS: candidate set and member set, `lds_words` words each (the widest graph of the
  // group: the fewer bytes a start holds, the more starts a CU runs side by side)
  extern __shared__ unsigned long long clq_lds[];
  unsigned long long* const cand = clq_lds;
  unsigned long long* const memb = clq_lds + lds_words;
  const int s = blockIdx.x, lane = threadIdx.x;
  const int2 sv = starts[s];
  const GraphDev G = graphs[sv.x];
  const int v = sv.y, n = G.n, words = G.words, mc = G.mc;
  const unsigned short* pref = pref_all + G.c_off;
  const unsigned long long* elig = elig_all + G.w_off;
  auto load_row = [&](int vtx, unsigned long long (&dst)[WPL]) {
    const unsigned long long* r = G.bits + (size_t)vtx * words;
#pragma unroll
    for (int q = 0; q < WPL; ++q) {
      const int w = lane + 64 * q;
      dst[q] = w < words ? r[w] : 0ull;
    }
  };
  unsigned long long row[WPL], ahead[WPL];
  load_row(v, row);
  int np = 0;
#pragma unroll
  for (int q = 0; q < WPL; ++q) {
    const int w = lane + 64 * q;
    if (w < words) {
      const unsigned long long x = row[q] & elig[w];
      cand[w] = x;
      memb[w] = 0;
      np += __builtin_popcountll(x);
    }
  }
  np = wave_sum(np);
  __syncthreads();
  int size = 0, rows = 1;  // rows of the bit matrix this start reads (statistics)
  if (np > mc) {
    int at = 0;                                        // first entry of the window in registers
    int u = lane < n ? (int)pref[lane] : -1;           // window [at, at + 64)
    int u_next = 64 + lane < n ? (int)pref[64 + lane] : -1;
    int skip = 0;                                      // lanes below it lie behind the cursor
    int ahead_of = -1;                                 // vertex whose row `ahead` holds
    for (;;) {
      // the first candidate in preference order from the cursor on
      const bool hit = lane >= skip && u >= 0 && ((cand[u >> 6] >> (u & 63)) & 1ull);
      const unsigned long long m = __ballot(hit);
      if (m == 0) {  // window used up: the next one is in registers already
        at += 64;
        if (at >= n) break;
        u = u_next;
        u_next = at + 64 + lane < n ? (int)pref[at + 64 + lane] : -1;
        skip = 0;
        continue;
      }
      const int first = __builtin_ctzll(m);
      const int pick = __builtin_amdgcn_readlane(u, first);
      skip = first + 1;
      const unsigned long long m2 = m & (m - 1);
      const int follower = m2 ? __builtin_amdgcn_readlane(u, __builtin_ctzll(m2)) : -1;
      if (lane == 0) memb[pick >> 6] |= 1ull << (pick & 63);
      ++size;
      if (pick == ahead_of) {  // wave-uniform
#pragma unroll
        for (int q = 0; q < WPL; ++q) row[q] = ahead[q];
      } else {
        load_row(pick, row);
        ++rows;
      }
      if (follower >= 0) {  // requested now, used (perhaps) in the next step
        load_row(follower, ahead);
        ++rows;
      }
      ahead_of = follower;
      int left = 0;
      __syncthreads();  // all lanes have read cand for the ballot above
#pragma unroll
      for (int q = 0; q < WPL; ++q) {
        const int w = lane + 64 * q;
        if (w < words) {
          const unsigned long long x = cand[w] & row[q];
          cand[w] = x;
          left += __builtin_popcountll(x);
        }
      }
      left = wave_sum(left);
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
  int ticket = -1, np_own = 0;
  if (size > 0 && size + 1 > mc) {
    {
      // neighbours whose bound exceeds `size`: a subset of the eligible ones (size >= mc)
      const unsigned short* cv = c_all + G.c_off;
      load_row(v, row);
#pragma unroll
      for (int q = 0; q < WPL; ++q) {
        const int w = lane + 64 * q;
        if (w >= words) continue;
        for (unsigned long long m = row[q] & elig[w]; m; m &= m - 1) {
          const int u = (w << 6) + __builtin_ctzll(m);
          np_own += (int)cv[u] + 1 > size;
        }
      }
      np_own = wave_sum(np_own);
    }
    // The host wants the members of the window's largest improving clique, the earliest start
    // among equals: a start keeps its set when it is the best of those finished so far (the one
    // the host ends up with always passes this test when it finishes; the others that do are few,
    // a sequence of running maxima).
    int t = CLQ_SLOTS;
    if (!kAlone) {
      const unsigned long long key = ((unsigned long long)size << 32) | (0xFFFFFFFFu - (unsigned)s);
      if (lane == 0 && atomicMax(&best_key[sv.x], key) < key) t = atomicAdd(&tickets[sv.x], 1);
      t = __builtin_amdgcn_readfirstlane(t);
    }
    if (kAlone || t < CLQ_SLOTS) {
      __syncthreads();  // lane 0's member bits
      unsigned long long* dst = sets + (size_t)G.list_off + (size_t)t * G.stride;
      for (int w = lane; w < words; w += 64) dst[w] = memb[w];
      ticket = t;
    }
  }
  if (lane == 0) {
    out[3 * s] = size;
    out[3 * s + 1] = ticket;
    out[3 * s + 2] = np_own;
    atomicAdd(rows_read, (unsigned long long)rows * (unsigned long long)words);  // in words
  }
}

}  // namespace lgicp

namespace lgicp_detail {

namespace {
struct PinBuf {  // grow-only pinned host array
  char* p = nullptr;
  size_t cap = 0;
  bool reserve(size_t bytes) {
    if (bytes <= cap) return true;
    park_pinned_memory(p);
    p = nullptr, cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) {
      p = nullptr;
      return false;
    }
    cap = want;
    return true;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr, cap = 0;
  }
};

}  // namespace

// buffers of the clique search, reused from group to group; a context owns one set per feature set
// (one clique search runs on a set at a time)
struct CliqueBuffers {
  int device = -1;
  DevBuf<unsigned short> c, pref;
  DevBuf<unsigned long long> elig, sets;
  DevBuf<lgicp::GraphDev> table;
  DevBuf<int2> starts;
  DevBuf<int> steps, out, tickets;
  DevBuf<unsigned long long> rows_read, best_key;
  hipEvent_t ev[2] = {nullptr, nullptr};
  PinBuf h_up, h_down;
  void release() {
    if (device >= 0) {
      int cur = -1;
      (void)hipGetDevice(&cur);
      if (cur != device) (void)hipSetDevice(device);
      c.release(), pref.release(), sets.release(), elig.release(), table.release();
      starts.release(), steps.release(), out.release(), tickets.release();
      h_up.release(), h_down.release(), rows_read.release(), best_key.release();
      for (hipEvent_t& e : ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
      }
      if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
    device = -1;
  }
  ~CliqueBuffers() { release(); }
};
void clique_buffers_delete(CliqueBuffers* b) { delete b; }

namespace {
constexpr int kMaxWindow = 65536;
// Windows of 1, 256 and 65,536 starts: a window lasts at least as long as its longest walk (the
// first start's, which finds a clique of hundreds to thousands of members step by step, and every
// later window holds starts that walk as far before they give up), so few windows; but a window's
// starts are all evaluated against the size at its beginning, so the second one must not be the
// large one (measured on 512-pair batches: growth 16 278 ms per batch, 64 285, 256 266; first
// windows of 4, 8 or 16 starts 271-283; 16 then everything 329).
constexpr int kWindowGrowth = 256;
}  // namespace

// largest graph the device path takes: vertex ids and core numbers travel as unsigned short, and the
// peeling kernel gives a row's 64-bit words one lane each of a workgroup of at most 1024 (65536
// vertices, the size the consistency graph is capped at anyway)
int clique_device_max_vertices() { return 65536; }

// teaser::greedy_clique on `count` bit matrices in device memory, all of them advancing together.
// Entries with n == 0 are skipped.  Returns false when the device refuses; cliques[i] comes back
// sorted ascending, equal to the host function's result for graph i.
bool greedy_clique_group(CliqueBuffers** bufs, int device, hipStream_t st, const CliqueGraph* gr, int count,
                         std::vector<int>* cliques, CliqueTimes* times, bool rerun_improving) {
  using lgicp::GraphDev;
  for (int i = 0; i < count; ++i) cliques[i].clear();
  std::vector<int> ids;
  for (int i = 0; i < count; ++i) {
    if (gr[i].n <= 0) continue;
    if (gr[i].n > clique_device_max_vertices() || gr[i].words > lgicp::CLQ_MAX_WORDS ||
        gr[i].words != (gr[i].n + 63) / 64)
      return false;
    ids.push_back(i);
  }
  const int P = (int)ids.size();
  if (P == 0) return true;
  if (!*bufs) *bufs = new CliqueBuffers();
  CliqueBuffers& B = **bufs;
  if (B.device != device) {
    B.release();
    if (hipSetDevice(device) != hipSuccess) return false;
    B.device = device;
  }
  for (hipEvent_t& e : B.ev)
    if (!e && hipEventCreate(&e) != hipSuccess) return false;
  if (B.rows_read.reserve(1) != hipSuccess ||
      hipMemsetAsync(B.rows_read.p, 0, sizeof(unsigned long long), st) != hipSuccess)
    return false;
  const bool trace = std::getenv("LGICP_TRACE") != nullptr;
  const auto t_0 = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
  };
  // ---- layout ----------------------------------------------------------------------------------
  std::vector<GraphDev> tab((size_t)P);
  std::vector<size_t> list_base((size_t)P);
  size_t total_c = 0, total_w = 0, total_l = 0;
  int max_n = 0, max_words = 0;
  for (int p = 0; p < P; ++p) {
    const CliqueGraph& g = gr[ids[p]];
    GraphDev& t = tab[p];
    t.bits = g.d_bits;
    t.n = g.n, t.words = g.words;
    t.n_pad = ((g.n + 7) & ~7) + 8;
    t.c_off = (int)total_c, t.w_off = (int)total_w;
    t.mc = 0, t.list_off = 0;
    t.nbits = 1;
    while ((1 << t.nbits) < g.n) ++t.nbits;  // values are at most n - 1
    t.stride = (g.words + 1) & ~1;
    list_base[p] = total_l;
    total_l += (size_t)2 * (lgicp::CLQ_SLOTS + 1) * t.stride;
    total_c += (size_t)t.n_pad;
    total_w += (size_t)g.words;
    max_n = std::max(max_n, g.n), max_words = std::max(max_words, g.words);
  }
  if (total_l > (size_t)INT_MAX || total_c > (size_t)INT_MAX) return false;
  if (B.c.reserve(total_c) != hipSuccess ||
      B.pref.reserve(total_c) != hipSuccess || B.sets.reserve(total_l) != hipSuccess ||
      B.elig.reserve(total_w) != hipSuccess || B.table.reserve((size_t)P) != hipSuccess ||
      B.steps.reserve((size_t)P) != hipSuccess ||
      B.tickets.reserve((size_t)P) != hipSuccess || B.best_key.reserve((size_t)P) != hipSuccess)
    return false;
  const size_t tab_bytes = sizeof(GraphDev) * (size_t)P;
  if (!B.h_up.reserve(tab_bytes + sizeof(unsigned short) * total_c) ||
      !B.h_down.reserve(std::max(sizeof(unsigned short) * total_c, sizeof(int) * (size_t)P)))
    return false;
  auto upload_table = [&]() {
    std::memcpy(B.h_up.p, tab.data(), tab_bytes);
    return hipMemcpyAsync(B.table.p, B.h_up.p, tab_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
  };
  // ---- core numbers ----------------------------------------------------------------------------
  if (hipMemsetAsync(B.c.p, 0, sizeof(unsigned short) * total_c, st) != hipSuccess || !upload_table()) return false;
  for (int p = 0; p < P; ++p)
    hipLaunchKernelGGL(lgicp::graph_degree_kernel, dim3((tab[p].n + 3) / 4), dim3(256), 0, st, tab[p].bits,
                       tab[p].n, tab[p].words, B.c.p + tab[p].c_off);
  int n_steps = 0;  // peeling steps of the graph that took the most
  {
    // one workgroup per graph, as many waves as the widest graph has 64-word pieces of a row
    (void)hipEventRecord(B.ev[0], st);
    hipLaunchKernelGGL(lgicp::core_peel_group_kernel, dim3((unsigned)P), dim3(64 * ((max_words + 63) / 64)), 0, st,
                       B.table.p, B.c.p, B.steps.p);
    (void)hipEventRecord(B.ev[1], st);
    int* h_steps = reinterpret_cast<int*>(B.h_down.p);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h_steps, B.steps.p, sizeof(int) * (size_t)P, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    for (int p = 0; p < P; ++p) n_steps = std::max(n_steps, h_steps[p]);
    if (times) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, B.ev[0], B.ev[1]) == hipSuccess) times->core_ms += ms;
      ++times->core_launches;
      // every vertex leaves once, and its row is read then
      for (int p = 0; p < P; ++p) times->core_matrix_bytes += (double)tab[p].n * tab[p].words * 8.0;
    }
  }
  unsigned short* h_c = reinterpret_cast<unsigned short*>(B.h_down.p);
  if (hipMemcpyAsync(h_c, B.c.p, sizeof(unsigned short) * total_c, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return false;
  const double ms_core = ms_since(t_0);
  const auto t_1 = std::chrono::steady_clock::now();
  // ---- orders (teaser::greedy_clique): starts by decreasing bound, ascending id among equals;
  // candidates by decreasing bound, then decreasing id.  Counting sorts over the bounds.
  struct Search {
    std::vector<int> K, order;
    int mc = 0, at = 0, win = 1;
    int best_v = -1, best_size = 0;
    size_t best_loc = 0;
    bool done = false;
    int seg_first = 0, seg_count = 0;   // its starts in the current round
  };
  std::vector<Search> S((size_t)P);
  unsigned short* h_pref = reinterpret_cast<unsigned short*>(B.h_up.p + tab_bytes);
  auto orders_of = [&](int p) {
    const int n = tab[p].n;
    Search& s = S[p];
    s.K.resize((size_t)n), s.order.resize((size_t)n);
    int max_k = 0;
    for (int v = 0; v < n; ++v) max_k = std::max(max_k, s.K[v] = (int)h_c[tab[p].c_off + v] + 1);
    std::vector<int> first((size_t)max_k + 2, 0);  // first[k] = vertices with a larger bound
    for (int v = 0; v < n; ++v) first[s.K[v]]++;
    int run = 0;
    for (int k = max_k; k >= 0; --k) {
      const int c = first[k];
      first[k] = run;
      run += c;
    }
    std::vector<int> next = first;
    for (int v = 0; v < n; ++v) s.order[next[s.K[v]]++] = v;
    next = first;
    unsigned short* pr = h_pref + tab[p].c_off;
    for (int v = n - 1; v >= 0; --v) pr[next[s.K[v]]++] = (unsigned short)v;
  };
  {
    // a few milliseconds of counting sorts for a group of large graphs: graphs side by side on host threads
    const int nt = std::min(P, (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    std::atomic<int> next_p{0};
    auto run = [&] {
      for (int p; (p = next_p.fetch_add(1)) < P;) orders_of(p);
    };
    std::vector<std::thread> pool;
    if (total_c >= 65536)
      for (int t = 1; t < nt; ++t) pool.emplace_back(run);
    run();
    for (std::thread& t : pool) t.join();
  }
  if (hipMemcpyAsync(B.pref.p, h_pref, sizeof(unsigned short) * total_c, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)  // h_up is rewritten below
    return false;
  const double ms_sort = ms_since(t_1);
  const auto t_2 = std::chrono::steady_clock::now();
  // ---- windows of starts -------------------------------------------------------------------------
  int n_windows = 0;
  long long n_starts = 0, n_alone = 0;
  const size_t clq_lds_bytes = sizeof(unsigned long long) * 2 * (size_t)max_words;
  // words of a row per lane, rounded up to the instantiated widths
  const int wpl = max_words <= 128 ? 2 : max_words <= 256 ? 4 : max_words <= 512 ? 8 : 16;
  auto launch_starts = [&](bool alone_mode, size_t count) {
#define LGICP_CLQ_LAUNCH(ALONE, W)                                                                              \
  hipLaunchKernelGGL((lgicp::clique_starts_group_kernel<ALONE, W>), dim3((unsigned)count), dim3(64), clq_lds_bytes, \
                     st, B.table.p, B.pref.p, B.elig.p, B.c.p, B.starts.p, B.out.p, B.sets.p, B.tickets.p, B.best_key.p,  \
                     max_words, B.rows_read.p)
    if (alone_mode) {
      if (wpl == 2) LGICP_CLQ_LAUNCH(true, 2); else if (wpl == 4) LGICP_CLQ_LAUNCH(true, 4);
      else if (wpl == 8) LGICP_CLQ_LAUNCH(true, 8); else LGICP_CLQ_LAUNCH(true, 16);
    } else {
      if (wpl == 2) LGICP_CLQ_LAUNCH(false, 2); else if (wpl == 4) LGICP_CLQ_LAUNCH(false, 4);
      else if (wpl == 8) LGICP_CLQ_LAUNCH(false, 8); else LGICP_CLQ_LAUNCH(false, 16);
    }
#undef LGICP_CLQ_LAUNCH
  };
  std::vector<int2> starts;
  for (int round = 0;; ++round) {
    starts.clear();
    const int parity = round & 1;
    for (int p = 0; p < P; ++p) {
      Search& s = S[p];
      s.seg_first = (int)starts.size(), s.seg_count = 0;
      if (s.done) continue;
      const int n = tab[p].n;
      int scan = s.at;
      while (scan < n && s.seg_count < s.win) {
        const int v = s.order[scan++];
        if (s.K[v] > s.mc) starts.push_back(make_int2(p, v)), ++s.seg_count;
      }
      if (s.seg_count == 0) {
        s.done = true;
        continue;
      }
      s.at = scan;
      tab[p].mc = s.mc;
      tab[p].list_off = (int)(list_base[p] + (size_t)parity * (lgicp::CLQ_SLOTS + 1) * tab[p].stride);
    }
    const size_t n_s = starts.size();
    if (n_s == 0) break;
    ++n_windows, n_starts += (long long)n_s;
    if (B.starts.reserve(n_s) != hipSuccess || B.out.reserve(3 * n_s) != hipSuccess ||
        !B.h_up.reserve(tab_bytes + sizeof(int2) * n_s) || !B.h_down.reserve(sizeof(int) * 3 * n_s))
      return false;
    std::memcpy(B.h_up.p + tab_bytes, starts.data(), sizeof(int2) * n_s);
    if (!upload_table() ||
        hipMemcpyAsync(B.starts.p, B.h_up.p + tab_bytes, sizeof(int2) * n_s, hipMemcpyHostToDevice, st) != hipSuccess)
      return false;
    hipLaunchKernelGGL(lgicp::clique_eligible_group_kernel, dim3((max_words + 255) / 256, P), dim3(256), 0, st,
                       B.table.p, B.c.p, B.elig.p, B.tickets.p, B.best_key.p);
    (void)hipEventRecord(B.ev[0], st);
    launch_starts(false, n_s);
    (void)hipEventRecord(B.ev[1], st);
    int* h_out = reinterpret_cast<int*>(B.h_down.p);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h_out, B.out.p, sizeof(int) * 3 * n_s, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return false;
    if (times) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, B.ev[0], B.ev[1]) == hipSuccess) times->starts_ms += ms;
      times->starts_launches += 1;
    }
    std::vector<int2> alone;  // accepted starts whose members were not kept: run once more
    for (int p = 0; p < P; ++p) {
      Search& s = S[p];
      if (s.done || s.seg_count == 0) continue;
      // The serial rule, applied to results that were all computed against the size mc0 of the
      // window's beginning.  Let mc be the running size.  A start whose clique (with the start
      // vertex) has more than mc members found exactly what it would have found against mc: all
      // its members have bounds above mc, so they stay eligible, stay first in preference order
      // at their step, and keep the candidate count above every give-up test; and a start that
      // does not beat mc against mc0 does not beat it against mc either (its walk can only end
      // sooner).  One test of the heuristic does not follow from a result against mc0: it starts
      // from a vertex only if MORE than mc of its neighbours are eligible, which matters for a
      // clique of exactly mc + 1 members; the kernel therefore reports that count for the bound
      // equal to the start's own result (out[3 s + 2]).  Every start is evaluated once.
      const int mc0 = s.mc;
      int accepted = -1;
      for (int k = 0; k < s.seg_count; ++k) {
        const int* o = h_out + 3 * (size_t)(s.seg_first + k);
        const int sz = o[0];
        if (sz <= 0 || sz + 1 <= s.mc) continue;
        if (sz + 1 >= s.mc + 2 || s.mc == mc0 || o[2] > s.mc) {
          accepted = k;
          s.mc = sz + 1;
        }
      }
      if (accepted >= 0) {
        const int at_s = s.seg_first + accepted;
        const int ticket = rerun_improving ? -1 : h_out[3 * (size_t)at_s + 1];
        s.best_size = h_out[3 * (size_t)at_s];
        s.best_v = starts[at_s].y;
        if (ticket >= 0) {
          s.best_loc = (size_t)tab[p].list_off + (size_t)ticket * tab[p].stride;
        } else {
          s.best_loc = (size_t)tab[p].list_off + (size_t)lgicp::CLQ_SLOTS * tab[p].stride;
          alone.push_back(starts[at_s]);
        }
      }
      s.win = std::min(s.win * kWindowGrowth, kMaxWindow);
    }
    if (!alone.empty()) {
      // same table (old mc, old eligibility) as the window they come from
      n_alone += (long long)alone.size();
      std::memcpy(B.h_up.p + tab_bytes, alone.data(), sizeof(int2) * alone.size());
      if (hipMemcpyAsync(B.starts.p, B.h_up.p + tab_bytes, sizeof(int2) * alone.size(), hipMemcpyHostToDevice, st) !=
          hipSuccess)
        return false;
      launch_starts(true, alone.size());
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return false;
    }
  }
  // ---- the members of every pair's best start ----------------------------------------------------
  size_t total_best = 0;
  for (int p = 0; p < P; ++p) total_best += (size_t)tab[p].words;
  if (!B.h_down.reserve(sizeof(unsigned long long) * (total_best + 1))) return false;
  unsigned long long* h_l = reinterpret_cast<unsigned long long*>(B.h_down.p);
  size_t o = 0;
  for (int p = 0; p < P; ++p) {
    if (S[p].best_size > 0 &&
        hipMemcpyAsync(h_l + o, B.sets.p + S[p].best_loc, sizeof(unsigned long long) * (size_t)tab[p].words,
                       hipMemcpyDeviceToHost, st) != hipSuccess)
      return false;
    o += (size_t)tab[p].words;
  }
  unsigned long long rows_words = 0;
  if (hipMemcpyAsync(&rows_words, B.rows_read.p, sizeof(rows_words), hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return false;
  if (times) times->starts_row_bytes += 8.0 * (double)rows_words;
  o = 0;
  for (int p = 0; p < P; ++p) {
    std::vector<int>& best = cliques[ids[p]];
    if (S[p].best_size > 0)
      for (int w = 0; w < tab[p].words; ++w)
        for (unsigned long long m = h_l[o + w]; m; m &= m - 1) best.push_back((w << 6) + __builtin_ctzll(m));
    o += (size_t)tab[p].words;
    if ((int)best.size() != S[p].best_size) return false;  // the set and the size the start reported disagree
    if (S[p].best_v >= 0) best.push_back(S[p].best_v);
    std::sort(best.begin(), best.end());
  }
  if (trace)
    std::fprintf(stderr, "[lgicp] clique search on the device: %d graphs (largest %d vertices), core numbers %.2f ms "
                 "(%d steps), orders %.2f ms, %d windows / %lld starts (%lld run alone) %.2f ms\n", P, max_n, ms_core,
                 n_steps, ms_sort, n_windows, n_starts, n_alone, ms_since(t_2));
  return true;
}

bool greedy_clique_device(CliqueBuffers** bufs, int device, hipStream_t st, const unsigned long long* d_bits,
                          int n, int words,
                          std::vector<int>* clique) {
  const CliqueGraph g{d_bits, n, words};
  return greedy_clique_group(bufs, device, st, &g, 1, clique);
}

}  // namespace lgicp_detail
