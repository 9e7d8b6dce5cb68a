// PairSharding — cutting one candidate list over several GPUs.
//
// The reference spreads the candidates of a tick over a pool of host threads, one PerformAlignment
// per task (loop_closure/src/IcpLoopComputation.cc:327-365, ThreadPool.h:17-115): pairs are
// independent.  Across GPUs the same independence holds, with one cost the pool does not have: a
// keyframe has to be resident on the GPU that aligns a pair it belongs to, and keyframes are shared
// by many candidates (proximity generation emits several per new node,
// loop_closure/src/ProximityLoopGeneration.cc:153-167).  PartitionPairs therefore balances the
// estimated cost (source points x target points) greedily, longest first, with a bonus for the
// device that already holds a pair's keyframes.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <numeric>
#include <set>
#include <vector>

namespace lamp_loop_closure {

struct PairKeys {
  uint64_t key_from;  // source
  uint64_t key_to;    // target
};

// Returns, per device, the ascending indices of the pairs it aligns.  Deterministic: pairs are
// placed in order of decreasing cost (ties: lower index first) on the device with the lowest
// load - 0.25 * mean_cost * (keyframes of the pair the device holds), ties to the lower device.
// held: keyframes resident on each device before the call (may be null or shorter than `devices`);
// on return (if not null) it also lists what the partition adds.
inline std::vector<std::vector<size_t>> PartitionPairs(
    const std::vector<PairKeys>& pairs, const std::function<size_t(uint64_t)>& cloud_size,
    int devices, std::vector<std::set<uint64_t>>* held = nullptr) {
  const int world = std::max(1, devices);
  std::vector<double> cost(pairs.size());
  for (size_t i = 0; i < pairs.size(); ++i)
    cost[i] = static_cast<double>(cloud_size(pairs[i].key_from)) *
              static_cast<double>(cloud_size(pairs[i].key_to));
  std::vector<size_t> order(pairs.size());
  std::iota(order.begin(), order.end(), size_t{0});
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
  double mean = 0;
  for (double c : cost) mean += c;
  mean = pairs.empty() ? 1.0 : mean / static_cast<double>(pairs.size());
  if (mean == 0) mean = 1.0;
  std::vector<std::set<uint64_t>> local;
  std::vector<std::set<uint64_t>>& h = held ? *held : local;
  h.resize(static_cast<size_t>(world));
  std::vector<double> load(static_cast<size_t>(world), 0.0);
  std::vector<std::vector<size_t>> out(static_cast<size_t>(world));
  for (size_t i : order) {
    int best = 0;
    double best_score = 0;
    for (int r = 0; r < world; ++r) {
      const int shared = static_cast<int>(h[r].count(pairs[i].key_from)) +
                         static_cast<int>(h[r].count(pairs[i].key_to));
      const double score = load[r] - 0.25 * mean * shared;
      if (r == 0 || score < best_score) best = r, best_score = score;
    }
    out[best].push_back(i);
    load[best] += cost[i];
    h[best].insert(pairs[i].key_from);
    h[best].insert(pairs[i].key_to);
  }
  for (auto& v : out) std::sort(v.begin(), v.end());
  return out;
}

}  // namespace lamp_loop_closure
