#include <cstdio>
#include <random>
#include "teaser.h"
using namespace lgicp::teaser;
int main() {
  std::mt19937_64 rng(7);
  for (int trial = 0; trial < 6; ++trial) {
    const int n = 600 + 500 * trial;
    BitGraph g; g.n = n; g.words = (n + 63) / 64; g.bits.assign((size_t)n * g.words, 0);
    std::uniform_real_distribution<double> U(0, 1);
    const double p = 0.3 + 0.1 * trial;
    // planted clique plus noise
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) {
      const bool e = (i % 7 == 0 && j % 7 == 0) || U(rng) < p;
      if (e) { g.bits[(size_t)i * g.words + (j >> 6)] |= 1ull << (j & 63); g.bits[(size_t)j * g.words + (i >> 6)] |= 1ull << (i & 63); }
    }
    const std::vector<int> a = greedy_clique(g, 1), b = greedy_clique(g, 8), c = greedy_clique(g, 3);
    std::printf("n %d clique %zu same %d %d\n", n, a.size(), (int)(a == b), (int)(a == c));
    if (a != b || a != c) return 1;
  }
  return 0;
}
