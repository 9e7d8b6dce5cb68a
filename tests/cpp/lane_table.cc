// LaneTable (lamp_amd/csrc/lane_table.h) from many threads, built with ThreadSanitizer:
//  * two grants never cover the same lane;
//  * a grant of all lanes never overlaps any other grant, and waits for those in flight;
//  * a caller that finds nothing in flight gets the lanes 0 .. wide-1, others one lane each;
//  * on_idle runs only with nothing in flight;
//  * the number of lanes can change while all lanes are held.
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "lane_table.h"

using lgicp_detail::LaneTable;

int main() {
  LaneTable table;
  {
    auto g = table.acquire_all([] {});
    table.set_lanes(4);
    table.release(g);
  }
  std::atomic<int> users[LaneTable::MAX_LANES];
  for (auto& u : users) u = 0;
  std::atomic<int> in_flight{0}, errors{0}, idle_calls{0}, wide_grants{0}, all_grants{0};
  std::atomic<bool> stop{false};
  auto enter = [&](const LaneTable::Grant& g) {
    for (int l = g.lane; l < g.lane + g.held; ++l)
      if (users[l].fetch_add(1) != 0) ++errors;
    in_flight.fetch_add(1);
  };
  auto leave = [&](const LaneTable::Grant& g) {
    in_flight.fetch_sub(1);
    for (int l = g.lane; l < g.lane + g.held; ++l) users[l].fetch_sub(1);
  };
  auto on_idle = [&] {
    if (in_flight.load() != 0) ++errors;  // runs under the table's lock, before the grant is used
    ++idle_calls;
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < 10; ++t)
    pool.emplace_back([&, t] {
      unsigned s = 1234567u * (t + 1);
      for (int i = 0; i < 3000 && !stop; ++i) {
        s = s * 1664525u + 1013904223u;
        const int wide = (t % 3 == 0) ? 1 : 3;
        LaneTable::Grant g = table.acquire_one(wide, on_idle);
        if (g.held < 1 || g.lane + g.held > table.lanes()) ++errors;
        if (g.held > 1) {
          if (!g.alone || g.lane != 0) ++errors;
          ++wide_grants;
        }
        enter(g);
        for (volatile int spin = 0; spin < (int)(s >> 24); ++spin) {
        }
        leave(g);
        table.release(g);
      }
    });
  for (int t = 0; t < 2; ++t)
    pool.emplace_back([&, t] {
      for (int i = 0; i < 300; ++i) {
        LaneTable::Grant g = table.acquire_all(on_idle);
        if (in_flight.load() != 0) ++errors;
        for (int l = 0; l < LaneTable::MAX_LANES; ++l)
          if (users[l].load() != 0) ++errors;
        table.set_lanes(2 + (i + t) % 5);  // 2..6 lanes
        ++all_grants;
        table.release(g);
        std::this_thread::yield();
      }
    });
  for (auto& th : pool) th.join();
  std::printf("errors %d idle_calls %d wide_grants %d all_grants %d\n", errors.load(), idle_calls.load(),
              wide_grants.load(), all_grants.load());
  return errors.load() == 0 && all_grants.load() == 600 && idle_calls.load() > 0 ? 0 : 1;
}
