// lane_table.h — which calls may work on a context at the same time.
//
// A context is cut into lanes (lgicp_set_submit_lanes): a submit takes one lane — or, when it
// finds nothing else in flight, the first `wide` lanes, i.e. the width a lone submit can use —
// while registration, lgicp_align and lgicp_set_submit_lanes take all of them and so wait for
// every submit in flight and hold new ones off.  The reference has a single spinner thread and a
// pool that is drained before ComputeTransforms returns (IcpLoopComputation.cc:358-365), i.e. one
// lane.  Host logic only (a mutex and a condition variable): tests/cpp/lane_table.cc exercises it
// from many threads under ThreadSanitizer.
#pragma once

#include <algorithm>
#include <condition_variable>
#include <mutex>

namespace lgicp_detail {

class LaneTable {
 public:
  static constexpr int MAX_LANES = 8;

  struct Grant {
    int lane = 0;   // first lane held
    int held = 0;   // lanes held, from `lane` on
    bool alone = false;  // nothing else was in flight when the grant was made
  };

  int lanes() const { return n_lanes_; }

  // Blocks until a lane is free.  With nothing in flight at all the grant covers lanes
  // 0 .. min(wide, lanes()) - 1 (wide <= 1: lane 0 alone) and on_idle() runs first, under the
  // table's lock: the moment at which memory that work in flight might still read may be freed.
  template <class F>
  Grant acquire_one(int wide, F on_idle) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] {
      for (int l = 0; l < n_lanes_; ++l)
        if (!busy_[l]) return true;
      return false;
    });
    Grant g;
    g.alone = true;
    for (int l = 0; l < n_lanes_; ++l) g.alone = g.alone && !busy_[l];
    if (g.alone) on_idle();
    if (g.alone && wide > 1) {
      g.lane = 0, g.held = std::min(wide, n_lanes_);
    } else {
      while (busy_[g.lane]) ++g.lane;
      g.held = 1;
    }
    for (int l = g.lane; l < g.lane + g.held; ++l) busy_[l] = true;
    return g;
  }

  // Blocks until nothing is in flight, then holds every lane (also those beyond lanes(): the
  // number of lanes may be changed while all are held); on_idle() as above.
  template <class F>
  Grant acquire_all(F on_idle) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] {
      for (int l = 0; l < n_lanes_; ++l)
        if (busy_[l]) return false;
      return true;
    });
    for (int l = 0; l < MAX_LANES; ++l) busy_[l] = true;
    on_idle();
    Grant g;
    g.lane = 0, g.held = MAX_LANES, g.alone = true;
    return g;
  }

  void release(const Grant& g) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int l = g.lane; l < g.lane + g.held && l < MAX_LANES; ++l) busy_[l] = false;
    }
    cv_.notify_all();
  }

  // only while every lane is held (acquire_all)
  void set_lanes(int n) {
    std::lock_guard<std::mutex> lk(mu_);
    n_lanes_ = std::max(1, std::min(n, (int)MAX_LANES));
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  bool busy_[MAX_LANES] = {};
  int n_lanes_ = 1;
};

}  // namespace lgicp_detail
