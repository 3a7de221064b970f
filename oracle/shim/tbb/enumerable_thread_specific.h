// oracle/shim/tbb/enumerable_thread_specific.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  One lazily constructed element per
// concurrent branch of the stand-in parallel_reduce (TBB: per worker thread; the stand-in starts fresh threads per call, so the
// branch number plays the worker's part and the pool stays as small as TBB's would).
#pragma once
#include <functional>
#include <map>
#include <mutex>
#include "parallel_reduce.h"
namespace tbb {
template <typename T> class enumerable_thread_specific {
 public:
  enumerable_thread_specific() : make_([] { return T(); }) {}
  template <typename F> explicit enumerable_thread_specific(F f) : make_(f) {}
  T& local() {
    std::lock_guard<std::mutex> lock(m_);
    typename std::map<int, T>::iterator it = items_.find(shim::worker_slot());
    if (it == items_.end()) it = items_.insert(std::make_pair(shim::worker_slot(), make_())).first;
    return it->second;
  }
  size_t size() const { std::lock_guard<std::mutex> lock(m_); return items_.size(); }
 private:
  std::function<T()> make_;
  mutable std::mutex m_;
  std::map<int, T> items_;                                  // (node-based: references stay valid while others are added)
};
}  // namespace tbb
