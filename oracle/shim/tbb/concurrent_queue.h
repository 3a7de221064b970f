// oracle/shim/tbb/concurrent_queue.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  The blocking queue between the reference's
// tracking thread and its keyframe-graph thread (keyframe_graph.cpp: push / blocking pop / empty).
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
namespace tbb {
template <typename T> class concurrent_bounded_queue {
 public:
  void push(const T& v) {
    { std::lock_guard<std::mutex> lock(m_); q_.push_back(v); }
    cv_.notify_one();
  }
  void pop(T& out) {
    std::unique_lock<std::mutex> lock(m_);
    cv_.wait(lock, [this] { return !q_.empty(); });
    out = q_.front();
    q_.pop_front();
  }
  bool try_pop(T& out) {
    std::lock_guard<std::mutex> lock(m_);
    if (q_.empty()) return false;
    out = q_.front();
    q_.pop_front();
    return true;
  }
  bool empty() const { std::lock_guard<std::mutex> lock(m_); return q_.empty(); }
  size_t size() const { std::lock_guard<std::mutex> lock(m_); return q_.size(); }
 private:
  mutable std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};
template <typename T> class concurrent_queue : public concurrent_bounded_queue<T> {};
}  // namespace tbb
