// oracle/shim/tbb/blocked_range.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  tbb::blocked_range with its published splitting
// rule (divisible while longer than the grain size; a split leaves the lower half in place and hands out the upper half).
#pragma once
#include <cstddef>
namespace tbb {
struct split {};
template <typename T> class blocked_range {
 public:
  blocked_range(T b, T e, std::size_t grain = 1) : b_(b), e_(e), grain_(grain ? grain : 1) {}
  blocked_range(blocked_range& r, split) : b_(r.b_ + (r.e_ - r.b_) / 2), e_(r.e_), grain_(r.grain_) { r.e_ = b_; }
  T begin() const { return b_; }
  T end() const { return e_; }
  std::size_t size() const { return std::size_t(e_ - b_); }
  std::size_t grainsize() const { return grain_; }
  bool empty() const { return !(b_ < e_); }
  bool is_divisible() const { return grain_ < size(); }
 private:
  T b_, e_;
  std::size_t grain_;
};
}  // namespace tbb
