// oracle/shim/tbb/blocked_range.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  Serial stand-in for the two TBB names the
// reference's (legacy, off-path) weight_calculation.cpp uses.
#pragma once
#include <cstddef>
namespace tbb {
struct split {};
template <typename T> class blocked_range {
 public:
  blocked_range(T b, T e, std::size_t = 1) : b_(b), e_(e) {}
  T begin() const { return b_; }
  T end() const { return e_; }
 private:
  T b_, e_;
};
template <typename R, typename B> void parallel_reduce(const R& r, B& body) { body(r); }
}  // namespace tbb
