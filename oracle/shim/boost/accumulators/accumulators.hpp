// oracle/shim/boost/accumulators/accumulators.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  A running mean, all the
// reference's stopwatch (dvo/util/stopwatch.h) asks of boost.accumulators.
#pragma once
#include <sstream>
namespace boost {
namespace accumulators {
namespace tag { struct mean {}; }
template <typename... T> struct stats {};
template <typename V, typename S> class accumulator_set {
 public:
  accumulator_set() : sum_(0), n_(0) {}
  void operator()(V v) { sum_ += v; ++n_; }
  V mean_value() const { return n_ ? sum_ / n_ : V(0); }
 private:
  V sum_;
  long n_;
};
template <typename A> double mean(const A& a) { return a.mean_value(); }
}  // namespace accumulators
}  // namespace boost
