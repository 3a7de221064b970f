// oracle/shim/tbb/parallel_invoke.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  Serial stand-in.
#pragma once
namespace tbb { template <typename F0, typename F1> void parallel_invoke(const F0& f0, const F1& f1) { f0(); f1(); } }
