// oracle/shim/g2o/core/optimization_algorithm_gauss_newton.h -- TEST INFRASTRUCTURE, see sparse_optimizer.h: constructed, never run.
#pragma once
#include "block_solver.h"
namespace g2o {
class OptimizationAlgorithmGaussNewton : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmGaussNewton(Solver* s) : s_(s) {}
  ~OptimizationAlgorithmGaussNewton() { delete s_; }
 private:
  Solver* s_;
};
}  // namespace g2o
