// oracle/shim/g2o/core/optimization_algorithm_dogleg.h -- TEST INFRASTRUCTURE, see sparse_optimizer.h: constructed, never run.
#pragma once
#include "block_solver.h"
namespace g2o {
class OptimizationAlgorithmDogleg : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmDogleg(Solver* s) : s_(s) {}
  ~OptimizationAlgorithmDogleg() { delete s_; }
 private:
  Solver* s_;
};
}  // namespace g2o
