// oracle/shim/g2o/core/optimization_algorithm_levenberg.h -- TEST INFRASTRUCTURE, see sparse_optimizer.h: constructed, never run.
#pragma once
#include "block_solver.h"
namespace g2o {
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
 public:
  explicit OptimizationAlgorithmLevenberg(Solver* s) : s_(s) {}
  ~OptimizationAlgorithmLevenberg() { delete s_; }
 private:
  Solver* s_;
};
}  // namespace g2o
