// oracle/shim/g2o/core/sparse_optimizer.h -- TEST INFRASTRUCTURE, see optimizable_graph.h.
#pragma once
#include "optimizable_graph.h"
namespace g2o {
class OptimizationAlgorithm { public: virtual ~OptimizationAlgorithm() {} };
class SparseOptimizer : public OptimizableGraph {
 public:
  SparseOptimizer() : algorithm_(0) {}
  ~SparseOptimizer() { delete algorithm_; }
  void setAlgorithm(OptimizationAlgorithm* a) { algorithm_ = a; }
  void setVerbose(bool) {}
  bool initializeOptimization() { std::abort(); }   // graph optimisation is out of scope
  void computeInitialGuess() { std::abort(); }
  int optimize(int) { std::abort(); }
 private:
  OptimizationAlgorithm* algorithm_;
};
}  // namespace g2o
