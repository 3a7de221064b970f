// oracle/shim/g2o/core/sparse_optimizer.h -- TEST INFRASTRUCTURE, see hyper_graph.h.  The OPTIMISER is not restated: graph
// optimisation is outside this engine's scope, and a pose-graph solver written here would be a second library to
// validate.  optimize() leaves every vertex estimate where its caller put it (the odometry chain) and reports 0 iterations; what the
// reference's callers do around it -- building the graph, proposing and validating loop closures with DenseTracker, inserting edges,
// weighing them with the robust kernel -- runs unchanged.
#pragma once
#include "optimizable_graph.h"
namespace g2o {
class OptimizationAlgorithm { public: virtual ~OptimizationAlgorithm() {} };
class SparseOptimizer : public OptimizableGraph {
 public:
  SparseOptimizer() : algorithm_(0), optimize_calls_(0) {}
  ~SparseOptimizer() { delete algorithm_; }
  void setAlgorithm(OptimizationAlgorithm* a) { delete algorithm_; algorithm_ = a; }
  void setVerbose(bool) {}
  bool initializeOptimization(int = 0) { return true; }
  bool initializeOptimization(HyperGraph::VertexSet&, int = 0) { return true; }
  void computeInitialGuess() {}
  int optimize(int, bool = false) { ++optimize_calls_; return 0; }
  int optimizeCalls() const { return optimize_calls_; }     // (stand-in only: how often the caller asked)
 private:
  OptimizationAlgorithm* algorithm_;
  int optimize_calls_;
};
}  // namespace g2o
