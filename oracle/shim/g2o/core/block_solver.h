// oracle/shim/g2o/core/block_solver.h -- TEST INFRASTRUCTURE, see optimizable_graph.h: solver objects are constructed, never run.
#pragma once
#include "sparse_optimizer.h"
namespace g2o {
template <typename M> class LinearSolver { public: virtual ~LinearSolver() {} };
class Solver { public: virtual ~Solver() {} };
class BlockSolver_6_3 : public Solver {
 public:
  typedef Eigen::Matrix<double, 6, 6> PoseMatrixType;
  explicit BlockSolver_6_3(LinearSolver<PoseMatrixType>* l) : l_(l) {}
  ~BlockSolver_6_3() { delete l_; }
 private:
  LinearSolver<PoseMatrixType>* l_;
};
}  // namespace g2o
