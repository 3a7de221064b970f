// oracle/shim/g2o/types/slam3d/vertex_se3.h -- TEST INFRASTRUCTURE, see ../../core/hyper_graph.h.
#pragma once
#include "../../core/optimizable_graph.h"
namespace g2o {
class VertexSE3 : public OptimizableGraph::Vertex {
 public:
  VertexSE3() { estimate_.setIdentity(); }
  void setEstimate(const Eigen::Isometry3d& e) { estimate_ = e; }
  const Eigen::Isometry3d& estimate() const { return estimate_; }
 private:
  Eigen::Isometry3d estimate_;
};
}  // namespace g2o
