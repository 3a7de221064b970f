#pragma once
#include "vertex_se3.h"
namespace g2o {
class EdgeSE3 : public OptimizableGraph::Edge {
 public:
  void setMeasurement(const Eigen::Isometry3d& m) { measurement_ = m; }
  const Eigen::Isometry3d& measurement() const { return measurement_; }
  void setInformation(const Eigen::Matrix<double, 6, 6>& i) { information_ = i; }
  const Eigen::Matrix<double, 6, 6>& information() const { return information_; }
 private:
  Eigen::Isometry3d measurement_;
  Eigen::Matrix<double, 6, 6> information_;
};
}  // namespace g2o
