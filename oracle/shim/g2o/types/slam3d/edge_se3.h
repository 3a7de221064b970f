// oracle/shim/g2o/types/slam3d/edge_se3.h -- TEST INFRASTRUCTURE, see ../../core/hyper_graph.h.  chi2 = e^T Omega e with g2o's error
// of a relative-pose edge (edge_se3.cpp / isometry3d_mappings.h: e = toVectorMQT(measurement^-1 * from^-1 * to): translation, then the
// vector part of the unit quaternion with w >= 0).
#pragma once
#include <cmath>
#include "vertex_se3.h"
namespace g2o {
class EdgeSE3 : public OptimizableGraph::Edge {
 public:
  EdgeSE3() { measurement_.setIdentity(); information_.setIdentity(); resize(2); }
  void setMeasurement(const Eigen::Isometry3d& m) { measurement_ = m; }
  const Eigen::Isometry3d& measurement() const { return measurement_; }
  void setInformation(const Eigen::Matrix<double, 6, 6>& i) { information_ = i; }
  const Eigen::Matrix<double, 6, 6>& information() const { return information_; }
  virtual double chi2() const {
    const VertexSE3* from = static_cast<const VertexSE3*>(vertex(0));
    const VertexSE3* to = static_cast<const VertexSE3*>(vertex(1));
    if (!from || !to) return 0.0;
    const Eigen::Isometry3d delta = measurement_.inverse() * (from->estimate().inverse() * to->estimate());
    Eigen::Quaterniond q(delta.rotation());
    double w = q.w(), e[6] = {delta(0, 3), delta(1, 3), delta(2, 3), q.x(), q.y(), q.z()};
    const double n = std::sqrt(w * w + e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
    const double s = (w < 0 ? -1.0 : 1.0) / (n > 0 ? n : 1.0);
    for (int i = 3; i < 6; ++i) e[i] *= s;
    double c = 0.0;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) c += e[i] * information_(i, j) * e[j];
    return c;
  }
 private:
  Eigen::Isometry3d measurement_;
  Eigen::Matrix<double, 6, 6> information_;
};
}  // namespace g2o
