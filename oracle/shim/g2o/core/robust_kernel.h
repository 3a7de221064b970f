// oracle/shim/g2o/core/robust_kernel.h -- TEST INFRASTRUCTURE, see hyper_graph.h.
#pragma once
#include <Eigen/Core>
namespace g2o {
class RobustKernel {
 public:
  RobustKernel() : delta_(1.0) {}
  virtual ~RobustKernel() {}
  // rho[0] = rho(e2), rho[1] = rho'(e2) (the weight), rho[2] = rho''(e2)
  virtual void robustify(double squared_error, Eigen::Vector3d& rho) const = 0;
  virtual void setDelta(double delta) { delta_ = delta; }
  double delta() const { return delta_; }
 protected:
  double delta_;
};
}  // namespace g2o
