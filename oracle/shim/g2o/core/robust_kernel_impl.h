// oracle/shim/g2o/core/robust_kernel_impl.h -- TEST INFRASTRUCTURE, see hyper_graph.h.  The Cauchy kernel as g2o publishes it
// (robust_kernel_impl.cpp: rho = delta^2 log(1 + e2 / delta^2)).
#pragma once
#include <cmath>
#include "robust_kernel.h"
namespace g2o {
class RobustKernelCauchy : public RobustKernel {
 public:
  virtual void robustify(double e2, Eigen::Vector3d& rho) const {
    const double dsqr = delta_ * delta_, dsqr_reci = 1.0 / dsqr, aux = dsqr_reci * e2 + 1.0;
    rho[0] = dsqr * std::log(aux);
    rho[1] = 1.0 / aux;
    rho[2] = -dsqr_reci * rho[1] * rho[1];
  }
};
}  // namespace g2o
