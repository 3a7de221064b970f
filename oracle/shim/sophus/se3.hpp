// oracle/shim/sophus/se3.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  Sophus is an un-vendored, unpinned dependency of
// the reference (sophus/Makefile:5-8 clones strasdat/Sophus at HEAD) and is not installed here.  This stand-in gives
// Sophus::SE3d the interface dense_tracking.cpp uses (:147-150, 238, 259-261, 302, 346, 371) on top of the oracle's own SE(3)
// closed forms (oracle/se3_oracle.h: unit quaternion + translation like Sophus::SO3/SE3, twist order (upsilon, omega)).
#pragma once

#include <sstream>   // (pulled in transitively by the real header; the reference relies on that)

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "../../se3_oracle.h"

namespace Sophus {

class SE3d {
 public:
  typedef Eigen::Matrix<double, 6, 1> Tangent;
  SE3d() {}
  SE3d(const Eigen::Matrix3d& R, const Eigen::Vector3d& t) {
    double M[16] = {R(0, 0), R(0, 1), R(0, 2), t(0), R(1, 0), R(1, 1), R(1, 2), t(1), R(2, 0), R(2, 1), R(2, 2), t(2), 0, 0, 0, 1};
    T_ = oracle::se3_from_matrix(M);
  }
  explicit SE3d(const Eigen::Matrix4d& m) {
    double M[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M[i * 4 + j] = m(i, j);
    T_ = oracle::se3_from_matrix(M);
  }
  static SE3d exp(const Tangent& x) {
    SE3d r;
    r.T_ = oracle::se3_exp(x.data());
    return r;
  }
  Tangent log() const {
    Tangent x;
    oracle::se3_log(T_, x.data());
    return x;
  }
  SE3d inverse() const {
    SE3d r;
    r.T_ = oracle::se3_inverse(T_);
    return r;
  }
  SE3d operator*(const SE3d& o) const {
    SE3d r;
    r.T_ = oracle::se3_mul(T_, o.T_);
    return r;
  }
  Eigen::Matrix4d matrix() const {
    double M[16];
    oracle::se3_to_matrix(T_, M);
    Eigen::Matrix4d m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m(i, j) = M[i * 4 + j];
    return m;
  }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(T_.t[0], T_.t[1], T_.t[2]); }

 private:
  oracle::SE3 T_;
};

typedef SE3d SE3;

}  // namespace Sophus
