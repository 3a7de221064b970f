// dvo/core/intrinsic_matrix.h -- dvo::core::IntrinsicMatrix (dvo_core/include/dvo/core/intrinsic_matrix.h:31-62).
#pragma once

namespace dvo {
namespace core {

struct IntrinsicMatrix {
  static IntrinsicMatrix create(float fx, float fy, float ox, float oy) {
    IntrinsicMatrix r;
    r.fx_ = fx; r.fy_ = fy; r.ox_ = ox; r.oy_ = oy;
    return r;
  }
  IntrinsicMatrix() : fx_(0), fy_(0), ox_(0), oy_(0) {}
  float fx() const { return fx_; }
  float fy() const { return fy_; }
  float ox() const { return ox_; }
  float oy() const { return oy_; }
  void invertOffset() { ox_ *= -1; oy_ *= -1; }
  void scale(float factor) { fx_ *= factor; fy_ *= factor; ox_ *= factor; oy_ *= factor; }   // intrinsic_matrix.cpp:90-93 (Q17)
 private:
  float fx_, fy_, ox_, oy_;
};

}  // namespace core
}  // namespace dvo
