// tests/dropin/shim/pcl/kdtree/kdtree_flann.h -- TEST INFRASTRUCTURE, see kdtree.h.
#pragma once
#include "kdtree.h"
namespace pcl {
template <typename P> class KdTreeFLANN : public KdTree<P> {
 public:
  explicit KdTreeFLANN(bool /*sorted*/ = true) {}
};
}  // namespace pcl
