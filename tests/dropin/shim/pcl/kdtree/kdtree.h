// tests/dropin/shim/pcl/kdtree/kdtree.h -- TEST INFRASTRUCTURE: the radius search over keyframe positions that the reference's
// NearestNeighborConstraintSearch (dvo_slam/src/keyframe_constraint_search.cpp) asks PCL for, by exhaustive comparison (unsorted
// results come back in index order -- KdTreeFLANN(false) leaves the order unspecified).
#pragma once
#include <map>      // (PCL's headers bring it in; keyframe_constraint_search.cpp relies on that)
#include <vector>
#include <boost/shared_ptr.hpp>
#include <boost/make_shared.hpp>
namespace pcl {
struct PointXYZ { float x, y, z; PointXYZ() : x(0), y(0), z(0) {} };
template <typename P> class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<P> > Ptr;
  typedef boost::shared_ptr<const PointCloud<P> > ConstPtr;
  std::vector<P> points;
};
template <typename P> class KdTree {
 public:
  typedef boost::shared_ptr<KdTree<P> > Ptr;
  virtual ~KdTree() {}
  virtual void setInputCloud(const typename PointCloud<P>::Ptr& cloud) { cloud_ = cloud; }
  virtual int radiusSearch(const P& p, double radius, std::vector<int>& indices, std::vector<float>& sqr_distances, unsigned int max_nn = 0) const {
    indices.clear();
    sqr_distances.clear();
    if (!cloud_) return 0;
    for (size_t i = 0; i < cloud_->points.size(); ++i) {
      const P& q = cloud_->points[i];
      const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z, d2 = dx * dx + dy * dy + dz * dz;
      if (d2 <= float(radius * radius) && (max_nn == 0 || indices.size() < max_nn)) { indices.push_back(int(i)); sqr_distances.push_back(d2); }
    }
    return int(indices.size());
  }
 protected:
  typename PointCloud<P>::Ptr cloud_;
};
}  // namespace pcl
