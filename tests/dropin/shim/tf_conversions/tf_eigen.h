// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of tf_conversions / tf transform_datatypes that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <cmath>
#include <Eigen/Geometry>
#include <geometry_msgs/Pose.h>
#include <tf/tf.h>
namespace tf {
inline void TransformTFToEigen(const Transform& t, Eigen::Affine3d& e) {
  const double x = t.rotation.x, y = t.rotation.y, z = t.rotation.z, w = t.rotation.w;
  e.setIdentity();
  e.matrix()(0, 0) = 1 - 2 * (y * y + z * z); e.matrix()(0, 1) = 2 * (x * y - z * w); e.matrix()(0, 2) = 2 * (x * z + y * w);
  e.matrix()(1, 0) = 2 * (x * y + z * w); e.matrix()(1, 1) = 1 - 2 * (x * x + z * z); e.matrix()(1, 2) = 2 * (y * z - x * w);
  e.matrix()(2, 0) = 2 * (x * z - y * w); e.matrix()(2, 1) = 2 * (y * z + x * w); e.matrix()(2, 2) = 1 - 2 * (x * x + y * y);
  e.matrix()(0, 3) = t.origin.x; e.matrix()(1, 3) = t.origin.y; e.matrix()(2, 3) = t.origin.z;
}
inline void TransformEigenToTF(const Eigen::Affine3d& e, Transform& t) {
  const double m00 = e.matrix()(0, 0), m11 = e.matrix()(1, 1), m22 = e.matrix()(2, 2), tr = m00 + m11 + m22;
  double x, y, z, w;
  if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (e.matrix()(2, 1) - e.matrix()(1, 2)) / s; y = (e.matrix()(0, 2) - e.matrix()(2, 0)) / s; z = (e.matrix()(1, 0) - e.matrix()(0, 1)) / s; }
  else if (m00 > m11 && m00 > m22) { const double s = std::sqrt(1.0 + m00 - m11 - m22) * 2; w = (e.matrix()(2, 1) - e.matrix()(1, 2)) / s; x = 0.25 * s; y = (e.matrix()(0, 1) + e.matrix()(1, 0)) / s; z = (e.matrix()(0, 2) + e.matrix()(2, 0)) / s; }
  else if (m11 > m22) { const double s = std::sqrt(1.0 + m11 - m00 - m22) * 2; w = (e.matrix()(0, 2) - e.matrix()(2, 0)) / s; x = (e.matrix()(0, 1) + e.matrix()(1, 0)) / s; y = 0.25 * s; z = (e.matrix()(1, 2) + e.matrix()(2, 1)) / s; }
  else { const double s = std::sqrt(1.0 + m22 - m00 - m11) * 2; w = (e.matrix()(1, 0) - e.matrix()(0, 1)) / s; x = (e.matrix()(0, 2) + e.matrix()(2, 0)) / s; y = (e.matrix()(1, 2) + e.matrix()(2, 1)) / s; z = 0.25 * s; }
  t.rotation = Quaternion(x, y, z, w);
  t.origin = Vector3(e.matrix()(0, 3), e.matrix()(1, 3), e.matrix()(2, 3));
}
inline void poseMsgToTF(const geometry_msgs::Pose& p, Transform& t) {
  t.origin = Vector3(p.position.x, p.position.y, p.position.z);
  t.rotation = Quaternion(p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w);
}
inline void poseTFToMsg(const Transform& t, geometry_msgs::Pose& p) {
  p.position.x = t.origin.x; p.position.y = t.origin.y; p.position.z = t.origin.z;
  p.orientation.x = t.rotation.x; p.orientation.y = t.rotation.y; p.orientation.z = t.rotation.z; p.orientation.w = t.rotation.w;
}
}  // namespace tf
