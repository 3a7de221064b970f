// tests/dropin/shim (see boost/thread/mutex.hpp): tf::Pose as dvo_benchmark/tools.h fills it, and the transform types the reference's
// live-camera front end (dvo_ros/src/camera_dense_tracking.cpp) converts to and from Eigen.
#pragma once
#include <string>
#include <ros/time.h>
namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion { double x, y, z, w; Quaternion(double a = 0, double b = 0, double c = 0, double d = 1) : x(a), y(b), z(c), w(d) {} };
struct Transform {
  Vector3 origin;
  Quaternion rotation;
  void setOrigin(const Vector3& v) { origin = v; }
  void setRotation(const Quaternion& q) { rotation = q; }
  static Transform getIdentity() { return Transform(); }
};
typedef Transform Pose;
struct StampedTransform : Transform {
  ros::Time stamp_;
  std::string frame_id_, child_frame_id_;
};
}  // namespace tf
