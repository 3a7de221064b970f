// tests/dropin/shim (see boost/thread/mutex.hpp): tf::Pose as dvo_benchmark/tools.h fills it.
#pragma once
namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion { double x, y, z, w; Quaternion(double a = 0, double b = 0, double c = 0, double d = 1) : x(a), y(b), z(c), w(d) {} };
struct Pose {
  Vector3 origin;
  Quaternion rotation;
  void setOrigin(const Vector3& v) { origin = v; }
  void setRotation(const Quaternion& q) { rotation = q; }
};
}  // namespace tf
