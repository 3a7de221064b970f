// tests/dropin/shim (see boost/thread/mutex.hpp): geometry_msgs/Pose as dvo_benchmark/tools.h fills it.
#pragma once
namespace geometry_msgs {
struct Point { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs
