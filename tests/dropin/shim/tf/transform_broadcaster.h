// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of tf that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <tf/tf.h>
namespace tf {
class TransformBroadcaster {   // the last transform sent is kept for the test to read
 public:
  void sendTransform(const StampedTransform& t) { last() = t; count() += 1; }
  static StampedTransform& last() { static StampedTransform t; return t; }
  static int& count() { static int n = 0; return n; }
};
}  // namespace tf
