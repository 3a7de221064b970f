// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of message_filters that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <functional>
#include <string>
#include <ros/ros.h>
namespace message_filters {
template <typename M> class Subscriber {
 public:
  Subscriber(ros::NodeHandle&, const std::string& topic, int) : topic_(topic) {}
  const std::string& topic() const { return topic_; }
 private:
  std::string topic_;
};
class Connection {
 public:
  Connection() : live_(nullptr) {}
  explicit Connection(bool* live) : live_(live) {}
  void disconnect() { if (live_) *live_ = false; }
 private:
  bool* live_;
};
}  // namespace message_filters
