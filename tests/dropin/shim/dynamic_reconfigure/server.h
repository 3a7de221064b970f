// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of dynamic_reconfigure that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <cstdint>
#include <boost/function.hpp>
#include <ros/ros.h>
namespace dynamic_reconfigure {
// setCallback calls the callback once with the defaults and every level bit set, like the real server does
template <typename Config> class Server {
 public:
  typedef boost::function<void(Config&, uint32_t)> CallbackType;
  explicit Server(const ros::NodeHandle& nh) : nh_(nh), config_(Config::__getDefault__()) { config_.__fromServer__(nh_); }
  void setCallback(const CallbackType& cb) { callback_ = cb; callback_(config_, ~uint32_t(0)); }
  Config& config() { return config_; }
 private:
  ros::NodeHandle nh_;
  Config config_;
  CallbackType callback_;
};
}  // namespace dynamic_reconfigure
