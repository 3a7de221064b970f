// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of sensor_msgs that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct CameraInfo {
  typedef std::shared_ptr<const CameraInfo> ConstPtr;
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::array<double, 9> K{};
  std::array<double, 12> P{};
};
}  // namespace sensor_msgs
