// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of sensor_msgs that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct Image {
  typedef std::shared_ptr<const Image> ConstPtr;
  typedef std::shared_ptr<Image> Ptr;
  std_msgs::Header header;
  uint32_t height = 0, width = 0, step = 0;
  std::string encoding;                    // "mono8", "bgr8", "16UC1", "32FC1"
  uint8_t is_bigendian = 0;
  std::vector<uint8_t> data;
};
}  // namespace sensor_msgs
