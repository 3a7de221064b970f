// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of cv_bridge that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <cstring>
#include <memory>
#include <opencv2/core/core.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
struct CvImage { cv::Mat image; };
typedef std::shared_ptr<const CvImage> CvImageConstPtr;
// a cv::Mat with the message's pixels (copied: the stand-in cv::Mat owns its memory)
inline CvImageConstPtr toCvShare(const sensor_msgs::Image::ConstPtr& msg) {
  int type = CV_8UC1, bytes = 1;
  if (msg->encoding == "bgr8" || msg->encoding == "rgb8") { type = CV_8UC3; bytes = 3; }
  else if (msg->encoding == "16UC1" || msg->encoding == "mono16") { type = CV_16UC1; bytes = 2; }
  else if (msg->encoding == "32FC1") { type = CV_32FC1; bytes = 4; }
  std::shared_ptr<CvImage> out(new CvImage);
  out->image = cv::Mat(int(msg->height), int(msg->width), type);
  for (uint32_t y = 0; y < msg->height; ++y)
    std::memcpy(out->image.data + size_t(y) * msg->width * bytes, msg->data.data() + size_t(y) * msg->step, size_t(msg->width) * bytes);
  return out;
}
}  // namespace cv_bridge
