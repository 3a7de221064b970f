// tests/dropin/shim (see boost/thread/mutex.hpp): TEST INFRASTRUCTURE -- the sliver of std_msgs that dvo_ros/src/camera_dense_tracking.cpp
// and camera_base.cpp (the reference's live-camera front end, compiled unmodified against this engine's facade) need to compile and to
// be driven by a test without a ROS master.  Not ROS.
#pragma once
#include <string>
#include <ros/time.h>
namespace std_msgs { struct Header { unsigned seq = 0; ros::Time stamp; std::string frame_id; }; }
