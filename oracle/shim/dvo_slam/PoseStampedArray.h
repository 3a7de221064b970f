// oracle/shim/dvo_slam/PoseStampedArray.h -- TEST INFRASTRUCTURE: generated ROS message type (dvo_slam/msg/PoseStampedArray.msg), name only.
#pragma once
namespace dvo_slam { struct PoseStampedArray {}; }
