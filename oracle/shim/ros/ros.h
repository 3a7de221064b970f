// oracle/shim/ros/ros.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  Topics the reference's front end publishes its
// diagnostics on (dvo_slam/src/keyframe_tracker.cpp:73-79): publishing is a no-op.
#pragma once
#include <fstream>
#include <string>
#include "console.h"
#include "time.h"
namespace ros {
class Publisher { public: template <typename M> void publish(const M&) const {} };
class NodeHandle { public: template <typename M> Publisher advertise(const std::string&, int) { return Publisher(); } };
}  // namespace ros
