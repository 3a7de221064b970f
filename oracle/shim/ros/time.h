// oracle/shim/ros/time.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  ros::Time as the reference's Keyframe uses it: seconds.
#pragma once
namespace ros {
class Time {
 public:
  Time() : sec_(0) {}
  explicit Time(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Time& fromSec(double s) { sec_ = s; return *this; }
  bool operator<(const Time& o) const { return sec_ < o.sec_; }
 private:
  double sec_;
};
}  // namespace ros
