// oracle/shim/ros/time.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  ros::Time as the reference's Keyframe uses it: seconds.
#pragma once
#include <ostream>
namespace ros {
class Time {
 public:
  Time() : sec_(0) {}
  explicit Time(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Time& fromSec(double s) { sec_ = s; return *this; }
  bool operator<(const Time& o) const { return sec_ < o.sec_; }
  bool isZero() const { return sec_ == 0.0; }
 private:
  double sec_;
};
inline std::ostream& operator<<(std::ostream& o, const Time& t) { return o << t.toSec(); }
}  // namespace ros
