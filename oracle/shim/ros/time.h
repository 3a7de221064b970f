// oracle/shim/ros/time.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  ros::Time as the reference's Keyframe uses it: seconds.
#pragma once
#include <cmath>
#include <cstdio>
#include <ostream>
namespace ros {
class Time {
 public:
  Time() : sec_(0) {}
  explicit Time(double s) : sec_(s) {}
  double toSec() const { return sec_; }
  Time& fromSec(double s) { sec_ = s; return *this; }
  bool operator<(const Time& o) const { return sec_ < o.sec_; }
  bool operator>(const Time& o) const { return sec_ > o.sec_; }
  bool operator<=(const Time& o) const { return sec_ <= o.sec_; }
  bool operator>=(const Time& o) const { return sec_ >= o.sec_; }
  bool operator==(const Time& o) const { return sec_ == o.sec_; }
  bool isZero() const { return sec_ == 0.0; }
 private:
  double sec_;
};
class Duration {
 public:
  Duration() : sec_(0) {}
  explicit Duration(double s) : sec_(s) {}
  double toSec() const { return sec_; }
 private:
  double sec_;
};
// printed like roscpp prints it: seconds '.' nanoseconds on nine digits (the trajectory files of benchmark_slam.cpp:494 carry it)
inline std::ostream& operator<<(std::ostream& o, const Time& t) {
  double sec = std::floor(t.toSec());
  long long nsec = (long long)std::llround((t.toSec() - sec) * 1e9);
  if (nsec >= 1000000000LL) { sec += 1; nsec -= 1000000000LL; }
  char buf[48];
  std::snprintf(buf, sizeof(buf), "%lld.%09lld", (long long)sec, nsec);
  return o << buf;
}
}  // namespace ros
