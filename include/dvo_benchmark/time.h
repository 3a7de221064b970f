// dvo_benchmark/time.h -- stand-in for ros::Time, the timestamp type of the reference's benchmark readers
// (dvo_benchmark/include/dvo_benchmark/rgbd_pair.h:44-48, groundtruth.h:44): seconds since the epoch with the
// handful of operations the drivers use (fromSec / toSec, ordering, difference, stream output).
#pragma once

#include <iomanip>
#include <ostream>

namespace dvo_benchmark {

class Duration {
 public:
  explicit Duration(double s = 0.0) : sec_(s) {}
  double toSec() const { return sec_; }

 private:
  double sec_;
};

class Time {
 public:
  Time() : sec_(0.0) {}
  explicit Time(double s) : sec_(s) {}
  Time& fromSec(double s) { sec_ = s; return *this; }
  double toSec() const { return sec_; }
  bool operator<(const Time& o) const { return sec_ < o.sec_; }
  bool operator>(const Time& o) const { return sec_ > o.sec_; }
  bool operator<=(const Time& o) const { return sec_ <= o.sec_; }
  bool operator>=(const Time& o) const { return sec_ >= o.sec_; }
  bool operator==(const Time& o) const { return sec_ == o.sec_; }
  Duration operator-(const Time& o) const { return Duration(sec_ - o.sec_); }

 private:
  double sec_;
};

inline std::ostream& operator<<(std::ostream& out, const Time& t) {
  const std::ios_base::fmtflags f = out.flags();
  const std::streamsize p = out.precision();
  out << std::fixed << std::setprecision(6) << t.toSec();   // TUM stamps carry microseconds
  out.flags(f);
  out.precision(p);
  return out;
}

}  // namespace dvo_benchmark
