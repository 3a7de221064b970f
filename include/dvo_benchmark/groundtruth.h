// dvo_benchmark/groundtruth.h -- one line of a TUM trajectory file: `stamp tx ty tz qx qy qz qw`
// (reference: dvo_benchmark/include/dvo_benchmark/groundtruth.h:37-79).
#pragma once

#include <istream>
#include <ostream>

#include "dvo_benchmark/time.h"

namespace dvo_benchmark {

class Groundtruth {
 public:
  Groundtruth() : px_(0), py_(0), pz_(0), qx_(0), qy_(0), qz_(0), qw_(1) {}
  const Time& Timestamp() const { return stamp_; }
  Groundtruth& Timestamp(const Time& t) { stamp_ = t; return *this; }
  double PositionX() const { return px_; }
  double PositionY() const { return py_; }
  double PositionZ() const { return pz_; }
  double OrientationX() const { return qx_; }
  double OrientationY() const { return qy_; }
  double OrientationZ() const { return qz_; }
  double OrientationW() const { return qw_; }
  Groundtruth& Position(double x, double y, double z) { px_ = x; py_ = y; pz_ = z; return *this; }
  Groundtruth& Orientation(double x, double y, double z, double w) { qx_ = x; qy_ = y; qz_ = z; qw_ = w; return *this; }

 private:
  Time stamp_;
  double px_, py_, pz_, qx_, qy_, qz_, qw_;
};

inline std::ostream& operator<<(std::ostream& out, const Groundtruth& g) {
  return out << g.Timestamp() << " " << g.PositionX() << " " << g.PositionY() << " " << g.PositionZ() << " " << g.OrientationX() << " "
             << g.OrientationY() << " " << g.OrientationZ() << " " << g.OrientationW() << std::endl;
}

inline std::istream& operator>>(std::istream& in, Groundtruth& g) {
  double v[8];
  for (int i = 0; i < 8; ++i)
    if (!(in >> v[i])) return in;
  g.Timestamp(Time(v[0])).Position(v[1], v[2], v[3]).Orientation(v[4], v[5], v[6], v[7]);
  return in;
}

}  // namespace dvo_benchmark
