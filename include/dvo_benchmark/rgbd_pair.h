// dvo_benchmark/rgbd_pair.h -- one line of a TUM association file: `rgb_stamp rgb_file depth_stamp depth_file`
// (reference: dvo_benchmark/include/dvo_benchmark/rgbd_pair.h:37-71).
#pragma once

#include <istream>
#include <ostream>
#include <string>

#include "dvo_benchmark/time.h"

namespace dvo_benchmark {

class RgbdPair {
 public:
  const Time& RgbTimestamp() const { return rgb_stamp_; }
  RgbdPair& RgbTimestamp(const Time& t) { rgb_stamp_ = t; return *this; }
  const std::string& RgbFile() const { return rgb_file_; }
  RgbdPair& RgbFile(const std::string& f) { rgb_file_ = f; return *this; }
  const Time& DepthTimestamp() const { return depth_stamp_; }
  RgbdPair& DepthTimestamp(const Time& t) { depth_stamp_ = t; return *this; }
  const std::string& DepthFile() const { return depth_file_; }
  RgbdPair& DepthFile(const std::string& f) { depth_file_ = f; return *this; }

 private:
  Time rgb_stamp_, depth_stamp_;
  std::string rgb_file_, depth_file_;
};

inline std::ostream& operator<<(std::ostream& out, const RgbdPair& p) {
  return out << p.RgbTimestamp() << " " << p.RgbFile() << " " << p.DepthTimestamp() << " " << p.DepthFile() << std::endl;
}

inline std::istream& operator>>(std::istream& in, RgbdPair& p) {
  double rgb_s = 0, depth_s = 0;
  std::string rgb_f, depth_f;
  if (in >> rgb_s >> rgb_f >> depth_s >> depth_f) p.RgbTimestamp(Time(rgb_s)).RgbFile(rgb_f).DepthTimestamp(Time(depth_s)).DepthFile(depth_f);
  return in;
}

}  // namespace dvo_benchmark
