// dvo/util/stopwatch.h -- dvo::util::stopwatch / stopwatch_collection with the interface of
// dvo_core/include/dvo/util/stopwatch.h:37-119 (start / stop / print / stopAndPrint; the mean of `interval` samples goes to
// stderr), on std::chrono instead of cv::getTickCount + boost.accumulators.  The reference's callers keep their live
// instances ("prepare", "m", "match", "online", ...: SURVEY.md section 5).
#pragma once

#include <chrono>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace dvo {
namespace util {

struct stopwatch {
 public:
  stopwatch(std::string name, int interval = 500) : name_(name + ": "), sum_(0.0), count_(0), interval_(interval) {}
  inline void start() { begin_ = clock::now(); }
  inline void stop() {
    sum_ += std::chrono::duration<double>(clock::now() - begin_).count();
    ++count_;
  }
  inline void print() {
    if (count_ == interval_) {
      std::cerr << name_ << sum_ / double(count_) << std::endl;
      sum_ = 0.0;
      count_ = 0;
    }
  }
  inline void stopAndPrint() {
    stop();
    print();
  }

 private:
  typedef std::chrono::steady_clock clock;
  std::string name_;
  clock::time_point begin_;
  double sum_;
  int count_, interval_;
};

struct stopwatch_collection {
 public:
  stopwatch_collection(const size_t num, std::string base_name, int interval = 500) : num_(num) {
    for (size_t idx = 0; idx < num; ++idx) {
      std::stringstream name;
      name << base_name << idx;
      watches_.push_back(new stopwatch(name.str(), interval));
    }
  }
  ~stopwatch_collection() {
    for (size_t idx = 0; idx < num_; ++idx) delete watches_[idx];
  }
  stopwatch& operator[](int idx) { return *watches_[size_t(idx)]; }

 private:
  size_t num_;
  std::vector<stopwatch*> watches_;
};

}  // namespace util
}  // namespace dvo
