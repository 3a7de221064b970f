// oracle/shim/tbb/tbb_thread.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  tbb::tbb_thread / this_tbb_thread::sleep over <thread>.
#pragma once
#include <chrono>
#include <thread>
namespace tbb {
class tick_count {
 public:
  class interval_t {
   public:
    explicit interval_t(double seconds = 0.0) : s_(seconds) {}
    double seconds() const { return s_; }
   private:
    double s_;
  };
};
class tbb_thread {
 public:
  tbb_thread() {}
  template <typename F> explicit tbb_thread(F f) : t_(f) {}
  ~tbb_thread() { if (t_.joinable()) t_.join(); }
  bool joinable() const { return t_.joinable(); }
  void join() { t_.join(); }
 private:
  std::thread t_;
};
namespace this_tbb_thread {
inline void sleep(const tick_count::interval_t& i) { std::this_thread::sleep_for(std::chrono::duration<double>(i.seconds())); }
inline void yield() { std::this_thread::yield(); }
}  // namespace this_tbb_thread
}  // namespace tbb
