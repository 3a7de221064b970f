// tests/dropin/shim -- TEST INFRASTRUCTURE: stand-ins for third-party headers the reference's dvo_benchmark/src/benchmark_slam.cpp
// and dvo_ros/src/camera_dense_tracking.cpp pull in (boost.thread, PCL, tf, ROS messages and node handles, generated
// dynamic_reconfigure configs, OpenCV image I/O), none of which is installed here.  Together with oracle/shim they let those files be
// compiled UNMODIFIED (tests/dropin/Makefile).
#pragma once
#include <mutex>
namespace boost {
class mutex : public std::mutex {
 public:
  class scoped_lock : public std::unique_lock<std::mutex> {
   public:
    explicit scoped_lock(boost::mutex& m) : std::unique_lock<std::mutex>(static_cast<std::mutex&>(m)) {}
  };
};
}  // namespace boost
