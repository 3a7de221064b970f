// tests/dropin/shim -- TEST INFRASTRUCTURE: stand-ins for third-party headers the reference's dvo_benchmark/src/benchmark_slam.cpp
// pulls in (boost.thread, PCL, tf, ROS messages and node handles, generated dynamic_reconfigure configs, OpenCV image I/O), none of
// which is installed here.  Together with oracle/shim they let that file be compiled UNMODIFIED (tests/dropin/Makefile).
#pragma once
#include <mutex>
namespace boost { typedef std::mutex mutex; }
