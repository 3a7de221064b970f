// oracle/shim/ros/ros.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  The sliver of roscpp the reference's callers use:
// the topics the front end publishes its diagnostics on (dvo_slam/src/keyframe_tracker.cpp:73-79; publishing is a no-op), and what
// benchmark_slam.cpp's main() and BenchmarkNode need to run as a plain process -- ros::init, node handles with private parameters,
// ok / spinOnce / Rate.
// Private parameters come from the command line in rosrun's own syntax (`_name:=value`), e.g.
//   benchmark_slam _rgbdpair_file:=/data/assoc.txt _estimate_trajectory:=true _trajectory_file:=out.txt _coarsest_level:=3
#pragma once
#include <cstdlib>
#include <fstream>   // (the real roscpp headers pull it in transitively; dvo_slam/serialization/map_serializer.h relies on that)
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <chrono>

#include <ros/console.h>
#include <ros/time.h>

namespace ros {
namespace init_options { enum InitOption { AnonymousName = 1 }; }

inline std::map<std::string, std::string>& private_params() {
  static std::map<std::string, std::string> p;
  return p;
}

inline void init(int& argc, char** argv, const std::string&, int = 0) {
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    const size_t eq = a.find(":=");
    if (a.size() > 1 && a[0] == '_' && eq != std::string::npos) private_params()[a.substr(1, eq - 1)] = a.substr(eq + 2);
  }
}
inline bool ok() { return true; }
inline void spinOnce() {}
class Rate {
 public:
  explicit Rate(double hz) : hz_(hz) {}
  void sleep() { std::this_thread::sleep_for(std::chrono::duration<double>(1.0 / hz_)); }
 private:
  double hz_;
};

// Publishing is a no-op but for a count and an optional tap: a test of the reference's live-camera front end
// (dvo_ros/src/camera_dense_tracking.cpp) reads what the node published through it.
class Publisher {
 public:
  template <typename M> void publish(const M& m) const {
    published() += 1;
    if (tap()) tap()(static_cast<const void*>(&m));
  }
  unsigned getNumSubscribers() const { return tap() ? 1u : 0u; }
  static int& published() { static int n = 0; return n; }
  typedef void (*Tap)(const void* message);
  static Tap& tap() { static Tap t = nullptr; return t; }
};
class Subscriber {};

class NodeHandle {
 public:
  NodeHandle() {}
  explicit NodeHandle(const std::string&) {}
  NodeHandle(const NodeHandle&, const std::string&) {}
  template <typename M> Publisher advertise(const std::string&, int) { return Publisher(); }
  template <typename M, typename T> Subscriber subscribe(const std::string&, int, void (T::*)(const std::shared_ptr<const M>&), T*) { return Subscriber(); }
  bool getParam(const std::string& name, std::string& value) const {
    std::map<std::string, std::string>::const_iterator it = private_params().find(name);
    if (it == private_params().end()) return false;
    value = it->second;
    return true;
  }
  template <typename T> bool getParam(const std::string& name, T& value) const {
    std::string s;
    if (!getParam(name, s)) return false;
    return parse(s, value);
  }
  template <typename T> void param(const std::string& name, T& value, const T& fallback) const {
    if (!getParam(name, value)) value = fallback;
  }
 private:
  static bool parse(const std::string& s, bool& v) { v = s == "true" || s == "True" || s == "1"; return true; }
  template <typename T> static bool parse(const std::string& s, T& v) { std::istringstream in(s); return bool(in >> v); }
};
}  // namespace ros
