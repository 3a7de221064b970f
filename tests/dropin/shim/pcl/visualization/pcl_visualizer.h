// tests/dropin/shim (see boost/thread/mutex.hpp): the two PCL types benchmark_slam.cpp names in its viewer-camera dump / restore
// code (processInput, only reached with a PCL viewer, which this build does not have).
#pragma once
#include <vector>
namespace pcl {
namespace visualization {
struct Camera {
  double clip[2], focal[3], pos[3], view[3], fovy, window_size[2], window_pos[2];
};
class PCLVisualizer {
 public:
  void getCameras(std::vector<Camera>& cameras) { cameras.assign(1, Camera()); }
  void getCameraParameters(int, char**) {}
  void updateCamera() {}
};
}  // namespace visualization
}  // namespace pcl
