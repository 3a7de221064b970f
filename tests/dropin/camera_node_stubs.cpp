// tests/dropin/camera_node_stubs.cpp -- TEST INFRASTRUCTURE: what has to exist at link time for the reference's live-camera front end
// (dvo_ros/src/camera_dense_tracking.cpp, compiled UNMODIFIED, see the Makefile) besides the tracker: cv::cvtColor (OpenCV is not
// installed; OpenCV's fixed-point BGR -> grey through include/dvo_benchmark/image_io.h) and the RViz visualiser class of dvo_ros
// (display only, out of scope: constructible, never reached with `reconstruction` off).
#include <cstdlib>

#include <dvo_ros/visualization/ros_camera_trajectory_visualizer.h>

#include <dvo_benchmark/image_io.h>

namespace cv {
void cvtColor(const Mat& src, Mat& dst, int code, int /*dst_channels*/) {
  if (code != CV_BGR2GRAY || src.type() != CV_8UC3) std::abort();
  Mat out(src.rows, src.cols, CV_8UC1);
  const size_t n = size_t(src.rows) * src.cols;
  for (size_t i = 0; i < n; ++i) out.data[i] = dvo_benchmark::greyFromRgb8(src.data[3 * i + 2], src.data[3 * i + 1], src.data[3 * i]);
  dst = out;
}
}  // namespace cv

namespace dvo_ros {
namespace visualization {
RosCameraTrajectoryVisualizer::RosCameraTrajectoryVisualizer(ros::NodeHandle&) {}
RosCameraTrajectoryVisualizer::~RosCameraTrajectoryVisualizer() {}
dvo::visualization::CameraVisualizer::Ptr RosCameraTrajectoryVisualizer::camera(std::string) { std::abort(); }
dvo::visualization::TrajectoryVisualizer::Ptr RosCameraTrajectoryVisualizer::trajectory(std::string) { std::abort(); }
void RosCameraTrajectoryVisualizer::reset() {}
}  // namespace visualization
}  // namespace dvo_ros
