// tests/dropin/benchmark_stubs.cpp -- TEST INFRASTRUCTURE: what has to exist at link time for the reference's built target
// dvo_benchmark/src/benchmark_slam.cpp (compiled UNMODIFIED, see the Makefile) besides the tracker itself:
//   * cv::imread / cv::cvtColor (OpenCV is not installed): PNG decoding and OpenCV's fixed-point BGR -> grey through this repo's
//     include/dvo_benchmark/image_io.h;
//   * the display / serialisation classes of subsystems that are out of scope (RViz and PCL visualisers, the pose-graph
//     serialisers of dvo_slam/src/serialization/map_serializer.cpp, which walk a g2o graph the stubbed back end does not build):
//     constructible, never reached on the no-visualisation path, the serialisers write a header line only.
#include <cstdlib>
#include <fstream>

#include <dvo/visualization/pcl_camera_trajectory_visualizer.h>
#include <dvo_ros/visualization/ros_camera_trajectory_visualizer.h>
#include <dvo_slam/serialization/map_serializer.h>
#include <dvo_slam/visualization/graph_visualizer.h>

#include <dvo_benchmark/image_io.h>   // this repo's (include/dvo_benchmark): zlib PNG reader

namespace cv {

Mat imread(const std::string& filename, int flags) {
  dvo_benchmark::PngImage png;
  try {
    png = dvo_benchmark::readPng(filename);
  } catch (...) {
    return Mat();
  }
  if (png.empty()) return Mat();
  if (flags < 0) {                                   // as stored: the 16-bit depth images
    if (png.channels == 1 && png.bit_depth == 16) {
      Mat m(png.height, png.width, CV_16UC1);
      unsigned short* o = m.ptr<unsigned short>();
      for (size_t i = 0; i < size_t(png.width) * png.height; ++i) o[i] = (unsigned short)((png.bytes[2 * i] << 8) | png.bytes[2 * i + 1]);
      return m;
    }
    if (png.channels == 1 && png.bit_depth == 8) {
      Mat m(png.height, png.width, CV_8UC1);
      std::memcpy(m.data, png.bytes.data(), size_t(png.width) * png.height);
      return m;
    }
  }
  // flags > 0: 8-bit, 3 channels, B G R order (a grey file is replicated), like cv::imread(file, 1)
  Mat m(png.height, png.width, CV_8UC3);
  const size_t n = size_t(png.width) * png.height, step = size_t(png.bit_depth / 8);
  for (size_t i = 0; i < n; ++i) {
    const unsigned char* s = png.bytes.data() + i * size_t(png.channels) * step;
    unsigned char r, g, b;
    if (png.channels >= 3) { r = s[0]; g = s[step]; b = s[2 * step]; }
    else { r = g = b = s[0]; }
    m.data[3 * i] = b; m.data[3 * i + 1] = g; m.data[3 * i + 2] = r;
  }
  return m;
}

void cvtColor(const Mat& src, Mat& dst, int code, int /*dst_channels*/) {
  if (code != CV_BGR2GRAY || src.type() != CV_8UC3) std::abort();
  Mat out(src.rows, src.cols, CV_8UC1);
  const size_t n = size_t(src.rows) * src.cols;
  for (size_t i = 0; i < n; ++i) out.data[i] = dvo_benchmark::greyFromRgb8(src.data[3 * i + 2], src.data[3 * i + 1], src.data[3 * i]);
  dst = out;
}

}  // namespace cv

namespace dvo {
namespace visualization {
pcl::visualization::PCLVisualizer& PclCameraTrajectoryVisualizer::visualizer() { std::abort(); }
}  // namespace visualization
}  // namespace dvo

namespace dvo_ros {
namespace visualization {
RosCameraTrajectoryVisualizer::RosCameraTrajectoryVisualizer(ros::NodeHandle&) {}
RosCameraTrajectoryVisualizer::~RosCameraTrajectoryVisualizer() {}
dvo::visualization::CameraVisualizer::Ptr RosCameraTrajectoryVisualizer::camera(std::string) { std::abort(); }
dvo::visualization::TrajectoryVisualizer::Ptr RosCameraTrajectoryVisualizer::trajectory(std::string) { std::abort(); }
void RosCameraTrajectoryVisualizer::reset() {}
}  // namespace visualization
}  // namespace dvo_ros

namespace dvo_slam {
namespace visualization {
namespace internal { class GraphVisualizerImpl {}; }
GraphVisualizer::GraphVisualizer(dvo_ros::visualization::RosCameraTrajectoryVisualizer&) {}
GraphVisualizer::~GraphVisualizer() {}
}  // namespace visualization

namespace serialization {
TrajectorySerializer::TrajectorySerializer(std::ostream& stream) : stream_(stream) {}
TrajectorySerializer::~TrajectorySerializer() {}
void TrajectorySerializer::serialize(const dvo_slam::KeyframeGraph&) { stream_ << "# pose graph back end not part of this build (tests/dropin)" << std::endl; }
EdgeErrorSerializer::EdgeErrorSerializer(std::ostream& stream) : stream_(stream) {}
EdgeErrorSerializer::~EdgeErrorSerializer() {}
void EdgeErrorSerializer::serialize(const dvo_slam::KeyframeGraph&) { stream_ << "# pose graph back end not part of this build (tests/dropin)" << std::endl; }
}  // namespace serialization
}  // namespace dvo_slam
