// tests/dropin/benchmark_stubs.cpp -- TEST INFRASTRUCTURE: what has to exist at link time for the reference's built target
// dvo_benchmark/src/benchmark_slam.cpp (compiled UNMODIFIED, see the Makefile) besides the tracker itself:
//   * cv::imread / cv::cvtColor (OpenCV is not installed): PNG decoding and OpenCV's fixed-point BGR -> grey through this repo's
//     include/dvo_benchmark/image_io.h;
//   * the display classes of subsystems that are out of scope (RViz and PCL visualisers): constructible, never reached on the
//     no-visualisation path;
//   * the two pose-graph serialisers benchmark_slam.cpp ends with (dvo_slam/src/serialization/map_serializer.cpp needs g2o's edge
//     error vectors and ROS pose messages): deterministic dumps of the graph's vertices and edges instead (below).
#include <cstdlib>
#include <map>
#include <fstream>

#include <dvo/visualization/pcl_camera_trajectory_visualizer.h>
#include <dvo_ros/visualization/ros_camera_trajectory_visualizer.h>
#include <dvo_slam/serialization/map_serializer.h>
#include <dvo_slam/visualization/graph_visualizer.h>
#include <dvo_slam/timestamped.h>
#include <g2o/types/slam3d/vertex_se3.h>
#include <g2o/types/slam3d/edge_se3.h>

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
// Deterministic dumps of whatever graph the build has (the counting stub's is empty; dvo_slam/src/keyframe_graph.cpp's holds the
// keyframe and odometry vertices and the relative-pose edges): one line per vertex in time order / per edge in id order, so that two
// builds of the same executable can be compared line by line (the reference's own serialisers walk pointer-ordered sets).
TrajectorySerializer::TrajectorySerializer(std::ostream& stream) : stream_(stream) {}
TrajectorySerializer::~TrajectorySerializer() {}
void TrajectorySerializer::serialize(const dvo_slam::KeyframeGraph& map) {
  stream_ << "# vertices of the pose graph (tests/dropin): stamp tx ty tz qx qy qz qw id" << std::endl;
  std::map<std::pair<double, int>, const g2o::VertexSE3*> ordered;
  for (g2o::HyperGraph::VertexIDMap::const_iterator it = map.graph().vertices().begin(); it != map.graph().vertices().end(); ++it) {
    const g2o::VertexSE3* v = static_cast<const g2o::VertexSE3*>(it->second);
    const dvo_slam::Timestamped* t = dynamic_cast<const dvo_slam::Timestamped*>(v->userData());
    ordered[std::make_pair(t ? t->timestamp.toSec() : 0.0, v->id())] = v;
  }
  stream_.precision(9);
  for (std::map<std::pair<double, int>, const g2o::VertexSE3*>::const_iterator it = ordered.begin(); it != ordered.end(); ++it) {
    const Eigen::Isometry3d& p = it->second->estimate();
    const Eigen::Quaterniond q(p.rotation());
    stream_ << std::fixed << it->first.first << std::defaultfloat << " " << p(0, 3) << " " << p(1, 3) << " " << p(2, 3) << " " << q.x() << " " << q.y() << " "
            << q.z() << " " << q.w() << " " << it->second->id() << std::endl;
  }
}
EdgeErrorSerializer::EdgeErrorSerializer(std::ostream& stream) : stream_(stream) {}
EdgeErrorSerializer::~EdgeErrorSerializer() {}
void EdgeErrorSerializer::serialize(const dvo_slam::KeyframeGraph& map) {
  stream_ << "# edges of the pose graph (tests/dropin): id vertex0 vertex1 level chi2 kernel_weight tx ty tz qx qy qz qw (measurement)" << std::endl;
  std::map<int, const g2o::EdgeSE3*> ordered;
  for (g2o::HyperGraph::EdgeSet::const_iterator it = map.graph().edges().begin(); it != map.graph().edges().end(); ++it)
    ordered[(*it)->id()] = static_cast<const g2o::EdgeSE3*>(*it);
  stream_.precision(9);
  for (std::map<int, const g2o::EdgeSE3*>::const_iterator it = ordered.begin(); it != ordered.end(); ++it) {
    const g2o::EdgeSE3* e = it->second;
    Eigen::Vector3d rho;
    rho.setOnes();
    if (e->robustKernel()) e->robustKernel()->robustify(e->chi2(), rho);
    const Eigen::Isometry3d& m = e->measurement();
    const Eigen::Quaterniond q(m.rotation());
    stream_ << e->id() << " " << e->vertex(0)->id() << " " << e->vertex(1)->id() << " " << e->level() << " " << e->chi2() << " " << rho(1) << " " << m(0, 3) << " "
            << m(1, 3) << " " << m(2, 3) << " " << q.x() << " " << q.y() << " " << q.z() << " " << q.w() << std::endl;
  }
}
}  // namespace serialization
}  // namespace dvo_slam
