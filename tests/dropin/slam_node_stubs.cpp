// tests/dropin/slam_node_stubs.cpp -- TEST INFRASTRUCTURE: what has to exist at link time for the reference's live SLAM node
// dvo_slam::CameraKeyframeTracker (dvo_slam/src/camera_keyframe_tracking.cpp, compiled UNMODIFIED, see the Makefile) besides the tracker,
// the front end and the keyframe graph:
//   * cv::cvtColor (OpenCV is not installed) through include/dvo_benchmark/image_io.h;
//   * the RViz visualiser of dvo_ros, which the node draws the current camera with every frame: the reference's own
//     NoopCameraTrajectoryVisualizer (dvo_core/src/visualization/camera_trajectory_visualizer.cpp, compiled) behind it;
//   * the graph display (constructible) and the message serialiser the node publishes the graph with on every map change
//     (dvo_slam/src/serialization/map_serializer.cpp needs ROS pose messages): it keeps what the driver reports instead.
#include <cstdlib>

#include <dvo/visualization/camera_trajectory_visualizer.h>
#include <dvo_ros/visualization/ros_camera_trajectory_visualizer.h>
#include <dvo_slam/keyframe_graph.h>
#include <dvo_slam/serialization/map_serializer.h>
#include <dvo_slam/visualization/graph_visualizer.h>

#include <dvo_benchmark/image_io.h>

int g_slam_node_map_changes = 0, g_slam_node_keyframes = 0, g_slam_node_edges = 0, g_slam_node_loop_closures = 0;

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
static dvo::visualization::NoopCameraTrajectoryVisualizer& noop() {
  static dvo::visualization::NoopCameraTrajectoryVisualizer v;
  return v;
}
RosCameraTrajectoryVisualizer::RosCameraTrajectoryVisualizer(ros::NodeHandle&) {}
RosCameraTrajectoryVisualizer::~RosCameraTrajectoryVisualizer() {}
dvo::visualization::CameraVisualizer::Ptr RosCameraTrajectoryVisualizer::camera(std::string name) { return noop().camera(name); }
dvo::visualization::TrajectoryVisualizer::Ptr RosCameraTrajectoryVisualizer::trajectory(std::string name) { return noop().trajectory(name); }
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
MessageSerializer::MessageSerializer(dvo_slam::PoseStampedArray& msg) : msg_(msg) {}
MessageSerializer::~MessageSerializer() {}
void MessageSerializer::serialize(const dvo_slam::KeyframeGraph& map) {
  g_slam_node_map_changes += 1;
  g_slam_node_keyframes = int(map.keyframes().size());
  g_slam_node_edges = int(map.graph().edges().size());
  int loops = 0;
  for (g2o::HyperGraph::EdgeSet::const_iterator it = map.graph().edges().begin(); it != map.graph().edges().end(); ++it) {
    const int a = (*it)->vertex(0)->id(), b = (*it)->vertex(1)->id();
    if (a > 0 && b > 0 && std::abs(a - b) > 1) ++loops;
  }
  g_slam_node_loop_closures = loops;
}
}  // namespace serialization
}  // namespace dvo_slam
