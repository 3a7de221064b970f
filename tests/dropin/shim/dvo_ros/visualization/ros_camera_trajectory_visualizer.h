// tests/dropin/shim (see boost/thread/mutex.hpp): the RViz visualizer of dvo_ros (display only, out of scope): the class
// benchmark_slam.cpp constructs when a visualization is asked for; this build has none and the members abort when reached.
#pragma once
#include <dvo/visualization/camera_trajectory_visualizer.h>
#include <ros/ros.h>
namespace dvo_ros {
namespace visualization {
class RosCameraTrajectoryVisualizer : public dvo::visualization::CameraTrajectoryVisualizerInterface {
 public:
  RosCameraTrajectoryVisualizer(ros::NodeHandle&);
  virtual ~RosCameraTrajectoryVisualizer();
  virtual dvo::visualization::CameraVisualizer::Ptr camera(std::string name);
  virtual dvo::visualization::TrajectoryVisualizer::Ptr trajectory(std::string name);
  virtual void reset();
};
}  // namespace visualization
}  // namespace dvo_ros
