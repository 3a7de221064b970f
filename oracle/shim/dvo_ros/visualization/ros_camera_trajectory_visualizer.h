// oracle/shim/dvo_ros/visualization/ros_camera_trajectory_visualizer.h -- TEST INFRASTRUCTURE: name only (RViz output is out of scope).
#pragma once
namespace dvo_ros { namespace visualization { class RosCameraTrajectoryVisualizer; } }
