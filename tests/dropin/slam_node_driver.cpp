// tests/dropin/slam_node_driver.cpp -- TEST INFRASTRUCTURE: drives the reference's live SLAM node dvo_slam::CameraKeyframeTracker
// (dvo_slam/src/camera_keyframe_tracking.cpp + dvo_ros/src/camera_base.cpp, compiled UNMODIFIED against this engine's facade, with the
// reference's front end and its keyframe graph) without a ROS master: the two reconfigure callbacks set the tracker and the SLAM
// parameters (handleTrackerConfig :108-152, handleSlamConfig :154-172), frames arrive as sensor messages (handleImages :186-286: mono8 +
// 32FC1 metres), the pose the node broadcasts after every frame is returned, and a last reconfigure call with graph_opt_final runs
// KeyframeTracker::finish() -> KeyframeGraph::finalOptimization().
#include <cstring>
#include <memory>

#include <dvo_slam/camera_keyframe_tracking.h>
#include <tf/transform_broadcaster.h>
#include <tf_conversions/tf_eigen.h>

extern int g_slam_node_map_changes, g_slam_node_keyframes, g_slam_node_edges, g_slam_node_loop_closures;

extern "C" int dropin_slam_node(int n, int w, int h, const float K[4], const unsigned char* const* grey, const float* const* depth_m, const double* stamps,
                                int coarsest, int finest, int max_iterations, double precision, double mu, int use_initial_estimate,
                                double max_translational_distance, double max_rotational_distance, double min_entropy_ratio_coarse,
                                double min_entropy_ratio_fine, double min_constraint_ratio, int robust_kernel, int multithreading, int finish,
                                double* out_pose /* n x 16, row-major */, int* out_counts /* 8 */) {
  g_slam_node_map_changes = g_slam_node_keyframes = g_slam_node_edges = g_slam_node_loop_closures = 0;
  const int sent_before = tf::TransformBroadcaster::count();
  ros::NodeHandle nh, nh_private("~");
  dvo_slam::CameraKeyframeTracker node(nh, nh_private);
  dvo_slam::KeyframeSlamConfig slam = dvo_slam::KeyframeSlamConfig::__getDefault__();
  slam.max_translational_distance = max_translational_distance;
  slam.max_rotational_distance = max_rotational_distance;
  slam.constraint_min_entropy_ratio_coarse = min_entropy_ratio_coarse;
  slam.constraint_min_entropy_ratio_fine = min_entropy_ratio_fine;
  slam.constraint_min_eq_sys_constraint_ratio = min_constraint_ratio;
  slam.graph_opt_robust = robust_kernel != 0;
  slam.use_multithreading = multithreading != 0;
  node.handleSlamConfig(slam, 1);
  dvo_ros::CameraDenseTrackerConfig cfg = dvo_ros::CameraDenseTrackerConfig::__getDefault__();
  cfg.run_dense_tracking = true;
  cfg.coarsest_level = coarsest;
  cfg.finest_level = finest;
  cfg.max_iterations = max_iterations;
  cfg.precision = precision;
  cfg.mu = mu;
  cfg.use_initial_estimate = use_initial_estimate != 0;
  cfg.reconstruction = false;
  node.handleTrackerConfig(cfg, dvo_ros::CameraDenseTracker_RunDenseTracking | dvo_ros::CameraDenseTracker_ConfigParam | dvo_ros::CameraDenseTracker_MiscParam);
  std::shared_ptr<sensor_msgs::CameraInfo> info(new sensor_msgs::CameraInfo);
  info->width = uint32_t(w);
  info->height = uint32_t(h);
  info->P[0] = K[0]; info->P[5] = K[1]; info->P[2] = K[2]; info->P[6] = K[3];      // (reset() reads P, camera_keyframe_tracking.cpp:88)
  for (int k = 0; k < n; ++k) {
    std::shared_ptr<sensor_msgs::Image> rgb(new sensor_msgs::Image), depth(new sensor_msgs::Image);
    rgb->width = depth->width = uint32_t(w);
    rgb->height = depth->height = uint32_t(h);
    rgb->encoding = "mono8"; rgb->step = uint32_t(w);
    rgb->data.assign(grey[k], grey[k] + size_t(w) * h);
    depth->encoding = "32FC1"; depth->step = uint32_t(w) * 4;
    depth->data.resize(size_t(w) * h * 4);
    std::memcpy(depth->data.data(), depth_m[k], size_t(w) * h * 4);
    rgb->header.stamp = depth->header.stamp = ros::Time(stamps[k]);
    node.handleImages(rgb, depth, info, info);
    Eigen::Affine3d pose;
    pose.setIdentity();
    if (tf::TransformBroadcaster::count() > sent_before) tf::TransformTFToEigen(tf::TransformBroadcaster::last(), pose);
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) out_pose[size_t(k) * 16 + a * 4 + b] = pose.matrix()(a, b);
  }
  out_counts[0] = g_slam_node_map_changes; out_counts[1] = g_slam_node_keyframes; out_counts[2] = g_slam_node_edges; out_counts[3] = g_slam_node_loop_closures;
  if (finish) {
    slam.graph_opt_final = true;
    node.handleSlamConfig(slam, 1);                                               // -> KeyframeTracker::finish() -> finalOptimization()
  }
  out_counts[4] = g_slam_node_map_changes; out_counts[5] = g_slam_node_keyframes; out_counts[6] = g_slam_node_edges; out_counts[7] = g_slam_node_loop_closures;
  return tf::TransformBroadcaster::count() - sent_before;
}
