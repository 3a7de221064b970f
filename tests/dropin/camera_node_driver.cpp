// tests/dropin/camera_node_driver.cpp -- TEST INFRASTRUCTURE: drives the reference's live-camera front end
// dvo_ros::CameraDenseTracker (dvo_ros/src/camera_dense_tracking.cpp + camera_base.cpp, compiled UNMODIFIED against this engine's
// facade, tests/dropin/Makefile) without a ROS master: a reconfigure call switches the synchronised image stream on
// (handleConfig, camera_dense_tracking.cpp:106-172), then frames are handed to handleImages (:187-309) as sensor_msgs -- 8-bit
// grey + 16-bit millimetre depth, the path through cv_bridge and SurfacePyramid::convertRawDepthImageSse -- and the accumulated
// camera pose the node broadcasts on tf after every frame (:310-323) is returned.
#include <cstring>
#include <memory>

#include <dvo_ros/camera_dense_tracking.h>
#include <tf/transform_broadcaster.h>
#include <tf_conversions/tf_eigen.h>

extern "C" int dropin_camera_node(int n, int w, int h, const float K[4], const unsigned char* const* grey, const unsigned short* const* depth_mm,
                                  int coarsest, int finest, int max_iterations, double precision, double mu, int use_initial_estimate,
                                  double* out_pose /* n x 16, row-major, pose after each frame */) {
  const int sent_before = tf::TransformBroadcaster::count();      // (the counter is shared by every library of the process that holds this stand-in)
  ros::NodeHandle nh, nh_private("~");
  dvo_ros::CameraDenseTracker node(nh, nh_private);
  dvo_ros::CameraDenseTrackerConfig cfg = dvo_ros::CameraDenseTrackerConfig::__getDefault__();
  cfg.run_dense_tracking = true;
  cfg.coarsest_level = coarsest;
  cfg.finest_level = finest;
  cfg.max_iterations = max_iterations;
  cfg.precision = precision;
  cfg.mu = mu;
  cfg.use_initial_estimate = use_initial_estimate != 0;
  cfg.reconstruction = false;
  node.handleConfig(cfg, dvo_ros::CameraDenseTracker_RunDenseTracking | dvo_ros::CameraDenseTracker_ConfigParam | dvo_ros::CameraDenseTracker_MiscParam);
  std::shared_ptr<sensor_msgs::CameraInfo> info(new sensor_msgs::CameraInfo);
  info->width = uint32_t(w);
  info->height = uint32_t(h);
  info->P[0] = K[0]; info->P[5] = K[1]; info->P[2] = K[2]; info->P[6] = K[3];      // (reset() reads P, camera_dense_tracking.cpp:89)
  for (int k = 0; k < n; ++k) {
    std::shared_ptr<sensor_msgs::Image> rgb(new sensor_msgs::Image), depth(new sensor_msgs::Image);
    rgb->width = depth->width = uint32_t(w);
    rgb->height = depth->height = uint32_t(h);
    rgb->encoding = "mono8"; rgb->step = uint32_t(w);
    rgb->data.assign(grey[k], grey[k] + size_t(w) * h);
    depth->encoding = "16UC1"; depth->step = uint32_t(w) * 2;
    depth->data.resize(size_t(w) * h * 2);
    std::memcpy(depth->data.data(), depth_mm[k], size_t(w) * h * 2);
    rgb->header.stamp = depth->header.stamp = ros::Time(double(k + 1) / 30.0);
    node.handleImages(rgb, depth, info, info);
    Eigen::Affine3d pose;
    pose.setIdentity();
    if (tf::TransformBroadcaster::count() > sent_before) tf::TransformTFToEigen(tf::TransformBroadcaster::last(), pose);
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) out_pose[size_t(k) * 16 + a * 4 + b] = pose.matrix()(a, b);
  }
  return tf::TransformBroadcaster::count() - sent_before;
}
