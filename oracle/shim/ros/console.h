// oracle/shim/ros/console.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  Logging macros of the reference's callers: no-ops.
#pragma once
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_WARN_COND(cond, ...) ((void)(cond))
#define ROS_WARN_STREAM_NAMED(...) ((void)0)
#define ROS_INFO(...) ((void)0)
