// tests/dropin/shim (see boost/thread/mutex.hpp): the ROS logging macros, printing to stderr.
#pragma once
#include <cstdio>
#include <iostream>
#define ROS_ERROR(...) (std::fprintf(stderr, "[ERROR] " __VA_ARGS__), std::fputc('\n', stderr))
#define ROS_WARN(...) (std::fprintf(stderr, "[WARN] " __VA_ARGS__), std::fputc('\n', stderr))
#define ROS_INFO(...) (std::fprintf(stderr, "[INFO] " __VA_ARGS__), std::fputc('\n', stderr))
#define ROS_WARN_COND(cond, ...) do { if (cond) ROS_WARN(__VA_ARGS__); } while (0)
#define ROS_WARN_STREAM_NAMED(name, args) do { std::cerr << "[WARN] [" << name << "] " << args << std::endl; } while (0)
#define ROS_INFO_STREAM(args) do { std::cerr << "[INFO] " << args << std::endl; } while (0)
#define ROS_WARN_STREAM(args) do { std::cerr << "[WARN] " << args << std::endl; } while (0)
