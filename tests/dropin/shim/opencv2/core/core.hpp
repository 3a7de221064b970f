// tests/dropin/shim (see boost/thread/mutex.hpp): the image I/O and colour conversion of OpenCV that benchmark_slam.cpp's load()
// calls (dvo_benchmark/src/benchmark_slam.cpp:46-93), on top of oracle/shim's cv::Mat.  Implemented in tests/dropin/benchmark_stubs.cpp
// with this repo's PNG reader and its OpenCV-exact BGR -> grey (include/dvo_benchmark/image_io.h).
#pragma once
#include_next <opencv2/core/core.hpp>
#include <string>
#define CV_BGR2GRAY 6
namespace cv {
Mat imread(const std::string& filename, int flags = 1);   // flags 1: 8-bit BGR; -1: as stored (16-bit depth PNGs stay 16-bit)
void cvtColor(const Mat& src, Mat& dst, int code, int dst_channels = 0);
}  // namespace cv
