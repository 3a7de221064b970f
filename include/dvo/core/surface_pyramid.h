// dvo/core/surface_pyramid.h -- SurfacePyramid::convertRawDepthImage[Sse] (dvo_core/include/dvo/core/surface_pyramid.h:36-52,
// src/core/surface_pyramid.cpp:42-105): CV_16UC1 raw depth -> CV_32FC1 metres, 0 -> NaN.  Host-side entry of the reference's
// loaders (benchmark_slam.cpp:77, camera_dense_tracking.cpp:235); frames handed to the engine as raw planes
// (dvo_hip_frame_create_raw) are converted on the device instead (pyramid_kernels.hip).
#pragma once

#include <cstdint>
#include <limits>

#include "../compat.h"

namespace dvo {
namespace core {

class SurfacePyramid {
 public:
  static void convertRawDepthImage(const dvo::compat::ImageMat& input, dvo::compat::ImageMat& output, float scale) {
    const int rows = dvo::compat::image_rows(input), cols = dvo::compat::image_cols(input);
    output = dvo::compat::image_create(rows, cols);
    const uint16_t* in = dvo::compat::image_ptr_u16(input);
    float* out = dvo::compat::image_ptr_mut(output);
    const float nan = std::numeric_limits<float>::quiet_NaN();
    for (size_t i = 0; i < size_t(rows) * cols; ++i) out[i] = in[i] == 0 ? nan : float(in[i]) * scale;
  }
  static void convertRawDepthImageSse(const dvo::compat::ImageMat& input, dvo::compat::ImageMat& output, float scale) {
    convertRawDepthImage(input, output, scale);   // same values: u16 -> f32 is exact, one rounding in the multiply either way
  }
  SurfacePyramid() {}
  virtual ~SurfacePyramid() {}
};

}  // namespace core
}  // namespace dvo
