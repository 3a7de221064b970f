// dvo/core/rgbd_image.h -- RgbdCamera / RgbdCameraPyramid / RgbdImage / RgbdImagePyramid with the signatures of
// dvo_core/include/dvo/core/rgbd_image.h:99-262, backed by device-resident pyramids of libdvo_hip (include/dvo_hip.h).
//
// Differences a caller can observe: derived planes (derivatives, point cloud, "acceleration structure") live on the
// GPU; the public cv::Mat-like fields of RgbdImage are HOST MIRRORS filled on first access through the accessor
// functions of the same name (intensity(), depth(), intensity_dx() ...).  buildPointCloud() / calculateDerivatives() are kept
// as no-ops (by-products of the device kernels) and buildAccelerationStructure() starts the asynchronous device build of the
// level's sampling planes, so dvo_slam/src/local_tracker.cpp:163-169 compiles and behaves unchanged.
#pragma once

#include <cassert>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../dvo_hip.h"
#include "datatypes.h"
#include "intrinsic_matrix.h"

namespace dvo {
namespace core {

// process-wide default context (one per thread of the reference == one DenseTracker per thread; share explicitly
// with DeviceContext::set_current when several trackers must use distinct streams)
class DeviceContext {
 public:
  static dvo_hip_context* current() {
    dvo_hip_context*& c = slot();
    if (!c) {
      int rc = dvo_hip_context_create(0, &c);
      if (rc != DVO_HIP_OK) throw std::runtime_error(std::string("dvo_hip_context_create: ") + dvo_hip_last_error(0));
    }
    return c;
  }
  static void set_current(dvo_hip_context* c) { slot() = c; }

 private:
  static dvo_hip_context*& slot() {
    static thread_local dvo_hip_context* ctx = 0;
    return ctx;
  }
};

inline void dvo_hip_check(dvo_hip_context* ctx, int rc, const char* what) {
  if (rc != DVO_HIP_OK) throw std::runtime_error(std::string(what) + ": " + dvo_hip_last_error(ctx));
}

class RgbdImage;
typedef std::shared_ptr<RgbdImage> RgbdImagePtr;
class RgbdImagePyramid;
typedef std::shared_ptr<RgbdImagePyramid> RgbdImagePyramidPtr;

class RgbdCamera {
 public:
  RgbdCamera(size_t width, size_t height, const IntrinsicMatrix& intrinsics) : width_(width), height_(height), intrinsics_(intrinsics) {}
  size_t width() const { return width_; }
  size_t height() const { return height_; }
  const IntrinsicMatrix& intrinsics() const { return intrinsics_; }

 private:
  size_t width_, height_;
  IntrinsicMatrix intrinsics_;
};
typedef std::shared_ptr<RgbdCamera> RgbdCameraPtr;

class RgbdCameraPyramid {
 public:
  RgbdCameraPyramid(size_t base_width, size_t base_height, const IntrinsicMatrix& base_intrinsics) {
    levels_.push_back(RgbdCameraPtr(new RgbdCamera(base_width, base_height, base_intrinsics)));
  }
  // rgbd_image.cpp:283-296: halves width, height and the whole intrinsic matrix per level
  void build(size_t levels) {
    for (size_t idx = levels_.size(); idx < levels; ++idx) {
      IntrinsicMatrix k(levels_[idx - 1]->intrinsics());
      k.scale(0.5f);
      levels_.push_back(RgbdCameraPtr(new RgbdCamera(levels_[idx - 1]->width() / 2, levels_[idx - 1]->height() / 2, k)));
    }
  }
  const RgbdCamera& level(size_t level) {
    build(level + 1);
    return *levels_[level];
  }
  size_t numLevels() const { return levels_.size(); }
  // intensity CV_32FC1 0..255, depth CV_32FC1 metres with NaN = invalid (asserts at rgbd_image.cpp:350, 356)
  inline RgbdImagePyramidPtr create(const dvo::compat::ImageMat& base_intensity, const dvo::compat::ImageMat& base_depth);

 private:
  std::vector<RgbdCameraPtr> levels_;
};
typedef std::shared_ptr<RgbdCameraPyramid> RgbdCameraPyramidPtr;

class RgbdImage {
 public:
  RgbdImage(RgbdImagePyramid* owner, int level, const RgbdCamera& camera) : width(camera.width()), height(camera.height()), timestamp(0),
      owner_(owner), level_(level), camera_(camera) {}
  const RgbdCamera& camera() const { return camera_; }
  size_t width, height;
  double timestamp;
  // host mirrors of the device planes (rgbd_image.h:161-179), downloaded on first use
  const dvo::compat::ImageMat& intensity() { return plane(0); }
  const dvo::compat::ImageMat& depth() { return plane(1); }
  const dvo::compat::ImageMat& intensity_dx() { return plane(2); }
  const dvo::compat::ImageMat& intensity_dy() { return plane(3); }
  const dvo::compat::ImageMat& depth_dx() { return plane(4); }
  const dvo::compat::ImageMat& depth_dy() { return plane(5); }
  // derivatives and the 3-D points are by-products of the device kernels; kept so callers compile unchanged
  void calculateDerivatives() {}
  void buildPointCloud() {}
  // rgbd_image.cpp:534-543.  On the device: the current-frame sampling planes of this level, built asynchronously on the
  // context's build stream (dvo_hip_frames_prepare) -- what LocalTracker does with a new image before handing it to its
  // trackers (local_tracker.cpp:163-169).  Optional: match() builds whatever is missing.
  inline void buildAccelerationStructure();

 private:
  inline const dvo::compat::ImageMat& plane(int idx);
  RgbdImagePyramid* owner_;
  int level_;
  const RgbdCamera& camera_;
  dvo::compat::ImageMat planes_[6];
};

class RgbdImagePyramid {
 public:
  typedef std::shared_ptr<RgbdImagePyramid> Ptr;
  RgbdImagePyramid(RgbdCameraPyramid& camera, const dvo::compat::ImageMat& intensity, const dvo::compat::ImageMat& depth)
      : camera_(camera), intensity_(intensity), depth_(depth), frame_(0), built_levels_(0), timestamp_(0) {
    assert(dvo::compat::image_is_float1(intensity) && dvo::compat::image_is_float1(depth));
    assert(dvo::compat::image_rows(intensity) == dvo::compat::image_rows(depth) && dvo::compat::image_cols(intensity) == dvo::compat::image_cols(depth));
    ctx_ = DeviceContext::current();
  }
  ~RgbdImagePyramid() { if (frame_) dvo_hip_frame_destroy(ctx_, frame_); }
  RgbdImagePyramid(const RgbdImagePyramid&) = delete;
  RgbdImagePyramid& operator=(const RgbdImagePyramid&) = delete;

  void compute(const size_t num_levels) { build(num_levels); }   // deprecated alias in the reference too
  // rgbd_image.cpp:156-172: idempotent, only ever grows
  void build(const size_t num_levels) {
    if (built_levels_ >= num_levels) return;
    camera_.build(num_levels);
    if (frame_) dvo_hip_frame_destroy(ctx_, frame_);
    frame_ = 0;
    const RgbdCamera& c0 = camera_.level(0);
    const float K[4] = {c0.intrinsics().fx(), c0.intrinsics().fy(), c0.intrinsics().ox(), c0.intrinsics().oy()};
    dvo_hip_check(ctx_, dvo_hip_frame_create_f32(ctx_, int(c0.width()), int(c0.height()), K, dvo::compat::image_ptr(intensity_),
                                                 dvo::compat::image_ptr(depth_), int(num_levels), &frame_), "dvo_hip_frame_create_f32");
    levels_.clear();
    for (size_t l = 0; l < num_levels; ++l) levels_.push_back(RgbdImagePtr(new RgbdImage(this, int(l), camera_.level(l))));
    for (size_t l = 0; l < num_levels; ++l) levels_[l]->timestamp = timestamp_;
    built_levels_ = num_levels;
  }
  RgbdImage& level(size_t idx) {
    build(idx + 1);
    return *levels_[idx];
  }
  double timestamp() const { return timestamp_; }
  void timestamp(double t) { timestamp_ = t; for (size_t l = 0; l < levels_.size(); ++l) levels_[l]->timestamp = t; }

  // engine handles (not in the reference API)
  dvo_hip_frame* device_frame() { return frame_; }
  dvo_hip_context* device_context() { return ctx_; }
  RgbdCameraPyramid& cameraPyramid() { return camera_; }

 private:
  RgbdCameraPyramid& camera_;
  dvo::compat::ImageMat intensity_, depth_;
  dvo_hip_context* ctx_;
  dvo_hip_frame* frame_;
  std::vector<RgbdImagePtr> levels_;
  size_t built_levels_;
  double timestamp_;
};

inline RgbdImagePyramidPtr RgbdCameraPyramid::create(const dvo::compat::ImageMat& base_intensity, const dvo::compat::ImageMat& base_depth) {
  return RgbdImagePyramidPtr(new RgbdImagePyramid(*this, base_intensity, base_depth));
}

inline void RgbdImage::buildAccelerationStructure() {
  dvo_hip_config c = {};
  c.first_level = c.last_level = level_;
  c.max_iterations_per_level = 1;
  dvo_hip_frame* one[1] = {owner_->device_frame()};
  dvo_hip_check(owner_->device_context(), dvo_hip_frames_prepare(owner_->device_context(), 1, one, DVO_HIP_ROLE_CURRENT, &c), "dvo_hip_frames_prepare");
}

inline const dvo::compat::ImageMat& RgbdImage::plane(int idx) {
  dvo::compat::ImageMat& m = planes_[idx];
  if (dvo::compat::image_rows(m) == 0) {
    m = dvo::compat::image_create(int(height), int(width));
    dvo_hip_check(owner_->device_context(), dvo_hip_frame_download_plane(owner_->device_context(), owner_->device_frame(), level_, idx,
                                                                         dvo::compat::image_ptr_mut(m)), "dvo_hip_frame_download_plane");
  }
  return m;
}

}  // namespace core
}  // namespace dvo
