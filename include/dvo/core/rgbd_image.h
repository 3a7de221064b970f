// dvo/core/rgbd_image.h -- RgbdCamera / RgbdCameraPyramid / RgbdImage / RgbdImagePyramid with the public surface of
// dvo_core/include/dvo/core/rgbd_image.h:90-262, backed by device-resident pyramids of libdvo_hip (include/dvo_hip.h).
//
// What a caller of the reference can rely on here:
//  * Same class names, constructors, methods, smart-pointer typedefs (boost::shared_ptr when boost is installed) and the same
//    PUBLIC FIELDS of RgbdImage (intensity, depth, *_dx, *_dy, rgb, normals, angles, pointcloud, acceleration, width, height,
//    timestamp): the reference's callers read and write them directly (dvo_benchmark/src/benchmark_slam.cpp:90, 334-339).
//  * The derived planes live on the GPU.  The fields are HOST MIRRORS: level 0 `intensity` / `depth` are the caller's own
//    matrices (no copy), `rgb` is whatever the caller stores; everything else is filled when the method that produces it in the
//    reference is called AND host mirrors are enabled (RgbdImage::hostMirrors(true), off by default), or on demand with
//    RgbdImage::syncHostMirrors().  Nothing on the alignment path reads them, and LocalTracker calls buildPointCloud() /
//    buildAccelerationStructure() on every level of every frame (dvo_slam/src/local_tracker.cpp:163-169): a download per call
//    would put a PCIe round trip on the tracking path for planes nobody looks at.
//  * calculateDerivatives() / buildPointCloud() are by-products of the device kernels; buildAccelerationStructure() starts the
//    asynchronous device build of the level's sampling planes, so local_tracker.cpp:163-169 compiles and behaves unchanged.
//  * No exceptions, like the reference: a device failure prints the library's message and aborts (the reference asserts on
//    misuse), unless an error handler is installed with DeviceContext::setErrorHandler (tests install one that throws).
#pragma once

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../dvo_hip.h"
#include "datatypes.h"
#include "intrinsic_matrix.h"

namespace dvo {
namespace core {

// One engine context per (device, host thread group).  current() is what new pyramids are bound to: the calling thread's
// context if one was set (set_current / Scope), else the process-wide context of device 0.  forDevice(i) hands out the
// process-wide context of device i (created on first use), so that a caller can spread independent frame pairs over the GPUs
// of a node inside one process, the reference's own model (dvo_slam/src/keyframe_graph.cpp:576-593 runs them on a TBB pool).
// The C library serialises concurrent calls on one context, so trackers on several threads may share it.
class DeviceContext {
 public:
  typedef void (*ErrorHandler)(const char* what, const char* message);

  static dvo_hip_context* current() {
    dvo_hip_context*& c = slot();
    return c ? c : forDevice(0);
  }
  static void set_current(dvo_hip_context* c) { slot() = c; }
  static dvo_hip_context* forDevice(int device) {
    std::lock_guard<std::mutex> lock(registry_mutex());
    std::vector<dvo_hip_context*>& r = registry();
    if (device < 0) device = 0;
    if (size_t(device) >= r.size()) r.resize(size_t(device) + 1, 0);
    if (!r[size_t(device)]) {
      const int rc = dvo_hip_context_create(device, &r[size_t(device)]);
      if (rc != DVO_HIP_OK) fail("dvo_hip_context_create", dvo_hip_last_error(0));
    }
    return r[size_t(device)];
  }
  static int deviceCount() { return dvo_hip_device_count(); }
  // a further, independent context on `device` (own streams and workspace), owned by the caller: dvo_hip_context_destroy
  static dvo_hip_context* createAdditional(int device) {
    dvo_hip_context* c = 0;
    if (dvo_hip_context_create(device, &c) != DVO_HIP_OK) fail("dvo_hip_context_create", dvo_hip_last_error(0));
    return c;
  }
  // binds the calling thread to a device (or to one particular context) for the lifetime of the object
  class Scope {
   public:
    explicit Scope(int device) : previous_(slot()) { slot() = forDevice(device); }
    explicit Scope(dvo_hip_context* context) : previous_(slot()) { slot() = context; }
    ~Scope() { slot() = previous_; }

   private:
    dvo_hip_context* previous_;
  };
  static void setErrorHandler(ErrorHandler h) { handler() = h; }
  static void fail(const char* what, const char* message) {
    if (handler()) {
      handler()(what, message);
      return;
    }
    std::fprintf(stderr, "dvo (MI355X engine): %s: %s\n", what, message ? message : "?");
    std::abort();
  }

 private:
  static dvo_hip_context*& slot() {
    static thread_local dvo_hip_context* ctx = 0;
    return ctx;
  }
  static std::vector<dvo_hip_context*>& registry() {
    static std::vector<dvo_hip_context*> r;
    return r;
  }
  static std::mutex& registry_mutex() {
    static std::mutex m;
    return m;
  }
  static ErrorHandler& handler() {
    static ErrorHandler h = 0;
    return h;
  }
};

// true when the call succeeded; otherwise the error handler has run (default: message + abort)
inline bool dvo_hip_check(dvo_hip_context* ctx, int rc, const char* what) {
  if (rc == DVO_HIP_OK) return true;
  DeviceContext::fail(what, dvo_hip_last_error(ctx));
  return false;
}

typedef dvo::compat::PointCloud PointCloud;

class RgbdImage;
typedef dvo::compat::shared_ptr<RgbdImage> RgbdImagePtr;
class RgbdImagePyramid;
typedef dvo::compat::shared_ptr<RgbdImagePyramid> RgbdImagePyramidPtr;

class RgbdCamera {
 public:
  RgbdCamera(size_t width, size_t height, const IntrinsicMatrix& intrinsics) : width_(width), height_(height), intrinsics_(intrinsics) {}
  ~RgbdCamera() {}
  size_t width() const { return width_; }
  size_t height() const { return height_; }
  const IntrinsicMatrix& intrinsics() const { return intrinsics_; }
  // a free-standing image of this camera (rgbd_image.cpp:206-232): host fields only, no device pyramid behind it
  inline RgbdImagePtr create(const dvo::compat::ImageMat& intensity, const dvo::compat::ImageMat& depth) const;
  inline RgbdImagePtr create() const;
  // p = ((x - ox) / fx, (y - oy) / fy, 1, 0) * z with w = 1  (rgbd_image.cpp:186-204, 245-262)
  void buildPointCloud(const dvo::compat::ImageMat& depth, PointCloud& pointcloud) const {
    assert(size_t(dvo::compat::image_rows(depth)) == height_ && size_t(dvo::compat::image_cols(depth)) == width_);
    dvo::compat::pointcloud_resize(pointcloud, width_ * height_);
    const float* z = dvo::compat::image_ptr(depth);
    float* out = dvo::compat::pointcloud_ptr(pointcloud);
    for (size_t y = 0; y < height_; ++y) {
      const float ty = (float(y) - intrinsics_.oy()) / intrinsics_.fy();
      for (size_t x = 0; x < width_; ++x, ++z, out += 4) {
        const float tx = (float(x) - intrinsics_.ox()) / intrinsics_.fx();
        out[0] = tx * *z;
        out[1] = ty * *z;
        out[2] = *z;
        out[3] = 1.0f;
      }
    }
  }

 private:
  size_t width_, height_;
  IntrinsicMatrix intrinsics_;
};
typedef dvo::compat::shared_ptr<RgbdCamera> RgbdCameraPtr;
typedef dvo::compat::shared_ptr<const RgbdCamera> RgbdCameraConstPtr;

class RgbdCameraPyramid {
 public:
  RgbdCameraPyramid(const RgbdCamera& base) { levels_.push_back(RgbdCameraPtr(new RgbdCamera(base))); }
  RgbdCameraPyramid(size_t base_width, size_t base_height, const IntrinsicMatrix& base_intrinsics) {
    levels_.push_back(RgbdCameraPtr(new RgbdCamera(base_width, base_height, base_intrinsics)));
  }
  ~RgbdCameraPyramid() {}
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
  const RgbdCamera& level(size_t level) const { return *levels_[level]; }
  size_t numLevels() const { return levels_.size(); }
  // intensity CV_32FC1 0..255, depth CV_32FC1 metres with NaN = invalid (asserts at rgbd_image.cpp:350, 356)
  inline RgbdImagePyramidPtr create(const dvo::compat::ImageMat& base_intensity, const dvo::compat::ImageMat& base_depth);

 private:
  std::vector<RgbdCameraPtr> levels_;
};
typedef dvo::compat::shared_ptr<RgbdCameraPyramid> RgbdCameraPyramidPtr;
typedef dvo::compat::shared_ptr<const RgbdCameraPyramid> RgbdCameraPyramidConstPtr;

class RgbdImage {
 public:
  typedef dvo::core::PointCloud PointCloud;
  typedef dvo::compat::Vec8f Vec8f;

  RgbdImage(const RgbdCamera& camera) : width(camera.width()), height(camera.height()), timestamp(0), owner_(0), level_(0), camera_(camera) {}
  virtual ~RgbdImage() {}
  const RgbdCamera& camera() const { return camera_; }

  // rgbd_image.h:161-179 -- host mirrors, see the header comment
  dvo::compat::ImageMat intensity, intensity_dx, intensity_dy;
  dvo::compat::ImageMat depth, depth_dx, depth_dy;
  dvo::compat::ImageMat normals, angles;
  dvo::compat::ImageMat rgb;
  PointCloud pointcloud;
  dvo::compat::AccelerationMat acceleration;
  size_t width, height;
  double timestamp;

  bool hasIntensity() const { return !dvo::compat::image_empty(intensity); }
  bool hasDepth() const { return !dvo::compat::image_empty(depth); }
  bool hasRgb() const { return !dvo::compat::image_empty(rgb); }
  void initialize() {}
  // derivatives and the 3-D points are by-products of the device kernels
  void calculateDerivatives() { if (hostMirrors()) syncHostMirrors(MirrorDerivatives); }
  bool calculateIntensityDerivatives() { calculateDerivatives(); return true; }
  void calculateDepthDerivatives() { calculateDerivatives(); }
  void calculateNormals() {}
  void buildPointCloud() { if (hostMirrors()) syncHostMirrors(MirrorPlanes | MirrorPointCloud); }
  // rgbd_image.cpp:534-543.  On the device: the current-frame sampling planes of this level, built asynchronously on the
  // context's build stream (dvo_hip_frames_prepare) -- what LocalTracker does with a new image before handing it to its
  // trackers (local_tracker.cpp:163-169).  Optional: match() builds whatever is missing.
  inline void buildAccelerationStructure();
  bool inImage(const float& x, const float& y) const { return x >= 0 && x < float(width) && y >= 0 && y < float(height); }

  // ---- not in the reference: control over the host mirrors ---------------------------------------------------------
  enum { MirrorPlanes = 1, MirrorDerivatives = 2, MirrorPointCloud = 4, MirrorAcceleration = 8, MirrorAll = 15 };
  static bool& hostMirrors() {   // process-wide; off = the fields of levels > 0 stay empty until syncHostMirrors()
    static bool enabled = false;
    return enabled;
  }
  static void hostMirrors(bool on) { hostMirrors() = on; }
  inline void syncHostMirrors(unsigned what = MirrorAll);

 private:
  friend class RgbdImagePyramid;
  RgbdImage(RgbdImagePyramid* owner, int level, const RgbdCamera& camera)
      : width(camera.width()), height(camera.height()), timestamp(0), owner_(owner), level_(level), camera_(camera) {}
  inline void download(int plane, dvo::compat::ImageMat& m);
  RgbdImagePyramid* owner_;   // null for free-standing images (RgbdCamera::create)
  int level_;
  const RgbdCamera& camera_;
};

class RgbdImagePyramid {
 public:
  typedef dvo::compat::shared_ptr<dvo::core::RgbdImagePyramid> Ptr;

  RgbdImagePyramid(RgbdCameraPyramid& camera, const dvo::compat::ImageMat& intensity, const dvo::compat::ImageMat& depth)
      : camera_(camera), intensity_(intensity), depth_(depth), frame_(0), timestamp_(0) {
    assert(dvo::compat::image_is_float1(intensity) && dvo::compat::image_is_float1(depth));
    assert(dvo::compat::image_rows(intensity) == dvo::compat::image_rows(depth) && dvo::compat::image_cols(intensity) == dvo::compat::image_cols(depth));
    ctx_ = DeviceContext::current();
  }
  virtual ~RgbdImagePyramid() { if (frame_) dvo_hip_frame_destroy(ctx_, frame_); }
  RgbdImagePyramid(const RgbdImagePyramid&) = delete;
  RgbdImagePyramid& operator=(const RgbdImagePyramid&) = delete;

  void compute(const size_t num_levels) { build(num_levels); }   // deprecated alias in the reference too
  // rgbd_image.cpp:156-172: idempotent, only ever grows.  The device frame is created once, with every level the image size
  // admits (a quarter of a level each: the pyramid above what a caller asks for is a few per cent of the frame), so asking for
  // more levels later neither re-uploads anything nor invalidates a handle a PointSelection holds; RgbdImage& references
  // returned by level() stay valid (the vector only grows and holds pointers).
  void build(const size_t num_levels) {
    camera_.build(num_levels);
    if (!frame_) {
      const RgbdCamera& c0 = camera_.level(0);
      int all = 1;
      while (all < DVO_HIP_MAX_LEVELS && (c0.width() >> all) >= 2 && (c0.height() >> all) >= 2) ++all;
      if (size_t(all) < num_levels) all = int(num_levels);   // frame_create rejects it with a message
      const float K[4] = {c0.intrinsics().fx(), c0.intrinsics().fy(), c0.intrinsics().ox(), c0.intrinsics().oy()};
      if (!dvo_hip_check(ctx_, dvo_hip_frame_create_f32(ctx_, int(c0.width()), int(c0.height()), K, dvo::compat::image_ptr(intensity_),
                                                        dvo::compat::image_ptr(depth_), all, &frame_), "dvo_hip_frame_create_f32"))
        frame_ = 0;
    }
    for (size_t l = levels_.size(); l < num_levels; ++l) {
      levels_.push_back(RgbdImagePtr(new RgbdImage(this, int(l), camera_.level(l))));
      levels_[l]->timestamp = timestamp_;
      if (l == 0) {   // the caller's own matrices: valid without a download
        levels_[0]->intensity = intensity_;
        levels_[0]->depth = depth_;
      }
    }
  }
  RgbdImage& level(size_t idx) {
    build(idx + 1);
    return *levels_[idx];
  }
  double timestamp() const { return levels_.empty() ? timestamp_ : levels_[0]->timestamp; }   // rgbd_image.cpp: level(0).timestamp
  void timestamp(double t) { timestamp_ = t; for (size_t l = 0; l < levels_.size(); ++l) levels_[l]->timestamp = t; }

  // engine handles (not in the reference API)
  dvo_hip_frame* device_frame() { build(1); return frame_; }
  size_t builtLevels() const { return levels_.size(); }
  dvo_hip_context* device_context() { return ctx_; }
  RgbdCameraPyramid& cameraPyramid() { return camera_; }

 private:
  RgbdCameraPyramid& camera_;
  dvo::compat::ImageMat intensity_, depth_;
  dvo_hip_context* ctx_;
  dvo_hip_frame* frame_;
  std::vector<RgbdImagePtr> levels_;
  double timestamp_;
};

inline RgbdImagePtr RgbdCamera::create(const dvo::compat::ImageMat& intensity, const dvo::compat::ImageMat& depth) const {
  RgbdImagePtr result(new RgbdImage(*this));
  result->intensity = intensity;
  result->depth = depth;
  return result;
}

inline RgbdImagePtr RgbdCamera::create() const { return RgbdImagePtr(new RgbdImage(*this)); }

inline RgbdImagePyramidPtr RgbdCameraPyramid::create(const dvo::compat::ImageMat& base_intensity, const dvo::compat::ImageMat& base_depth) {
  return RgbdImagePyramidPtr(new RgbdImagePyramid(*this, base_intensity, base_depth));
}

inline void RgbdImage::buildAccelerationStructure() {
  if (!owner_) return;
  // The callers walk the levels of a new image (dvo_slam/src/local_tracker.cpp:163-169): the first call prepares every level that has
  // been built from this one upwards in ONE background launch, the calls for the other levels find their planes in place.  The
  // reference role is prepared along with it, speculatively (negative thresholds: the selection thresholds of the context's last
  // match, and only if the image holds no selection yet -- a key frame that was selected for a tracker's thresholds keeps it):
  // in a tracking front end every image is the odometry tracker's reference one frame later (local_tracker.cpp:198-206), and
  // deriving it then would sit inside that match.
  dvo_hip_config c = {};
  c.last_level = level_;
  c.first_level = int(owner_->builtLevels()) > level_ ? int(owner_->builtLevels()) - 1 : level_;
  c.max_iterations_per_level = 1;
  dvo_hip_frame* one[1] = {owner_->device_frame()};
  dvo_hip_check(owner_->device_context(), dvo_hip_frames_prepare(owner_->device_context(), 1, one, DVO_HIP_ROLE_CURRENT, &c), "dvo_hip_frames_prepare");
  c.intensity_derivative_threshold = c.depth_derivative_threshold = -1.0f;
  dvo_hip_check(owner_->device_context(), dvo_hip_frames_prepare(owner_->device_context(), 1, one, DVO_HIP_ROLE_REFERENCE, &c), "dvo_hip_frames_prepare");
  if (hostMirrors()) syncHostMirrors(MirrorAcceleration);
}

inline void RgbdImage::download(int plane, dvo::compat::ImageMat& m) {
  m = dvo::compat::image_create(int(height), int(width));
  dvo_hip_check(owner_->device_context(), dvo_hip_frame_download_plane(owner_->device_context(), owner_->device_frame(), level_, plane,
                                                                       dvo::compat::image_ptr_mut(m)), "dvo_hip_frame_download_plane");
}

// fills the requested host fields from the device planes (bit-identical to what the reference computes on the host:
// tests/test_gpu_parity.py compares every plane with the oracle)
inline void RgbdImage::syncHostMirrors(unsigned what) {
  if (!owner_) {   // free-standing image: only the point cloud can be derived, on the host
    if ((what & MirrorPointCloud) && hasDepth()) camera_.buildPointCloud(depth, pointcloud);
    return;
  }
  const bool need_planes = (what & (MirrorPlanes | MirrorPointCloud | MirrorAcceleration)) != 0;
  if (need_planes && level_ > 0 && dvo::compat::image_empty(intensity)) {
    download(0, intensity);
    download(1, depth);
  }
  if ((what & (MirrorDerivatives | MirrorAcceleration)) && dvo::compat::image_empty(intensity_dx)) {
    download(2, intensity_dx);
    download(3, intensity_dy);
    download(4, depth_dx);
    download(5, depth_dy);
  }
  if (what & MirrorPointCloud) camera_.buildPointCloud(depth, pointcloud);
  if (what & MirrorAcceleration) {   // {I, Z, Idx, Idy, Zdx, Zdy, 0, 0} interleaved (rgbd_image.cpp:534-543)
    const dvo::compat::ImageMat* planes[6] = {&intensity, &depth, &intensity_dx, &intensity_dy, &depth_dx, &depth_dy};
    dvo::compat::acceleration_fill(acceleration, int(height), int(width), planes);
  }
}

}  // namespace core
}  // namespace dvo
