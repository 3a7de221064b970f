// dvo/compat.h -- minimal stand-ins for the third-party types in the reference's public API, used ONLY when the real
// libraries are not installed (this build image has neither Eigen nor OpenCV nor boost).  With
// -DDVO_HIP_USE_EIGEN / -DDVO_HIP_USE_OPENCV (or when <Eigen/Geometry> / <opencv2/core/core.hpp> are found) the facade
// uses Eigen::Affine3d / cv::Mat directly and dvo_benchmark compiles against it unchanged.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Geometry>) && !defined(DVO_HIP_NO_EIGEN)
#define DVO_HIP_USE_EIGEN 1
#endif
#if __has_include(<opencv2/core/core.hpp>) && !defined(DVO_HIP_NO_OPENCV)
#define DVO_HIP_USE_OPENCV 1
#endif
#if __has_include(<boost/shared_ptr.hpp>) && !defined(DVO_HIP_NO_BOOST)
#define DVO_HIP_USE_BOOST 1
#endif
#endif

#ifdef DVO_HIP_USE_EIGEN
#include <Eigen/Geometry>
#endif
#ifdef DVO_HIP_USE_OPENCV
#include <opencv2/core/core.hpp>
#endif
#ifdef DVO_HIP_USE_BOOST
#include <boost/shared_ptr.hpp>
#endif

namespace dvo {
namespace compat {

// eigenvalues of a symmetric 6x6 matrix (row-major), cyclic Jacobi rotations in float64; unsorted
inline void sym6_eigenvalues(const double* m, double* ev) {
  double a[36];
  std::memcpy(a, m, sizeof a);
  for (int sweep = 0; sweep < 50; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) (i == j ? diag : off) += a[i * 6 + j] * a[i * 6 + j];
    if (!(off > 1e-30 * diag)) break;   // also leaves on NaN
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        const double apq = a[p * 6 + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {   // A <- A J
          const double akp = a[k * 6 + p], akq = a[k * 6 + q];
          a[k * 6 + p] = c * akp - s * akq;
          a[k * 6 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 6; ++k) {   // A <- J^T A
          const double apk = a[p * 6 + k], aqk = a[q * 6 + k];
          a[p * 6 + k] = c * apk - s * aqk;
          a[q * 6 + k] = s * apk + c * aqk;
        }
      }
  }
  for (int i = 0; i < 6; ++i) ev[i] = a[i * 6 + i];
}

// the reference's smart-pointer typedefs are boost::shared_ptr (rgbd_image.h:93-97): the same type when boost is there
#ifdef DVO_HIP_USE_BOOST
using boost::shared_ptr;
#else
using std::shared_ptr;
#endif

#ifdef DVO_HIP_USE_EIGEN
typedef Eigen::Affine3d Affine3d;
typedef Eigen::Matrix<double, 6, 1> Vector6d;
typedef Eigen::Matrix<double, 6, 6> Matrix6d;
typedef Eigen::Matrix2d Matrix2d;
typedef Eigen::Vector2d Vector2d;
inline void affine_to_rowmajor(const Affine3d& T, double* m) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i * 4 + j] = T.matrix()(i, j);
}
inline void affine_from_rowmajor(const double* m, Affine3d& T) {
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T.matrix()(i, j) = m[i * 4 + j];
}
inline double determinant6(const Matrix6d& m) { return m.determinant(); }
typedef Eigen::Matrix<float, 4, Eigen::Dynamic, Eigen::ColMajor> PointCloud;   // rgbd_image.h:90
inline void pointcloud_resize(PointCloud& p, size_t n) { p.resize(4, int(n)); }
inline float* pointcloud_ptr(PointCloud& p) { return p.data(); }
#else
// fixed-size dense matrix with (i, j) access, just enough for Result / Stats
template <int R, int C>
struct Mat {
  double d[R * C];
  Mat() { for (int i = 0; i < R * C; ++i) d[i] = 0.0; }
  double& operator()(int i, int j) { return d[i * C + j]; }
  double operator()(int i, int j) const { return d[i * C + j]; }
  double& operator()(int i) { return d[i]; }
  double operator()(int i) const { return d[i]; }
  void setZero() { for (int i = 0; i < R * C; ++i) d[i] = 0.0; }
  void setIdentity() { setZero(); for (int i = 0; i < (R < C ? R : C); ++i) d[i * C + i] = 1.0; }
  void setConstant(double v) { for (int i = 0; i < R * C; ++i) d[i] = v; }
  double sum() const { double s = 0; for (int i = 0; i < R * C; ++i) s += d[i]; return s; }
  const double* data() const { return d; }
  double* data() { return d; }
};
typedef Mat<6, 1> Vector6d;
typedef Mat<6, 6> Matrix6d;
typedef Mat<2, 2> Matrix2d;
typedef Mat<2, 1> Vector2d;

// rigid transform with the subset of Eigen::Affine3d the reference's callers use
struct Affine3d {
  Mat<4, 4> m;
  Affine3d() { m.setIdentity(); }
  Mat<4, 4>& matrix() { return m; }
  const Mat<4, 4>& matrix() const { return m; }
  void setIdentity() { m.setIdentity(); }
  static Affine3d Identity() { return Affine3d(); }
  Affine3d operator*(const Affine3d& o) const {
    Affine3d r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += m(i, k) * o.m(k, j);
        r.m(i, j) = s;
      }
    return r;
  }
  Affine3d inverse() const {   // rigid inverse
    Affine3d r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.m(i, j) = m(j, i);
    for (int i = 0; i < 3; ++i) r.m(i, 3) = -(r.m(i, 0) * m(0, 3) + r.m(i, 1) * m(1, 3) + r.m(i, 2) * m(2, 3));
    return r;
  }
  double translation(int i) const { return m(i, 3); }
};
// determinant by LU with partial pivoting (Eigen's Matrix::determinant() for sizes > 4 does the same)
inline double determinant6(const Mat<6, 6>& m) {
  double a[36];
  std::memcpy(a, m.d, sizeof a);
  double det = 1.0;
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(a[i * 6 + k]) > std::fabs(a[piv * 6 + k])) piv = i;
    if (a[piv * 6 + k] == 0.0) return 0.0;
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { const double t = a[k * 6 + j]; a[k * 6 + j] = a[piv * 6 + j]; a[piv * 6 + j] = t; }
      det = -det;
    }
    det *= a[k * 6 + k];
    for (int i = k + 1; i < 6; ++i) {
      const double f = a[i * 6 + k] / a[k * 6 + k];
      for (int j = k + 1; j < 6; ++j) a[i * 6 + j] -= f * a[k * 6 + j];
    }
  }
  return det;
}
inline void affine_to_rowmajor(const Affine3d& T, double* out) { std::memcpy(out, T.m.d, sizeof(double) * 16); }
inline void affine_from_rowmajor(const double* in, Affine3d& T) { std::memcpy(T.m.d, in, sizeof(double) * 16); }
// 4 x n column-major float points
struct PointCloud {
  std::vector<float> d;
  int cols() const { return int(d.size() / 4); }
  float& operator()(int i, int j) { return d[size_t(j) * 4 + i]; }
  float operator()(int i, int j) const { return d[size_t(j) * 4 + i]; }
  float* data() { return d.data(); }
  const float* data() const { return d.data(); }
};
inline void pointcloud_resize(PointCloud& p, size_t n) { p.d.resize(n * 4); }
inline float* pointcloud_ptr(PointCloud& p) { return p.d.data(); }
#endif

#ifdef DVO_HIP_USE_OPENCV
typedef cv::Mat ImageMat;
inline const float* image_ptr(const ImageMat& m) { return m.ptr<float>(); }
inline int image_rows(const ImageMat& m) { return m.rows; }
inline int image_cols(const ImageMat& m) { return m.cols; }
inline bool image_is_float1(const ImageMat& m) { return m.type() == CV_32FC1 && m.isContinuous(); }
inline ImageMat image_create(int rows, int cols) { return ImageMat(rows, cols, CV_32FC1); }
inline float* image_ptr_mut(ImageMat& m) { return m.ptr<float>(); }
inline bool image_empty(const ImageMat& m) { return m.total() == 0; }
inline const uint16_t* image_ptr_u16(const ImageMat& m) { return m.ptr<uint16_t>(); }
typedef cv::Vec<float, 8> Vec8f;
typedef cv::Mat_<Vec8f> AccelerationMat;                                        // rgbd_image.h:175-176
inline void acceleration_fill(AccelerationMat& a, int rows, int cols, const ImageMat* const planes[6]) {
  a.create(rows, cols);
  for (int y = 0; y < rows; ++y) {
    Vec8f* out = a.ptr<Vec8f>(y);
    for (int x = 0; x < cols; ++x) {
      for (int c = 0; c < 6; ++c) out[x].val[c] = planes[c]->ptr<float>(y)[x];
      out[x].val[6] = out[x].val[7] = 0.0f;
    }
  }
}
#else
// single-channel float image with shared storage (cv::Mat_<float> stand-in)
struct ImageMat {
  int rows, cols;
  std::shared_ptr<std::vector<float> > buf;
  ImageMat() : rows(0), cols(0) {}
  ImageMat(int r, int c) : rows(r), cols(c), buf(new std::vector<float>(size_t(r) * c)) {}
  ImageMat(int r, int c, const float* src) : rows(r), cols(c), buf(new std::vector<float>(src, src + size_t(r) * c)) {}
  bool empty() const { return rows == 0 || cols == 0; }
  size_t total() const { return size_t(rows) * cols; }
  template <typename T> T* ptr() { return reinterpret_cast<T*>(buf->data()); }
  template <typename T> const T* ptr() const { return reinterpret_cast<const T*>(buf->data()); }
  template <typename T> T& at(int y, int x) { return ptr<T>()[size_t(y) * cols + x]; }
};
inline const float* image_ptr(const ImageMat& m) { return m.ptr<float>(); }
inline int image_rows(const ImageMat& m) { return m.rows; }
inline int image_cols(const ImageMat& m) { return m.cols; }
inline bool image_is_float1(const ImageMat&) { return true; }
inline ImageMat image_create(int rows, int cols) { return ImageMat(rows, cols); }
inline float* image_ptr_mut(ImageMat& m) { return m.ptr<float>(); }
inline bool image_empty(const ImageMat& m) { return m.empty(); }
inline const uint16_t* image_ptr_u16(const ImageMat& m) { return m.ptr<uint16_t>(); }   // (a 16-bit plane stored two per float slot)
struct Vec8f { float val[8]; };
struct AccelerationMat {
  int rows, cols;
  std::vector<Vec8f> d;
  AccelerationMat() : rows(0), cols(0) {}
};
inline void acceleration_fill(AccelerationMat& a, int rows, int cols, const ImageMat* const planes[6]) {
  a.rows = rows; a.cols = cols;
  a.d.resize(size_t(rows) * cols);
  for (size_t i = 0; i < a.d.size(); ++i) {
    for (int c = 0; c < 6; ++c) a.d[i].val[c] = planes[c]->ptr<float>()[i];
    a.d[i].val[6] = a.d[i].val[7] = 0.0f;
  }
}
#endif

}  // namespace compat
}  // namespace dvo
