// oracle/shim/opencv2/core/core.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  The sliver of cv::Mat the reference's
// hot-path translation units and their headers need to compile: a reference-counted dense 2-D array with typed row / element
// pointers.  Not OpenCV.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>   // the real header pulls it in transitively; the reference relies on that
#include <memory>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

typedef unsigned char uchar;
typedef unsigned short ushort;

namespace cv {

template <typename T, int N> struct Vec {
  T val[N];
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<double, 3> Vec3d;
typedef Vec<float, 3> Vec3f;
typedef Vec<unsigned char, 3> Vec3b;

template <typename T> struct DataType { enum { type = -1 }; };
template <> struct DataType<unsigned char> { enum { type = CV_8UC1 }; };
template <> struct DataType<unsigned short> { enum { type = CV_16UC1 }; };
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<double> { enum { type = CV_64FC1 }; };
template <typename T, int N> struct DataType<Vec<T, N> > { enum { type = CV_MAKETYPE(DataType<T>::type & 7, N) }; };

struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} int area() const { return width * height; } };

class Mat {
 public:
  int rows, cols;
  unsigned char* data;
  size_t step;   // bytes per row
  Mat() : rows(0), cols(0), data(0), step(0), type_(0), elem_(0) {}
  Mat(int r, int c, int type) : rows(0), cols(0), data(0), step(0), type_(0), elem_(0) { create(r, c, type); }
  void create(int r, int c, int type) {
    static const int depth_bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    type_ = type;
    elem_ = size_t(depth_bytes[type & 7]) * size_t((type >> 3) + 1);
    rows = r; cols = c; step = elem_ * size_t(c);
    void* p = 0;
    if (posix_memalign(&p, 64, step * size_t(r) ? step * size_t(r) : 64) != 0) std::abort();
    buf_.reset(static_cast<unsigned char*>(p), std::free);
    data = buf_.get();
  }
  int type() const { return type_; }
  int channels() const { return (type_ >> 3) + 1; }
  bool empty() const { return data == 0 || rows * cols == 0; }
  size_t total() const { return size_t(rows) * cols; }
  size_t elemSize() const { return elem_; }
  bool isContinuous() const { return true; }
  Size size() const { return Size(cols, rows); }
  template <typename T> T* ptr(int y = 0) { return reinterpret_cast<T*>(data + step * size_t(y)); }
  template <typename T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + step * size_t(y)); }
  template <typename T> T* ptr(int y, int x) { return reinterpret_cast<T*>(data + step * size_t(y) + elem_ * size_t(x)); }
  template <typename T> const T* ptr(int y, int x) const { return reinterpret_cast<const T*>(data + step * size_t(y) + elem_ * size_t(x)); }
  template <typename T> T& at(int y, int x) { return *reinterpret_cast<T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
  template <typename T> const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
  Mat clone() const { Mat m(rows, cols, type_); if (data) std::memcpy(m.data, data, step * size_t(rows)); return m; }
 protected:
  int type_;
  size_t elem_;
  std::shared_ptr<unsigned char> buf_;
};

template <typename T> class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) { Mat::create(r, c, DataType<T>::type); elem_ = sizeof(T); step = sizeof(T) * size_t(c); }
  void create(int r, int c) { Mat::create(r, c, DataType<T>::type); }
  static Mat_ zeros(int r, int c) { Mat_ m(r, c); std::memset(m.data, 0, m.step * size_t(r)); return m; }
  T& operator()(int y, int x) { return *reinterpret_cast<T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
  const T& operator()(int y, int x) const { return *reinterpret_cast<const T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
};
typedef Mat_<unsigned char> Mat1b;
typedef Mat_<float> Mat1f;

}  // namespace cv
