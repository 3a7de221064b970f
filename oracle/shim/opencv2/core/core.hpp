// oracle/shim/opencv2/core/core.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  The sliver of cv::Mat the reference's
// hot-path translation units and their headers need to compile: a reference-counted dense 2-D array with typed row / element
// pointers.  Not OpenCV.
#pragma once

#include <cassert>
#include <chrono>   // (pulled in transitively by the real header; the reference relies on that)
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
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)
#define CV_64FC1 CV_MAKETYPE(CV_64F, 1)

typedef unsigned char uchar;
typedef unsigned short ushort;

namespace cv {

template <typename T, int N> struct Vec {
  T val[N];
  Vec() {}
  Vec(T a, T b, T c) { static_assert(N == 3, "3 elements"); val[0] = a; val[1] = b; val[2] = c; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<double, 3> Vec3d;
typedef Vec<float, 3> Vec3f;
typedef Vec<float, 4> Vec4f;
typedef Vec<unsigned char, 3> Vec3b;

template <typename T> struct DataType { enum { type = -1 }; };
template <> struct DataType<unsigned char> { enum { type = CV_8UC1 }; };
template <> struct DataType<unsigned short> { enum { type = CV_16UC1 }; };
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<double> { enum { type = CV_64FC1 }; };
template <typename T, int N> struct DataType<Vec<T, N> > { enum { type = CV_MAKETYPE(DataType<T>::type & 7, N) }; };

struct Size {
  int width, height;
  Size(int w = 0, int h = 0) : width(w), height(h) {}
  int area() const { return width * height; }
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};

// (dvo::util::stopwatch times with these: nanoseconds of the steady clock)
inline long long getTickCount() { return static_cast<long long>(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
inline double getTickFrequency() { return 1e9; }

class Mat {
 public:
  int rows, cols;
  unsigned char* data;
  size_t step;   // bytes per row
  Mat() : rows(0), cols(0), data(0), step(0), type_(0), elem_(0) {}
  Mat(int r, int c, int type) : rows(0), cols(0), data(0), step(0), type_(0), elem_(0) { create(r, c, type); }
  Mat(Size s, int type) : rows(0), cols(0), data(0), step(0), type_(0), elem_(0) { create(s.height, s.width, type); }
  void create(Size s, int type) { create(s.height, s.width, type); }
  static Mat zeros(Size s, int type) { Mat m(s.height, s.width, type); std::memset(m.data, 0, m.step * size_t(m.rows)); return m; }
  static Mat zeros(int r, int c, int type) { return zeros(Size(c, r), type); }
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
  unsigned char* ptr(int y = 0) { return data + step * size_t(y); }
  const unsigned char* ptr(int y = 0) const { return data + step * size_t(y); }
  template <typename T> T& at(size_t i) { return reinterpret_cast<T*>(data)[i]; }
  template <typename T> const T& at(size_t i) const { return reinterpret_cast<const T*>(data)[i]; }
  // image arithmetic: only the reference's legacy weighting (off the alignment path) uses it -- compile, abort when called
  Mat mul(const Mat&) const { std::abort(); }
  template <typename V> void setTo(const V&, const Mat& = Mat()) { std::abort(); }
  void copyTo(Mat&) const { std::abort(); }
  // element-wise conversion between the depths the reference's loaders use (u8 / u16 / f32 sources to f32), any channel count
  void convertTo(Mat& dst, int rtype, double alpha = 1, double beta = 0) const {
    const int cn = channels(), ddepth = rtype & 7;
    if (ddepth != CV_32F) std::abort();
    Mat out(rows, cols, CV_MAKETYPE(CV_32F, cn));
    const size_t n = size_t(rows) * cols * cn;
    float* o = reinterpret_cast<float*>(out.data);
    switch (type_ & 7) {
      case CV_8U: for (size_t i = 0; i < n; ++i) o[i] = float(double(data[i]) * alpha + beta); break;
      case CV_16U: for (size_t i = 0; i < n; ++i) o[i] = float(double(reinterpret_cast<const unsigned short*>(data)[i]) * alpha + beta); break;
      case CV_32F: for (size_t i = 0; i < n; ++i) o[i] = float(double(reinterpret_cast<const float*>(data)[i]) * alpha + beta); break;
      default: std::abort();
    }
    dst = out;
  }
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
  Mat_(Size s, const T& v) { Mat::create(s.height, s.width, DataType<T>::type); for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) (*this)(y, x) = v; }
  void create(int r, int c) { Mat::create(r, c, DataType<T>::type); }
  static Mat_ zeros(int r, int c) { Mat_ m(r, c); std::memset(m.data, 0, m.step * size_t(r)); return m; }
  T& operator()(int y, int x) { return *reinterpret_cast<T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
  const T& operator()(int y, int x) const { return *reinterpret_cast<const T*>(data + step * size_t(y) + sizeof(T) * size_t(x)); }
};
struct Scalar { double v[4]; double operator()(int i) const { return v[i]; } double operator[](int i) const { return v[i]; } };
inline Mat operator!=(const Mat&, const Mat&) { std::abort(); }
inline Mat operator==(const Mat&, const Mat&) { std::abort(); }
inline Mat operator-(const Mat&, const Mat&) { std::abort(); }
inline Mat operator*(const Mat&, double) { std::abort(); }
inline Mat operator*(double, const Mat&) { std::abort(); }
inline Mat operator/(const Mat&, double) { std::abort(); }
inline void log(const Mat&, Mat&) { std::abort(); }
inline Scalar sum(const Mat&) { std::abort(); }
inline int countNonZero(const Mat&) { std::abort(); }
inline Mat abs(const Mat&) { std::abort(); }
inline void absdiff(const Mat&, const Mat&, Mat&) { std::abort(); }
template <typename A, typename B> inline void meanStdDev(const Mat&, A&, B&, const Mat& = Mat()) { std::abort(); }
inline void medianBlur(const Mat&, Mat&, int) { std::abort(); }
enum { BORDER_REPLICATE = 1 };
inline void Sobel(const Mat&, Mat&, int, int, int, int = 3, double = 1, double = 0, int = 0) { std::abort(); }
inline int waitKey(int = 0) { return -1; }

// cv::merge: n single-channel planes of one type interleaved into an n-channel image (RgbdImage::buildAccelerationStructure)
inline void merge(const Mat* planes, size_t n, Mat& dst) {
  const int depth = planes[0].type() & 7;
  dst.create(planes[0].rows, planes[0].cols, CV_MAKETYPE(depth, int(n)));
  const size_t es = planes[0].elemSize();
  for (int y = 0; y < dst.rows; ++y)
    for (int x = 0; x < dst.cols; ++x)
      for (size_t c = 0; c < n; ++c)
        std::memcpy(dst.data + dst.step * size_t(y) + (size_t(x) * n + c) * es, planes[c].data + planes[c].step * size_t(y) + size_t(x) * es, es);
}
typedef Mat_<unsigned char> Mat1b;
typedef Mat_<float> Mat1f;

}  // namespace cv
