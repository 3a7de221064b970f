/*
 * dvo_oracle.cpp -- CPU ORACLE (test infrastructure; parity status: dvo_oracle.h -- the REF_SSE mode is pinned
 * bit for bit against the reference's own translation units, oracle/_ref).
 *
 * A from-scratch restatement of the reference's dense RGB-D alignment path.  Every function cites
 * the reference lines (relative to /root/reference) whose behaviour it reproduces.  Build with
 *   g++ -O3 -march=native -ffp-contract=off -frounding-math        (oracle/Makefile)
 * -ffp-contract=off: the reference's numerics are those of a no-FMA SSE3 build;
 * -frounding-math : the REF_SSE residual pass runs under MXCSR round-toward-zero.
 */
#include "dvo_oracle.h"
#include "se3_oracle.h"

#include <immintrin.h>
#include <pmmintrin.h>
#include <xmmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace oracle {

static const float kNaN = std::numeric_limits<float>::quiet_NaN();

// 16-byte aligned float storage (the reference's AoS records are EIGEN_ALIGN16)
template <typename T>
struct AlignedBuf {
  T* p = nullptr;
  size_t n = 0;
  void resize(size_t count) {
    if (count <= n) return;
    std::free(p);
    p = static_cast<T*>(std::aligned_alloc(64, ((count * sizeof(T) + 63) / 64) * 64));
    n = count;
  }
  ~AlignedBuf() { std::free(p); }
  AlignedBuf() = default;
  AlignedBuf(const AlignedBuf&) = delete;
  AlignedBuf& operator=(const AlignedBuf&) = delete;
  AlignedBuf(AlignedBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  AlignedBuf& operator=(AlignedBuf&& o) noexcept {
    if (this != &o) { std::free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
};

struct Intrinsics { float fx, fy, ox, oy; };

// one selected reference point: 16 B point + 32 B {i,z,idx,idy,zdx,zdy,-,-}  (rgbd_image.h:39-62)
struct alignas(16) RefPoint {
  float p[4];
  float v[8];
};

struct Level {
  int w = 0, h = 0;
  Intrinsics K{};
  std::vector<float> plane[6];           // I Z Idx Idy Zdx Zdy
  AlignedBuf<float> accel;               // 8 interleaved channels  (rgbd_image.cpp:534-543)
  std::vector<float> cloud;              // 4 floats per pixel      (rgbd_image.cpp:245-262)
  bool derived = false;
  // cached selection (point_selection.cpp:89-117)
  AlignedBuf<RefPoint> sel;
  int n_sel = -1;
  float sel_ithr = 0, sel_dthr = 0;
};

}  // namespace oracle

struct oracle_pyramid {
  std::vector<oracle::Level> levels;
};

namespace oracle {

// ---- image model -------------------------------------------------------------------------------

// rgbd_image.cpp:38-55 : 2x2 mean, summed left to right then / 4.0f
static void pyr_down_mean(const std::vector<float>& in, int w, int h, std::vector<float>& out) {
  const int ow = w / 2, oh = h / 2;
  out.resize(size_t(ow) * oh);
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      const float* r0 = &in[size_t(2 * y) * w + 2 * x];
      const float* r1 = r0 + w;
      out[size_t(y) * ow + x] = (r0[0] + r0[1] + r1[0] + r1[1]) / 4.0f;
    }
}

// rgbd_image.cpp:127-139 : top-left sample, NaN holes are kept
static void pyr_down_subsample(const std::vector<float>& in, int w, int h, std::vector<float>& out) {
  const int ow = w / 2, oh = h / 2;
  out.resize(size_t(ow) * oh);
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) out[size_t(y) * ow + x] = in[size_t(2 * y) * w + 2 * x];
}

// rgbd_image.cpp:419-489, rgbd_image_sse.cpp:241-284 : 0.5*(next-prev), clamped borders
static void derivative_x(const std::vector<float>& in, int w, int h, std::vector<float>& out) {
  out.resize(in.size());
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int prev = std::max(x - 1, 0), next = std::min(x + 1, w - 1);
      out[size_t(y) * w + x] = (in[size_t(y) * w + next] - in[size_t(y) * w + prev]) * 0.5f;
    }
}
static void derivative_y(const std::vector<float>& in, int w, int h, std::vector<float>& out) {
  out.resize(in.size());
  for (int y = 0; y < h; ++y) {
    int prev = std::max(y - 1, 0), next = std::min(y + 1, h - 1);
    for (int x = 0; x < w; ++x)
      out[size_t(y) * w + x] = (in[size_t(next) * w + x] - in[size_t(prev) * w + x]) * 0.5f;
  }
}

// rgbd_image.cpp:534-543 (+ derivatives) and :186-204,245-262 (point cloud)
static void derive_level(Level& L) {
  if (L.derived) return;
  derivative_x(L.plane[0], L.w, L.h, L.plane[2]);
  derivative_y(L.plane[0], L.w, L.h, L.plane[3]);
  derivative_x(L.plane[1], L.w, L.h, L.plane[4]);
  derivative_y(L.plane[1], L.w, L.h, L.plane[5]);
  const size_t n = size_t(L.w) * L.h;
  L.accel.resize(n * 8);
  for (size_t i = 0; i < n; ++i) {
    float* a = L.accel.p + i * 8;
    for (int c = 0; c < 6; ++c) a[c] = L.plane[c][i];
    a[6] = 0.0f; a[7] = 0.0f;
  }
  L.cloud.resize(n * 4);
  size_t idx = 0;
  for (int y = 0; y < L.h; ++y)
    for (int x = 0; x < L.w; ++x, ++idx) {
      const float tx = (float(x) - L.K.ox) / L.K.fx;       // pointcloud_template_, rgbd_image.cpp:198-199
      const float ty = (float(y) - L.K.oy) / L.K.fy;
      const float z = L.plane[1][idx];
      L.cloud[idx * 4 + 0] = tx * z;
      L.cloud[idx * 4 + 1] = ty * z;
      L.cloud[idx * 4 + 2] = 1.0f * z;
      L.cloud[idx * 4 + 3] = 1.0f;
    }
  L.derived = true;
}

// point_selection.cpp:119-152 with the predicate of point_selection.h:63-66
static int select_points(Level& L, float ithr, float dthr, uint8_t* mask) {
  derive_level(L);
  if (L.n_sel >= 0 && L.sel_ithr == ithr && L.sel_dthr == dthr && !mask) return L.n_sel;
  const size_t n = size_t(L.w) * L.h;
  L.sel.resize(n);
  int cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* a = L.accel.p + i * 8;
    const float z = L.cloud[i * 4 + 2];
    const float idx = a[2], idy = a[3], zdx = a[4], zdy = a[5];
    const bool ok = z == z && zdx == zdx && zdy == zdy &&
                    (std::fabs(idx) > ithr || std::fabs(idy) > ithr || std::fabs(zdx) > dthr || std::fabs(zdy) > dthr);
    if (mask) mask[i] = ok ? 1 : 0;
    if (!ok) continue;
    RefPoint& r = L.sel.p[cnt++];
    std::memcpy(r.p, &L.cloud[i * 4], 16);
    std::memcpy(r.v, a, 32);
  }
  L.n_sel = cnt;
  L.sel_ithr = ithr;
  L.sel_dthr = dthr;
  return cnt;
}

// ---- per-iteration passes ------------------------------------------------------------------------

struct Scratch {
  AlignedBuf<RefPoint> points_error;   // dense_tracking.cpp:160-165
  AlignedBuf<float> residuals;         // 2 floats per valid point
  AlignedBuf<float> weights;
  std::vector<int> pixel_of;           // not in the reference: pixel index for residual dumps
};

struct Weights8 {
  alignas(16) float wcur[8];
  alignas(16) float wref[8];
};

// dense_tracking.cpp:215-220
static Weights8 make_weights(const Intrinsics& K) {
  Weights8 W;
  const float wcur_id = 0.5f, wref_id = 0.5f, wcur_zd = 1.0f, wref_zd = 0.0f;
  float c[8] = {1.0f / 255.0f, 1.0f, wcur_id * K.fx / 255.0f, wcur_id * K.fy / 255.0f, wcur_zd * K.fx, wcur_zd * K.fy, 0.0f, 0.0f};
  float r[8] = {-1.0f / 255.0f, -1.0f, wref_id * K.fx / 255.0f, wref_id * K.fy / 255.0f, wref_zd * K.fx, wref_zd * K.fy, 0.0f, 0.0f};
  std::memcpy(W.wcur, c, sizeof(c));
  std::memcpy(W.wref, r, sizeof(r));
  return W;
}

// the quirk bits of a semantic mode (dvo_oracle.h): REF_SSE = all of them, MATH = none, DVO_ORACLE_QUIRKS | bits = those bits
static inline int quirks_of(int mode) {
  if (mode == DVO_ORACLE_REF_SSE) return DVO_ORACLE_Q_ALL;
  if (mode == DVO_ORACLE_MATH) return 0;
  return mode & DVO_ORACLE_Q_ALL;
}
// the residual pass of a mode: MATH proper keeps its own scalar restatement (residual_pass_math), everything else runs the
// SSE restatement with the mode's quirk bits
static inline bool uses_math_residuals(int mode) { return mode == DVO_ORACLE_MATH; }

// KT = K * T(3x4) in float, Eigen coefficient order (dense_tracking_impl.cpp:142-148)
static void make_KT(const Intrinsics& K, const float T[12], float KT[12]) {
  for (int j = 0; j < 4; ++j) {
    KT[0 * 4 + j] = K.fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j] + K.ox * T[2 * 4 + j];
    KT[1 * 4 + j] = 0.0f * T[0 * 4 + j] + K.fy * T[1 * 4 + j] + K.oy * T[2 * 4 + j];
    KT[2 * 4 + j] = 0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j] + 1.0f * T[2 * 4 + j];
  }
}

// dense_tracking_impl.cpp:122-128
static inline float depth_std_dev_z(float depth) {
  float s = depth - 0.4f;
  s = 0.0012f + 0.0019f * s * s;
  return s;
}

// sum of a 4-vector's lanes in the order two _mm_hadd_ps produce: (l0+l1)+(l2+l3)
static inline __m128 hsum2(__m128 a, __m128 b, __m128 c, __m128 d) {
  return _mm_hadd_ps(_mm_hadd_ps(a, b), _mm_hadd_ps(c, d));   // [sum a, sum b, sum c, sum d]
}

/*
 * Pass 1, REF_SSE semantics (dense_tracking_impl.cpp:133-393): approximate reciprocal, MXCSR
 * round-toward-zero for every add/mul/convert in the loop, inclusive bounds 0<=u<=W-2, NaN test on
 * all 8 interpolated channels, occlusion test, odd trailing point dropped.  Points are independent
 * of their pair partner, so they are processed one at a time with the same lane arithmetic.
 */
static int residual_pass_ref(const RefPoint* pts, int n_pts, const Level& cur, const float T[12], const Weights8& W,
                             Scratch& S, const int* pix_in, bool track_pixels, int quirks = DVO_ORACLE_Q_ALL) {
  // quirk by quirk (dvo_oracle.h): without Q1 the projection divides exactly, without Q2 the loop rounds to nearest (and
  // converts with truncation, = floor on the accepted range), without Q3 an odd trailing point is kept.  No quirk = the
  // arithmetic of residual_pass_math, bit for bit (tests/test_oracle.py).
  const bool approx_rcp = (quirks & DVO_ORACLE_Q_RCP_PROJECTION) != 0, rz = (quirks & DVO_ORACLE_Q_ROUND_TOWARD_ZERO) != 0;
  float KT[12];
  make_KT(cur.K, T, KT);
  const __m128 r1 = _mm_loadu_ps(KT + 0), r2 = _mm_loadu_ps(KT + 4), r3 = _mm_loadu_ps(KT + 8);
  const __m128 wcur_a = _mm_load_ps(W.wcur), wcur_b = _mm_load_ps(W.wcur + 4);
  const __m128 wref_a = _mm_load_ps(W.wref), wref_b = _mm_load_ps(W.wref + 4);
  const __m128 lo = _mm_set1_ps(0.0f);
  const __m128 hi = _mm_setr_ps(float(cur.w - 2), float(cur.h - 2), float(cur.w - 2), float(cur.h - 2));
  const __m128 ones = _mm_set1_ps(1.0f);
  const __m128 zsel = _mm_castsi128_ps(_mm_setr_epi32(0, -1, 0, 0));

  const unsigned old_mode = _MM_GET_ROUNDING_MODE();
  if (rz) _MM_SET_ROUNDING_MODE(_MM_ROUND_TOWARD_ZERO);

  const int n_used = (quirks & DVO_ORACLE_Q_DROP_ODD) ? (n_pts & ~1) : n_pts;
  int out = 0;
  for (int i = 0; i < n_used; ++i) {
    const RefPoint& rp = pts[i];
    const __m128 p = _mm_load_ps(rp.p);
    const __m128 xyz = hsum2(_mm_mul_ps(r1, p), _mm_mul_ps(r2, p), _mm_mul_ps(r3, p), _mm_mul_ps(r3, p));  // [x y z z]
    const __m128 zz = _mm_shuffle_ps(xyz, xyz, _MM_SHUFFLE(2, 2, 2, 2));
    const __m128 uv = approx_rcp ? _mm_mul_ps(xyz, _mm_rcp_ps(zz)) : _mm_div_ps(xyz, zz);   // [u v . .]
    const __m128i uvi = rz ? _mm_cvtps_epi32(uv) : _mm_cvttps_epi32(uv);     // truncation under RZ
    const __m128 uv0 = _mm_cvtepi32_ps(uvi);
    const __m128 f1 = _mm_sub_ps(uv, uv0);
    const __m128 f0 = _mm_sub_ps(ones, f1);
    const int inb = _mm_movemask_ps(_mm_and_ps(_mm_cmpge_ps(uv, lo), _mm_cmple_ps(uv, hi)));
    if ((inb & 3) != 3) continue;
    const int u0 = _mm_cvtsi128_si32(uvi);
    const int v0 = _mm_cvtsi128_si32(_mm_shuffle_epi32(uvi, 1));
    const float* t00 = cur.accel.p + (size_t(v0) * cur.w + u0) * 8;
    const float* t01 = t00 + size_t(cur.w) * 8;
    const __m128 a0 = _mm_shuffle_ps(f0, f0, 0x00), a1 = _mm_shuffle_ps(f1, f1, 0x00);   // u weights
    const __m128 b0 = _mm_shuffle_ps(f0, f0, 0x55), b1 = _mm_shuffle_ps(f1, f1, 0x55);   // v weights
    const __m128 top_a = _mm_mul_ps(b0, _mm_add_ps(_mm_mul_ps(a0, _mm_load_ps(t00 + 0)), _mm_mul_ps(a1, _mm_load_ps(t00 + 8))));
    const __m128 top_b = _mm_mul_ps(b0, _mm_add_ps(_mm_mul_ps(a0, _mm_load_ps(t00 + 4)), _mm_mul_ps(a1, _mm_load_ps(t00 + 12))));
    const __m128 bot_a = _mm_mul_ps(b1, _mm_add_ps(_mm_mul_ps(a0, _mm_load_ps(t01 + 0)), _mm_mul_ps(a1, _mm_load_ps(t01 + 8))));
    const __m128 bot_b = _mm_mul_ps(b1, _mm_add_ps(_mm_mul_ps(a0, _mm_load_ps(t01 + 4)), _mm_mul_ps(a1, _mm_load_ps(t01 + 12))));
    const __m128 ia = _mm_add_ps(top_a, bot_a), ib = _mm_add_ps(top_b, bot_b);
    if (_mm_movemask_ps(_mm_cmpunord_ps(ia, ib)) != 0) continue;
    __m128 ref_a = _mm_load_ps(rp.v);
    ref_a = _mm_or_ps(_mm_and_ps(zsel, zz), _mm_andnot_ps(zsel, ref_a));     // ref depth := transformed z
    const __m128 res_a = _mm_add_ps(_mm_mul_ps(wcur_a, ia), _mm_mul_ps(wref_a, ref_a));
    RefPoint& o = S.points_error.p[out];
    _mm_store_ps(o.p, p);
    _mm_store_ps(o.v, res_a);
    if (!(o.v[1] > -20.0f * depth_std_dev_z(rp.v[1]))) continue;
    const __m128 res_b = _mm_add_ps(_mm_mul_ps(wcur_b, ib), _mm_mul_ps(wref_b, _mm_load_ps(rp.v + 4)));
    _mm_store_ps(o.v + 4, res_b);
    S.residuals.p[2 * out + 0] = o.v[0];
    S.residuals.p[2 * out + 1] = o.v[1];
    if (track_pixels) S.pixel_of[out] = pix_in[i];
    ++out;
  }
  _MM_SET_ROUNDING_MODE(old_mode);
  return out;
}

/* Pass 1, MATH semantics: exact division, round-to-nearest, floor, every selected point. */
static int residual_pass_math(const RefPoint* pts, int n_pts, const Level& cur, const float T[12], const Weights8& W,
                              Scratch& S, const int* pix_in, bool track_pixels) {
  float KT[12];
  make_KT(cur.K, T, KT);
  const float umax = float(cur.w - 2), vmax = float(cur.h - 2);
  int out = 0;
  for (int i = 0; i < n_pts; ++i) {
    const RefPoint& rp = pts[i];
    const float X = rp.p[0], Y = rp.p[1], Z = rp.p[2];
    const float qx = (KT[0] * X + KT[1] * Y) + (KT[2] * Z + KT[3]);
    const float qy = (KT[4] * X + KT[5] * Y) + (KT[6] * Z + KT[7]);
    const float qz = (KT[8] * X + KT[9] * Y) + (KT[10] * Z + KT[11]);
    const float u = qx / qz, v = qy / qz;
    if (!(u >= 0.0f && u <= umax && v >= 0.0f && v <= vmax)) continue;
    const float uf = std::floor(u), vf = std::floor(v);
    const int u0 = int(uf), v0 = int(vf);
    const float a1 = u - uf, a0 = 1.0f - a1, b1 = v - vf, b0 = 1.0f - b1;
    const float* t00 = cur.accel.p + (size_t(v0) * cur.w + u0) * 8;
    const float* t01 = t00 + size_t(cur.w) * 8;
    float c[8];
    bool nan = false;
    for (int k = 0; k < 8; ++k) {
      c[k] = b0 * (a0 * t00[k] + a1 * t00[8 + k]) + b1 * (a0 * t01[k] + a1 * t01[8 + k]);
      nan |= (c[k] != c[k]);
    }
    if (nan) continue;
    RefPoint& o = S.points_error.p[out];
    std::memcpy(o.p, rp.p, 16);
    float refv[8];
    std::memcpy(refv, rp.v, 32);
    refv[1] = qz;
    for (int k = 0; k < 8; ++k) o.v[k] = W.wcur[k] * c[k] + W.wref[k] * refv[k];
    if (!(o.v[1] > -20.0f * depth_std_dev_z(rp.v[1]))) continue;
    S.residuals.p[2 * out + 0] = o.v[0];
    S.residuals.p[2 * out + 1] = o.v[1];
    if (track_pixels) S.pixel_of[out] = pix_in[i];
    ++out;
  }
  return out;
}

// r^T P r with Eigen's (r^T P) r grouping; P row-major
static inline float mahalanobis(const float* r, const float P[4]) {
  return (r[0] * P[0] + r[1] * P[2]) * r[0] + (r[0] * P[1] + r[1] * P[3]) * r[1];
}

// Pass 2 (dense_tracking_impl.cpp:640-707): t-distribution weights, nu = 5, mean = 0
static void weights_pass(const float* res, int n, const float P[4], float* w, int mode) {
  int i = 0;
  if (quirks_of(mode) & DVO_ORACLE_Q_RCP_WEIGHTS) {
    const int n4 = n & ~3;
    for (; i < n4; i += 4) {   // groups of four use the approximate reciprocal
      alignas(16) float d[4];
      for (int k = 0; k < 4; ++k) d[k] = 5.0f + mahalanobis(res + 2 * (i + k), P);
      _mm_store_ps(w + i, _mm_mul_ps(_mm_set1_ps(7.0f), _mm_rcp_ps(_mm_load_ps(d))));
    }
  }
  for (; i < n; ++i) w[i] = float((2.0 + 5.0f) / (5.0f + mahalanobis(res + 2 * i, P)));
}

// Pass 3 (dense_tracking_impl.cpp:566-638): C = sum w r r^T / (n-3).  C = {c00, c01, c11, c10}: the reference's scalar tail
// (odd n) adds the full outer product (w d) d^T coefficient by coefficient, so c10 = (w y) x can differ from c01 = (w x) y
// in the last bit; the SSE part writes one value to both.
static void scale_pass(const float* res, const float* w, int n, int mode, float C[4]) {
  if (quirks_of(mode) & DVO_ORACLE_Q_SCALE_PAIRING) {
    const float scale = 1.0f / float(size_t(n) - 2 - 1);
    float a0 = 0, a1 = 0, a3 = 0;
    const int n2 = n & ~1;
    for (int i = 0; i < n2; i += 2) {   // the first residual of each pair is used twice
      const float x = res[2 * i], y = res[2 * i + 1];
      const float xx = x * x, yx = y * x, yy = y * y;
      a0 = a0 + (scale * (w[i] * xx) + scale * (w[i + 1] * xx));
      a1 = a1 + (scale * (w[i] * yx) + scale * (w[i + 1] * yx));
      a3 = a3 + (scale * (w[i] * yy) + scale * (w[i + 1] * yy));
    }
    float a2 = a1;                              // covariance(1,0) = tmp[1]  (:629)
    if (n & 1) {
      const float x = res[2 * n2], y = res[2 * n2 + 1], ww = w[n2];
      a0 += scale * ((ww * x) * x);
      a1 += scale * ((ww * x) * y);
      a2 += scale * ((ww * y) * x);
      a3 += scale * ((ww * y) * y);
    }
    C[0] = a0; C[1] = a1; C[2] = a3; C[3] = a2;
  } else {
    double s0 = 0, s1 = 0, s3 = 0;
    if (mode & DVO_ORACLE_X_PAIRING_F64) {
      // experiment (tests/golden/make_quirk_table.py): the pairing of Q6 as a formula -- sum_k (w_2k + w_2k+1) r_2k r_2k^T, a
      // function of the compaction RANKS only, which a GPU could compute with a scan -- without the float sequential rounding
      const int n2 = n & ~1;
      for (int i = 0; i < n2; i += 2) {
        const double x = res[2 * i], y = res[2 * i + 1], ww = double(w[i]) + double(w[i + 1]);
        s0 += ww * x * x; s1 += ww * x * y; s3 += ww * y * y;
      }
      if (n & 1) { const double x = res[2 * n2], y = res[2 * n2 + 1], ww = w[n2]; s0 += ww * x * x; s1 += ww * x * y; s3 += ww * y * y; }
    } else
    for (int i = 0; i < n; ++i) {
      const double x = res[2 * i], y = res[2 * i + 1], ww = w[i];
      s0 += ww * x * x; s1 += ww * x * y; s3 += ww * y * y;
    }
    const double d = double(n) - 3.0;
    C[0] = float(s0 / d); C[1] = float(s1 / d); C[2] = float(s3 / d); C[3] = C[1];
  }
}

// Eigen 2x2 inverse (dense_tracking.cpp:295) of {c00, c01, c11, c10}; P row-major
static void invert2(const float C[4], float P[4]) {
  const float det = C[0] * C[2] - C[3] * C[1];
  const float inv = 1.0f / det;
  P[0] = C[2] * inv;
  P[1] = -C[1] * inv;
  P[2] = -C[3] * inv;
  P[3] = C[0] * inv;
}

// Pass 4 (dense_tracking_impl.cpp:406-425). Returns ll.
static double loglik_pass(const float* res, int n, const float P[4], int mode) {
  if (quirks_of(mode) & DVO_ORACLE_Q_LOGLIK_TAIL) {
    double sum = 0.0, acc = 1.0;
    for (int i = 0; i < n; ++i) {
      acc *= (1.0 + 0.2 * mahalanobis(res + 2 * i, P));
      if (((i + 1) % 50) == 0) { sum += std::log(acc); acc = 1.0; }   // tail dropped
    }
    const float det = P[0] * P[3] - P[1] * P[2];
    return float(0.5 * n * std::log(det) - 0.5 * (5.0 + 2.0) * sum);   // result is cast to float at :297
  }
  double sum = 0.0;
  for (int i = 0; i < n; ++i) sum += std::log1p(0.2 * double(mahalanobis(res + 2 * i, P)));
  const double det = double(P[0]) * P[3] - double(P[1]) * P[2];
  return float(0.5 * n * std::log(det) - 3.5 * sum);   // a float in the reference (:297): the rounding is not a quirk
}

// dense_tracking.cpp:448-476 and :333-340
static inline void jacobian_rows(const RefPoint& e, float J0[6], float J1[6]) {
  const float x = e.p[0], y = e.p[1], z = e.p[2];
  const float iz = 1.0f / z, iz2 = 1.0f / (z * z);
  float a[6], b[6];
  a[0] = iz; a[1] = 0.0f; a[2] = -x * iz2; a[3] = a[2] * y; a[4] = 1.0f - a[2] * x; a[5] = -y * iz;
  b[0] = 0.0f; b[1] = iz; b[2] = -y * iz2; b[3] = -1.0f + b[2] * y; b[4] = -a[3]; b[5] = x * iz;
  const float jz[6] = {0.0f, 0.0f, 1.0f, y, -x, 0.0f};
  const float gix = e.v[2], giy = e.v[3], gzx = e.v[4], gzy = e.v[5];
  for (int k = 0; k < 6; ++k) {
    J0[k] = gix * a[k] + giy * b[k];
    J1[k] = (gzx * a[k] + gzy * b[k]) - jz[k];
  }
}

// Pass 5 (least_squares.cpp:58-64, math_sse.cpp:82-207). A row-major 6x6 (symmetric), b.
static void normal_equations(const RefPoint* pe, const float* w, int n, const float P[4], int mode, double A[36], double b[6]) {
  if (quirks_of(mode) & DVO_ORACLE_Q_FLOAT_NORMAL_EQ) {
    float blk[24];   // six row-major 2x2 blocks (0,0)(0,2)(0,4)(2,2)(2,4)(4,4)
    float bf[6];
    std::memset(blk, 0, sizeof(blk));
    std::memset(bf, 0, sizeof(bf));
    for (int i = 0; i < n; ++i) {
      float J0[6], J1[6];
      jacobian_rows(pe[i], J0, J1);
      const float W00 = w[i] * P[0], W01 = w[i] * P[1], W10 = w[i] * P[2], W11 = w[i] * P[3];
      float ua[6], ub[6];   // u = alpha^T v per column
      for (int k = 0; k < 6; ++k) {
        ua[k] = J0[k] * W00 + J1[k] * W10;
        ub[k] = J0[k] * W01 + J1[k] * W11;
      }
      int o = 0;
      for (int bi = 0; bi < 6; bi += 2)
        for (int bj = bi; bj < 6; bj += 2) {
          blk[o + 0] += ua[bi] * J0[bj] + ub[bi] * J1[bj];
          blk[o + 1] += ua[bi] * J0[bj + 1] + ub[bi] * J1[bj + 1];
          blk[o + 2] += ua[bi + 1] * J0[bj] + ub[bi + 1] * J1[bj];
          blk[o + 3] += ua[bi + 1] * J0[bj + 1] + ub[bi + 1] * J1[bj + 1];
          o += 4;
        }
      const float r0 = pe[i].v[0], r1 = pe[i].v[1];
      for (int k = 0; k < 6; ++k) {   // b -= (J^T W) r
        const float m0 = J0[k] * W00 + J1[k] * W10, m1 = J0[k] * W01 + J1[k] * W11;
        bf[k] -= m0 * r0 + m1 * r1;
      }
    }
    float Af[36];
    int o = 0;
    for (int bi = 0; bi < 6; bi += 2)
      for (int bj = bi; bj < 6; bj += 2) {
        Af[bi * 6 + bj] = blk[o]; Af[bi * 6 + bj + 1] = blk[o + 1];
        Af[(bi + 1) * 6 + bj] = blk[o + 2]; Af[(bi + 1) * 6 + bj + 1] = blk[o + 3];
        o += 4;
      }
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) { A[i * 6 + j] = Af[i * 6 + j]; A[j * 6 + i] = Af[i * 6 + j]; }   // selfadjointView<Upper>
    for (int k = 0; k < 6; ++k) b[k] = bf[k];
  } else {
    double Ad[36], bd[6];
    std::memset(Ad, 0, sizeof(Ad));
    std::memset(bd, 0, sizeof(bd));
    for (int i = 0; i < n; ++i) {
      float J0[6], J1[6];
      jacobian_rows(pe[i], J0, J1);
      const double W00 = double(w[i]) * P[0], W01 = double(w[i]) * P[1], W11 = double(w[i]) * P[3];
      const double r0 = pe[i].v[0], r1 = pe[i].v[1];
      for (int k = 0; k < 6; ++k) {
        const double ua = J0[k] * W00 + J1[k] * W01, ub = J0[k] * W01 + J1[k] * W11;
        for (int l = k; l < 6; ++l) Ad[k * 6 + l] += ua * J0[l] + ub * J1[l];
        bd[k] -= ua * r0 + ub * r1;
      }
    }
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) { A[i * 6 + j] = Ad[i * 6 + j]; A[j * 6 + i] = Ad[i * 6 + j]; }
    for (int k = 0; k < 6; ++k) b[k] = bd[k];
  }
}

static void ensure_scratch(Scratch& S, size_t n, bool track) {
  S.points_error.resize(n + 1);
  S.residuals.resize(2 * n + 4);
  S.weights.resize(n + 4);
  if (track && S.pixel_of.size() < n + 1) S.pixel_of.resize(n + 1);
}

// ---- driver (dense_tracking.cpp:131-376) ----------------------------------------------------------

static void fill_nan(double* p, int n) { for (int i = 0; i < n; ++i) p[i] = std::numeric_limits<double>::quiet_NaN(); }

static int match_impl(oracle_pyramid* ref, oracle_pyramid* cur, const oracle_config& cfg, oracle_result* result,
                      oracle_level_stats* levels, int cap_levels, oracle_iteration_stats* iters, int cap_iters, Scratch& S) {
  const int num_levels = cfg.first_level + 1;   // dense_tracking_config.cpp:44-47
  if (int(ref->levels.size()) < num_levels || int(cur->levels.size()) < num_levels) return -1;
  if (cfg.first_level < cfg.last_level) return -2;

  SE3 inc;   // "our first increment is the given guess" (:147)
  if (cfg.use_initial_estimate) inc = se3_from_matrix(result->transformation);
  SE3 initial = inc, initial_old = inc;
  SE3 estimate, estimate_old;

  const size_t max_pts = size_t(ref->levels[0].w) * ref->levels[0].h;
  ensure_scratch(S, std::max(max_pts, size_t(16)), false);

  int n_levels_out = 0, n_iters_out = 0;
  oracle_iteration_stats dummy_it;
  oracle_level_stats dummy_lv;
  oracle_level_stats* last_level = nullptr;
  oracle_iteration_stats* last_level_first_iter = nullptr;

  float P[4] = {0, 0, 0, 0};
  for (int level = cfg.first_level; level >= cfg.last_level; --level) {
    oracle_level_stats* ls = (n_levels_out < cap_levels && levels) ? &levels[n_levels_out] : &dummy_lv;
    ++n_levels_out;
    std::memset(ls, 0, sizeof(*ls));
    ls->first_iteration_index = n_iters_out;
    last_level = ls;
    last_level_first_iter = iters ? iters + n_iters_out : nullptr;

    P[0] = P[1] = P[2] = P[3] = 0.0f;
    int iteration = 0;
    double error = DBL_MAX, last_error;

    Level& C = cur->levels[level];
    Level& R = ref->levels[level];
    derive_level(C);
    const Weights8 W = make_weights(C.K);
    const int n_sel = select_points(R, cfg.intensity_derivative_threshold, cfg.depth_derivative_threshold, nullptr);

    ls->id = level;
    ls->max_valid_pixels = int(double(max_pts) * std::pow(0.25, double(level)));   // point_selection.cpp:68-71
    ls->valid_pixels = n_sel;
    ls->termination = -1;

    double A[36], b[6], x[6];
    se3_log(inc, x);   // :238
    bool accept = true;
    auto inf_norm = [](const double* v) {
      double m = 0;
      for (int i = 0; i < 6; ++i) { if (v[i] != v[i]) return std::numeric_limits<double>::quiet_NaN(); m = std::max(m, std::fabs(v[i])); }
      return m;
    };
    do {
      oracle_iteration_stats* is = (n_iters_out < cap_iters && iters) ? &iters[n_iters_out] : &dummy_it;
      ++n_iters_out;
      ++ls->n_iterations;
      is->id = iteration;
      is->valid_constraints = 0;
      is->tdist_loglik = is->prior_loglik = std::numeric_limits<double>::quiet_NaN();
      is->tdist_mean[0] = is->tdist_mean[1] = 0.0;
      fill_nan(is->tdist_precision, 4);
      fill_nan(is->increment, 6);
      fill_nan(is->information, 36);

      inc = se3_exp(x);                                         // :259
      initial_old = initial; initial = se3_mul(se3_inverse(inc), initial);   // :260
      estimate_old = estimate; estimate = se3_mul(inc, estimate);            // :261
      double M[16];
      se3_to_matrix(estimate, M);
      float Tf[12];
      for (int i = 0; i < 12; ++i) Tf[i] = float(M[i]);        // :263

      const int n = !uses_math_residuals(cfg.mode)
                        ? residual_pass_ref(R.sel.p, n_sel, C, Tf, W, S, nullptr, false, quirks_of(cfg.mode))
                        : residual_pass_math(R.sel.p, n_sel, C, Tf, W, S, nullptr, false);
      is->valid_constraints = n;
      if (n < 6) {                                              // :276-284
        initial = initial_old; estimate = estimate_old;
        ls->termination = DVO_ORACLE_TOO_FEW_CONSTRAINTS;
        break;
      }
      if (iteration == 0) std::fill(S.weights.p, S.weights.p + n, 1.0f);   // :286-289
      else weights_pass(S.residuals.p, n, P, S.weights.p, cfg.mode);
      float Cv[4];
      scale_pass(S.residuals.p, S.weights.p, n, cfg.mode, Cv);
      invert2(Cv, P);                                           // :295
      const double ll = loglik_pass(S.residuals.p, n, P, cfg.mode);
      is->tdist_loglik = -ll;
      for (int i = 0; i < 4; ++i) is->tdist_precision[i] = P[i];
      double li[6];
      se3_log(initial, li);
      double sq = 0;
      for (int i = 0; i < 6; ++i) sq += li[i] * li[i];
      is->prior_loglik = cfg.mu * sq;                           // :302

      last_error = error;
      error = -ll;
      accept = error < last_error;                              // :312
      if (!accept) {
        initial = initial_old; estimate = estimate_old;
        ls->termination = DVO_ORACLE_LOGLIKELIHOOD_DECREASED;
        break;
      }
      normal_equations(S.points_error.p, S.weights.p, n, P, cfg.mode, A, b);
      for (int i = 0; i < 6; ++i) { A[i * 6 + i] += cfg.mu; b[i] += cfg.mu * li[i]; }   // :345-346
      ldlt_solve6(A, b, x);
      for (int i = 0; i < 6; ++i) is->increment[i] = x[i];
      std::memcpy(is->information, A, sizeof(A));
      ++iteration;
    } while (accept && inf_norm(x) > cfg.precision && !(iteration >= cfg.max_iterations_per_level));   // :357

    // :359-363 run after the loop however it was left, so they can overwrite the criterion set at a
    // break (x is then the previous solve, or log(inc) if the break came on the level's first pass)
    if (inf_norm(x) <= cfg.precision) ls->termination = DVO_ORACLE_INCREMENT_TOO_SMALL;
    if (iteration >= cfg.max_iterations_per_level) ls->termination = DVO_ORACLE_ITERATIONS_EXCEEDED;
  }

  // :368-373
  double M[16];
  se3_to_matrix(se3_inverse(estimate), M);
  std::memcpy(result->transformation, M, sizeof(M));
  fill_nan(result->information, 36);
  result->loglik = std::numeric_limits<double>::quiet_NaN();
  if (last_level && last_level_first_iter) {
    int idx = last_level->n_iterations - 1;
    if (last_level->termination == DVO_ORACLE_LOGLIKELIHOOD_DECREASED) idx -= 1;
    const int abs_idx = last_level->first_iteration_index + idx;
    if (idx >= 0 && abs_idx < cap_iters) {
      const oracle_iteration_stats& li = iters[abs_idx];
      for (int i = 0; i < 36; ++i) result->information[i] = li.information[i] * 0.008 * 0.008;
      result->loglik = li.tdist_loglik + li.prior_loglik;
    }
  }
  result->n_levels = n_levels_out;
  result->n_iterations_total = n_iters_out;
  return 0;
}

}  // namespace oracle

// =================================================================================================
// C interface
// =================================================================================================
using namespace oracle;

extern "C" {

const char* oracle_version(void) { return "dvo-oracle 3 (REF_SSE pinned bit-exactly against the reference's own translation units, oracle/_ref)"; }

oracle_pyramid* oracle_pyramid_create(int width, int height, const float K[4], const float* intensity, const float* depth, int levels) {
  if (width <= 0 || height <= 0 || levels < 1) return nullptr;
  oracle_pyramid* p = new oracle_pyramid();
  p->levels.resize(levels);
  Level& L0 = p->levels[0];
  L0.w = width; L0.h = height;
  L0.K = Intrinsics{K[0], K[1], K[2], K[3]};
  L0.plane[0].assign(intensity, intensity + size_t(width) * height);
  L0.plane[1].assign(depth, depth + size_t(width) * height);
  for (int l = 1; l < levels; ++l) {   // rgbd_image.cpp:156-172, :283-296
    Level& a = p->levels[l - 1];
    Level& b = p->levels[l];
    b.w = a.w / 2; b.h = a.h / 2;
    b.K = Intrinsics{a.K.fx * 0.5f, a.K.fy * 0.5f, a.K.ox * 0.5f, a.K.oy * 0.5f};   // intrinsic_matrix.cpp:90-93 (Q17)
    pyr_down_mean(a.plane[0], a.w, a.h, b.plane[0]);
    pyr_down_subsample(a.plane[1], a.w, a.h, b.plane[1]);
  }
  return p;
}

void oracle_pyramid_destroy(oracle_pyramid* p) { delete p; }

int oracle_pyramid_num_levels(const oracle_pyramid* p) { return int(p->levels.size()); }

const float* oracle_pyramid_plane(oracle_pyramid* p, int level, int plane, int* w, int* h, float K[4]) {
  if (!p || level < 0 || level >= int(p->levels.size()) || plane < 0 || plane > 5) return nullptr;
  Level& L = p->levels[level];
  derive_level(L);
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (K) { K[0] = L.K.fx; K[1] = L.K.fy; K[2] = L.K.ox; K[3] = L.K.oy; }
  return L.plane[plane].data();
}

int oracle_pyramid_select(oracle_pyramid* p, int level, float ithr, float dthr, uint8_t* mask) {
  if (!p || level < 0 || level >= int(p->levels.size())) return -1;
  return select_points(p->levels[level], ithr, dthr, mask);
}

void oracle_convert_raw_depth(const uint16_t* raw, float* out, int n, float scale) {
  for (int i = 0; i < n; ++i) out[i] = raw[i] == 0 ? kNaN : float(raw[i]) * scale;
}

void oracle_bgr_to_grey(const uint8_t* bgr, float* out, int n) {
  // OpenCV 2 CV_BGR2GRAY for 8-bit: fixed point, shift 14, coefficients B 1868, G 9617, R 4899
  for (int i = 0; i < n; ++i) {
    const int v = (bgr[3 * i] * 1868 + bgr[3 * i + 1] * 9617 + bgr[3 * i + 2] * 4899 + (1 << 13)) >> 14;
    out[i] = float(v);
  }
}

int oracle_match(oracle_pyramid* ref, oracle_pyramid* cur, const oracle_config* cfg, oracle_result* result,
                 oracle_level_stats* levels, int cap_levels, oracle_iteration_stats* iters, int cap_iters) {
  Scratch S;
  std::vector<oracle_iteration_stats> own;
  if (!iters) {   // the driver needs the per-iteration record of the last level for :369-373
    own.resize(size_t(cfg->first_level - cfg->last_level + 1) * size_t(cfg->max_iterations_per_level + 1));
    iters = own.data();
    cap_iters = int(own.size());
  }
  return match_impl(ref, cur, *cfg, result, levels, cap_levels, iters, cap_iters, S);
}

double oracle_match_batch(int n, oracle_pyramid** refs, oracle_pyramid** curs, const oracle_config* cfg, oracle_result* results, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  // derived planes / selections are cached inside the pyramids; build them up front (single thread)
  // so that worker threads only read shared pyramids (local_tracker.cpp:163-169 does the same).
  for (int i = 0; i < n; ++i)
    for (int l = cfg->last_level; l <= cfg->first_level; ++l) {
      derive_level(curs[i]->levels[l]);
      select_points(refs[i]->levels[l], cfg->intensity_derivative_threshold, cfg->depth_derivative_threshold, nullptr);
    }
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    Scratch S;
    std::vector<oracle_iteration_stats> it(size_t(cfg->first_level - cfg->last_level + 1) * size_t(cfg->max_iterations_per_level + 1));
    for (;;) {
      int i = next.fetch_add(1);
      if (i >= n) break;
      match_impl(refs[i], curs[i], *cfg, &results[i], nullptr, 0, it.data(), int(it.size()), S);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int oracle_level_iteration(oracle_pyramid* ref, oracle_pyramid* cur, int level, int mode, float ithr, float dthr,
                           const float T34[12], const float P_prev[4], int first_iteration_on_level,
                           oracle_iteration_out* out, float* residuals) {
  if (!ref || !cur || level < 0 || level >= int(ref->levels.size()) || level >= int(cur->levels.size())) return -1;
  Level& C = cur->levels[level];
  Level& R = ref->levels[level];
  derive_level(C);
  const size_t npx = size_t(R.w) * R.h;
  std::vector<uint8_t> mask(npx);
  const int n_sel = select_points(R, ithr, dthr, mask.data());
  std::vector<int> pix(n_sel);
  for (size_t i = 0, k = 0; i < npx; ++i) if (mask[i]) pix[k++] = int(i);
  Scratch S;
  ensure_scratch(S, std::max(npx, size_t(16)), true);
  const Weights8 W = make_weights(C.K);
  const int n = !uses_math_residuals(mode) ? residual_pass_ref(R.sel.p, n_sel, C, T34, W, S, pix.data(), true, quirks_of(mode))
                                           : residual_pass_math(R.sel.p, n_sel, C, T34, W, S, pix.data(), true);
  std::memset(out, 0, sizeof(*out));
  out->n = n;
  out->n_selected = n_sel;
  if (residuals) {
    for (size_t i = 0; i < 2 * npx; ++i) residuals[i] = kNaN;
    for (int i = 0; i < n; ++i) {
      residuals[2 * size_t(S.pixel_of[i]) + 0] = S.residuals.p[2 * i];
      residuals[2 * size_t(S.pixel_of[i]) + 1] = S.residuals.p[2 * i + 1];
    }
  }
  if (n < 6) return 1;
  if (first_iteration_on_level) std::fill(S.weights.p, S.weights.p + n, 1.0f);
  else weights_pass(S.residuals.p, n, P_prev, S.weights.p, mode);
  double sw = 0;
  for (int i = 0; i < n; ++i) sw += S.weights.p[i];
  out->sum_w = sw;
  float Cv[4];
  scale_pass(S.residuals.p, S.weights.p, n, mode, Cv);
  invert2(Cv, out->precision);
  std::memcpy(out->scale_cov, Cv, 3 * sizeof(float));
  out->neg_loglik = -loglik_pass(S.residuals.p, n, out->precision, mode);
  normal_equations(S.points_error.p, S.weights.p, n, out->precision, mode, out->A, out->b);
  return 0;
}

// ---- the passes one by one on caller-supplied arrays (pinning against oracle/_ref, tests/test_oracle_ref.py) ----------
int oracle_pass_residuals(int mode, int n, const float* points, const float* accel, int w, int h, const float K[4], const float T[12],
                          float* out_points, float* out_residuals) {
  Level cur;
  cur.w = w; cur.h = h;
  cur.K = Intrinsics{K[0], K[1], K[2], K[3]};
  cur.accel.resize(size_t(w) * h * 8);
  std::memcpy(cur.accel.p, accel, size_t(w) * h * 8 * sizeof(float));
  AlignedBuf<RefPoint> pts;
  pts.resize(size_t(n) + 1);
  for (int i = 0; i < n; ++i) {
    std::memcpy(pts.p[i].p, points + size_t(i) * 12, 4 * sizeof(float));
    std::memcpy(pts.p[i].v, points + size_t(i) * 12 + 4, 8 * sizeof(float));
  }
  Scratch S;
  ensure_scratch(S, size_t(n) + 2, false);
  const Weights8 W = make_weights(cur.K);
  const int n_out = !uses_math_residuals(mode) ? residual_pass_ref(pts.p, n, cur, T, W, S, nullptr, false, quirks_of(mode))
                                               : residual_pass_math(pts.p, n, cur, T, W, S, nullptr, false);
  for (int i = 0; i < n_out; ++i) {
    std::memcpy(out_points + size_t(i) * 12, S.points_error.p[i].p, 4 * sizeof(float));
    std::memcpy(out_points + size_t(i) * 12 + 4, S.points_error.p[i].v, 8 * sizeof(float));
    out_residuals[2 * i] = S.residuals.p[2 * i];
    out_residuals[2 * i + 1] = S.residuals.p[2 * i + 1];
  }
  return n_out;
}

void oracle_pass_weight_vectors(const float K[4], float reference_weight[8], float current_weight[8]) {
  const Weights8 W = make_weights(Intrinsics{K[0], K[1], K[2], K[3]});
  std::memcpy(reference_weight, W.wref, sizeof(W.wref));
  std::memcpy(current_weight, W.wcur, sizeof(W.wcur));
}

void oracle_pass_weights(int mode, int n, const float* residuals, const float P[4], float* weights) { weights_pass(residuals, n, P, weights, mode); }

void oracle_pass_scale(int mode, int n, const float* residuals, const float* weights, float C[4]) { scale_pass(residuals, weights, n, mode, C); }

double oracle_pass_loglik(int mode, int n, const float* residuals, const float P[4]) { return loglik_pass(residuals, n, P, mode); }

void oracle_se3_exp(const double x[6], double T[16]) { se3_to_matrix(se3_exp(x), T); }
void oracle_se3_log(const double T[16], double x[6]) { se3_log(se3_from_matrix(T), x); }
int oracle_solve6(const double A[36], const double b[6], double x[6]) { return ldlt_solve6(A, b, x); }

void oracle_rank_update_2x6(const float* J, int n, const float alpha[4], int mode, double A[36]) {
  // the fixture shape of dvo_core/src/sse_test.cpp:32-102 : A = sum_i J_i^T alpha J_i
  std::vector<RefPoint> dummy;
  if (quirks_of(mode) & DVO_ORACLE_Q_FLOAT_NORMAL_EQ) {
    float blk[24];
    std::memset(blk, 0, sizeof(blk));
    for (int i = 0; i < n; ++i) {
      const float* J0 = J + size_t(i) * 12;
      const float* J1 = J0 + 6;
      float ua[6], ub[6];
      for (int k = 0; k < 6; ++k) {
        ua[k] = J0[k] * alpha[0] + J1[k] * alpha[2];
        ub[k] = J0[k] * alpha[1] + J1[k] * alpha[3];
      }
      int o = 0;
      for (int bi = 0; bi < 6; bi += 2)
        for (int bj = bi; bj < 6; bj += 2) {
          blk[o + 0] += ua[bi] * J0[bj] + ub[bi] * J1[bj];
          blk[o + 1] += ua[bi] * J0[bj + 1] + ub[bi] * J1[bj + 1];
          blk[o + 2] += ua[bi + 1] * J0[bj] + ub[bi + 1] * J1[bj];
          blk[o + 3] += ua[bi + 1] * J0[bj + 1] + ub[bi + 1] * J1[bj + 1];
          o += 4;
        }
    }
    float Af[36];
    std::memset(Af, 0, sizeof(Af));
    int o = 0;
    for (int bi = 0; bi < 6; bi += 2)
      for (int bj = bi; bj < 6; bj += 2) {
        Af[bi * 6 + bj] = blk[o]; Af[bi * 6 + bj + 1] = blk[o + 1];
        Af[(bi + 1) * 6 + bj] = blk[o + 2]; Af[(bi + 1) * 6 + bj + 1] = blk[o + 3];
        o += 4;
      }
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) { A[i * 6 + j] = Af[i * 6 + j]; A[j * 6 + i] = Af[i * 6 + j]; }
  } else {
    double Ad[36];
    std::memset(Ad, 0, sizeof(Ad));
    for (int i = 0; i < n; ++i) {
      const float* J0 = J + size_t(i) * 12;
      const float* J1 = J0 + 6;
      for (int k = 0; k < 6; ++k) {
        const double ua = double(J0[k]) * alpha[0] + double(J1[k]) * alpha[2];
        const double ub = double(J0[k]) * alpha[1] + double(J1[k]) * alpha[3];
        for (int l = 0; l < 6; ++l) Ad[k * 6 + l] += ua * J0[l] + ub * J1[l];
      }
    }
    std::memcpy(A, Ad, sizeof(Ad));
  }
}

}  // extern "C"
