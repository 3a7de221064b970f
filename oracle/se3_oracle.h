/*
 * se3_oracle.h -- float64 SE(3) exp/log/compose and a pivoted 6x6 LDL^T solve for the CPU oracle.
 *
 * TEST INFRASTRUCTURE (see dvo_oracle.h).  Restates the closed forms of the un-vendored Sophus
 * dependency (sophus/Makefile:5-8 clones strasdat/Sophus at HEAD, no pinned revision) as used at
 * dvo_core/src/dense_tracking.cpp:147-150, 238, 259-261, 302, 346, 371, and Eigen's
 * A.ldlt().solve(b) at dense_tracking.cpp:347.  Twist ordering is Sophus': x = (upsilon, omega).
 * Rotation goes through a unit quaternion exactly as Sophus::SO3 does.
 */
#ifndef SE3_ORACLE_H_
#define SE3_ORACLE_H_

#include <cmath>
#include <cstring>

namespace oracle {

struct Quat { double w, x, y, z; };

struct SE3 {
  Quat q;        // unit quaternion
  double t[3];
  SE3() : q{1, 0, 0, 0}, t{0, 0, 0} {}
};

inline Quat quat_normalized(Quat q) {
  double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return Quat{q.w / n, q.x / n, q.y / n, q.z / n};
}

inline Quat quat_mul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
              a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}

inline void quat_to_matrix(const Quat& q, double R[9]) {
  const double w = q.w, x = q.x, y = q.y, z = q.z;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// rotation matrix -> unit quaternion (Shepperd), as Eigen::Quaterniond(R) does for Sophus::SO3(R)
inline Quat matrix_to_quat(const double R[9]) {
  Quat q;
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q.w = 0.5 * s;
    s = 0.5 / s;
    q.x = (R[7] - R[5]) * s; q.y = (R[2] - R[6]) * s; q.z = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * s;
    s = 0.5 / s;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * s;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return quat_normalized(q);
}

inline void rotate(const Quat& q, const double v[3], double out[3]) {
  double R[9];
  quat_to_matrix(q, R);
  double o0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double o1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double o2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  out[0] = o0; out[1] = o1; out[2] = o2;
}

inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r;
  r.q = quat_normalized(quat_mul(a.q, b.q));
  double rb[3];
  rotate(a.q, b.t, rb);
  for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + rb[i];
  return r;
}

inline SE3 se3_inverse(const SE3& a) {
  SE3 r;
  r.q = Quat{a.q.w, -a.q.x, -a.q.y, -a.q.z};
  double rt[3];
  rotate(r.q, a.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = -rt[i];
  return r;
}

inline void hat(const double w[3], double O[9]) {
  O[0] = 0;     O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2];  O[4] = 0;     O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0];  O[8] = 0;
}

inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  std::memcpy(C, T, sizeof(T));
}

// exp: x = (upsilon, omega)
inline SE3 se3_exp(const double x[6]) {
  const double* u = x;
  const double* w = x + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = std::sqrt(th2);
  SE3 r;
  double imag, real;
  if (th < 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
    real = 1.0 - th2 / 8.0 + th4 / 384.0;
  } else {
    imag = std::sin(0.5 * th) / th;
    real = std::cos(0.5 * th);
  }
  r.q = quat_normalized(Quat{real, imag * w[0], imag * w[1], imag * w[2]});
  double O[9], O2[9];
  hat(w, O);
  mat3_mul(O, O, O2);
  double c1, c2;   // V = I + c1*O + c2*O^2
  if (th < 1e-5) {
    c1 = 0.5 - th2 / 24.0;
    c2 = 1.0 / 6.0 - th2 / 120.0;
  } else {
    c1 = (1.0 - std::cos(th)) / th2;
    c2 = (th - std::sin(th)) / (th2 * th);
  }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 3; ++j) {
      double V = (i == j ? 1.0 : 0.0) + c1 * O[i * 3 + j] + c2 * O2[i * 3 + j];
      s += V * u[j];
    }
    r.t[i] = s;
  }
  return r;
}

inline void so3_log(const Quat& q, double w[3]) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  const double n = std::sqrt(n2);
  double f;
  if (n < 1e-10) {
    f = 2.0 / q.w - 2.0 * n2 / (3.0 * q.w * q.w * q.w);
  } else if (std::fabs(q.w) < 1e-10) {
    f = (q.w > 0 ? M_PI : -M_PI) / n;
  } else {
    f = 2.0 * std::atan(n / q.w) / n;
  }
  w[0] = f * q.x; w[1] = f * q.y; w[2] = f * q.z;
}

inline void se3_log(const SE3& T, double x[6]) {
  double w[3];
  so3_log(T.q, w);
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = std::sqrt(th2);
  double O[9], O2[9];
  hat(w, O);
  mat3_mul(O, O, O2);
  double c;   // Vinv = I - 0.5*O + c*O^2
  if (th < 1e-5) {
    c = 1.0 / 12.0 + th2 / 720.0;
  } else {
    const double half = 0.5 * th;
    c = (1.0 - th * std::cos(half) / (2.0 * std::sin(half))) / th2;
  }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 3; ++j) {
      double Vi = (i == j ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + c * O2[i * 3 + j];
      s += Vi * T.t[j];
    }
    x[i] = s;
  }
  x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}

inline void se3_to_matrix(const SE3& T, double M[16]) {
  double R[9];
  quat_to_matrix(T.q, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
    M[i * 4 + 3] = T.t[i];
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
}

inline SE3 se3_from_matrix(const double M[16]) {
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = M[i * 4 + j];
  SE3 T;
  T.q = matrix_to_quat(R);
  T.t[0] = M[3]; T.t[1] = M[7]; T.t[2] = M[11];
  return T;
}

// Symmetric 6x6 solve by LDL^T with diagonal pivoting (largest remaining |diagonal|), the
// strategy of Eigen::LDLT.  Returns 0 on success, 1 if a zero pivot was met (x then holds the
// solution with the corresponding components set to 0, like Eigen's semidefinite handling).
inline int ldlt_solve6(const double Ain[36], const double bin[6], double x[6]) {
  double A[36], b[6];
  int perm[6];
  std::memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 6; ++i) { perm[i] = i; b[i] = bin[i]; }
  double D[6];
  int rc = 0;
  // work on the full symmetric matrix for simplicity
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = std::fabs(A[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i * 6 + i]) > best) { best = std::fabs(A[i * 6 + i]); p = i; }
    if (p != k) {
      for (int j = 0; j < 6; ++j) std::swap(A[k * 6 + j], A[p * 6 + j]);
      for (int j = 0; j < 6; ++j) std::swap(A[j * 6 + k], A[j * 6 + p]);
      std::swap(perm[k], perm[p]);
    }
    D[k] = A[k * 6 + k];
    if (D[k] == 0.0 || !std::isfinite(D[k])) {
      if (!std::isfinite(D[k])) { for (int i = 0; i < 6; ++i) x[i] = NAN; return 2; }
      rc = 1;
      for (int i = k + 1; i < 6; ++i) A[i * 6 + k] = 0;
      continue;
    }
    for (int i = k + 1; i < 6; ++i) A[i * 6 + k] /= D[k];   // L(i,k)
    for (int i = k + 1; i < 6; ++i)
      for (int j = k + 1; j < 6; ++j) A[i * 6 + j] -= A[i * 6 + k] * D[k] * A[j * 6 + k];
  }
  double y[6];
  for (int i = 0; i < 6; ++i) y[i] = b[perm[i]];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * 6 + j] * y[j];
  for (int i = 0; i < 6; ++i) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
  for (int i = 5; i >= 0; --i)
    for (int j = i + 1; j < 6; ++j) y[i] -= A[j * 6 + i] * y[j];
  for (int i = 0; i < 6; ++i) x[perm[i]] = y[i];
  return rc;
}

}  // namespace oracle
#endif
