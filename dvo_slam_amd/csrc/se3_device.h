// se3_device.h -- float64 SE(3) bookkeeping and the 6x6 solve, executed by one lane per frame pair
// on the device so that the Gauss-Newton loop never round-trips through the host.
//
// Replaces the Sophus::SE3d calls of dvo_core/src/dense_tracking.cpp:147-150, 238, 259-261, 302, 346,
// 371 and Eigen's A.ldlt().solve(b) of :347.  Rotation-matrix (Rodrigues) formulation with
// Gaussian elimination -- deliberately a different algorithm from the oracle's quaternion/LDL^T
// restatement so that the parity tests compare two independent implementations.
#pragma once

#include "device_types.h"

namespace dvo_hip {

DVO_HD void se3_identity(SE3d& T) {
  for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  T.t[0] = T.t[1] = T.t[2] = 0.0;
}

DVO_HD void mat3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}

// C = A * B
DVO_HD void se3_mul(const SE3d& A, const SE3d& B, SE3d& C) {
  SE3d r;
  mat3_mul(A.R, B.R, r.R);
  for (int i = 0; i < 3; ++i) r.t[i] = A.R[i * 3] * B.t[0] + A.R[i * 3 + 1] * B.t[1] + A.R[i * 3 + 2] * B.t[2] + A.t[i];
  C = r;
}

DVO_HD void se3_inverse(const SE3d& A, SE3d& C) {
  SE3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = A.R[j * 3 + i];
  for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[i * 3] * A.t[0] + r.R[i * 3 + 1] * A.t[1] + r.R[i * 3 + 2] * A.t[2]);
  C = r;
}

DVO_HD void hat3(const double* w, double* O) {
  O[0] = 0;     O[1] = -w[2]; O[2] = w[1];
  O[3] = w[2];  O[4] = 0;     O[5] = -w[0];
  O[6] = -w[1]; O[7] = w[0];  O[8] = 0;
}

// x = (upsilon, omega)
DVO_HD void se3_exp(const double* x, SE3d& T) {
  const double* u = x;
  const double* w = x + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = sqrt(th2);
  double a, b, c;   // R = I + a O + b O^2 ; V = I + b O + c O^2
  if (th < 1e-5) {
    a = 1.0 - th2 / 6.0;
    b = 0.5 - th2 / 24.0;
    c = 1.0 / 6.0 - th2 / 120.0;
  } else {
    // (one division instead of three: a float64 division is ~150 cycles of one lane's serial time, and this lane is the solver step)
    const double s = sin(th), co = cos(th);
    const double inv2 = 1.0 / th2, inv1 = th * inv2;          // 1 / th^2, 1 / th
    a = s * inv1;
    b = (1.0 - co) * inv2;
    c = (th - s) * (inv2 * inv1);
  }
  double O[9], O2[9];
  hat3(w, O);
  mat3_mul(O, O, O2);
  for (int i = 0; i < 9; ++i) T.R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) + b * O[i * 3 + j] + c * O2[i * 3 + j]) * u[j];
    T.t[i] = s;
  }
}

DVO_HD void se3_log(const SE3d& T, double* x) {
  const double* R = T.R;
  // axis * sin(theta) from the skew part, cos(theta) from the trace
  double v[3] = {0.5 * (R[7] - R[5]), 0.5 * (R[2] - R[6]), 0.5 * (R[3] - R[1])};
  const double s = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double c = 0.5 * (R[0] + R[4] + R[8] - 1.0);
  const double th = atan2(s, c);
  double w[3];
  if (s < 1e-7) {
    if (c > 0) {   // theta ~ 0 : theta / sin(theta) ~ 1 + theta^2/6
      const double f = 1.0 + th * th / 6.0;
      for (int i = 0; i < 3; ++i) w[i] = f * v[i];
    } else {       // theta ~ pi : axis from the symmetric part
      double d[3] = {0.5 * (R[0] + 1.0), 0.5 * (R[4] + 1.0), 0.5 * (R[8] + 1.0)};
      int k = 0;
      if (d[1] > d[k]) k = 1;
      if (d[2] > d[k]) k = 2;
      double ax[3];
      ax[k] = sqrt(d[k] > 0 ? d[k] : 0);
      for (int i = 0; i < 3; ++i)
        if (i != k) ax[i] = 0.25 * (R[i * 3 + k] + R[k * 3 + i]) / (ax[k] > 0 ? ax[k] : 1.0);
      for (int i = 0; i < 3; ++i) w[i] = th * ax[i];
    }
  } else {
    const double f = th / s;
    for (int i = 0; i < 3; ++i) w[i] = f * v[i];
  }
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double t = sqrt(th2);
  double cc;   // Vinv = I - 0.5 O + cc O^2
  if (t < 1e-5) {
    cc = 1.0 / 12.0 + th2 / 720.0;
  } else {
    const double half = 0.5 * t;
    cc = (1.0 - t * cos(half) / (2.0 * sin(half))) / th2;
  }
  double O[9], O2[9];
  hat3(w, O);
  mat3_mul(O, O, O2);
  for (int i = 0; i < 3; ++i) {
    double sum = 0;
    for (int j = 0; j < 3; ++j) sum += (((i == j) ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + cc * O2[i * 3 + j]) * T.t[j];
    x[i] = sum;
  }
  x[3] = w[0]; x[4] = w[1]; x[5] = w[2];
}

// Symmetric 6x6 solve (Eigen's A.ldlt().solve(b) at dense_tracking.cpp:347).  Fast path: unpivoted LDL^T with
// compile-time indices only, so the whole factorisation stays in registers (a runtime-indexed 6x7 array lives in
// scratch memory and made this solve half of the solver kernel's serial time).  A = J^T W J + mu I is symmetric
// positive definite whenever the alignment is well posed, and LDL^T without pivoting is backward stable for SPD
// input.  If a pivot is not positive (rank-deficient or indefinite A) the pivoted elimination below takes over.
DVO_HD bool solve6_pivoted(const double* Ain, const double* bin, double* x);

DVO_HD bool solve6(const double* A, const double* b, double* x) {
  double L[6][6], D[6], Dinv[6], y[6];
  bool spd = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k] * D[k];
    D[j] = d;
    spd = spd && (d > 0.0);
    const double inv = 1.0 / d;
    Dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * D[k];
      L[i][j] = v * inv;
    }
  }
  if (!spd) return solve6_pivoted(A, b, x);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] *= Dinv[i];              // (the pivots' reciprocals of the factorisation: six divisions fewer)
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
    x[i] = v;
  }
  return true;
}

// Gaussian elimination with partial pivoting on the 6x6 system; returns false on a singular matrix
// (x is then NaN, which ends the Gauss-Newton loop exactly like a NaN from Eigen's LDLT would).
DVO_HD bool solve6_pivoted(const double* Ain, const double* bin, double* x) {
  double M[6][7];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) M[i][j] = Ain[i * 6 + j];
    M[i][6] = bin[i];
  }
  bool ok = true;
  for (int k = 0; k < 6; ++k) {
    int p = k;
    double best = fabs(M[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(M[i][k]) > best) { best = fabs(M[i][k]); p = i; }
    if (!(best > 0.0)) { ok = false; break; }
    if (p != k)
      for (int j = 0; j < 7; ++j) { double t = M[k][j]; M[k][j] = M[p][j]; M[p][j] = t; }
    const double inv = 1.0 / M[k][k];
    for (int i = k + 1; i < 6; ++i) {
      const double f = M[i][k] * inv;
      for (int j = k; j < 7; ++j) M[i][j] -= f * M[k][j];
    }
  }
  if (!ok) {
    for (int i = 0; i < 6; ++i) x[i] = __builtin_nan("");
    return false;
  }
  for (int i = 5; i >= 0; --i) {
    double s = M[i][6];
    for (int j = i + 1; j < 6; ++j) s -= M[i][j] * x[j];
    x[i] = s / M[i][i];
  }
  return true;
}

// det of a symmetric 6x6 (row-major) as the product of the pivots of an unpivoted LDL^T (exact whenever the leading
// minors are non-zero, definite or not; ~100 flops in registers).
DVO_HD double sym6_determinant(const double* A) {
  double L[6][6], D[6], det = 1.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k] * D[k];
    D[j] = d;
    det *= d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k] * D[k];
      L[i][j] = v * inv;
    }
  }
  return det;
}

// Eigenvalues of a symmetric 6x6 (row-major) by cyclic Jacobi rotations, unsorted.  Every (p, q) index is a compile-time
// constant after unrolling, so on the device the matrix lives in registers (a runtime-indexed local array would go to
// scratch memory).  Relative accuracy ~1e-15 for the positive definite information matrices this is used on.
DVO_HD void sym6_eigenvalues(const double* A, double* ev) {
  double a[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) a[i][j] = A[i * 6 + j];
#pragma unroll 1
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      diag += a[p][p] * a[p][p];
#pragma unroll
      for (int q = p + 1; q < 6; ++q) off += a[p][q] * a[p][q];
    }
    if (off <= 1e-32 * diag) break;                       // NaN input: never true, the sweeps run out and NaN is returned
#pragma unroll
    for (int p = 0; p < 5; ++p) {
#pragma unroll
      for (int q = p + 1; q < 6; ++q) {
        const double apq = a[p][q];
        if (apq != 0.0) {
          const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
          a[p][p] -= t * apq;
          a[q][q] += t * apq;
          a[p][q] = 0.0;
          a[q][p] = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            if (k != p && k != q) {
              const double akp = a[k][p], akq = a[k][q];
              a[k][p] = c * akp - sn * akq;
              a[p][k] = a[k][p];
              a[k][q] = sn * akp + c * akq;
              a[q][k] = a[k][q];
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) ev[i] = a[i][i];
}

}  // namespace dvo_hip
