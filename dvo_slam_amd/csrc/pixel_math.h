// pixel_math.h -- the per-pixel arithmetic of the alignment kernels (host/device inline functions).
//
// Used by align_kernels.hip on the GPU.  tests/ also compiles this header with the host compiler to
// emulate the device code path pixel by pixel when no GPU is present (test-only).
//
// Semantics = the oracle's MATH mode: the reference algorithm (dvo_core/src/dense_tracking_impl.cpp:133-393,
// dvo_core/src/dense_tracking.cpp:217-220, 448-476) with exact division instead of _mm_rcp_ps and
// round-to-nearest instead of MXCSR round-toward-zero.  Everything that decides
// validity or feeds the residual is written with floating-point contraction OFF and in the oracle's
// operation order, so that the valid-pixel count and the residuals agree bit-for-bit with it.
#pragma once

#include "device_types.h"

namespace dvo_hip {

// reciprocal for the Jacobian (not parity-critical): v_rcp_f32 (1 ulp) on the device
DVO_HD float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// sqrt of the t-distribution weight in two instructions: sqrt(7 / (5 + q)) = sqrt(7) * rsq(5 + q) (v_rsq_f32, 1 ulp)
DVO_HD float fast_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}

// _mm_rcp_ps of the host CPU, from the table dumped at run time (LevelGeom::rcp_table): the instruction returns a value that depends
// on the leading mantissa bits of its operand only (2^11 table entries on Intel, 2^12 on AMD Zen 5 -- probed, never assumed) and
// scales exactly with its exponent.  Zero, subnormal, infinite and NaN operands are not reproduced (a depth is none of them).
DVO_HD float rcp_like_the_host(const float* table, int shift, float x) {
  union { float f; unsigned u; } in, t;
  in.f = x;
  t.f = table[(in.u & 0x7fffffu) >> shift];             // rcp(1.m), in (0.5, 1]
  t.u = (t.u + ((127u - ((in.u >> 23) & 0xffu)) << 23)) | (in.u & 0x80000000u);   // x 2^-(e - 127), sign of x
  return t.f;
}

struct PixelTerms {
  float r0, r1;       // intensity / depth residual
  float gix, giy;     // intensity gradient row  (0.5 fx (Icx + Irx)/255 , 0.5 fy (Icy + Iry)/255)
  float gzx, gzy;     // depth gradient row      (fx Zcx, fy Zcy)
  float X, Y, Z;      // untransformed reference point
};

// r^T P r with Eigen's (r^T P) r grouping; P row-major (dense_tracking_impl.cpp:415, :643)
DVO_HD float mahalanobis(float r0, float r1, const float* P) {
#pragma clang fp contract(off)
  return (r0 * P[0] + r1 * P[2]) * r0 + (r0 * P[1] + r1 * P[3]) * r1;
}

// t-distribution weight, nu = 5: (2+5)/(5 + r^T P r)  (dense_tracking_impl.cpp:640-644)
DVO_HD float tdist_weight(float r0, float r1, const float* P) {
#pragma clang fp contract(off)
  return 7.0f / (5.0f + mahalanobis(r0, r1, P));
}

// its square root (the matrix-core schedule scales the per-pixel vector by sqrt(w) on both operand sides)
DVO_HD float tdist_weight_sqrt(float r0, float r1, const float* P) {
  return 2.6457513110645906f * fast_rsqrt(5.0f + mahalanobis(r0, r1, P));
}

// C = S/(n-3) rounded to float, P = C^-1 in Eigen's 2x2 inverse order (dense_tracking.cpp:295)
DVO_HD void scale_to_precision(double c00, double c01, double c11, float* C, float* P) {
#pragma clang fp contract(off)
  C[0] = float(c00); C[1] = float(c01); C[2] = float(c11);
  const float det = C[0] * C[2] - C[1] * C[1];
  const float inv = 1.0f / det;
  P[0] = C[2] * inv;
  P[1] = -C[1] * inv;
  P[2] = -C[1] * inv;
  P[3] = C[0] * inv;
}

// float(K * T) in Eigen's coefficient order (dense_tracking_impl.cpp:142-148); T row-major 3x4 float
DVO_HD void make_KT(float fx, float fy, float ox, float oy, const float* T, float* KT) {
#pragma clang fp contract(off)
  for (int j = 0; j < 4; ++j) {
    KT[0 * 4 + j] = fx * T[0 * 4 + j] + 0.0f * T[1 * 4 + j] + ox * T[2 * 4 + j];
    KT[1 * 4 + j] = 0.0f * T[0 * 4 + j] + fy * T[1 * 4 + j] + oy * T[2 * 4 + j];
    KT[2 * 4 + j] = 0.0f * T[0 * 4 + j] + 0.0f * T[1 * 4 + j] + 1.0f * T[2 * 4 + j];
  }
}

// ---- the residual of one reference pixel, in three stages so that the kernel can software-pipeline them ------
// (1) pixel_project: warp + projection + bounds test          (dense_tracking_impl.cpp:148-203)
// (2) pixel_fetch:   the 4 + 4 bilinear taps                   (:212-213, :229-251)
// (3) pixel_finish:  bilinear blend, NaN / occlusion tests, residual and gradient rows (:256-281)
struct PixelProj {
  float X, Y, Z;      // untransformed reference point
  float qz;           // transformed depth
  float a1, b1;       // bilinear weights of the +1 taps
  int base;           // index of tap (u0, v0) in the current-frame planes
  bool ok;
};

struct PixelTaps {
  float4 A00, A10, A01, A11;
  float2 B00, B10, B01, B11;
};

// tx = (u - ox) / fx and ty = (v - oy) / fy of the pixel, from the per-level tables (rgbd_image.cpp:198-201)
DVO_HD PixelProj pixel_project_at(const LevelGeom& g, const float* KT, const float4 ref, float tx, float ty) {
#pragma clang fp contract(off)
  PixelProj p;
  p.ok = false;
  p.base = 0;
  p.a1 = p.b1 = 0.0f;
  const float Z = ref.x;
  p.Z = Z;
  const float X = tx * Z;                           // rgbd_image.cpp:258
  const float Y = ty * Z;
  p.X = X; p.Y = Y;
  const float qx = (KT[0] * X + KT[1] * Y) + (KT[2] * Z + KT[3]);
  const float qy = (KT[4] * X + KT[5] * Y) + (KT[6] * Z + KT[7]);
  const float qz = (KT[8] * X + KT[9] * Y) + (KT[10] * Z + KT[11]);
  p.qz = qz;
  if (!(Z == Z)) return p;                          // not selected / no depth (Q19)
  float u, v;
  if (g.rcp_table) {                                // reference-compatible (option "ref_compat")
    const float r = rcp_like_the_host(g.rcp_table, g.rcp_shift, qz);
    u = qx * r;
    v = qy * r;
  } else {
    u = qx / qz;                                    // correctly rounded division (MATH semantics, Q1)
    v = qy / qz;
  }
  if (!(u >= 0.0f && u <= float(g.w - 2) && v >= 0.0f && v <= float(g.h - 2))) return p;   // Q4
  const float uf = floorf(u), vf = floorf(v);
  p.a1 = u - uf;
  p.b1 = v - vf;
  p.base = int(vf) * g.w + int(uf);
  p.ok = true;
  return p;
}

DVO_HD PixelProj pixel_project(const LevelGeom& g, const float* KT, const float4 ref, int u_r, int v_r) {
  return pixel_project_at(g, KT, ref, g.tx[u_r], g.ty[v_r]);
}

template <typename PtrA, typename PtrB>
DVO_HD void pixel_fetch(const LevelGeom& g, PtrA curA, PtrB curB, const PixelProj& p, PixelTaps& t) {
  const int base = p.base;
  t.A00 = curA[base]; t.A10 = curA[base + 1]; t.A01 = curA[base + g.w]; t.A11 = curA[base + g.w + 1];
  t.B00 = curB[base]; t.B10 = curB[base + 1]; t.B01 = curB[base + g.w]; t.B11 = curB[base + g.w + 1];
}

DVO_HD bool pixel_finish(const LevelGeom& g, const float4 ref, const PixelProj& p, const PixelTaps& t, PixelTerms& o) {
#pragma clang fp contract(off)
  const float a1 = p.a1, a0 = 1.0f - a1, b1 = p.b1, b0 = 1.0f - b1;
  // intensity and depth feed the residual and the validity tests: reference operation order, no contraction
#define DVO_BILERP(f) (b0 * (a0 * t.A00.f + a1 * t.A10.f) + b1 * (a0 * t.A01.f + a1 * t.A11.f))
  const float cI = DVO_BILERP(x), cZ = DVO_BILERP(y);
#undef DVO_BILERP
  // the four gradient channels only feed the Jacobian (and the NaN test): same formula with fused multiply-adds
#define DVO_BILERP_FMA(v00, v10, v01, v11) fmaf(b1, fmaf(a1, v11, a0 * (v01)), b0 * fmaf(a1, v10, a0 * (v00)))
  const float cIx = DVO_BILERP_FMA(t.A00.z, t.A10.z, t.A01.z, t.A11.z);
  const float cIy = DVO_BILERP_FMA(t.A00.w, t.A10.w, t.A01.w, t.A11.w);
  const float cZx = DVO_BILERP_FMA(t.B00.x, t.B10.x, t.B01.x, t.B11.x);
  const float cZy = DVO_BILERP_FMA(t.B00.y, t.B10.y, t.B01.y, t.B11.y);
#undef DVO_BILERP_FMA
  if (!(cI == cI && cZ == cZ && cIx == cIx && cIy == cIy && cZx == cZx && cZy == cZy)) return false;   // Q9
  // residual = wcur * interpolated + wref * reference with the weights of dense_tracking.cpp:217-220
  const float inv255 = 1.0f / 255.0f;
  o.r0 = inv255 * cI + (-inv255) * ref.y;
  o.r1 = 1.0f * cZ + (-1.0f) * p.qz;                 // reference depth := transformed z (:269)
  float sigma = p.Z - 0.4f;                          // dense_tracking_impl.cpp:122-128
  sigma = 0.0012f + 0.0019f * sigma * sigma;
  if (!(o.r1 > -20.0f * sigma)) return false;        // occlusion test (Q5)
  const float wi_x = g.wi_x, wi_y = g.wi_y;          // 0.5 fx / 255, 0.5 fy / 255 (dense_tracking.cpp:219-220)
  o.gix = wi_x * cIx + wi_x * ref.z;                 // ESM-style average of both gradients (Q10)
  o.giy = wi_y * cIy + wi_y * ref.w;
  o.gzx = (1.0f * g.fx) * cZx;                       // wref_zd = 0: current-frame depth gradient only
  o.gzy = (1.0f * g.fy) * cZy;
  o.X = p.X; o.Y = p.Y; o.Z = p.Z;
  return true;
}

// the three stages back to back
template <typename PtrA, typename PtrB>
DVO_HD bool pixel_residual(const LevelGeom& g, const float* KT, PtrA curA, PtrB curB, const float4 ref, int u_r, int v_r,
                           PixelTerms& o) {
  const PixelProj p = pixel_project(g, KT, ref, u_r, v_r);
  if (!p.ok) return false;
  PixelTaps t;
  pixel_fetch(g, curA, curB, p, t);
  return pixel_finish(g, ref, p, t, o);
}

// 2x6 Jacobian rows at the UNtransformed reference point (Q10; dense_tracking.cpp:448-476, :333-340), scaled by `s`
// (1 for the plain rows; sqrt(w) for the matrix-core schedule).  Written out term by term: the generic
// "g.x * Jw(0,k) + g.y * Jw(1,k)" wastes a multiply-add on each of the four structural zeros of Jw.
//   Jw row 0 = [1/z, 0, -x/z^2, -xy/z^2, 1 + x^2/z^2, -y/z]     Jw row 1 = [0, 1/z, -y/z^2, -(1 + y^2/z^2), xy/z^2, x/z]
//   J0 = gI . Jw      J1 = gZ . Jw - [0, 0, 1, y, -x, 0]
DVO_HD void jacobian_rows_scaled(const PixelTerms& t, float s, float* J0, float* J1) {
  const float iz = fast_rcp(t.Z), iz2 = iz * iz;
  const float a2 = -t.X * iz2, b2 = -t.Y * iz2;
  const float a3 = a2 * t.Y;                 // Jw(0,3); Jw(1,4) = -a3
  const float a4 = 1.0f - a2 * t.X;          // Jw(0,4)
  const float b3 = -1.0f + b2 * t.Y;         // Jw(1,3)
  const float a5 = -t.Y * iz, b5 = t.X * iz; // Jw(0,5), Jw(1,5)
  const float gix = s * t.gix, giy = s * t.giy, gzx = s * t.gzx, gzy = s * t.gzy;
  J0[0] = gix * iz;
  J0[1] = giy * iz;
  J0[2] = gix * a2 + giy * b2;
  J0[3] = gix * a3 + giy * b3;
  J0[4] = gix * a4 - giy * a3;
  J0[5] = gix * a5 + giy * b5;
  J1[0] = gzx * iz;
  J1[1] = gzy * iz;
  J1[2] = (gzx * a2 + gzy * b2) - s;
  J1[3] = (gzx * a3 + gzy * b3) - s * t.Y;
  J1[4] = (gzx * a4 - gzy * a3) + s * t.X;
  J1[5] = gzx * a5 + gzy * b5;
}

DVO_HD void jacobian_rows(const PixelTerms& t, float* J0, float* J1) { jacobian_rows_scaled(t, 1.0f, J0, J1); }

// ---- the same stages as straight-line code (what the matrix-core sweep runs) ------------------------------------------------
// No early exits: every lane computes everything and carries one predicate, so the row has no nested divergent regions and
// no per-exit default values.  The arithmetic is that of the stages above, bit for bit (valid count and residuals equal the
// oracle's MATH mode: correctly rounded divisions, no contraction, the reference's operation order) -- tests/emul checks the
// two forms against each other.  (Measured and dropped: fused multiply-adds throughout, u = qx * rcp(qz), the blend as four
// shared tap weights -- 35 fewer vector instructions per pixel, not a microsecond faster in the sweep, which is not bound by
// vector-ALU issue)
// COMPAT: the reference's u = x * rcp(z) with the host CPU's reciprocal table (option "ref_compat"; launch path and, since round 4,
// the resident kernel)
template <bool COMPAT = false>
DVO_HD PixelProj pixel_project_flat(const LevelGeom& g, const float* KT, float Z, float tx, float ty) {
#pragma clang fp contract(off)
  PixelProj p;
  p.Z = Z;                                          // NaN = not selected / no depth / outside the image (Q19): fails the bounds test
  const float X = tx * Z, Y = ty * Z;               // rgbd_image.cpp:258
  p.X = X; p.Y = Y;
  const float qx = (KT[0] * X + KT[1] * Y) + (KT[2] * Z + KT[3]);
  const float qy = (KT[4] * X + KT[5] * Y) + (KT[6] * Z + KT[7]);
  const float qz = (KT[8] * X + KT[9] * Y) + (KT[10] * Z + KT[11]);
  p.qz = qz;
  float u, v;
  if (COMPAT) {                                     // reference-compatible: u = x * rcp(z) (dense_tracking_impl.cpp:192, Q1)
    const float r = rcp_like_the_host(g.rcp_table, g.rcp_shift, qz);
    u = qx * r;
    v = qy * r;
  } else {
    u = qx / qz;                                    // correctly rounded division (MATH semantics)
    v = qy / qz;
  }
  p.ok = u >= 0.0f && u <= float(g.w - 2) && v >= 0.0f && v <= float(g.h - 2);   // Q4; false for NaN
  const float uf = floorf(u), vf = floorf(v);
  p.a1 = u - uf;
  p.b1 = v - vf;
  p.base = int(vf) * g.w + int(uf);                 // meaningless unless ok
  return p;
}

// nx / d and ny / d, CORRECTLY ROUNDED like the IEEE division (what the oracle's MATH mode computes), without the range scaling and
// special-case fix-up of the compiler's division sequence (v_div_scale x 2, v_div_fmas, v_div_fixup per quotient): one refined
// reciprocal shared by both quotients, then per quotient the two remainder corrections of that very sequence -- q = n r,
// q += r (n - d q) twice, every remainder exact in a fused multiply-add.  13 instead of ~24 vector instructions per pixel.  Valid
// where the division's own scaling is not needed: |d| and |n / d| well inside the normal range -- here d is a depth in metres and
// only quotients in [0, 32768) are ever used; zero, infinite and NaN operands give a quotient that fails the bounds test like the
// IEEE result does.  Checked bit for bit against the IEEE division on the device (scripts/ubench/div_check.hip: 2^32 operand pairs)
// and by every parity test of the sweep that uses it (valid counts and residuals bit-exact against the oracle).
DVO_HD void divide2_correctly_rounded(float nx, float ny, float d, float& u, float& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(d);                 // 1 ulp
  r = fmaf(fmaf(-d, r, 1.0f), r, r);                  // one Newton step
  float q = nx * r;
  q = fmaf(fmaf(-d, q, nx), r, q);
  u = fmaf(fmaf(-d, q, nx), r, q);
  q = ny * r;
  q = fmaf(fmaf(-d, q, ny), r, q);
  v = fmaf(fmaf(-d, q, ny), r, q);
#else
  u = nx / d;
  v = ny / d;
#endif
}

// The same projection for the sweep that stages the current frame in LDS (align_window.hip): returns the tap corner as (u0, v0)
// instead of a plane index, and leaves X / Y to the caller (it recomputes them where it needs them).
// DIVISION: 0 = the compiler's IEEE division, 1 = divide2_correctly_rounded (the same bits), 2 = the reference's x * rcp(z) with the
// host CPU's reciprocal table (option "ref_compat")
template <int DIVISION = 0>
DVO_HD PixelProj pixel_project_uv_flat(const LevelGeom& g, const float* KT, float Z, float tx, float ty, int& u0, int& v0) {
#pragma clang fp contract(off)
  PixelProj p;
  p.Z = Z;
  const float X = tx * Z, Y = ty * Z;
  p.X = X; p.Y = Y;
  const float qx = (KT[0] * X + KT[1] * Y) + (KT[2] * Z + KT[3]);
  const float qy = (KT[4] * X + KT[5] * Y) + (KT[6] * Z + KT[7]);
  const float qz = (KT[8] * X + KT[9] * Y) + (KT[10] * Z + KT[11]);
  p.qz = qz;
  float u, v;
  if (DIVISION == 2) {
    const float r = rcp_like_the_host(g.rcp_table, g.rcp_shift, qz);
    u = qx * r;
    v = qy * r;
  } else if (DIVISION == 1) {
    divide2_correctly_rounded(qx, qy, qz, u, v);
  } else {
    u = qx / qz;
    v = qy / qz;
  }
  p.ok = u >= 0.0f && u <= float(g.w - 2) && v >= 0.0f && v <= float(g.h - 2);
  const float uf = floorf(u), vf = floorf(v);
  p.a1 = u - uf;
  p.b1 = v - vf;
  u0 = int(uf);
  v0 = int(vf);
  p.base = 0;
  return p;
}

// blend, NaN / occlusion tests, residual and gradient rows of a lane whose taps were fetched; everything is computed, the
// return value says whether the pixel is a constraint
DVO_HD bool pixel_finish_flat(const LevelGeom& g, const float4 ref, const PixelProj& p, const PixelTaps& t, PixelTerms& o) {
#pragma clang fp contract(off)
  const float a1 = p.a1, a0 = 1.0f - a1, b1 = p.b1, b0 = 1.0f - b1;
#define DVO_BILERP(f) (b0 * (a0 * t.A00.f + a1 * t.A10.f) + b1 * (a0 * t.A01.f + a1 * t.A11.f))
  const float cI = DVO_BILERP(x), cZ = DVO_BILERP(y);
#undef DVO_BILERP
#define DVO_BILERP_FMA(v00, v10, v01, v11) fmaf(b1, fmaf(a1, v11, a0 * (v01)), b0 * fmaf(a1, v10, a0 * (v00)))
  const float cIx = DVO_BILERP_FMA(t.A00.z, t.A10.z, t.A01.z, t.A11.z);
  const float cIy = DVO_BILERP_FMA(t.A00.w, t.A10.w, t.A01.w, t.A11.w);
  const float cZx = DVO_BILERP_FMA(t.B00.x, t.B10.x, t.B01.x, t.B11.x);
  const float cZy = DVO_BILERP_FMA(t.B00.y, t.B10.y, t.B01.y, t.B11.y);
#undef DVO_BILERP_FMA
  const float inv255 = 1.0f / 255.0f;
  o.r0 = inv255 * cI + (-inv255) * ref.y;
  o.r1 = 1.0f * cZ + (-1.0f) * p.qz;
  float sigma = p.Z - 0.4f;
  sigma = 0.0012f + 0.0019f * sigma * sigma;
  o.gix = g.wi_x * cIx + g.wi_x * ref.z;
  o.giy = g.wi_y * cIy + g.wi_y * ref.w;
  o.gzx = (1.0f * g.fx) * cZx;
  o.gzy = (1.0f * g.fy) * cZy;
  o.X = p.X; o.Y = p.Y; o.Z = p.Z;
  return (cI == cI && cZ == cZ) && (cIx == cIx && cIy == cIy) && (cZx == cZx && cZy == cZy) && o.r1 > -20.0f * sigma;   // Q9, Q5
}

// The same for taps whose four gradient channels are plain differences next - previous, i.e. TWICE the stored central differences
// (the sweep that derives them from staged {I, Z} pixels, align_window.hip): the factor 0.5 is applied to the four blended
// channels.  Bit-identical to blending the halved differences: a multiplication by 0.5 is exact and commutes with the rounding of
// every product and fused multiply-add of the blend (no value here comes near the subnormal range: intensities are multiples of
// 2^-8 at the finest level a sweep of this kind handles, depths of 2e-4, weights differences of floats in [0, 640)).
// TAP_WEIGHTS: the four gradient channels (which only feed the Jacobian) are blended with four tap weights b a formed once, and every
// factor 0.5 of the central differences (taps and reference gradient: the caller passes ref.z / ref.w as plain differences) rides on the
// level constants -- twelve vector instructions fewer per pixel, last-bit differences in those channels; intensity and depth keep
// the reference's order.
// (Measured and not taken: intensity and depth blended with the same four tap weights -- ten instructions fewer per row, residuals off
// the reference's order in the last bit; the residuals of every schedule stay bit-identical to the oracle's.)
template <bool TAP_WEIGHTS = false>
DVO_HD bool pixel_finish_flat_d(const LevelGeom& g, const float4 ref, const PixelProj& p, const PixelTaps& t, PixelTerms& o) {
#pragma clang fp contract(off)
  const float a1 = p.a1, a0 = 1.0f - a1, b1 = p.b1, b0 = 1.0f - b1;
#define DVO_BILERP(f) (b0 * (a0 * t.A00.f + a1 * t.A10.f) + b1 * (a0 * t.A01.f + a1 * t.A11.f))
  const float cI = DVO_BILERP(x), cZ = DVO_BILERP(y);
#undef DVO_BILERP
  float cIx, cIy, cZx, cZy;
  if (TAP_WEIGHTS) {
    // TWICE the gradient channels: the taps are differences without the central difference's 0.5, and so is the reference
    // gradient the caller hands over in ref.z / ref.w; the factor rides on the level constants below (0.5 wi, 0.5 f)
    const float w00 = b0 * a0, w10 = b0 * a1, w01 = b1 * a0, w11 = b1 * a1;
#define DVO_BLEND4(v00, v10, v01, v11) fmaf(w11, v11, fmaf(w01, v01, fmaf(w10, v10, w00 * (v00))))
    cIx = DVO_BLEND4(t.A00.z, t.A10.z, t.A01.z, t.A11.z);
    cIy = DVO_BLEND4(t.A00.w, t.A10.w, t.A01.w, t.A11.w);
    cZx = DVO_BLEND4(t.B00.x, t.B10.x, t.B01.x, t.B11.x);
    cZy = DVO_BLEND4(t.B00.y, t.B10.y, t.B01.y, t.B11.y);
#undef DVO_BLEND4
  } else {
#define DVO_BILERP_FMA(v00, v10, v01, v11) fmaf(b1, fmaf(a1, v11, a0 * (v01)), b0 * fmaf(a1, v10, a0 * (v00)))
    cIx = 0.5f * DVO_BILERP_FMA(t.A00.z, t.A10.z, t.A01.z, t.A11.z);
    cIy = 0.5f * DVO_BILERP_FMA(t.A00.w, t.A10.w, t.A01.w, t.A11.w);
    cZx = 0.5f * DVO_BILERP_FMA(t.B00.x, t.B10.x, t.B01.x, t.B11.x);
    cZy = 0.5f * DVO_BILERP_FMA(t.B00.y, t.B10.y, t.B01.y, t.B11.y);
#undef DVO_BILERP_FMA
  }
  const float inv255 = 1.0f / 255.0f;
  o.r0 = inv255 * cI + (-inv255) * ref.y;
  o.r1 = 1.0f * cZ + (-1.0f) * p.qz;
  float sigma = p.Z - 0.4f;
  sigma = 0.0012f + 0.0019f * sigma * sigma;
  if (TAP_WEIGHTS) {
    o.gix = g.half_wi_x * (cIx + ref.z);                      // (these rows only feed the Jacobian: one rounding fewer, not the reference's order)
    o.giy = g.half_wi_y * (cIy + ref.w);
    o.gzx = g.half_fx * cZx;
    o.gzy = g.half_fy * cZy;
  } else {
    o.gix = g.wi_x * cIx + g.wi_x * ref.z;
    o.giy = g.wi_y * cIy + g.wi_y * ref.w;
    o.gzx = (1.0f * g.fx) * cZx;
    o.gzy = (1.0f * g.fy) * cZy;
  }
  o.X = p.X; o.Y = p.Y; o.Z = p.Z;
  // intensities are never NaN: the three depth channels carry every hole of the 12-pixel neighbourhood (Q9)
  return (cI == cI && cZ == cZ) && (cIx == cIx && cIy == cIy) && (cZx == cZx && cZy == cZy) && o.r1 > -20.0f * sigma;   // Q9, Q5
}

// sqrt(7 / (5 + r^T P r)) through the symmetric form, P2x = P01 + P10 (the weight only scales the normal equations; it does not
// decide validity or enter the stored residuals)
DVO_HD float tdist_weight_sqrt_fast(float r0, float r1, float P00, float P2x, float P11) {
  const float t = fmaf(P2x, r1, P00 * r0);
  return 2.6457513110645906f * fast_rsqrt(fmaf(t, r0, fmaf(P11 * r1, r1, 5.0f)));
}

// reference-compatible: the square root of w = 7 * rcp(5 + r^T P r) with the host CPU's reciprocal (dense_tracking_impl.cpp:700, Q1)
DVO_HD float tdist_weight_sqrt_compat(const float* table, int shift, float r0, float r1, const float* P) {
  return sqrtf(7.0f * rcp_like_the_host(table, shift, 5.0f + mahalanobis(r0, r1, P)));
}

// The scaled Jacobian rows in normalised coordinates: with x = tx z, y = ty z the warp Jacobian is
//   Jw row 0 = [1/z, 0, -tx/z, -tx ty, 1 + tx^2, -ty]     Jw row 1 = [0, 1/z, -ty/z, -(1 + ty^2), tx ty, tx]
// cx = 1 + tx^2 (a constant of the column), cy = 1 + ty^2 (of the row).
DVO_HD void jacobian_rows_fast(const PixelTerms& t, float s, float tx, float ty, float cx, float cy, float* J0, float* J1) {
  const float iz = fast_rcp(t.Z);
  const float txy = tx * ty;
  const float gix = s * t.gix, giy = s * t.giy, gzx = s * t.gzx, gzy = s * t.gzy;
  // negations are written on the operands (source modifiers), not on the results (an extra instruction each)
  J0[0] = gix * iz;
  J0[1] = giy * iz;
  J0[2] = fmaf(-ty, J0[1], -tx * J0[0]);
  J0[3] = fmaf(-giy, cy, -gix * txy);
  J0[4] = fmaf(gix, cx, giy * txy);
  J0[5] = fmaf(giy, tx, -gix * ty);
  J1[0] = gzx * iz;
  J1[1] = gzy * iz;
  J1[2] = fmaf(-ty, J1[1], fmaf(-tx, J1[0], -s));
  J1[3] = fmaf(-gzy, cy, fmaf(-gzx, txy, -s * t.Y));
  J1[4] = fmaf(gzx, cx, fmaf(gzy, txy, s * t.X));
  J1[5] = fmaf(gzy, tx, -gzx * ty);
}

// Rank update of the P-independent Gram sums (layout in device_types.h) with one pixel's rows.
DVO_HD void accumulate_pixel(float* acc, const PixelTerms& t, float w) {
  float J0[6], J1[6], wJ0[6], wJ1[6];
  jacobian_rows(t, J0, J1);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    wJ0[i] = w * J0[i];
    wJ1[i] = w * J1[i];
  }
  // explicit fused multiply-adds: one instruction per accumulated product
  const float wr0 = w * t.r0, wr1 = w * t.r1;
  acc[kAccN] += 1.0f;
  acc[kAccS + 0] = fmaf(wr0, t.r0, acc[kAccS + 0]);
  acc[kAccS + 1] = fmaf(wr0, t.r1, acc[kAccS + 1]);
  acc[kAccS + 2] = fmaf(wr1, t.r1, acc[kAccS + 2]);
  int o = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = i; j < 6; ++j) {
      acc[kAccJ00 + o] = fmaf(wJ0[i], J0[j], acc[kAccJ00 + o]);
      acc[kAccJ11 + o] = fmaf(wJ1[i], J1[j], acc[kAccJ11 + o]);
      acc[kAccJ01 + o] = fmaf(wJ0[j], J1[i], fmaf(wJ0[i], J1[j], acc[kAccJ01 + o]));
      ++o;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    acc[kAccB00 + i] = fmaf(wJ0[i], t.r0, acc[kAccB00 + i]);
    acc[kAccB01 + i] = fmaf(wJ1[i], t.r0, fmaf(wJ0[i], t.r1, acc[kAccB01 + i]));
    acc[kAccB11 + i] = fmaf(wJ1[i], t.r1, acc[kAccB11 + i]);
  }
}

}  // namespace dvo_hip
