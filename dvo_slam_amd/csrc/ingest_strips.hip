// ingest_strips.hip -- the ingest of raw sensor planes (level 0 in the frame's role + pyramid levels 1-3) with one 128 x 8 pixel
// strip per WAVEFRONT, everything in registers: no LDS, no barriers.
//
// Same arithmetic as k_build_from_raw (pyramid_kernels.hip), which remains the path for odd widths and unaligned planes -- the
// reference's ingest (dvo_core/src/core/surface_pyramid.cpp:65-105), 2x2-mean pyr-down and depth subsampling
// (dvo_core/src/core/rgbd_image.cpp:38-55, 127-139), clamped central differences (rgbd_image.cpp:419-489) and the selection predicate
// (dvo_core/src/core/point_selection.cpp:89-152): bit-identical planes.  What changed is how the work is laid out on the machine.
// k_build_from_raw gives a 64 x 16 tile to a 256-thread workgroup that converts it into LDS and meets at three barriers; a
// workgroup has 3 KB of loads in flight in two dependent rounds and the kernel moves 14 B per pixel at 1.7 of the 8 TB/s (1024
// frames of 640 x 480: 2.6 ms, the largest single item of a bench step after the finest-level sweeps).  Here
//   lane l of a wavefront owns pixel columns 2l, 2l + 1 of the strip and all 8 (+ 2 halo) rows: 16-20 independent loads per lane are
//     issued before the first is used (6 KB in flight per wavefront, 100+ KB per CU);
//   a lane's pixel pair is 16 B of every 8-B-per-pixel plane (R, C, B): one 16-B store per lane and row, 1 KiB per instruction;
//   the horizontal neighbours of a pair are the adjacent lanes' registers (DPP wave shifts; lanes 0 and 63 load the strip's edge
//     columns), the vertical ones the lane's own rows; the 2 x 2 means of levels 1-3 are in-lane sums and DPP row shifts;
//   the four wavefronts of a workgroup take four vertically adjacent strips, so their shared halo rows meet in the CU's cache.
// A current frame that is only read by the window sweep (plane C = {I, Z}, align_window.hip) needs no neighbours at all: TAPS = false
// drops the halo rows, the edge columns and the differences.
#include "global_ptr.h"
#include "launch.h"

namespace dvo_hip {

namespace {

constexpr int kStripW = 128, kStripH = 8, kStripsPerGroup = 4;
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138, kDppRowShl1 = 0x101, kDppRowShl2 = 0x102;

// lane i <- lane i + 1 of the wavefront; lane 63 keeps `edge`
__device__ __forceinline__ float from_next_lane(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), kDppWaveShl1, 0xf, 0xf, false));
}
// lane i <- lane i - 1 of the wavefront; lane 0 keeps `edge`
__device__ __forceinline__ float from_previous_lane(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), kDppWaveShr1, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float row_shifted(float v) {   // within a row of 16 lanes; the lanes that use it always have a source lane
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

}  // namespace

// The role planes of a strip's 8 rows from its register rows: I / Z[j] = image row y0 + kFirst + j (clamped), kFirst = -1 with TAPS;
// eI / eZ[r]: the strip's edge columns (lane 0: the column left of the strip, lane 63: the column right of it).  Returns the number of
// selected pixels of the wavefront (ROLE 1).
template <int ROLE, bool TAPS>
__device__ __forceinline__ int strip_role_planes(const float (&I)[TAPS ? kStripH + 2 : kStripH][2], const float (&Z)[TAPS ? kStripH + 2 : kStripH][2],
                                                 const float (&eI)[kStripH], const float (&eZ)[kStripH], int x, int y0, int w0, int h0, bool in_x,
                                                 Global<float2> R0, Global<float4> A0, Global<float2> B0, Global<float2> C0, int cur_flavor,
                                                 float ithr, float dthr) {
#pragma clang fp contract(off)
  constexpr int kFirst = TAPS ? -1 : 0;
  const float nanv = __builtin_nanf("");
  int count = 0;
#pragma unroll
      for (int r = 0; r < kStripH; ++r) {
        const int j = r - kFirst, y = y0 + r;
        const bool inside = in_x && y < h0;
        const size_t at = size_t(y) * w0 + x;
        const float i0 = I[j][0], i1 = I[j][1], z0 = Z[j][0], z1 = Z[j][1];
        if (TAPS) {
          // column x - 1 of the pair's first pixel: the previous lane's second pixel (lane 0: the edge column), clamped at the image
          // border like the reference's derivative code; column x + 2 of the second pixel likewise
          const float ie = eI[r], ze = eZ[r];
          const float il_n = from_previous_lane(i1, ie), zl_n = from_previous_lane(z1, ze);
          const float ir_n = from_next_lane(i0, ie), zr_n = from_next_lane(z0, ze);
          const float il = x > 0 ? il_n : i0, zl = x > 0 ? zl_n : z0;
          const float ir = x + 2 < w0 ? ir_n : i1, zr = x + 2 < w0 ? zr_n : z1;
          const float idx0 = (i1 - il) * 0.5f, idx1 = (ir - i0) * 0.5f;
          const float zdx0 = (z1 - zl) * 0.5f, zdx1 = (zr - z0) * 0.5f;
          const float idy0 = (I[j + 1][0] - I[j - 1][0]) * 0.5f, idy1 = (I[j + 1][1] - I[j - 1][1]) * 0.5f;
          const float zdy0 = (Z[j + 1][0] - Z[j - 1][0]) * 0.5f, zdy1 = (Z[j + 1][1] - Z[j - 1][1]) * 0.5f;
          if (ROLE == 1) {
            const bool ok0 = inside && z0 == z0 && zdx0 == zdx0 && zdy0 == zdy0 &&
                             (fabsf(idx0) > ithr || fabsf(idy0) > ithr || fabsf(zdx0) > dthr || fabsf(zdy0) > dthr);
            const bool ok1 = inside && z1 == z1 && zdx1 == zdx1 && zdy1 == zdy1 &&
                             (fabsf(idx1) > ithr || fabsf(idy1) > ithr || fabsf(zdx1) > dthr || fabsf(zdy1) > dthr);
            if (inside) gstore_pair(R0 + at, make_float4(ok0 ? z0 : nanv, i0, ok1 ? z1 : nanv, i1));
            count += __popcll(__ballot(ok0)) + __popcll(__ballot(ok1));   // wave-uniform
          } else if (inside) {
            if (cur_flavor & kCurAB) {
              gstore(A0 + at, make_float4(i0, z0, idx0, idy0));
              gstore(A0 + at + 1, make_float4(i1, z1, idx1, idy1));
              gstore_pair(B0 + at, make_float4(zdx0, zdy0, zdx1, zdy1));
            }
            if (cur_flavor & kCurC) gstore_pair(C0 + at, make_float4(i0, z0, i1, z1));
          }
        } else if (inside) {
          gstore_pair(C0 + at, make_float4(i0, z0, i1, z1));
        }
      }
  return count;
}

// ROLE: -1 = none (raw copy + pyramid only), 0 = current, 1 = reference (R + selection count, counter zeroed before).
// TAPS: level 0 needs the central differences (reference role; current role with the gathered taps A + B).
// c_levels: bit l set = pyramid level l (1-3) also gets the current role's {I, Z} plane C.
template <int ROLE, bool TAPS>
__global__ __launch_bounds__(256) void k_ingest_strips(const FrameBuildPtrs* __restrict__ tbl, float scale, int w0, int h0, int levels,
                                                       float ithr, float dthr, int groups_x, int groups_y, int n_frames, int cur_flavor, int c_levels) {
#pragma clang fp contract(off)
  static_assert(ROLE != 1 || TAPS, "the selection predicate needs the differences");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w1 = w0 >> 1, h1 = h0 >> 1, w2 = w1 >> 1, h2 = h1 >> 1, w3 = w2 >> 1, h3 = h2 >> 1;
  const float nanv = __builtin_nanf("");
  const int per_frame = groups_x * groups_y, total = per_frame * n_frames;
  for (int gi = blockIdx.x; gi < total; gi += gridDim.x) {
    const int frame = gi / per_frame, t = gi - frame * per_frame;
    const int sx = t % groups_x, sy = (t / groups_x) * kStripsPerGroup + wave;
    const int y0 = sy * kStripH;
    if (y0 >= h0) continue;                                     // (wave-uniform; nothing below synchronises the workgroup)
    const FrameBuildPtrs& f = tbl[frame];
    const auto grey = global_ptr(f.grey);
    const auto raw = global_ptr(f.raw);
    const auto keep_grey = global_ptr(f.keep_grey);
    const auto keep_raw = global_ptr(f.keep_raw);
    // (the pointers of every plane the strip writes, read before the loads are issued: a scalar load further down would wait behind them)
    const auto R0 = global_ptr(f.R[0]);
    const auto A0 = global_ptr(f.A[0]);
    const auto B0 = global_ptr(f.B[0]);
    const auto C0 = global_ptr(f.C[0]);
    const auto I1 = global_ptr(f.I[1]), I2 = global_ptr(f.I[2]), I3 = global_ptr(f.I[3]);
    const auto Z1 = global_ptr(f.Z[1]), Z2 = global_ptr(f.Z[2]), Z3 = global_ptr(f.Z[3]);
    const auto C1 = global_ptr(f.C[1]), C2 = global_ptr(f.C[2]), C3 = global_ptr(f.C[3]);
    const bool want_c1 = (c_levels & 2) && C1, want_c2 = (c_levels & 4) && C2, want_c3 = (c_levels & 8) && C3;   // (a frame without the plane: skipped)
    const auto sel_count = global_ptr(f.sel_count);
    const int x = sx * kStripW + 2 * lane;
    const bool in_x = x < w0;                                   // widths are even here: the pair is inside or outside as a whole
    const int xl = in_x ? x : w0 - 2;                           // lanes past the right border load the last pair (and store nothing)
    auto depth_of = [&](unsigned d) { return d == 0 ? nanv : float(d) * scale; };

    // ---- every load of the strip, before anything is used ----
    constexpr int kFirst = TAPS ? -1 : 0, kRows = TAPS ? kStripH + 2 : kStripH;   // register row j = image row y0 + kFirst + j (clamped)
    unsigned g[kRows], d[kRows];
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      const int y = min(max(y0 + kFirst + j, 0), h0 - 1);       // (scalar)
      const size_t row = size_t(y) * w0;
      g[j] = *(Global<const uint16_t>)(grey + row + xl);
      d[j] = *(Global<const uint32_t>)(raw + row + xl);
    }
    unsigned ge[kStripH], de[kStripH];                          // the strip's edge columns, rows y0 .. y0 + 7: lane 0 left, the others right
    if (TAPS) {
      const int xe = lane == 0 ? max(sx * kStripW - 1, 0) : min(sx * kStripW + kStripW, w0 - 1);
      const bool edge_lane = lane == 0 || lane == 63;
#pragma unroll
      for (int r = 0; r < kStripH; ++r) {
        const size_t row = size_t(min(y0 + r, h0 - 1)) * w0;
        ge[r] = 0u; de[r] = 0u;
        if (edge_lane) {
          ge[r] = grey[row + xe];
          de[r] = raw[row + xe];
        }
      }
    }
    if (keep_grey && in_x) {                                    // the frame's own copy of its raw planes (for the other role, later)
#pragma unroll
      for (int r = 0; r < kStripH; ++r) {
        const int y = y0 + r;
        if (y < h0) {
          const size_t at = size_t(y) * w0 + x;
          *(Global<uint16_t>)(keep_grey + at) = uint16_t(g[r - kFirst]);
          *(Global<uint32_t>)(keep_raw + at) = d[r - kFirst];
        }
      }
    }
    float I[kRows][2], Z[kRows][2];
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      I[j][0] = float(g[j] & 0xffu); I[j][1] = float(g[j] >> 8);
      Z[j][0] = depth_of(d[j] & 0xffffu); Z[j][1] = depth_of(d[j] >> 16);
    }

    // ---- level 0 in the frame's role ----
    float eI[kStripH], eZ[kStripH];
    if (TAPS) {
#pragma unroll
      for (int r = 0; r < kStripH; ++r) { eI[r] = float(ge[r]); eZ[r] = depth_of(de[r]); }
    }
    const int count = ROLE >= 0 ? strip_role_planes<ROLE, TAPS>(I, Z, eI, eZ, x, y0, w0, h0, in_x, R0, A0, B0, C0, cur_flavor, ithr, dthr) : 0;
    if (ROLE == 1 && lane == 0 && count) atomicAdd((int*)sel_count, count);

    // ---- pyramid levels 1-3: 64 x 4, 32 x 2 and 16 x 1 pixels per strip; an out-of-image quad is never written ----
    if (levels < 2) continue;
    const int x1 = sx * (kStripW / 2) + lane;
    float m1[4], z1v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = 2 * k - kFirst, y1 = sy * 4 + k;
      m1[k] = (I[j][0] + I[j][1] + I[j + 1][0] + I[j + 1][1]) / 4.0f;   // same summation order as the reference
      z1v[k] = Z[j][0];                                                 // top-left sample, NaN holes kept (Q18)
      if (x1 < w1 && y1 < h1) {
        const size_t at = size_t(y1) * w1 + x1;
        I1[at] = m1[k];
        Z1[at] = z1v[k];
        if (want_c1) gstore(C1 + at, make_float2(m1[k], z1v[k]));
      }
    }
    if (levels < 3) continue;
    const int x2 = x1 >> 1;
    float m2[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int y2 = sy * 2 + m;
      m2[m] = (m1[2 * m] + row_shifted<kDppRowShl1>(m1[2 * m]) + m1[2 * m + 1] + row_shifted<kDppRowShl1>(m1[2 * m + 1])) / 4.0f;
      if ((lane & 1) == 0 && x2 < w2 && y2 < h2) {
        const size_t at = size_t(y2) * w2 + x2;
        I2[at] = m2[m];
        Z2[at] = z1v[2 * m];
        if (want_c2) gstore(C2 + at, make_float2(m2[m], z1v[2 * m]));
      }
    }
    if (levels < 4) continue;
    const int x3 = x1 >> 2, y3 = sy;
    const float m3 = (m2[0] + row_shifted<kDppRowShl2>(m2[0]) + m2[1] + row_shifted<kDppRowShl2>(m2[1])) / 4.0f;
    if ((lane & 3) == 0 && x3 < w3 && y3 < h3) {
      const size_t at = size_t(y3) * w3 + x3;
      I3[at] = m3;
      Z3[at] = z1v[0];
      if (want_c3) gstore(C3 + at, make_float2(m3, z1v[0]));
    }
  }
}

// The role planes of ONE pyramid level from the float planes I / Z (levels >= 1, and level 0 of frames created from float planes): the
// same strips, the same role code, 8-byte loads of pixel pairs instead of the raw planes' 2 + 4 bytes.  Replaces k_derive_current /
// k_derive_reference (one pixel per thread, ten scattered 4-byte loads each: 3 TB/s) for batches; a single camera frame's levels go
// through k_derive_levels (one launch for all of them).
template <int ROLE, bool TAPS>
__global__ __launch_bounds__(256) void k_derive_strips(const FrameBuildPtrs* __restrict__ tbl, int level, int w0, int h0, float ithr, float dthr,
                                                       int groups_x, int groups_y, int n_frames, int cur_flavor) {
#pragma clang fp contract(off)
  static_assert(ROLE == 0 || ROLE == 1, "current or reference");
  static_assert(ROLE != 1 || TAPS, "the selection predicate needs the differences");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_frame = groups_x * groups_y, total = per_frame * n_frames;
  for (int gi = blockIdx.x; gi < total; gi += gridDim.x) {
    const int frame = gi / per_frame, t = gi - frame * per_frame;
    const int sx = t % groups_x, sy = (t / groups_x) * kStripsPerGroup + wave;
    const int y0 = sy * kStripH;
    if (y0 >= h0) continue;
    const FrameBuildPtrs& f = tbl[frame];
    const auto Ip = global_ptr<const float>(f.I[level]);
    const auto Zp = global_ptr<const float>(f.Z[level]);
    const auto Rl = global_ptr(f.R[level]);
    const auto Al = global_ptr(f.A[level]);
    const auto Bl = global_ptr(f.B[level]);
    const auto Cl = global_ptr(f.C[level]);
    const auto sel_count = global_ptr(f.sel_count);
    const int x = sx * kStripW + 2 * lane;
    const bool in_x = x < w0;
    const int xl = in_x ? x : w0 - 2;
    constexpr int kFirst = TAPS ? -1 : 0, kRows = TAPS ? kStripH + 2 : kStripH;
    float I[kRows][2], Z[kRows][2];
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
      const int y = min(max(y0 + kFirst + j, 0), h0 - 1);
      const size_t row = size_t(y) * w0;
      const GlobalF32x2 iv = *(Global<const GlobalF32x2>)(Ip + row + xl);
      const GlobalF32x2 zv = *(Global<const GlobalF32x2>)(Zp + row + xl);
      I[j][0] = iv.x; I[j][1] = iv.y;
      Z[j][0] = zv.x; Z[j][1] = zv.y;
    }
    float eI[kStripH], eZ[kStripH];
    if (TAPS) {
      const int xe = lane == 0 ? max(sx * kStripW - 1, 0) : min(sx * kStripW + kStripW, w0 - 1);
      const bool edge_lane = lane == 0 || lane == 63;
#pragma unroll
      for (int r = 0; r < kStripH; ++r) {
        const size_t row = size_t(min(y0 + r, h0 - 1)) * w0;
        eI[r] = 0.0f; eZ[r] = 0.0f;
        if (edge_lane) {
          eI[r] = Ip[row + xe];
          eZ[r] = Zp[row + xe];
        }
      }
    }
    const int count = strip_role_planes<ROLE, TAPS>(I, Z, eI, eZ, x, y0, w0, h0, in_x, Rl, Al, Bl, Cl, cur_flavor, ithr, dthr);
    if (ROLE == 1 && lane == 0 && count) atomicAdd((int*)sel_count + level, count);
  }
}

bool derive_strips_supports(int w) { return w % 2 == 0 && w >= 4; }

// role 0: current (flavours cur_flavor), role 1: reference (the level's counters zeroed before by the caller's k_zero_counts)
void launch_derive_strips(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int role, float ithr, float dthr,
                          int max_workgroups, int cur_flavor) {
  const int gx = (w + kStripW - 1) / kStripW, gy = (h + kStripH * kStripsPerGroup - 1) / (kStripH * kStripsPerGroup);
  const long long total = (long long)gx * gy * n_frames;
  const dim3 grid(int(max_workgroups > 0 && total > max_workgroups ? max_workgroups : total)), block(256);
  if (role == 1) k_derive_strips<1, true><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, gx, gy, n_frames, cur_flavor);
  else if (cur_flavor & kCurAB) k_derive_strips<0, true><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, gx, gy, n_frames, cur_flavor);
  else k_derive_strips<0, false><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, gx, gy, n_frames, cur_flavor);
}

bool ingest_strips_supports(int w0, bool wide) { return wide && w0 % 4 == 0; }

void launch_ingest_strips(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, float scale, int w0, int h0, int levels, int role,
                          float ithr, float dthr, int max_workgroups, int cur_flavor, int c_levels) {
  const int gx = (w0 + kStripW - 1) / kStripW, gy = (h0 + kStripH * kStripsPerGroup - 1) / (kStripH * kStripsPerGroup);
  const long long total = (long long)gx * gy * n_frames;
  const dim3 grid(int(max_workgroups > 0 && total > max_workgroups ? max_workgroups : total)), block(256);
  const int lv = levels < 4 ? levels : 4;
#define DVO_LAUNCH_STRIPS(ROLE, TAPS) \
  k_ingest_strips<ROLE, TAPS><<<grid, block, 0, s>>>(tbl, scale, w0, h0, lv, ithr, dthr, gx, gy, n_frames, cur_flavor, c_levels)
  if (role == 1) DVO_LAUNCH_STRIPS(1, true);
  else if (role == 0 && (cur_flavor & kCurAB)) DVO_LAUNCH_STRIPS(0, true);
  else if (role == 0) DVO_LAUNCH_STRIPS(0, false);
  else DVO_LAUNCH_STRIPS(-1, false);
#undef DVO_LAUNCH_STRIPS
}

}  // namespace dvo_hip
