// pyramid_kernels.hip -- device-resident image data model, batched over frames.
//
// What the reference does on the host per frame and per level
//   ingest        dvo_benchmark/src/benchmark_slam.cpp:46-93, dvo_core/src/core/surface_pyramid.cpp:65-105
//   pyr-down      dvo_core/src/core/rgbd_image.cpp:38-55 (2x2 mean), :127-139 (depth subsample), :156-172
//   derivatives   dvo_core/src/core/rgbd_image.cpp:419-489, rgbd_image_sse.cpp:241-284
//   accel struct  dvo_core/src/core/rgbd_image.cpp:534-543   (8 interleaved float channels, 32 B/pixel)
//   selection     dvo_core/src/core/point_selection.cpp:89-152, point_selection.h:49-67
// is done here on the GPU, one launch per pyramid level for a whole batch of frames (blockIdx.z = frame),
// so a frame upload is two raw planes and every derived plane stays in HBM.  Layout (all float32):
//   I, Z          planar, 4 B/pixel each (pyr-down and derivative source; built when the frame is ingested)
//   A             float4 {I, Z, Idx, Idy}  current-side sampling plane, one 16-B tap per bilinear corner
//   B             float2 {Zdx, Zdy}        current-side sampling plane,  8-B tap
//   R             float4 {Zsel, I, Idx, Idy} reference-side stream; Zsel = NaN where the selection
//                 predicate rejects the pixel, so the reduce kernel needs no separate mask or list
// These are bandwidth-trivial elementwise kernels; they are written for coalescing only.
#include "global_ptr.h"
#include "launch.h"

namespace dvo_hip {

// Work distribution of the frame-build kernels: a linear index over (tile, frame) walked with a grid stride.  Launched with
// one workgroup per tile this is the plain one-tile-per-workgroup kernel; launched with FEWER workgroups (option
// "build_workgroups") the same kernel becomes a background job that leaves most wave slots of every CU free, so the
// short dependent kernels of an alignment running concurrently on the other stream are dispatched at once instead of queueing
// behind tens of thousands of elementwise workgroups.
template <typename Body>
__device__ __forceinline__ void for_each_tile(int tiles_x, int tiles_y, int n_frames, Body body) {
  const int per_frame = tiles_x * tiles_y, total = per_frame * n_frames;
  for (int i = blockIdx.x; i < total; i += gridDim.x) {
    const int frame = i / per_frame, t = i - frame * per_frame;
    body(t % tiles_x, t / tiles_x, frame);
  }
}

// Level 0 straight from the raw planes, in the role the frame is about to play, plus pyramid levels 1-3: one pass over
// 3 B per pixel instead of  raw -> float I, Z (8 B written)  followed by  I, Z (+ halo) -> role planes (8 B read again).
// A 32 x 8 workgroup owns a 64 x 16 tile; the tile and its one-pixel border are converted into LDS with 4- and 8-byte
// loads (WIDE: rows and plane addresses are 4-pixel aligned), border coordinates clamped like the reference's derivative
// code (rgbd_image.cpp:419-489), so pixels past the image edge hold the edge value.  Then every thread derives its 2 x 2
// quad from LDS and folds pyramid levels 1-3 (2 x 2 means through LDS, same summation order as k_pyr_down).  Float I / Z planes
// of level 0 are never written:
// a frame built this way keeps either its current-role planes (which hold everything) or a 3-B copy of the raw planes
// (keep_grey / keep_raw) from which the other role can be derived later by the same kernel.
// ROLE: -1 = none (copy + pyramid only), 0 = current (A, B), 1 = reference (R + selection count, counter zeroed before).
// Arithmetic = the reference's ingest (surface_pyramid.cpp:65-105), k_pyr_down and derive_at: bit-identical planes.
constexpr int kB0W = 64, kB0H = 16, kB0Stride = 68;

template <int ROLE, bool WIDE>
__global__ __launch_bounds__(256) void k_build_from_raw(const FrameBuildPtrs* __restrict__ tbl, float scale, int w0, int h0, int levels,
                                                        float ithr, float dthr, int tiles_x, int tiles_y, int n_frames, int cur_flavor) {
#pragma clang fp contract(off)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8 threads, one 2 x 2 quad each
  const int w1 = w0 >> 1, h1 = h0 >> 1;
  const int w2 = w1 >> 1, h2 = h1 >> 1, w3 = w2 >> 1, h3 = h2 >> 1;
  __shared__ float sI[kB0H + 2][kB0Stride];
  __shared__ float sZ[kB0H + 2][kB0Stride];
  __shared__ float s1[8][32];
  __shared__ float s2[4][16];
  __shared__ int wave_counts[4];
  const float nanv = __builtin_nanf("");
  for_each_tile(tiles_x, tiles_y, n_frames, [&](int bx, int by, int frame) {
    const FrameBuildPtrs& f = tbl[frame];
    // (global_ptr.h: every plane pointer read once, as a pointer into the global address space)
    const auto grey = global_ptr(f.grey);
    const auto raw = global_ptr(f.raw);
    const auto keep_grey = global_ptr(f.keep_grey);
    const auto keep_raw = global_ptr(f.keep_raw);
    const auto A0 = global_ptr(f.A[0]);
    const auto B0 = global_ptr(f.B[0]);
    const auto C0 = global_ptr(f.C[0]);
    const auto R0 = global_ptr(f.R[0]);
    const auto I1 = global_ptr(f.I[1]), I2 = global_ptr(f.I[2]), I3 = global_ptr(f.I[3]);
    const auto Z1 = global_ptr(f.Z[1]), Z2 = global_ptr(f.Z[2]), Z3 = global_ptr(f.Z[3]);
    const int x0 = bx * kB0W, y0 = by * kB0H;
    auto depth_of = [&](uint16_t d) { return d == 0 ? nanv : float(d) * scale; };
    // ---- tile + border into LDS (column c of the slab = image column x0 - 1 + c, row r = image row y0 - 1 + r, clamped) ----
    if (WIDE) {
      for (int i = threadIdx.x; i < (kB0H + 2) * (kB0W / 4); i += 256) {
        const int r = i / (kB0W / 4), qd = i - r * (kB0W / 4);
        const int yy = y0 - 1 + r, y = min(max(yy, 0), h0 - 1);
        const int x = x0 + 4 * qd;
        const size_t at = size_t(y) * w0 + x;
        if (x + 3 < w0) {
          const unsigned gbits = *(Global<const unsigned>)(grey + at);
          const GlobalU32x2 dbits = *(Global<const GlobalU32x2>)(raw + at);
          const uchar4 gq = make_uchar4(gbits & 0xffu, gbits >> 8 & 0xffu, gbits >> 16 & 0xffu, gbits >> 24);
          const ushort4 dq = make_ushort4(dbits.x & 0xffffu, dbits.x >> 16, dbits.y & 0xffffu, dbits.y >> 16);
          float* di = &sI[r][1 + 4 * qd];
          float* dz = &sZ[r][1 + 4 * qd];
          di[0] = float(gq.x); di[1] = float(gq.y); di[2] = float(gq.z); di[3] = float(gq.w);
          dz[0] = depth_of(dq.x); dz[1] = depth_of(dq.y); dz[2] = depth_of(dq.z); dz[3] = depth_of(dq.w);
          if (keep_grey && yy == y && r >= 1 && r <= kB0H) {
            *(Global<unsigned>)(keep_grey + at) = gbits;
            *(Global<GlobalU32x2>)(keep_raw + at) = dbits;
          }
        } else {
          for (int k = 0; k < 4; ++k) {
            const int xc = min(x + k, w0 - 1);
            const uint8_t gv = grey[size_t(y) * w0 + xc];
            const uint16_t dv = raw[size_t(y) * w0 + xc];
            sI[r][1 + 4 * qd + k] = float(gv);
            sZ[r][1 + 4 * qd + k] = depth_of(dv);
            if (keep_grey && yy == y && r >= 1 && r <= kB0H && x + k < w0) {
              keep_grey[size_t(y) * w0 + xc] = gv;
              keep_raw[size_t(y) * w0 + xc] = dv;
            }
          }
        }
      }
      if (threadIdx.x < 2 * (kB0H + 2)) {                       // the two border columns
        const int r = threadIdx.x >> 1, side = threadIdx.x & 1;
        const int y = min(max(y0 - 1 + r, 0), h0 - 1);
        const int x = side ? min(x0 + kB0W, w0 - 1) : max(x0 - 1, 0);
        sI[r][side ? kB0W + 1 : 0] = float(grey[size_t(y) * w0 + x]);
        sZ[r][side ? kB0W + 1 : 0] = depth_of(raw[size_t(y) * w0 + x]);
      }
    } else {
      for (int i = threadIdx.x; i < (kB0H + 2) * (kB0W + 2); i += 256) {
        const int r = i / (kB0W + 2), c = i - r * (kB0W + 2);
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const int y = min(max(yy, 0), h0 - 1), x = min(max(xx, 0), w0 - 1);
        const uint8_t gv = grey[size_t(y) * w0 + x];
        const uint16_t dv = raw[size_t(y) * w0 + x];
        sI[r][c] = float(gv);
        sZ[r][c] = depth_of(dv);
        if (keep_grey && yy == y && xx == x && r >= 1 && r <= kB0H && c >= 1 && c <= kB0W) {
          keep_grey[size_t(y) * w0 + x] = gv;
          keep_raw[size_t(y) * w0 + x] = dv;
        }
      }
    }
    __syncthreads();
    // ---- level 0 in the frame's role ----
    // wavefront w writes rows w, w + 4, ... of the tile, lane = column: every store instruction covers 64 consecutive pixels
    // (1 KiB of A or R, 512 B of B) instead of every other pixel of two rows
    const int x1 = bx * 32 + tx, y1 = by * 8 + ty;              // level-1 pixel = level-0 quad (pyramid part below)
    int count = 0;
    if (ROLE >= 0) {
      const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < kB0H / 4; ++k) {
        const int x = x0 + lx, y = y0 + ly + 4 * k;
        const int r = ly + 4 * k + 1, c = lx + 1;
        bool ok = false;
        if (x < w0 && y < h0) {
          const float i0 = sI[r][c], z0 = sZ[r][c];
          const float idx = (sI[r][c + 1] - sI[r][c - 1]) * 0.5f, idy = (sI[r + 1][c] - sI[r - 1][c]) * 0.5f;
          const float zdx = (sZ[r][c + 1] - sZ[r][c - 1]) * 0.5f, zdy = (sZ[r + 1][c] - sZ[r - 1][c]) * 0.5f;
          const size_t at = size_t(y) * w0 + x;
          if (ROLE == 0) {
            // the current role comes in two flavours (device_types.h kCurAB / kCurC): the gathered taps of the gathering sweep and
            // the resident kernel, and / or the 8-byte {I, Z} plane the window sweep stages in LDS (align_window.hip)
            if (cur_flavor & kCurAB) {
              gstore(A0 + at, make_float4(i0, z0, idx, idy));
              gstore(B0 + at, make_float2(zdx, zdy));
            }
            if (cur_flavor & kCurC) gstore(C0 + at, make_float2(i0, z0));
          } else {
            ok = z0 == z0 && zdx == zdx && zdy == zdy && (fabsf(idx) > ithr || fabsf(idy) > ithr || fabsf(zdx) > dthr || fabsf(zdy) > dthr);
            gstore(R0 + at, make_float2(ok ? z0 : nanv, i0));
          }
        }
        if (ROLE == 1) count += __popcll(__ballot(ok));         // wave-uniform
      }
    }
    if (ROLE == 1 && (threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = count;
    // ---- pyramid levels 1-3 (16 x 4 and 8 x 2 pixels of levels 2 and 3 per tile; an out-of-image quad is never written) ----
    const int r = 2 * ty + 1, c = 2 * tx + 1;
    const float i1 = (sI[r][c] + sI[r][c + 1] + sI[r + 1][c] + sI[r + 1][c + 1]) / 4.0f;   // same summation order as the reference
    const float z00 = sZ[r][c];
    if (levels >= 2 && x1 < w1 && y1 < h1) {
      I1[size_t(y1) * w1 + x1] = i1;
      Z1[size_t(y1) * w1 + x1] = z00;                       // top-left sample, NaN holes kept (Q18)
    }
    s1[ty][tx] = i1;
    __syncthreads();                                             // s1 and wave_counts complete; sI / sZ free for the next tile
    if (ROLE == 1 && threadIdx.x == 0) {
      const int total = (wave_counts[0] + wave_counts[1]) + (wave_counts[2] + wave_counts[3]);
      if (total) atomicAdd(f.sel_count, total);
    }
    const int x2 = x1 >> 1, y2 = y1 >> 1;
    if ((tx & 1) == 0 && (ty & 1) == 0) {
      const float i2 = (s1[ty][tx] + s1[ty][tx + 1] + s1[ty + 1][tx] + s1[ty + 1][tx + 1]) / 4.0f;
      if (levels >= 3 && x2 < w2 && y2 < h2) {
        I2[size_t(y2) * w2 + x2] = i2;
        Z2[size_t(y2) * w2 + x2] = z00;
      }
      s2[ty >> 1][tx >> 1] = i2;
    }
    __syncthreads();
    if (levels >= 4 && (tx & 3) == 0 && (ty & 3) == 0) {
      const int x3 = x1 >> 2, y3 = y1 >> 2, cx = tx >> 1, cy = ty >> 1;
      if (x3 < w3 && y3 < h3) {
        I3[size_t(y3) * w3 + x3] = (s2[cy][cx] + s2[cy][cx + 1] + s2[cy + 1][cx] + s2[cy + 1][cx + 1]) / 4.0f;
        Z3[size_t(y3) * w3 + x3] = z00;
      }
    }
    __syncthreads();                                             // s1, s2, wave_counts free for the next tile of this workgroup
  });
}

__global__ void k_pyr_down(const FrameBuildPtrs* __restrict__ tbl, int level, int w, int h) {
#pragma clang fp contract(off)
  const FrameBuildPtrs& f = tbl[blockIdx.z];
  const auto I = global_ptr<const float>(f.I[level - 1]);       // (global_ptr.h: read once, global address space)
  const auto Z = global_ptr<const float>(f.Z[level - 1]);
  const auto outI = global_ptr(f.I[level]);
  const auto outZ = global_ptr(f.Z[level]);
  const int ow = w >> 1, oh = h >> 1;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= ow || y >= oh) return;
  const auto r0 = I + size_t(2 * y) * w + 2 * x;
  const auto r1 = r0 + w;
  outI[size_t(y) * ow + x] = (r0[0] + r0[1] + r1[0] + r1[1]) / 4.0f;   // same summation order as the reference
  outZ[size_t(y) * ow + x] = Z[size_t(2 * y) * w + 2 * x];             // top-left sample, NaN holes kept (Q18)
}

// Central differences with clamped borders (rgbd_image.cpp:419-489).  The derived planes are built per ROLE, like the
// reference builds them lazily: a frame that is only ever a current frame gets A + B (buildAccelerationStructure,
// rgbd_image.cpp:534-543), a frame that is only ever a reference gets R + the selection count (PointSelection::select,
// point_selection.cpp:89-152).
struct Derivs {
  float i0, z0, idx, idy, zdx, zdy;
};

__device__ __forceinline__ Derivs derive_at(Global<const float> I, Global<const float> Z, int w, int h, int x, int y) {
#pragma clang fp contract(off)
  const int xp = max(x - 1, 0), xn = min(x + 1, w - 1);
  const int yp = max(y - 1, 0), yn = min(y + 1, h - 1);
  const size_t row = size_t(y) * w;
  Derivs d;
  d.i0 = I[row + x];
  d.z0 = Z[row + x];
  d.idx = (I[row + xn] - I[row + xp]) * 0.5f;
  d.idy = (I[size_t(yn) * w + x] - I[size_t(yp) * w + x]) * 0.5f;
  d.zdx = (Z[row + xn] - Z[row + xp]) * 0.5f;
  d.zdy = (Z[size_t(yn) * w + x] - Z[size_t(yp) * w + x]) * 0.5f;
  return d;
}

// current-frame role: the two sampling planes
__global__ void k_derive_current(const FrameBuildPtrs* __restrict__ tbl, int level, int w, int h, int tiles_x, int tiles_y, int n_frames, int cur_flavor) {
  for_each_tile(tiles_x, tiles_y, n_frames, [&](int bx, int by, int frame) {
    const FrameBuildPtrs& f = tbl[frame];
    const auto I = global_ptr<const float>(f.I[level]);         // (global_ptr.h: read once, global address space)
    const auto Z = global_ptr<const float>(f.Z[level]);
    const auto A = global_ptr(f.A[level]);
    const auto B = global_ptr(f.B[level]);
    const auto C = global_ptr(f.C[level]);
    const int x = bx * 64 + threadIdx.x;
    const int y = by * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    if (cur_flavor == kCurC) {                                  // (uniform) only the {I, Z} pair: no neighbours to read
      gstore(C + size_t(y) * w + x, make_float2(I[size_t(y) * w + x], Z[size_t(y) * w + x]));
      return;
    }
    const Derivs d = derive_at(I, Z, w, h, x, y);
    gstore(A + size_t(y) * w + x, make_float4(d.i0, d.z0, d.idx, d.idy));
    gstore(B + size_t(y) * w + x, make_float2(d.zdx, d.zdy));
    if (cur_flavor & kCurC) gstore(C + size_t(y) * w + x, make_float2(d.i0, d.z0));
  });
}

// A frame that holds only one flavour of the current role at a level gets the other: the taps A + B from the {I, Z} plane C (the
// clamped central differences of derive_at, same operation order: bit-identical planes), or C from A.  MODE 2: the reference role
// (R + selection count, counter zeroed before) from C -- PointSelection over a frame that has been a current frame of the window
// sweep so far.
template <int MODE>
__global__ void k_from_current_plane(const FrameBuildPtrs* __restrict__ tbl, int level, int w, int h, float ithr, float dthr,
                                     int tiles_x, int tiles_y, int n_frames) {
#pragma clang fp contract(off)
  __shared__ int wave_counts[4];
  for_each_tile(tiles_x, tiles_y, n_frames, [&](int bx, int by, int frame) {
    const FrameBuildPtrs& f = tbl[frame];
    const auto A = global_ptr(f.A[level]);                      // (global_ptr.h: read once, global address space)
    const auto B = global_ptr(f.B[level]);
    const auto R = global_ptr(f.R[level]);
    const auto Cw = global_ptr(f.C[level]);
    const int x = bx * 64 + threadIdx.x;
    const int y = by * 4 + threadIdx.y;
    bool ok = false;
    if (x < w && y < h) {
      const size_t at = size_t(y) * w + x;
      if (MODE == 1) {
        const float4 a = gload((Global<const float4>)(A + at));
        gstore(Cw + at, make_float2(a.x, a.y));
      } else {
        const auto C = global_ptr<const float2>(f.C[level]);
        const int xp = max(x - 1, 0), xn = min(x + 1, w - 1), yp = max(y - 1, 0), yn = min(y + 1, h - 1);
        const float2 c = gload(C + at), l = gload(C + size_t(y) * w + xp), r = gload(C + size_t(y) * w + xn), u = gload(C + size_t(yp) * w + x), d = gload(C + size_t(yn) * w + x);
        const float idx = (r.x - l.x) * 0.5f, idy = (d.x - u.x) * 0.5f, zdx = (r.y - l.y) * 0.5f, zdy = (d.y - u.y) * 0.5f;
        if (MODE == 0) {
          gstore(A + at, make_float4(c.x, c.y, idx, idy));
          gstore(B + at, make_float2(zdx, zdy));
        } else {
          ok = c.y == c.y && zdx == zdx && zdy == zdy && (fabsf(idx) > ithr || fabsf(idy) > ithr || fabsf(zdx) > dthr || fabsf(zdy) > dthr);
          gstore(R + at, make_float2(ok ? c.y : __builtin_nanf(""), c.x));
        }
      }
    }
    if (MODE == 2) {
      const int count = __popcll(__ballot(ok));
      if (threadIdx.x == 0) wave_counts[threadIdx.y] = count;
      __syncthreads();
      if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int total = (wave_counts[0] + wave_counts[1]) + (wave_counts[2] + wave_counts[3]);
        if (total) atomicAdd(f.sel_count + level, total);
      }
      __syncthreads();
    }
  });
}

// reference role: the streamed plane with the selection predicate folded into Z; counts the selected pixels.
// The counter of (frame, level) must have been zeroed on the stream before.  A workgroup sweeps a 64 x 16 pixel
// tile and issues ONE atomic for it (one atomic per wavefront-row serialised the whole kernel on 128 addresses).
__global__ void k_derive_reference(const FrameBuildPtrs* __restrict__ tbl, int level, int w, int h, float ithr, float dthr,
                                   int tiles_x, int tiles_y, int n_frames) {
  __shared__ int wave_counts[4];
  for_each_tile(tiles_x, tiles_y, n_frames, [&](int bx, int by, int frame) {
    const FrameBuildPtrs& f = tbl[frame];
    const auto I = global_ptr<const float>(f.I[level]);         // (global_ptr.h: read once, global address space)
    const auto Z = global_ptr<const float>(f.Z[level]);
    const auto R = global_ptr(f.R[level]);
    const int x = bx * 64 + threadIdx.x;
    int count = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = (by * 4 + r) * 4 + threadIdx.y;
      bool ok = false;
      if (x < w && y < h) {
        const Derivs d = derive_at(I, Z, w, h, x, y);
        ok = d.z0 == d.z0 && d.zdx == d.zdx && d.zdy == d.zdy &&
             (fabsf(d.idx) > ithr || fabsf(d.idy) > ithr || fabsf(d.zdx) > dthr || fabsf(d.zdy) > dthr);
        gstore(R + size_t(y) * w + x, make_float2(ok ? d.z0 : __builtin_nanf(""), d.i0));
      }
      count += __popcll(__ballot(ok));      // wave-uniform
    }
    if (threadIdx.x == 0) wave_counts[threadIdx.y] = count;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
      const int total = (wave_counts[0] + wave_counts[1]) + (wave_counts[2] + wave_counts[3]);
      if (total) atomicAdd(f.sel_count + level, total);
    }
    __syncthreads();                        // wave_counts is rewritten by the next tile
  });
}

// The role planes of SEVERAL pyramid levels of a set of frames in one launch (a single camera frame: levels 1-3 are 6000 pixels
// together; one launch per level, each with its table upload and counter reset, is 9-12 launches of 4 microseconds of work and 8 of
// launch latency each -- the largest item of a tracking front end's frame after the match itself).  Same arithmetic as
// k_derive_current / k_derive_reference (derive_at): bit-identical planes.  A workgroup takes a 64 x 16 tile of one level of one frame.
// ROLE 0: current (flavour per level: LevelSpan::flavor), ROLE 1: reference (counters of the levels zeroed before).
template <int ROLE>
__global__ __launch_bounds__(256) void k_derive_levels(const FrameBuildPtrs* __restrict__ tbl, const LevelSpan span, int n_frames, float ithr, float dthr) {
  __shared__ int wave_counts[4];
  const int per_frame = span.tile0[span.l1 + 1], total = per_frame * n_frames;
  for (int gi = blockIdx.x; gi < total; gi += gridDim.x) {
    const int frame = gi / per_frame, t = gi - frame * per_frame;
    int level = span.l0;
    while (level < span.l1 && t >= span.tile0[level + 1]) ++level;
    const int w = span.w[level], h = span.h[level], tiles_x = (w + 63) / 64;
    const int tile = t - span.tile0[level], bx = tile % tiles_x, by = tile / tiles_x;
    const FrameBuildPtrs& f = tbl[frame];
    const auto I = global_ptr<const float>(f.I[level]);
    const auto Z = global_ptr<const float>(f.Z[level]);
    const auto A = global_ptr(f.A[level]);
    const auto B = global_ptr(f.B[level]);
    const auto C = global_ptr(f.C[level]);
    const auto R = global_ptr(f.R[level]);
    const int flavor = span.flavor[level];
    const int x = bx * 64 + threadIdx.x;
    int count = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = (by * 4 + r) * 4 + threadIdx.y;
      bool ok = false;
      if (x < w && y < h) {
        const size_t at = size_t(y) * w + x;
        if (ROLE == 0 && flavor == kCurC) {                     // (uniform) only the {I, Z} pair: no neighbours to read
          gstore(C + at, make_float2(I[at], Z[at]));
        } else {
          const Derivs d = derive_at(I, Z, w, h, x, y);
          if (ROLE == 0) {
            gstore(A + at, make_float4(d.i0, d.z0, d.idx, d.idy));
            gstore(B + at, make_float2(d.zdx, d.zdy));
            if (flavor & kCurC) gstore(C + at, make_float2(d.i0, d.z0));
          } else {
            ok = d.z0 == d.z0 && d.zdx == d.zdx && d.zdy == d.zdy &&
                 (fabsf(d.idx) > ithr || fabsf(d.idy) > ithr || fabsf(d.zdx) > dthr || fabsf(d.zdy) > dthr);
            gstore(R + at, make_float2(ok ? d.z0 : __builtin_nanf(""), d.i0));
          }
        }
      }
      if (ROLE == 1) count += __popcll(__ballot(ok));           // wave-uniform
    }
    if (ROLE == 1) {
      if (threadIdx.x == 0) wave_counts[threadIdx.y] = count;
      __syncthreads();
      if (threadIdx.x == 0 && threadIdx.y == 0) {
        const int sum = (wave_counts[0] + wave_counts[1]) + (wave_counts[2] + wave_counts[3]);
        if (sum) atomicAdd(f.sel_count + level, sum);
      }
      __syncthreads();                                          // wave_counts is rewritten by the next tile
    }
  }
}

__global__ void k_zero_counts_levels(const FrameBuildPtrs* __restrict__ tbl, int n_frames, int l0, int l1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_frames)
    for (int l = l0; l <= l1; ++l) tbl[i].sel_count[l] = 0;
}

__global__ void k_zero_counts(const FrameBuildPtrs* __restrict__ tbl, int n_frames, int level) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_frames) tbl[i].sel_count[level] = 0;
}

// re-selection with other thresholds (PointSelection with a different predicate), one frame
__global__ void k_select_pack(const float4* __restrict__ A, const float2* __restrict__ B, int n, float ithr, float dthr,
                              float2* __restrict__ R, int* __restrict__ count, uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < n) {
    const float4 a = A[i];
    const float2 b = B[i];
    ok = a.y == a.y && b.x == b.x && b.y == b.y &&
         (fabsf(a.z) > ithr || fabsf(a.w) > ithr || fabsf(b.x) > dthr || fabsf(b.y) > dthr);
    R[i] = make_float2(ok ? a.y : __builtin_nanf(""), a.x);
    if (mask) mask[i] = ok ? 1 : 0;
  }
  const unsigned long long ballot = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && ballot) atomicAdd(count, __popcll(ballot));
}

__global__ void k_unpack_plane(const float4* __restrict__ A, const float2* __restrict__ B, int n, int plane, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v;
  switch (plane) {
    case 0: v = A[i].x; break;
    case 1: v = A[i].y; break;
    case 2: v = A[i].z; break;
    case 3: v = A[i].w; break;
    case 4: v = B[i].x; break;
    default: v = B[i].y; break;
  }
  out[i] = v;
}

// grid of a frame-build kernel: one workgroup per tile unless a cap is given (background build)
static int capped_grid(int tiles_x, int tiles_y, int n_frames, int max_workgroups) {
  const long long total = (long long)tiles_x * tiles_y * n_frames;
  return int(max_workgroups > 0 && total > max_workgroups ? max_workgroups : total);
}

void launch_build_from_raw(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, float scale, int w0, int h0, int levels, int role, bool wide,
                           float ithr, float dthr, int max_workgroups, int cur_flavor, int c_levels) {
  if (ingest_strips_supports(w0, wide)) {
    if (role == 1) k_zero_counts<<<dim3((n_frames + 63) / 64), dim3(64), 0, s>>>(tbl, n_frames, 0);
    launch_ingest_strips(s, tbl, n_frames, scale, w0, h0, levels, role, ithr, dthr, max_workgroups, cur_flavor, c_levels);
    return;
  }
  const int tx = (w0 + kB0W - 1) / kB0W, ty = (h0 + kB0H - 1) / kB0H;
  const dim3 grid(capped_grid(tx, ty, n_frames, max_workgroups)), block(256);
  const int lv = levels < 4 ? levels : 4;
  if (role == 1) k_zero_counts<<<dim3((n_frames + 63) / 64), dim3(64), 0, s>>>(tbl, n_frames, 0);
#define DVO_LAUNCH_B0(ROLE, WIDE) k_build_from_raw<ROLE, WIDE><<<grid, block, 0, s>>>(tbl, scale, w0, h0, lv, ithr, dthr, tx, ty, n_frames, cur_flavor)
  if (wide) {
    if (role == 0) DVO_LAUNCH_B0(0, true); else if (role == 1) DVO_LAUNCH_B0(1, true); else DVO_LAUNCH_B0(-1, true);
  } else {
    if (role == 0) DVO_LAUNCH_B0(0, false); else if (role == 1) DVO_LAUNCH_B0(1, false); else DVO_LAUNCH_B0(-1, false);
  }
#undef DVO_LAUNCH_B0
}

void launch_pyr_down(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h) {
  const int ow = w / 2, oh = h / 2;
  k_pyr_down<<<dim3((ow + 63) / 64, (oh + 3) / 4, n_frames), dim3(64, 4), 0, s>>>(tbl, level, w, h);
}

// (even widths: the strip form, ingest_strips.hip)
void launch_derive_current(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int max_workgroups, int cur_flavor) {
  if (derive_strips_supports(w)) {
    launch_derive_strips(s, tbl, n_frames, level, w, h, 0, 0.0f, 0.0f, max_workgroups, cur_flavor);
    return;
  }
  const int tx = (w + 63) / 64, ty = (h + 3) / 4;
  k_derive_current<<<dim3(capped_grid(tx, ty, n_frames, max_workgroups)), dim3(64, 4), 0, s>>>(tbl, level, w, h, tx, ty, n_frames, cur_flavor);
}

void launch_from_current_plane(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int mode, float ithr, float dthr,
                               int max_workgroups) {
  const int tx = (w + 63) / 64, ty = (h + 3) / 4;
  const dim3 grid(capped_grid(tx, ty, n_frames, max_workgroups)), block(64, 4);
  if (mode == 2) k_zero_counts<<<dim3((n_frames + 63) / 64), dim3(64), 0, s>>>(tbl, n_frames, level);
  if (mode == 0) k_from_current_plane<0><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, tx, ty, n_frames);
  else if (mode == 1) k_from_current_plane<1><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, tx, ty, n_frames);
  else k_from_current_plane<2><<<grid, block, 0, s>>>(tbl, level, w, h, ithr, dthr, tx, ty, n_frames);
}

void launch_derive_levels(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, const LevelSpan& span, int role, float ithr, float dthr,
                          int max_workgroups) {
  const int total = span.tile0[span.l1 + 1] * n_frames;
  const dim3 grid(max_workgroups > 0 && total > max_workgroups ? max_workgroups : total), block(64, 4);
  if (role == 1) {
    k_zero_counts_levels<<<dim3((n_frames + 63) / 64), dim3(64), 0, s>>>(tbl, n_frames, span.l0, span.l1);
    k_derive_levels<1><<<grid, block, 0, s>>>(tbl, span, n_frames, ithr, dthr);
  } else {
    k_derive_levels<0><<<grid, block, 0, s>>>(tbl, span, n_frames, ithr, dthr);
  }
}

void launch_derive_reference(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, float ithr, float dthr,
                             int max_workgroups) {
  k_zero_counts<<<dim3((n_frames + 63) / 64), dim3(64), 0, s>>>(tbl, n_frames, level);
  if (derive_strips_supports(w)) {
    launch_derive_strips(s, tbl, n_frames, level, w, h, 1, ithr, dthr, max_workgroups, 0);
    return;
  }
  const int tx = (w + 63) / 64, ty = (h + 15) / 16;
  k_derive_reference<<<dim3(capped_grid(tx, ty, n_frames, max_workgroups)), dim3(64, 4), 0, s>>>(tbl, level, w, h, ithr, dthr, tx, ty, n_frames);
}

void launch_select_pack(hipStream_t s, const float4* A, const float2* B, int n, float ithr, float dthr, float2* R, int* count, uint8_t* mask) {
  k_select_pack<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(A, B, n, ithr, dthr, R, count, mask);
}

void launch_unpack_plane(hipStream_t s, const float4* A, const float2* B, int n, int plane, float* out) {
  k_unpack_plane<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(A, B, n, plane, out);
}

}  // namespace dvo_hip
