// align_small.hip -- the sweep of a SMALL pyramid level (80 x 60 here: what the contracted window sweep does not take) with the WHOLE
// current level staged in LDS.
//
// Until round 6 these levels ran the gathering sweep (align_mfma.hip): every lane fetches its eight taps from memory -- two dependent
// round trips per pixel row -- and the launch sat at 0.3-0.5 of the kernel's roofline, the worst of the step's kernels (1024 pairs:
// fourteen launches of 60 us, 0.86 of a step's 11.2 ms; profiles/r06_kernel_rooflines.md).  A level this small is 38 KB of {I, Z}:
// a workgroup copies ALL of it into LDS once (with a one-cell clamped border: the reference's border rule for its central differences,
// dvo_core/src/core/rgbd_image.cpp:419-489), and every reference pixel it sweeps finds its twelve cells there -- no window to place,
// no bounding box, no fall-back path, one memory round trip per row (the reference pixel and its four neighbours).  The rest is the
// contracted window sweep's own code (fast_sweep.h): projection, blend, residual pair, weight, Jacobian at the untransformed point,
// f16 Gram on the matrix pipe, epilogue -- dense_tracking_impl.cpp:148-281, dense_tracking.cpp:448-476 as there.
//
// The level is walked as ONE row of w * h pixels in 64-pixel segments (LevelGeom::linear: an 80-pixel-wide level walked in 64-column
// tiles would idle 37 % of its lanes); a workgroup takes 4 * rows_per_wave consecutive segments -- 2 to 8 workgroups per pair, so that
// a batch fills the chip's three workgroups per compute unit (53 KB of LDS each) in whole rounds (BatchPolicy::small_level_tiles).
// Residual pairs by pixel (NaN where there is no constraint), partial rows in the canonical layout: the log-likelihood pass and the
// solver step read them like the gathering sweep's.
#include "fast_sweep.h"
#include "launch.h"

namespace dvo_hip {

constexpr int kSmallCells = 5376;                            // (w + 2) x (h + 2) cells of 8 B at most: 43 KB (80 x 60: 82 x 62 = 5084)
constexpr int kSmallLoads = kSmallCells / kBlock;            // cells a thread copies (21)

__global__ __launch_bounds__(kBlock, 3) void k_sweep_small(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd, int rows_per_wave, int* __restrict__ f16_range_flag) {
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;
  const PairState& st = states[pair];
  if (!st.active || st.level != g.level) return;            // (not on this level: finished it, and maybe begun the next)
  const PairPtrs pp = pairs[pair];
  __shared__ __attribute__((aligned(16))) float slab[4][kSlabFloatsF16];
  __shared__ __attribute__((aligned(16))) float2 img[kSmallCells];
  __shared__ int counts[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_px = g.w * g.h, plane_bytes = n_px * 8;
  const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t curC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.curC), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t resid = __builtin_amdgcn_make_buffer_rsrc(scratch + size_t(pair) * size_t(n_px), 0, plane_bytes, 0x00020000);

  // ---- the level into LDS: cell (cx, cy) = pixel (cx - 1, cy - 1), coordinates clamped to the image ------------------------------
  const int pitch = g.w + 2, cells = pitch * (g.h + 2);
  {
    const float inv_pitch = 1.0f / float(pitch);
    f32x2 v[kSmallLoads];
#pragma unroll
    for (int j = 0; j < kSmallLoads; ++j) {                  // every load before the first store: one round trip
      const int c = min(int(threadIdx.x) + j * kBlock, cells - 1);
      int cy = int(float(c) * inv_pitch);                    // (c < 2^24: one float multiply lands within one row of the quotient)
      int cx = c - cy * pitch;
      if (cx < 0) { cx += pitch; cy -= 1; }
      if (cx >= pitch) { cx -= pitch; cy += 1; }
      const int x = min(max(cx - 1, 0), g.w - 1), y = min(max(cy - 1, 0), g.h - 1);
      v[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(curC, (y * g.w + x) * 8, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < kSmallLoads; ++j) {
      const int c = int(threadIdx.x) + j * kBlock;
      if (c < cells) img[c] = make_float2(v[j].x, v[j].y);
    }
  }

  // ---- the segments of this workgroup ------------------------------------------------------------------------------------------------
  float KT[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
  const FastWeights wt(st);
  const FastRcpSource no_table = {__builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, 0x00020000), nullptr, 0};
  const unsigned gram_entries = kGramEntryTable.e[min(int(threadIdx.x), kNumAcc - 1)];
  const unsigned w2_bits = __builtin_bit_cast(unsigned, float(g.w - 2)), h2_bits = __builtin_bit_cast(unsigned, float(g.h - 2));
  const float inv_w = 1.0f / float(g.w);
  const float nanv = __builtin_nanf("");
  const int segments = (n_px + kTileW - 1) / kTileW;
  const int seg0 = tile * (kWavesPerBlock * rows_per_wave) + wave;
  float* my = slab[wave];
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;
  auto load_f = [&](int pixel) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, pixel * 8 + 4, 0, 0)); };
  struct Ref {
    float z, i, left, right, up, down, tx, ty;
  };
  // the reference pixel of this lane in segment `seg` and its four neighbours (clamped at the image border), the normalised coordinates
  auto load_ref = [&](int seg) {
    const int pc = min(seg * kTileW + lane, n_px - 1);
    int row = int(float(pc) * inv_w);
    int col = pc - row * g.w;
    if (col < 0) { col += g.w; row -= 1; }
    if (col >= g.w) { col -= g.w; row += 1; }
    Ref r;
    const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, pc * 8, 0, 0));
    r.z = zi.x; r.i = zi.y;
    r.left = load_f(pc - (col > 0 ? 1 : 0));
    r.right = load_f(pc + (col < g.w - 1 ? 1 : 0));
    r.up = load_f(pc - (row > 0 ? g.w : 0));
    r.down = load_f(pc + (row < g.h - 1 ? g.w : 0));
    r.tx = g.tx[col];
    r.ty = g.ty[row];
    return r;
  };
  __syncthreads();                                           // the level is in LDS
  Ref next = load_ref(min(seg0, segments - 1));
#pragma unroll 1
  for (int k = 0; k < rows_per_wave; ++k) {
    const int seg = seg0 + k * kWavesPerBlock;               // (uniform)
    if (seg >= segments) break;
    const Ref ref = next;
    if (k + 1 < rows_per_wave) next = load_ref(min(seg + kWavesPerBlock, segments - 1));   // the next row's round trip passes beside this row
    const int idx = seg * kTileW + lane;
    const bool in_image = idx < n_px;
    const float z = in_image ? ref.z : nanv;
    const float tx = ref.tx, ty = ref.ty;
    float u, v, qz;
    fast_project<0>(g, no_table, KT, fmaf(KT[0], tx, KT[2]), fmaf(KT[4], tx, KT[6]), fmaf(KT[8], tx, KT[10]), z, tx, ty, u, v, qz);
    // 0 <= u <= w - 2 on the integer image of the float: negative numbers and NaNs compare above every non-negative bound (Q4, Q19)
    const unsigned long long ok_mask = __builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned, u) <= w2_bits) &
                                       __builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned, v) <= h2_bits);
    const bool ok = __builtin_amdgcn_inverse_ballot_w64(ok_mask);
    const int u0 = int(u), v0 = int(v);
    FastRow r;
    r.z = z; r.i = ref.i;
    r.gx = ref.right - ref.left;                             // TWICE the central differences (the factor rides in LevelGeom::half_wi_*)
    r.gy = ref.down - ref.up;
    r.qz = qz;
    r.a1 = __builtin_amdgcn_fractf(u);
    r.b1 = __builtin_amdgcn_fractf(v);
    r.idx = v0 * pitch + u0;                                 // the neighbourhood's upper left cell: pixel (u0 - 1, v0 - 1)
    f32x2 P[4][4];
    {
      FastLdsCellPtr q = (FastLdsCellPtr)(img + (ok ? r.idx : 0));
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) P[rr][cc] = q[rr * pitch + cc];
    }
    fast_row_tail<2, false, 0, false>(g, no_table, P, r, ok_mask, tx, ty, fmaf(tx, tx, 1.0f), wt, resid, in_image ? idx * 8 : 0x7ffffff8, 0, my, lane, acc0, acc1, n_valid);
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i];
    my[256 + ((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc1[i];
  }
  if (lane == 0) counts[wave] = n_valid;
  __syncthreads();
  fast_epilogue(slab, counts, gram_entries, partials + (size_t(pair) * tiles + tile) * kAccStride, f16_range_flag ? f16_range_flag + pair : nullptr);
}

// the levels the kernel takes: even widths (the plane C of a level is built in pixel pairs), the whole level + border in kSmallCells
bool small_sweep_takes(int w, int h) {
  return w % 2 == 0 && w >= 4 && h >= 4 && (w + 2) * (h + 2) <= kSmallCells;
}

void launch_sweep_small(hipStream_t s, int rows_per_wave, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                        float* partials, float2* scratch, int* f16_range_flag) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  k_sweep_small<<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, rows_per_wave, f16_range_flag);
}

}  // namespace dvo_hip
