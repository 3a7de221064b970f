// align_window.hip -- the fused residual / Jacobian / reduce sweep with the current frame's pixels STAGED IN LDS
// (schedule variants 6 and 7 of the sweep; same per-pixel arithmetic and the same outputs as align_mfma.hip, bit for bit).
//
// Why: the counters of the gathering sweep (profiles/r02_c_pmc_utilisation.md) show the vector-memory path 78 % / 93 % busy:
// every reference pixel requests 4 x 16 B + 4 x 8 B = 96 B of bilinear taps through the 64-B-per-clock vector L1 although the
// planes hold only 24 B per pixel -- the 4x re-read that neighbouring lanes cause IS the bottleneck.  Here a workgroup (4
// wavefronts, a 64 x 16 tile of reference pixels) first projects its pixels, takes the bounding box of their taps in the current
// image, loads that window of the 8-byte {I, Z} plane ONCE, coalesced, into LDS (256 B per clock), and every lane then reads the
// 4 x 4 neighbourhood of its tap corner from LDS (12 x ds_read_b64) and derives the four gradient channels itself -- the clamped
// central differences of rgbd_image.cpp:419-489, same operation order as the frame build, hence the same bits as the stored
// planes A / B (pyramid_kernels.hip::derive_at).  HBM / L2 traffic of the current frame drops from 24 B to 8 B per pixel (x the
// window's halo), the vector L1 sees ~30 B per pixel instead of 120.
//
//   phase A  per wavefront: stream its 4 reference rows, project (dense_tracking_impl.cpp:148-203), keep the projection in
//            registers (8 per row), reduce the tap bounding box (DPP + one LDS exchange between the four wavefronts)
//   phase B  the workgroup loads the window [x0, x0 + 80) x [y0, y0 + 28) -- coordinates clamped to the image, so that the
//            clamped central differences at the image border come out of the same subtraction -- into LDS
//   phase C  per wavefront and row: 12 LDS reads, gradients, bilinear blend, residual, weight, Jacobian, Gram accumulation on
//            the matrix cores (sweep_parts.h); lanes whose taps fall outside the window (a tile that straddles a depth
//            discontinuity under a large motion) fetch their 12 pixels from memory instead -- correct for any motion
//
// Variant 7 additionally accumulates the Gram matrix on the f16 matrix pipe: every component is split exactly into hi + lo
// (two f16), G = H H^T + H L^T + L H^T through v_mfma_f32_16x16x32_f16 (4 instead of 16 matrix instructions per row; the f32
// matrix instruction shares the vector ALU's issue time, the f16 one does not).  f32-class accuracy (scripts/ubench/gram_f16.hip).
#include "sweep_parts.h"

namespace dvo_hip {

constexpr int kWinRPW = 4;                              // rows per wavefront: a 64 x 16 tile per workgroup
constexpr int kWinPitch = 80;                           // window columns: 64 + the taps' reach (3) + 13 of motion / parallax
constexpr int kWinRows = kWavesPerBlock * kWinRPW + 12; // window rows
constexpr int kWinCells = kWinPitch * kWinRows;         // 2240 cells x 8 B = 17.9 KB
constexpr int kWinLoads = (kWinCells + kBlock - 1) / kBlock;

// wavefront-wide minimum of a signed int through DPP (row_shr 1, 2, 4, 8; row_bcast 15, 31): valid in lane 63.  Lanes without a
// source keep their own value (old = v), which is neutral for min.
__device__ __forceinline__ int wave_min_lane63(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
  return v;
}

struct WinRowState {                                    // what phase A leaves for phase C, per row and lane
  float z, i, gx, gy;                                   // the reference quad {Zsel, I, Idx, Idy}
  float qz, a1, b1;                                     // transformed depth, bilinear weights of the +1 taps
  int uv;                                               // u0 | v0 << 16 of tap (u0, v0); -1: no usable projection
};

typedef _Float16 __attribute__((ext_vector_type(8))) f16x8;
typedef __fp16 __attribute__((__vector_size__(4 * sizeof(__fp16)))) fp16x4;
typedef __fp16 __attribute__((ext_vector_type(2))) fp16x2;

// F16: Gram accumulation on the f16 matrix pipe (variant 7)
template <bool F16>
__global__ __launch_bounds__(kBlock, 4) void k_sweep_window(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd, unsigned long long* __restrict__ fallback_count) {
  constexpr int RPW = kWinRPW;
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;
  const PairState& st = states[pair];
  if (!st.active) return;
  const PairPtrs pp = pairs[pair];
  float KT[12], Pp[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) Pp[i] = st.P_prev[i];
  const bool first = st.first != 0;
  const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, g.w * g.h * 8, 0x00020000);
  // the current frame's {I, Z}: its own 8-byte plane when the frame has one, else the first half of the 16-byte taps
  const int cshift = pp.curC ? 3 : 4;
  const __amdgpu_buffer_rsrc_t curC = __builtin_amdgcn_make_buffer_rsrc(
      pp.curC ? static_cast<void*>(const_cast<float2*>(pp.curC)) : static_cast<void*>(const_cast<float4*>(pp.curA)), 0, (g.w * g.h) << cshift, 0x00020000);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int u_r = (tile % g.tiles_x) * kTileW + lane;
  const int row0 = (tile / g.tiles_x) * (kWavesPerBlock * RPW) + wave;
  const size_t pix_base = size_t(pair) * size_t(g.w) * g.h;
  const float nanv = __builtin_nanf("");
  const bool col_ok = u_r < g.w;
  const float tx_u = g.tx[col_ok ? u_r : 0];
  const float cx_u = fmaf(tx_u, tx_u, 1.0f);
  const float P00 = Pp[0], P11 = Pp[3], P2x = Pp[1] + Pp[2];
  const int u_c = min(u_r, g.w - 1);

  __shared__ __attribute__((aligned(16))) float slab[kWavesPerBlock][kSlabFloats];
  __shared__ __attribute__((aligned(16))) float2 win[kWinCells];
  __shared__ __attribute__((aligned(16))) int bbox[kWavesPerBlock][4];
  __shared__ int counts[kWavesPerBlock];
  float* my = slab[wave];

  // ---- phase A: reference rows, projection, tap bounding box -------------------------------------------------------------------
  auto load_f = [&](int pixel) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, pixel * 8 + 4, 0, 0)); };
  WinRowState rs[RPW];
  int umin = 0x7fff, vmin = 0x7fff, umax_n = 0x7fff, vmax_n = 0x7fff;     // maxima as minima of the negated value
  {
    float zv[RPW], iv[RPW];
    float up[RPW], down[RPW], edge[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {                            // all loads of the four rows first: one round trip
      const int v = min(row0 + k * kWavesPerBlock, g.h - 1);
      const int idx = v * g.w + u_c;
      const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, idx * 8, 0, 0));
      zv[k] = zi.x;
      iv[k] = zi.y;
      up[k] = load_f(idx - (v > 0 ? g.w : 0));
      down[k] = load_f(idx + (v < g.h - 1 ? g.w : 0));
      edge[k] = 0.0f;
      if (lane == 0) edge[k] = load_f(idx - (u_c > 0 ? 1 : 0));
      if (lane == 63) edge[k] = load_f(idx + (u_c < g.w - 1 ? 1 : 0));
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int v_r = row0 + k * kWavesPerBlock;
      const int ic = __builtin_bit_cast(int, iv[k]), ie = __builtin_bit_cast(int, edge[k]);
      const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x130, 0xf, 0xf, false));   // wave_shl:1
      const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x138, 0xf, 0xf, false));    // wave_shr:1
      const bool in_image = col_ok && v_r < g.h;
      const float ty_p = g.ty[min(v_r, g.h - 1)];
      int u0, v0;
      const PixelProj p = pixel_project_uv_flat(g, KT, in_image ? zv[k] : nanv, tx_u, ty_p, u0, v0);
      rs[k].z = in_image ? zv[k] : nanv;
      rs[k].i = iv[k];
      rs[k].gx = (right - left) * 0.5f;
      rs[k].gy = (down[k] - up[k]) * 0.5f;
      rs[k].qz = p.qz;
      rs[k].a1 = p.a1;
      rs[k].b1 = p.b1;
      rs[k].uv = p.ok ? (u0 | (v0 << 16)) : -1;
      if (p.ok) {
        umin = min(umin, u0); vmin = min(vmin, v0);
        umax_n = min(umax_n, -u0); vmax_n = min(vmax_n, -v0);
      }
    }
  }
  {
    const int a = wave_min_lane63(umin), b = wave_min_lane63(vmin), c = wave_min_lane63(umax_n), d = wave_min_lane63(vmax_n);
    if (lane == 63) {
      bbox[wave][0] = a; bbox[wave][1] = b; bbox[wave][2] = c; bbox[wave][3] = d;
    }
  }
  __syncthreads();
  int x0, y0, ww, wh;                                          // window origin (image coordinates, may be -1) and the extent in use
  {
    int a = 0x7fff, b = 0x7fff, c = 0x7fff, d = 0x7fff;
#pragma unroll
    for (int w4 = 0; w4 < kWavesPerBlock; ++w4) {
      a = min(a, bbox[w4][0]); b = min(b, bbox[w4][1]); c = min(c, bbox[w4][2]); d = min(d, bbox[w4][3]);
    }
    a = __builtin_amdgcn_readfirstlane(a); b = __builtin_amdgcn_readfirstlane(b);
    c = __builtin_amdgcn_readfirstlane(c); d = __builtin_amdgcn_readfirstlane(d);
    x0 = a - 1; y0 = b - 1;
    ww = a == 0x7fff ? 0 : min(-c + 3 - x0, kWinPitch);        // columns x0 .. umax + 2
    wh = a == 0x7fff ? 0 : min(-d + 3 - y0, kWinRows);
  }

  // ---- phase B: the window into LDS ----------------------------------------------------------------------------------------------
  {
    f32x2 cell[kWinLoads];
#pragma unroll
    for (int j = 0; j < kWinLoads; ++j) {
      const int e = int(threadIdx.x) + j * kBlock;
      const int cy = e / kWinPitch, cx = e - cy * kWinPitch;
      cell[j] = f32x2{0.0f, 0.0f};
      if (cy < wh && cx < ww) {
        const int x = min(max(x0 + cx, 0), g.w - 1), y = min(max(y0 + cy, 0), g.h - 1);
        cell[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(curC, (y * g.w + x) << cshift, 0, 0));
      }
    }
#pragma unroll
    for (int j = 0; j < kWinLoads; ++j) {
      const int e = int(threadIdx.x) + j * kBlock;
      const int cy = e / kWinPitch, cx = e - cy * kWinPitch;
      if (cy < wh && cx < ww) win[e] = make_float2(cell[j].x, cell[j].y);
    }
  }
  __syncthreads();

  // ---- phase C: taps from LDS, residual, weight, Jacobian, Gram accumulation ----------------------------------------------------
  f32x4* wr = reinterpret_cast<f32x4*>(my + lane * 4);
  const float* rd = my + ((lane >> 2) & 3) * kQuadStride + (lane >> 4) * 4 + (lane & 3);
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;
  unsigned n_fallback = 0;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int v_r = row0 + k * kWavesPerBlock;
    const WinRowState& r = rs[k];
    const bool ok = r.uv >= 0;
    const int u0 = r.uv & 0xffff, v0 = r.uv >> 16;
    const int cx = u0 - x0 - 1, cy = v0 - y0 - 1;              // the 4 x 4 neighbourhood's corner in the window (>= 0 by construction)
    const bool in_win = ok && cx + 3 < kWinPitch && cy + 3 < kWinRows;
    f32x2 P[4][4];
    {
      // every lane reads (lanes without a neighbourhood in the window: cell 0; their values are never used): no divergent region
      // around the twelve reads, no uninitialised registers
      const float2* q = win + (in_win ? cy * kWinPitch + cx : 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) {
            const float2 t = q[rr * kWinPitch + cc];
            P[rr][cc] = f32x2{t.x, t.y};
          }
    }
    if (ok && !in_win) {                                        // rare: the neighbourhood from memory, coordinates clamped like the window's
      n_fallback += 1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) {
            const int x = min(max(u0 - 1 + cc, 0), g.w - 1), y = min(max(v0 - 1 + rr, 0), g.h - 1);
            P[rr][cc] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(curC, (y * g.w + x) << cshift, 0, 0));
          }
    }
    PixelTaps t;
    {
#pragma clang fp contract(off)
      // tap (i, j) = P[j + 1][i + 1]; its gradients are the clamped central differences of the frame build (derive_at): (next - previous) * 0.5
#define DVO_TAP(i, j, Aq, Bq)                                                                                        \
  Aq = make_float4(P[j + 1][i + 1].x, P[j + 1][i + 1].y, (P[j + 1][i + 2].x - P[j + 1][i].x) * 0.5f,               \
                   (P[j + 2][i + 1].x - P[j][i + 1].x) * 0.5f);                                                     \
  Bq = make_float2((P[j + 1][i + 2].y - P[j + 1][i].y) * 0.5f, (P[j + 2][i + 1].y - P[j][i + 1].y) * 0.5f);
      DVO_TAP(0, 0, t.A00, t.B00)
      DVO_TAP(1, 0, t.A10, t.B10)
      DVO_TAP(0, 1, t.A01, t.B01)
      DVO_TAP(1, 1, t.A11, t.B11)
#undef DVO_TAP
    }
    const float ty_p = g.ty[min(v_r, g.h - 1)];
    PixelProj p;
    p.Z = r.z; p.X = tx_u * r.z; p.Y = ty_p * r.z;            // rgbd_image.cpp:258 (the products pixel_project_flat forms)
    p.qz = r.qz; p.a1 = r.a1; p.b1 = r.b1; p.base = 0; p.ok = ok;
    const float4 ref = make_float4(r.z, r.i, r.gx, r.gy);
    PixelTerms o;
    const bool valid = pixel_finish_flat(g, ref, p, t, o) && ok;
    n_valid += __popcll(__ballot(valid));
    const bool in_image = col_ok && v_r < g.h;
    if (in_image) scratch[pix_base + size_t(v_r) * g.w + u_r] = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);
    if (valid) {
      const float sw = first ? 1.0f : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
      float J0[6], J1[6];
      jacobian_rows_fast(o, sw, tx_u, ty_p, cx_u, fmaf(ty_p, ty_p, 1.0f), J0, J1);
      wr[0] = f32x4{J0[0], J0[1], J0[2], J0[3]};
      wr[kQuadStride / 4] = f32x4{J0[4], J0[5], J1[0], J1[1]};
      wr[2 * (kQuadStride / 4)] = f32x4{J1[2], J1[3], J1[4], J1[5]};
      wr[3 * (kQuadStride / 4)] = f32x4{sw * o.r0, sw * o.r1, 0.0f, 0.0f};
    } else {
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      wr[0] = zero;
      wr[kQuadStride / 4] = zero;
      wr[2 * (kQuadStride / 4)] = zero;
      wr[3 * (kQuadStride / 4)] = zero;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int grp = 0; grp < 16; grp += 2) {
      const float a0 = rd[grp * 16], a1 = rd[grp * 16 + 16];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
  if (lane == 0) counts[wave] = n_valid;
  if (fallback_count) {
    const unsigned long long lanes = __ballot(n_fallback != 0);
    if (lanes) {
      unsigned t = n_fallback;
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0) atomicAdd(fallback_count, (unsigned long long)t);
    }
  }
  __syncthreads();
  const int kk = threadIdx.x;
  if (kk < kNumAcc) {
    auto G = [&](int e) { return (slab[0][e] + slab[1][e]) + (slab[2][e] + slab[3][e]); };
    float v;
    if (kk == kAccN) {
      v = float((counts[0] + counts[1]) + (counts[2] + counts[3]));
    } else {
      int e1, e2;
      gram_entries_of_accumulator(kk, e1, e2);
      v = G(e1);
      if (e2 >= 0) v += G(e2);
    }
    partials[(size_t(pair) * tiles + tile) * kAccStride + kk] = v;
  }
}

bool window_sweep_supports(const LevelGeom& g) { return !g.linear && g.w % kTileW == 0 && g.w < 32768 && g.h < 32768; }

void launch_sweep_window(hipStream_t s, bool f16, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                         float* partials, float2* scratch, unsigned long long* fallback_count) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(kBlock);
  (void)f16;
  k_sweep_window<false><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count);
}

}  // namespace dvo_hip
