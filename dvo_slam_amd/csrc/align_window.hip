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
//   phase B  the workgroup loads the window [x0, x0 + 80) x [y0, y0 + 28) of the current frame's 8-byte {I, Z} plane (PairPtrs::curC)
//            -- coordinates clamped to the image, so that the clamped central differences at the image border come out of the
//            same subtraction -- into LDS, 16 bytes per load
//   phase C  per wavefront and row: 12 LDS reads, gradients, bilinear blend, residual, weight, Jacobian, Gram accumulation on
//            the matrix cores (sweep_parts.h); lanes whose taps fall outside the window (a tile that straddles a depth
//            discontinuity under a large motion) fetch their 12 pixels from memory instead -- correct for any motion
//
// Variant 7 additionally accumulates the Gram matrix on the f16 matrix pipe: every component is split exactly into hi + lo
// (two f16), G = H H^T + H L^T + L H^T through v_mfma_f32_16x16x32_f16 (4 instead of 16 matrix instructions per row; the f32
// matrix instruction shares the vector ALU's issue time, the f16 one does not).  f32-class accuracy (scripts/ubench/gram_f16.hip).
#include "gram_f16.h"

namespace dvo_hip {

constexpr int kWinTileRows = 16;                        // a 64 x 16 tile per workgroup: WAVES wavefronts x 16 / WAVES rows each
constexpr int kWinPitch = 80;                           // window columns: 64 + the taps' reach (3) + 13 of motion / parallax
// phase B: a thread owns one PAIR of window columns (16 B: the window starts at an even image column) and every sixth row
constexpr int kWinPairs = kWinPitch / 2;                // 40 column pairs
constexpr int kWinRows = 30;                            // window rows: 16 + the taps' reach (3) + 11
constexpr int kWinCells = kWinPitch * kWinRows;         // 2400 cells x 8 B = 19.2 KB
constexpr int kNoProjection = 0x7fff7fff;               // WinRowState::uv of a lane without a usable projection

typedef short __attribute__((ext_vector_type(2))) i16x2;

// wavefront-wide minimum / maximum of packed pairs of 16-bit integers through DPP (row_shr 1, 2, 4, 8; row_bcast 15, 31): valid in
// lane 63.  Lanes without a source keep their own value, which is neutral.
__device__ __forceinline__ void wave_minmax_pk16_lane63(int& lo, int& hi) {
#define DVO_STAGE(ctrl, rmask)                                                                                     \
  {                                                                                                                \
    const int a = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rmask, 0xf, false);                                    \
    const int b = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rmask, 0xf, false);                                    \
    lo = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(i16x2, lo), __builtin_bit_cast(i16x2, a))); \
    hi = __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2, hi), __builtin_bit_cast(i16x2, b))); \
  }
  DVO_STAGE(0x111, 0xf) DVO_STAGE(0x112, 0xf) DVO_STAGE(0x114, 0xf) DVO_STAGE(0x118, 0xf) DVO_STAGE(0x142, 0xa) DVO_STAGE(0x143, 0xc)
#undef DVO_STAGE
}

struct WinRowState {                                    // what phase A leaves for phase C, per row and lane
  float z, i, gx, gy;                                   // the reference quad {Zsel, I, Idx, Idy}
  float qz, a1, b1;                                     // transformed depth, bilinear weights of the +1 taps
  int uv;                                               // u0 | v0 << 16 of tap (u0, v0); kNoProjection: no usable projection
};

// WAVES: wavefronts per workgroup (4: four rows each, 8: two rows each -- less projection state per lane, one wavefront per SIMD more).
// F16: Gram accumulation on the f16 matrix pipe (variant 7).  COMPAT: the reference's x * rcp(z) in projection and weights with the
// host CPU's reciprocal table (option "ref_compat", LevelGeom::rcp_table).
template <bool F16, bool COMPAT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 6 : (F16 ? 5 : 4)) void k_sweep_window(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd, unsigned long long* __restrict__ fallback_count, int* __restrict__ f16_range_flag) {
  // (Measured and dropped: a workgroup sweeping several vertically adjacent tiles in a loop, so that the pair's state, plane pointers
  // and transform are fetched once per group of tiles -- inside a loop the compiler wants 155 vector registers for the same body and
  // spills 59 of them at the five-wavefront budget.)
  constexpr int RPW = kWinTileRows / WAVES;
  constexpr int kThreads = WAVES * 64;
  constexpr int kWinRowGroups = kThreads / kWinPairs;     // 6 (240 of 256 threads load) / 12 (480 of 512)
  constexpr int kWinLoads = (kWinRows + kWinRowGroups - 1) / kWinRowGroups;   // loads of 16 B per thread: 5 / 3
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;
  const int tile_x = tile % g.tiles_x, tile_y = tile / g.tiles_x;
  const PairState& st = states[pair];
  // (Measured and dropped: requesting the pair's plane pointers together with its activity flag and pinning both above the branch,
  // which helps the gathering sweep's short tiles -- here it cost 10 %.)
  if (!st.active || st.level != g.level) return;
  const PairPtrs pp = pairs[pair];
  const int plane_bytes = g.w * g.h * 8;
  const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t curC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.curC), 0, plane_bytes, 0x00020000);
  // the residual pairs of this pair (for the log-likelihood pass): one store per row, the row offset in a scalar register
  const __amdgpu_buffer_rsrc_t resid = __builtin_amdgcn_make_buffer_rsrc(scratch + size_t(pair) * size_t(g.w) * g.h, 0, plane_bytes, 0x00020000);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned gram_entries = kGramEntryTable.e[min(int(threadIdx.x), kNumAcc - 1)];   // (for the epilogue; loaded here, used there)
  const int u_r = tile_x * kTileW + lane;                      // < g.w: the level's width is a multiple of the tile's
  const int row_bytes = g.w * 8;
  const float nanv = __builtin_nanf("");
  const float tx_u = g.tx[u_r];
  const float cx_u = fmaf(tx_u, tx_u, 1.0f);

  constexpr int kMySlabFloats = F16 ? kSlabFloatsF16 : kSlabFloats;
  __shared__ __attribute__((aligned(16))) float slab[WAVES][kMySlabFloats];
  __shared__ __attribute__((aligned(16))) float2 win[kWinCells];
  __shared__ __attribute__((aligned(16))) int bbox[WAVES][2];
  __shared__ int counts[WAVES];
  float* my = slab[wave];
  const int off_px = u_r * 8;
  const int off_edge = (lane == 0 ? max(u_r - 1, 0) : lane == 63 ? min(u_r + 1, g.w - 1) : u_r) * 8 + 4;
  unsigned n_fallback = 0;

  const int row0 = tile_y * (WAVES * RPW) + wave;
  float ty_rows[RPW];                                          // (wave-uniform: scalar loads, once for both phases)
#pragma unroll
  for (int k = 0; k < RPW; ++k) ty_rows[k] = g.ty[min(row0 + k * WAVES, g.h - 1)];
  float KT[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];

  // ---- phase A: reference rows, projection, tap bounding box -------------------------------------------------------------------
  // Loads: the lane's byte offset in a row is a constant of the kernel, the row offset a scalar.  The horizontal neighbours of a
  // pixel come from the adjacent lanes; lanes 0 and 63 load the pixel beside the row segment (clamped at the image border like the
  // reference's derivative code, rgbd_image.cpp:419-489), the other lanes load their own pixel again: one unconditional load.
  WinRowState rs[RPW];
  int uv_min = kNoProjection, uv_max = 0;                      // packed (u0, v0) extremes over the lanes with a projection
  {
    float zv[RPW], iv[RPW], up[RPW], down[RPW], edge[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {                            // all loads of the four rows first: one round trip
      const int v = min(row0 + k * WAVES, g.h - 1);
      const int soff = v * row_bytes;
      const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, off_px, soff, 0));
      zv[k] = zi.x;
      iv[k] = zi.y;
      up[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, off_px + 4, soff - (v > 0 ? row_bytes : 0), 0));
      down[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, off_px + 4, soff + (v < g.h - 1 ? row_bytes : 0), 0));
      edge[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, off_edge, soff, 0));
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int v_r = row0 + k * WAVES;
      const int ic = __builtin_bit_cast(int, iv[k]), ie = __builtin_bit_cast(int, edge[k]);
      const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x130, 0xf, 0xf, false));   // wave_shl:1
      const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x138, 0xf, 0xf, false));    // wave_shr:1
      const float z = v_r < g.h ? zv[k] : nanv;                // rows below the image: no depth
      const float ty_p = ty_rows[k];
      int u0, v0;
      const PixelProj p = pixel_project_uv_flat<COMPAT ? 2 : 1>(g, KT, z, tx_u, ty_p, u0, v0);
      const int uv = u0 | (v0 << 16);
      rs[k].z = z;
      rs[k].i = iv[k];
      // (f16 schedule: plain differences, the 0.5 rides on a level constant -- pixel_finish_flat_d<true>)
      rs[k].gx = F16 ? right - left : (right - left) * 0.5f;
      rs[k].gy = F16 ? down[k] - up[k] : (down[k] - up[k]) * 0.5f;
      rs[k].qz = p.qz;
      rs[k].a1 = p.a1;
      rs[k].b1 = p.b1;
      rs[k].uv = p.ok ? uv : kNoProjection;
      uv_min = __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(i16x2, uv_min), __builtin_bit_cast(i16x2, rs[k].uv)));
      uv_max = __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2, uv_max), __builtin_bit_cast(i16x2, p.ok ? uv : 0)));
    }
  }
  wave_minmax_pk16_lane63(uv_min, uv_max);
  if (lane == 63) {
    bbox[wave][0] = uv_min; bbox[wave][1] = uv_max;
  }
  __syncthreads();
  int x0, y0, ww, wh;                                          // window origin (image coordinates, may be -1 / -2) and the extent in use
  {
    i16x2 lo = __builtin_bit_cast(i16x2, bbox[0][0]), hi = __builtin_bit_cast(i16x2, bbox[0][1]);
#pragma unroll
    for (int w4 = 1; w4 < WAVES; ++w4) {
      lo = __builtin_elementwise_min(lo, __builtin_bit_cast(i16x2, bbox[w4][0]));
      hi = __builtin_elementwise_max(hi, __builtin_bit_cast(i16x2, bbox[w4][1]));
    }
    const int lo_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)), hi_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi));
    const int umin = lo_s & 0xffff, vmin = (lo_s >> 16) & 0xffff, umax = hi_s & 0xffff, vmax = (hi_s >> 16) & 0xffff;
    x0 = (umin - 1) & ~1; y0 = vmin - 1;                       // even: a 16-byte load holds two window cells
    ww = umin == 0x7fff ? 0 : min(umax + 3 - x0, kWinPitch);   // columns x0 .. umax + 2
    wh = umin == 0x7fff ? 0 : min(vmax + 3 - y0, kWinRows);
  }

  // ---- phase B: the window into LDS ----------------------------------------------------------------------------------------------
  // Thread t loads column pair t % 40 of rows t / 40, t / 40 + 6, ...: the column arithmetic (clamping, the two border cases) is
  // done once per thread, a load costs a row clamp and one multiply-add.  Image width and window origin are even, so a pair lies
  // entirely inside the image, entirely left of it (both cells = column 0) or entirely right of it (both = column w - 1).  Cells
  // outside the extent in use are requested beyond the end of the plane: the buffer load returns zeros without touching memory.
  {
    const int t = threadIdx.x;
    const int rg = t / kWinPairs, cxp = t - rg * kWinPairs;
    if (rg < kWinRowGroups) {
      const int x = x0 + 2 * cxp;
      const int xl = min(max(x, 0), g.w - 2) * 8;
      const bool col_needed = 2 * cxp < ww;
      f32x4 cell[kWinLoads];
#pragma unroll
      for (int j = 0; j < kWinLoads; ++j) {
        const int cy = rg + j * kWinRowGroups;                  // (cy >= kWinRows, past the window: >= wh as well)
        const int y = min(max(y0 + cy, 0), g.h - 1);
        const int off = col_needed && cy < wh ? y * row_bytes + xl : 0x7ffffff0;
        cell[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(curC, off, 0, 0));
      }
      if (x < 0) {
#pragma unroll
        for (int j = 0; j < kWinLoads; ++j) { cell[j].z = cell[j].x; cell[j].w = cell[j].y; }
      }
      if (x >= g.w) {
#pragma unroll
        for (int j = 0; j < kWinLoads; ++j) { cell[j].x = cell[j].z; cell[j].y = cell[j].w; }
      }
      f32x4* dst = reinterpret_cast<f32x4*>(win) + t;
#pragma unroll
      for (int j = 0; j < kWinLoads; ++j)
        if (kWinRowGroups * kWinLoads == kWinRows || rg + j * kWinRowGroups < kWinRows) dst[j * kWinRowGroups * kWinPairs] = cell[j];
    }
  }
  __syncthreads();

  // ---- phase C: taps from LDS, residual, weight, Jacobian, Gram accumulation ----------------------------------------------------
  const float P00 = st.P_prev[0], P11 = st.P_prev[3], P2x = st.P_prev[1] + st.P_prev[2];
  const bool first = st.first != 0;
  const int lane_c = lane;
  f32x4* wr = reinterpret_cast<f32x4*>(my + lane_c * 4);
  const float* rd = my + ((lane_c >> 2) & 3) * kQuadStride + (lane_c >> 4) * 4 + (lane_c & 3);
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int v_r = row0 + k * WAVES;
    const WinRowState& r = rs[k];
    const bool ok = r.uv != kNoProjection;
    const int u0 = r.uv & 0xffff, v0 = r.uv >> 16;
    const int cx = u0 - x0 - 1, cy = v0 - y0 - 1;              // the 4 x 4 neighbourhood's corner in the window (>= 0 by construction)
    // (no projection: u0 = 0x7fff, cx out of range)
    const bool in_win = unsigned(cx) < unsigned(kWinPitch - 3) && unsigned(cy) < unsigned(kWinRows - 3);
    f32x2 P[4][4];
    {
      // every lane reads (lanes without a neighbourhood in the window: cell 0; their values are never used): no divergent region
      // around the twelve reads, no uninitialised registers
      // (volatile: twelve ds_read_b64, two LDS cycles each; the compiler otherwise pairs them into ds_read2_b64, eight cycles a pair)
      typedef const volatile __attribute__((address_space(3))) f32x2* LdsCellPtr;
      LdsCellPtr q = (LdsCellPtr)(win + (in_win ? cy * kWinPitch + cx : 0));
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) P[rr][cc] = q[rr * kWinPitch + cc];
    }
    if (ok && !in_win) {                                        // rare: the neighbourhood from memory, coordinates clamped like the window's
      n_fallback += 1;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) {
            const int x = min(max(u0 - 1 + cc, 0), g.w - 1), y = min(max(v0 - 1 + rr, 0), g.h - 1);
            P[rr][cc] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(curC, (y * g.w + x) * 8, 0, 0));
          }
    }
    PixelTaps t;
    {
#pragma clang fp contract(off)
      // tap (i, j) = P[j + 1][i + 1]; its gradient channels are the clamped central differences of the frame build (derive_at),
      // (next - previous) * 0.5 -- here WITHOUT the factor 0.5, which pixel_finish_flat_d applies to the four blended channels
      // instead of to the sixteen differences (a multiplication by a power of two commutes with every rounding of the blend)
#define DVO_TAP(i, j, Aq, Bq)                                                                                        \
  Aq = make_float4(P[j + 1][i + 1].x, P[j + 1][i + 1].y, P[j + 1][i + 2].x - P[j + 1][i].x, P[j + 2][i + 1].x - P[j][i + 1].x); \
  Bq = make_float2(P[j + 1][i + 2].y - P[j + 1][i].y, P[j + 2][i + 1].y - P[j][i + 1].y);
      DVO_TAP(0, 0, t.A00, t.B00)
      DVO_TAP(1, 0, t.A10, t.B10)
      DVO_TAP(0, 1, t.A01, t.B01)
      DVO_TAP(1, 1, t.A11, t.B11)
#undef DVO_TAP
    }
    const float ty_p = ty_rows[k];
    PixelProj p;
    p.Z = r.z; p.X = tx_u * r.z; p.Y = ty_p * r.z;            // rgbd_image.cpp:258 (the products pixel_project_flat forms)
    p.qz = r.qz; p.a1 = r.a1; p.b1 = r.b1; p.base = 0; p.ok = ok;
    const float4 ref = make_float4(r.z, r.i, r.gx, r.gy);
    PixelTerms o;
    const bool valid = pixel_finish_flat_d<F16>(g, ref, p, t, o) && ok;   // (f32 Gram: bit-identical to the gathering sweep)
    if constexpr (!F16) n_valid += __popcll(__ballot(valid));
    if (v_r < g.h) {                                           // (uniform)
      const f32x2 rr2 = valid ? f32x2{o.r0, o.r1} : f32x2{nanv, nanv};
      typedef unsigned __attribute__((__vector_size__(2 * sizeof(unsigned)))) u32v2;
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32v2, rr2), resid, off_px, v_r * row_bytes, 0);
    }
    if constexpr (F16) {
      // No branch on `valid`: an invalid lane's weight is zero and every product that could meet one of its NaN terms is a LEGACY
      // multiply (0 x anything = 0), so its operand rows are zeros without a second control-flow path (14 zero moves, the exec
      // juggling and the register shuffles where the two paths met: about 30 instructions per row)
      const float sw_any = first ? 1.0f : COMPAT ? tdist_weight_sqrt_compat(g.rcp_table, g.rcp_shift, o.r0, o.r1, st.P_prev) : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
      const float sw_lane = valid ? sw_any : 0.0f;
      n_valid += __popcll(__builtin_amdgcn_ballot_w64(sw_lane > 0.0f));   // (a weight is positive; counted off one compare instead of the flag's round trip through a register)
      gram_f16_row(my, lane_c, o, sw_lane, tx_u, ty_p, cx_u, acc0, acc1);
      continue;
    }
    if (valid) {
      const float sw = first ? 1.0f : COMPAT ? tdist_weight_sqrt_compat(g.rcp_table, g.rcp_shift, o.r0, o.r1, st.P_prev) : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
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

  if constexpr (F16) {
    gram_f16_finish(my, lane, acc0, acc1, f16_range_flag ? f16_range_flag + pair : nullptr);   // (one word per pair)
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
  }
  if (lane == 0) counts[wave] = n_valid;
  __syncthreads();
  const int kk = threadIdx.x;
  if (kk < kNumAcc) {
    auto G = [&](int e) {
      float t = (slab[0][e] + slab[1][e]) + (slab[2][e] + slab[3][e]);
      if (WAVES == 8) t += (slab[4][e] + slab[5][e]) + (slab[6][e] + slab[7][e]);
      return t;
    };
    float v;
    if (kk == kAccN) {
      int c = (counts[0] + counts[1]) + (counts[2] + counts[3]);
      if (WAVES == 8) c += (counts[4] + counts[5]) + (counts[6] + counts[7]);
      v = float(c);
    } else {
      const int e1 = gram_entries & 0xff, e2 = gram_entries >> 8;   // gram_entries_of_accumulator(kk), from its table
      v = G(e1);
      if (e2 != 0xff) v += G(e2);
    }
    partials[(size_t(pair) * tiles + tile) * kAccStride + kk] = v;
  }
  if (fallback_count) {
    const unsigned long long lanes = __ballot(n_fallback != 0);
    if (lanes) {
      unsigned t = n_fallback;
      for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
      if (lane == 0) atomicAdd(fallback_count, (unsigned long long)t);
    }
  }
}

bool window_sweep_supports(const LevelGeom& g) { return !g.linear && g.w % kTileW == 0 && g.w < 32768 && g.h < 32768; }

void launch_sweep_window(hipStream_t s, bool f16, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                         float* partials, float2* scratch, unsigned long long* fallback_count, int* f16_range_flag) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8);
  // (Measured and dropped: eight wavefronts of two rows each -- WAVES = 8, 69 registers, six wavefronts per SIMD instead of five:
  // 3.10 instead of 2.57 ms per 1024-pair launch; the barriers of the three phases span twice the wavefronts and every wavefront's
  // fixed work is spread over half the rows.)
  const dim3 block(256);
  if (g.rcp_table) {
    if (f16) k_sweep_window<true, true, 4><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag);
    else k_sweep_window<false, true, 4><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag);
  } else {
    if (f16) k_sweep_window<true, false, 4><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag);
    else k_sweep_window<false, false, 4><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag);
  }
}

}  // namespace dvo_hip
