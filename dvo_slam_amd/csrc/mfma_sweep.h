// mfma_sweep.h -- the device code of the gathering sweep with the normal-equation accumulation on the matrix cores (the design notes are at
// the top of align_mfma.hip), as a function of one tile (mfma_sweep_tile).  Included by align_mfma.hip and align_coarse.hip.
#pragma once
#include "gram_f16.h"

namespace dvo_hip {

// One row of the reference plane for a wavefront: the {Zsel, I} pair of every lane's pixel (64 lanes x 8 B = 512 B contiguous)
// and the three intensities the central differences need from outside the wavefront's row: the pixels above and below (clamped
// at the image border like the reference's derivative code, rgbd_image.cpp:419-489) and, in lanes 0 / 63, the pixel left / right
// of the row segment.  The horizontal neighbours of the other lanes come from the adjacent lanes (DPP wave shift).
// (Linear walk of a level narrower than a tile: all four neighbours are loaded, a segment wraps around image rows.)
struct RefRow {
  float z, i, up, down, left_or_edge, right;      // tiled: left_or_edge = the edge pixel (lanes 0 and 63 only), right unused
};

// LINEAR: the level is walked as one row of w*h pixels in 64-pixel segments (LevelGeom::linear) -- same per-pixel arithmetic,
// the pixel coordinates come from a division instead of the tile position.
// MODE 0: f32 Gram (the f32 matrix instruction); 1: the Gram accumulation on the f16 matrix pipe (gram_f16.h, schedule variant 7); 2 (round 4:
// what the default schedule, variant 8, runs on the levels the window sweep does not take): 1 with the CONTRACTED per-pixel arithmetic
// of align_fast.hip -- projection as z (KT.col0 tx + KT.col1 ty + KT.col2) + KT.col3 in fused multiply-adds, u = qx rcp(qz), the
// bounds test on the float's bits, the six channels blended in lerp form on packed f32 -- the same function, a few ulp of the tap
// coordinate apart (tests/test_gpu_parity.py::test_contracted_gathering_sweep_against_the_exact_one).
// One tile (kWavesPerBlock * RPW rows of 64 pixels, or as many 64-pixel segments of a level walked linearly) of one pair at one level:
// everything of k_residual_reduce_mfma behind "which pair, which tile, is the pair on this level".  Shared by that kernel (one workgroup
// per tile and launch) and the fused coarse-level kernel (align_coarse.hip: one workgroup walks all tiles of its pair) -- the same
// instructions, hence the same bits.  KT_src / P_prev_src / first: PairState::KT, ::P_prev, ::first (uniform); slab_mem: kWavesPerBlock
// slabs of mfma_slab_floats<MODE>() floats; counts: kWavesPerBlock ints.  Two tiles on the same LDS need a barrier between them (the
// fold at the end reads every wavefront's slab).
template <int MODE>
__host__ __device__ constexpr int mfma_slab_floats() { return MODE >= 1 ? kSlabFloatsF16 : kSlabFloats; }

// WT: partial row and residual pairs are stored write-through (agent-scope stores, sc1): the pair's step runs in this very launch
// (solver_step.h, the sweeps' tail)
template <int RPW, bool LINEAR, int MODE, bool WT = false>
__device__ __forceinline__ void mfma_sweep_tile(const LevelGeom& g, const float* KT_src, const float* P_prev_src, bool first, const PairPtrs& pp, int pair,
                                                int tile, float* __restrict__ partials, float2* __restrict__ scratch, float* slab_mem, int* counts,
                                                int* __restrict__ f16_range_flag) {
  constexpr bool F16 = MODE >= 1;
  constexpr bool FAST = MODE == 2;
  const int tiles = g.tiles_x * g.tiles_y;
  float KT[12], Pp[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = KT_src[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) Pp[i] = P_prev_src[i];
  const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, g.w * g.h * 8, 0x00020000);
  TapPlanes taps;
  taps.A = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(pp.curA), 0, g.w * g.h * 16, 0x00020000);
  taps.B = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.curB), 0, g.w * g.h * 8, 0x00020000);
  taps.rowA = g.w * 16;
  taps.rowB = g.w * 8;

  // the wavefront index is uniform: keeping it (and every row index derived from it) in scalar registers moves the row
  // bounds test, the row offsets and the ty table load from the vector ALU to the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // tiled: lane = column u_r of the tile, rows row0, row0 + 4, ...;  linear: 64-pixel segments row0, row0 + 4, ... of the
  // flattened level, lane = offset in the segment
  const int u_r = LINEAR ? lane : (tile % g.tiles_x) * kTileW + lane;
  // wavefront w sweeps rows w, w+4, w+8, ... of the tile: the four waves work on ADJACENT rows at the same time, so the
  // lower tap row of one wave is the upper tap row of the next and is served by the CU's L1 instead of a second L2 request
  const int row0 = (tile / g.tiles_x) * (kWavesPerBlock * RPW) + wave;
  const size_t pix_base = size_t(pair) * size_t(g.w) * g.h;
  const float nanv = __builtin_nanf("");
  const bool col_ok = LINEAR || u_r < g.w;
  const int n_px = g.w * g.h;
  const float inv_w = 1.0f / float(g.w);
  const float tx_u = LINEAR ? 0.0f : g.tx[col_ok ? u_r : 0];  // column term of the back-projection: constant over the rows
  const float cx_u = fmaf(tx_u, tx_u, 1.0f);
  const float P2x = Pp[1] + Pp[2];

  constexpr int kMySlabFloats = F16 ? kSlabFloatsF16 : kSlabFloats;
  float (*slab)[kMySlabFloats] = reinterpret_cast<float (*)[kMySlabFloats]>(slab_mem);
  float* my = slab[wave];
  // write side: component quad q of pixel `lane` at my[q*kQuadStride + lane*4 .. +3]
  f32x4* wr = reinterpret_cast<f32x4*>(my + lane * 4);
  // read side: MFMA operand lane l <- component c = l&15 of pixel 4g + (l>>4); g enters as a constant offset of 16 floats
  const float* rd = my + ((lane >> 2) & 3) * kQuadStride + (lane >> 4) * 4 + (lane & 3);

  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;

  // The row is straight-line code with two divergent regions (tap fetch; constraint / no constraint): no nested early exits
  // and no per-exit default values (pixel_math.h, the *_flat stages); the reference rows alternate between two register quads
  // (rows are processed in pairs) instead of being copied.  The reference row of iteration k+1 is requested before row k is
  // processed: one of the two dependent memory round trips of a row (reference pixel -> projected tap addresses) is off the
  // critical path.  (Measured and dropped: reading all sixteen matrix operands of a row behind a scheduling barrier before
  // the first matrix instruction, +15 %; one residual store per branch instead of a select, no gain; tap fetches of lanes
  // without a usable projection redirected to tap 0 instead of branched around, no gain; the three stages of a row software-
  // pipelined over rows (taps of row k+1 in flight during the second half of row k): 104 registers, 4 waves per SIMD, +10 %.)
  const float P00 = Pp[0], P11 = Pp[3];
  const int u_c = LINEAR ? lane : min(u_r, g.w - 1);
  auto load_f = [&](int pixel) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, pixel * 8 + 4, 0, 0)); };
  auto load_ref = [&](int v_r) {                              // clamped: rows / segments past the end are masked by in_image
    RefRow r;
    if constexpr (LINEAR) {
      const int idx = min(v_r * kTileW + lane, n_px - 1);
      int row = int(float(idx) * inv_w);                      // idx < 2^24: one float multiply lands within one row of the quotient
      int col = idx - row * g.w;
      if (col < 0) { col += g.w; row -= 1; }
      if (col >= g.w) { col -= g.w; row += 1; }
      const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, idx * 8, 0, 0));
      r.z = zi.x; r.i = zi.y;
      r.left_or_edge = load_f(idx - (col > 0 ? 1 : 0));
      r.right = load_f(idx + (col < g.w - 1 ? 1 : 0));
      r.up = load_f(idx - (row > 0 ? g.w : 0));
      r.down = load_f(idx + (row < g.h - 1 ? g.w : 0));
    } else {
      const int v = min(v_r, g.h - 1);
      const int idx = v * g.w + u_c;
      const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, idx * 8, 0, 0));   // 512 B contiguous per wave
      r.z = zi.x; r.i = zi.y;
      r.up = load_f(idx - (v > 0 ? g.w : 0));
      r.down = load_f(idx + (v < g.h - 1 ? g.w : 0));
      r.left_or_edge = 0.0f;
      r.right = 0.0f;
      if (lane == 0) r.left_or_edge = load_f(idx - (u_c > 0 ? 1 : 0));
      if (lane == 63) r.left_or_edge = load_f(idx + (u_c < g.w - 1 ? 1 : 0));
    }
    return r;
  };
  // the reference quad {Zsel, I, Idx, Idy} the per-pixel stages work on: the gradient is the reference's central difference
  // 0.5 (next - previous), same operation order as the frame build (pyramid_kernels.hip::derive_at), hence the same bits
  auto ref_quad = [&](const RefRow& r) {
    float left, right;
    if constexpr (LINEAR) {
      left = r.left_or_edge;
      right = r.right;
    } else {
      const int ic = __builtin_bit_cast(int, r.i), ie = __builtin_bit_cast(int, r.left_or_edge);
      right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x130, 0xf, 0xf, false));   // wave_shl:1 -- lane l <- lane l + 1; lane 63 keeps the edge
      left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x138, 0xf, 0xf, false));    // wave_shr:1 -- lane l <- lane l - 1; lane 0 keeps the edge
    }
    return make_float4(r.z, r.i, (right - left) * 0.5f, (r.down - r.up) * 0.5f);
  };
  auto sweep_row = [&](int v_r, const RefRow& ref_row) __attribute__((always_inline)) {
    const float4 ref = ref_quad(ref_row);
    bool in_image;
    size_t pix;                                               // index of this lane's pixel in the level
    float tx_p, ty_p, cx;
    if constexpr (LINEAR) {
      const int idx = v_r * kTileW + lane;
      in_image = idx < n_px;
      pix = size_t(idx);
      const int pc = in_image ? idx : 0;
      int row = int(float(pc) * inv_w);                       // idx < 2^24: one float multiply lands within one row of the quotient
      int col = pc - row * g.w;
      if (col < 0) { col += g.w; row -= 1; }
      if (col >= g.w) { col -= g.w; row += 1; }
      tx_p = g.tx[col];
      ty_p = g.ty[row];
      cx = fmaf(tx_p, tx_p, 1.0f);
    } else {
      in_image = col_ok && v_r < g.h;
      pix = size_t(v_r) * g.w + u_r;                          // scalar row offset + lane
      tx_p = tx_u;
      ty_p = g.ty[min(v_r, g.h - 1)];
      cx = cx_u;
    }
    PixelProj p;
    if constexpr (FAST) {
      const float Z = in_image ? ref.x : nanv;
      const float qx = fmaf(Z, fmaf(KT[1], ty_p, fmaf(KT[0], tx_p, KT[2])), KT[3]);
      const float qy = fmaf(Z, fmaf(KT[5], ty_p, fmaf(KT[4], tx_p, KT[6])), KT[7]);
      const float qz = fmaf(Z, fmaf(KT[9], ty_p, fmaf(KT[8], tx_p, KT[10])), KT[11]);
      const float rq = __builtin_amdgcn_rcpf(qz);
      const float u = qx * rq, v = qy * rq;
      // 0 <= u <= w - 2 on the integer image of the float (negative numbers and NaNs compare above every non-negative bound; Q4, Q19)
      p.ok = __builtin_bit_cast(unsigned, u) <= __builtin_bit_cast(unsigned, float(g.w - 2)) &&
             __builtin_bit_cast(unsigned, v) <= __builtin_bit_cast(unsigned, float(g.h - 2));
      p.Z = Z; p.X = tx_p * Z; p.Y = ty_p * Z; p.qz = qz;
      p.a1 = __builtin_amdgcn_fractf(u); p.b1 = __builtin_amdgcn_fractf(v);
      p.base = int(v) * g.w + int(u);                          // (meaningless unless ok)
    } else {
      p = g.rcp_table ? pixel_project_flat<true>(g, KT, in_image ? ref.x : nanv, tx_p, ty_p)      // (uniform; option "ref_compat")
                      : pixel_project_flat<false>(g, KT, in_image ? ref.x : nanv, tx_p, ty_p);
    }
    PixelTaps t;
    if (p.ok) taps.fetch(p.base, t);                          // lanes without a usable projection are masked out of `valid`
    PixelTerms o;
    bool valid;
    if constexpr (FAST) {
      // the six channels {I, Z}, {Idx, Idy}, {Zdx, Zdy} as three register pairs through one lerp formula (packed f32)
      const f32x2 a1 = {p.a1, p.a1}, b1 = {p.b1, p.b1};
      auto blend = [&](f32x2 v00, f32x2 v10, f32x2 v01, f32x2 v11) {
        const f32x2 top = __builtin_elementwise_fma(a1, v10 - v00, v00), bot = __builtin_elementwise_fma(a1, v11 - v01, v01);
        return __builtin_elementwise_fma(b1, bot - top, top);
      };
      const f32x2 cIZ = blend(f32x2{t.A00.x, t.A00.y}, f32x2{t.A10.x, t.A10.y}, f32x2{t.A01.x, t.A01.y}, f32x2{t.A11.x, t.A11.y});
      const f32x2 cIg = blend(f32x2{t.A00.z, t.A00.w}, f32x2{t.A10.z, t.A10.w}, f32x2{t.A01.z, t.A01.w}, f32x2{t.A11.z, t.A11.w});
      const f32x2 cZg = blend(f32x2{t.B00.x, t.B00.y}, f32x2{t.B10.x, t.B10.y}, f32x2{t.B01.x, t.B01.y}, f32x2{t.B11.x, t.B11.y});
      o.r0 = (cIZ.x - ref.y) * (1.0f / 255.0f);
      o.r1 = cIZ.y - p.qz;
      const float dz = p.Z - 0.4f;
      o.gix = g.wi_x * (cIg.x + ref.z);
      o.giy = g.wi_y * (cIg.y + ref.w);
      o.gzx = g.fx * cZg.x;
      o.gzy = g.fy * cZg.y;
      o.X = p.X; o.Y = p.Y; o.Z = p.Z;
      // Q9: a hole under any tap makes cZ (hence r1), cZx or cZy not-a-number; Q5: the occlusion threshold -20 (0.0012 + 0.0019 (z - 0.4)^2)
      valid = p.ok && o.r1 > fmaf(dz * -0.038f, dz, -0.024f) && !__builtin_isunordered(cZg.x, cZg.y);
    } else {
      valid = pixel_finish_flat(g, ref, p, t, o) && p.ok;
    }
    n_valid += __popcll(__ballot(valid));                     // exact count on the scalar unit
    if (in_image) {                                           // one full-width store
      const float2 rr = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);
      if constexpr (WT) __hip_atomic_store(reinterpret_cast<unsigned long long*>(scratch + pix_base + pix), __builtin_bit_cast(unsigned long long, rr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else scratch[pix_base + pix] = rr;
    }
    if constexpr (F16) {
      // (no branch on `valid`: zero weight and legacy multiplies, gram_f16.h)
      const float sw_any = first ? 1.0f : g.rcp_table ? tdist_weight_sqrt_compat(g.rcp_table, g.rcp_shift, o.r0, o.r1, Pp)
                                                      : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
      gram_f16_row(my, lane, o, valid ? sw_any : 0.0f, tx_p, ty_p, cx, acc0, acc1);
      return;
    }
    if (valid) {
      // t-distribution weight with the PREVIOUS pass' precision (Q11); first pass on a level: w = 1.  sqrt(w) is folded into
      // the four gradient factors of the Jacobian rows.
      const float sw = first ? 1.0f : g.rcp_table ? tdist_weight_sqrt_compat(g.rcp_table, g.rcp_shift, o.r0, o.r1, Pp)   // (uniform; option "ref_compat")
                                                  : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
      float J0[6], J1[6];
      jacobian_rows_fast(o, sw, tx_p, ty_p, cx, fmaf(ty_p, ty_p, 1.0f), J0, J1);
      wr[0] = f32x4{J0[0], J0[1], J0[2], J0[3]};
      wr[kQuadStride / 4] = f32x4{J0[4], J0[5], J1[0], J1[1]};
      wr[2 * (kQuadStride / 4)] = f32x4{J1[2], J1[3], J1[4], J1[5]};
      wr[3 * (kQuadStride / 4)] = f32x4{sw * o.r0, sw * o.r1, 0.0f, 0.0f};
    } else {
      // a pixel without a constraint contributes a zero vector: four stores of one zero quad instead of clearing the
      // fourteen component registers on every row
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      wr[0] = zero;
      wr[kQuadStride / 4] = zero;
      wr[2 * (kQuadStride / 4)] = zero;
      wr[3 * (kQuadStride / 4)] = zero;
    }
    // the slab is private to this wavefront and LDS executes a wavefront's operations in order: only the compiler has to
    // be kept from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int grp = 0; grp < 16; grp += 2) {                   // 4 pixels per MFMA, two independent accumulator chains
      const float a0 = rd[grp * 16], a1 = rd[grp * 16 + 16];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  RefRow ref_a = load_ref(row0), ref_b = ref_a;
#pragma unroll 1
  for (int k = 0; k < RPW; k += 2) {
    const int v_r = row0 + k * kWavesPerBlock;                // scalar: image row (tiled) or segment (linear)
    if (k + 1 < RPW) ref_b = load_ref(v_r + kWavesPerBlock);
    sweep_row(v_r, ref_a);
    if (k + 1 < RPW) {
      if (k + 2 < RPW) ref_a = load_ref(v_r + 2 * kWavesPerBlock);
      sweep_row(v_r + kWavesPerBlock, ref_b);
    }
  }

  // lane l, register i holds G[row (l>>4)*4 + i][col l&15] of this wavefront's rows
#pragma unroll
  // (the wavefront is done with its slab: its Gram matrix goes into the first 256 floats)
  for (int i = 0; i < 4; ++i)
    if (!F16) my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
  if constexpr (F16) gram_f16_finish(my, lane, acc0, acc1, f16_range_flag ? f16_range_flag + pair : nullptr);   // (one word per pair)
  if (lane == 0) counts[wave] = n_valid;
  __syncthreads();
  // fold the four wavefront Gram matrices into the canonical partial row (device_types.h); vector layout:
  // components 0..5 = J0, 6..11 = J1, 12 = r0, 13 = r1
  const int k = threadIdx.x;
  if (k < kNumAcc) {
    auto G = [&](int e) { return (slab[0][e] + slab[1][e]) + (slab[2][e] + slab[3][e]); };
    float v;
    if (k == kAccN) {
      v = float((counts[0] + counts[1]) + (counts[2] + counts[3]));
    } else {
      int e1, e2;
      gram_entries_of_accumulator(k, e1, e2);
      v = G(e1);
      if (e2 >= 0) v += G(e2);
    }
    if constexpr (WT) __hip_atomic_store(partials + (size_t(pair) * tiles + tile) * kAccStride + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partials[(size_t(pair) * tiles + tile) * kAccStride + k] = v;
  }
}

}  // namespace dvo_hip
