// fast_sweep.h -- the device code of the contracted window sweep (schedule variants 8 / 9; the design notes are at the top of
// align_fast.hip): window, per-pixel arithmetic, operand rows on the f16 matrix pipe, epilogue, and the tile function that puts them
// together for one 64 x 16 tile (fast_sweep_tile).  Included by align_fast.hip and align_coarse.hip.
#pragma once
#include <type_traits>

#include "gram_f16.h"


namespace dvo_hip {

constexpr int kFastTileRows = 16;                       // a 64 x 16 tile per workgroup, four rows per wavefront
constexpr int kFastPitch = 96;                          // window pitch in cells (768 B: every window row starts at bank 0)
constexpr int kFastCols = 84;                           // window columns in use: 64 + the taps' reach (3) + 17 of motion / parallax
constexpr int kFastPairs = kFastCols / 2;               // 42 column pairs (16-byte loads)
constexpr int kFastRows = 28;                           // window rows: 16 + the taps' reach (3) + 9
constexpr int kFastRowGroups = 6;                       // 6 x 42 = 252 of the 256 threads fill the window, five rows each at most
constexpr int kFastLoads = (kFastRows + kFastRowGroups - 1) / kFastRowGroups;
constexpr int kFastCells = kFastPitch * kFastRows;      // 2688 cells x 8 B = 21 504 B (+ 4 x 2560 B of operand slabs = 31.8 KB: five workgroups per CU)

typedef short __attribute__((ext_vector_type(2))) fast_i16x2;
typedef unsigned __attribute__((ext_vector_type(2))) fast_u32x2;
typedef unsigned __attribute__((__vector_size__(2 * sizeof(unsigned)))) fast_u32v2;
typedef const volatile __attribute__((address_space(3))) f32x2* FastLdsCellPtr;

// wavefront-wide minima (a, b) and maxima (c, d) of four integers through DPP, fused into the min / max instruction (row_shr 1, 2,
// 4, 8; row_bcast 15, 31): valid in lane 63.  A lane without a source does not execute and keeps its own value, which is neutral.
// Four registers per stage: an instruction never reads what the one before it wrote (the two wait states a DPP read needs after a
// vector write are covered by its neighbours; the first stage follows unknown code: s_nop).
__device__ __forceinline__ void fast_wave_min2max2_lane63(int& a, int& b, int& c, int& d) {
#define DVO_STAGE(ctrl)                                                                                            \
  asm volatile("s_nop 1\n\t"                                                                                       \
               "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"                                                              \
               "v_min_i32_dpp %1, %1, %1 " ctrl "\n\t"                                                              \
               "v_max_i32_dpp %2, %2, %2 " ctrl "\n\t"                                                              \
               "v_max_i32_dpp %3, %3, %3 " ctrl                                                                     \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  DVO_STAGE("row_shr:1 row_mask:0xf bank_mask:0xf")
  DVO_STAGE("row_shr:2 row_mask:0xf bank_mask:0xf")
  DVO_STAGE("row_shr:4 row_mask:0xf bank_mask:0xf")
  DVO_STAGE("row_shr:8 row_mask:0xf bank_mask:0xf")
  DVO_STAGE("row_bcast:15 row_mask:0xa bank_mask:0xf")
  DVO_STAGE("row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef DVO_STAGE
}

struct FastRow {                                        // what phase A leaves for phase C, per row and lane (8 registers)
  float z, i, gx, gy;                                   // reference {Zsel, I} and TWICE the central differences of I
  float qz, a1, b1;                                     // transformed depth, bilinear weights of the +1 taps
  int idx;                                              // u0 + kFastPitch * v0 of tap corner (u0, v0)
};

__device__ __forceinline__ float fast_lerp(float p, float q, float t) { return fmaf(t, q - p, p); }
__device__ __forceinline__ f32x2 fast_lerp2(f32x2 p, f32x2 q, f32x2 t) { return __builtin_elementwise_fma(t, q - p, p); }

// The host CPU's _mm_rcp_ps from its table (pixel_math.h::rcp_like_the_host, option "ref_compat"), the exponent and the sign put back
// with integer arithmetic.  COMPAT 2: the table's 16-bit copy in LDS (LevelGeom::rcp_packed; 4-8 KB, filled by every workgroup) -- the
// weights' operand 5 + r^T P r takes any mantissa, so its 64 lanes ask for 64 different cache lines of a table in memory and the
// texture addresser, not the vector ALU, paces the sweep (round 5: 17.6 ms per 1024-pair step that way, against 11.6 without the
// table).  COMPAT 1: a table that does not pack -- one 4-byte gather per lane through a buffer resource.
struct FastRcpSource {
  __amdgpu_buffer_rsrc_t table;
  const unsigned short* lds;
  int shift;
};

template <int COMPAT>
__device__ __forceinline__ float fast_rcp_host(const FastRcpSource& src, float x) {
  const unsigned b = __builtin_bit_cast(unsigned, x);
  const unsigned idx = (b & 0x7fffffu) >> src.shift;
  unsigned t;                                                                                          // rcp(1.m), in (0.5, 1]
  if constexpr (COMPAT == 2) t = 0x3f000000u + (unsigned(src.lds[idx]) << 8);
  else t = __builtin_amdgcn_raw_buffer_load_b32(src.table, idx << 2, 0, 0);
  t = (t + (0x3f800000u - (b & 0x7f800000u))) | (b & 0x80000000u);                                     // x 2^-(e - 127), sign of x
  return __builtin_bit_cast(float, t);
}

// q = K T (tx z, ty z, z, 1) as z (KT.col0 tx + KT.col1 ty + KT.col2) + KT.col3 and u = qx rcp(qz), v = qy rcp(qz) (v_rcp_f32, 1 ulp).
// c0..c2: the column's part of the bracket, fmaf(KT[4 i], tx, KT[4 i + 2]).  One instruction sequence for phase A and for the lanes that
// look their tap corner up again (the checked path): the same bits.
// COMPAT (option "ref_compat", the reference's u = x * _mm_rcp_ps(z), dense_tracking_impl.cpp:192): the reciprocal is the host CPU's
// table value.  That is a STEP function of qz (2^11 - 2^12 steps per binade), so qz is formed in the reference's operation order
// without contraction -- the very float the exact schedule and the oracle hand to the table, hence the same table entry -- while qx
// and qy stay contracted: u and v then differ from the exact schedule's by the few ulp they do in the default mode, not by a table step.
template <int COMPAT>
__device__ __forceinline__ void fast_project(const LevelGeom& g, const FastRcpSource& table, const float* KT, float c0, float c1, float c2, float z,
                                             float tx, float ty, float& u, float& v, float& qz) {
  const float qx = fmaf(z, fmaf(KT[1], ty, c0), KT[3]);
  const float qy = fmaf(z, fmaf(KT[5], ty, c1), KT[7]);
  float r;
  if constexpr (COMPAT != 0) {
    {
#pragma clang fp contract(off)
      const float X = tx * z, Y = ty * z;                     // rgbd_image.cpp:258; pixel_math.h::pixel_project_uv_flat
      qz = (KT[8] * X + KT[9] * Y) + (KT[10] * z + KT[11]);
    }
    r = fast_rcp_host<COMPAT>(table, qz);
  } else {
    qz = fmaf(z, fmaf(KT[9], ty, c2), KT[11]);
    r = __builtin_amdgcn_rcpf(qz);
  }
  u = qx * r;
  v = qy * r;
}

// ---- the window ------------------------------------------------------------------------------------------------------------------
struct FastWindow {
  int x0, y0, ww, wh;                                   // origin (image coordinates, may be -1 / -2) and the extent in use
  bool all_in;                                          // every projected neighbourhood of the tile lies inside the window
};

// from the four wavefronts' tap bounding boxes (packed u | v << 16: minima in [w][0], maxima in [w][1]; 0x7fff = no projection)
__device__ __forceinline__ FastWindow fast_window_of(const int (*bbox)[2]) {
  fast_i16x2 lo = __builtin_bit_cast(fast_i16x2, bbox[0][0]), hi = __builtin_bit_cast(fast_i16x2, bbox[0][1]);
#pragma unroll
  for (int w4 = 1; w4 < 4; ++w4) {
    lo = __builtin_elementwise_min(lo, __builtin_bit_cast(fast_i16x2, bbox[w4][0]));
    hi = __builtin_elementwise_max(hi, __builtin_bit_cast(fast_i16x2, bbox[w4][1]));
  }
  const int lo_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)), hi_s = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi));
  const int bu0 = lo_s & 0xffff, bv0 = (lo_s >> 16) & 0xffff, bu1 = hi_s & 0xffff, bv1 = (hi_s >> 16) & 0xffff;
  FastWindow wnd;
  wnd.x0 = (bu0 - 1) & ~1; wnd.y0 = bv0 - 1;            // even: a 16-byte load holds two window cells
  const bool any = bu0 != 0x7fff;
  wnd.ww = any ? min(bu1 + 3 - wnd.x0, kFastCols) : 0;  // columns x0 .. umax + 2
  wnd.wh = any ? min(bv1 + 3 - wnd.y0, kFastRows) : 0;
  wnd.all_in = !any || (bu1 + 3 - wnd.x0 <= kFastCols && bv1 + 3 - wnd.y0 <= kFastRows);
  return wnd;
}

// phase B: thread t loads column pair t % 42 of rows t / 42, t / 42 + 6, ...  Cells outside the extent in use are not loaded at all.
__device__ __forceinline__ void fast_fill_window(const LevelGeom& g, __amdgpu_buffer_rsrc_t curC, float2* win, const FastWindow& wnd, int row_bytes) {
  const int t = threadIdx.x;
  const int rg = t / kFastPairs, cxp = t - rg * kFastPairs;
  f32x4* dst = reinterpret_cast<f32x4*>(win) + rg * (kFastPitch / 2) + cxp;
  const int x0 = wnd.x0, y0 = wnd.y0, wh = wnd.wh;
  const bool interior = x0 >= 0 && x0 + kFastCols <= g.w && y0 >= 0 && y0 + kFastRows <= g.h;   // (uniform)
  if (rg < kFastRowGroups && 2 * cxp < wnd.ww) {
    if (interior) {
      // no clamping anywhere: one vector offset per thread, the row group's offset is a scalar
      const int voff = (y0 + rg) * row_bytes + (x0 + 2 * cxp) * 8;
      // (the last round only has window rows for the first kFastRows - 24 row groups: the others ask past the end of the resource -- the
      // scalar row offset is not range-checked, and rows y0 + 28, y0 + 29 may lie below the plane)
      const int voff_last = rg + (kFastLoads - 1) * kFastRowGroups < kFastRows ? voff : 0x7ffffff0;
      f32x4 cell[kFastLoads];
#pragma unroll
      for (int j = 0; j < kFastLoads; ++j)
        if (j * kFastRowGroups < wh)                                                              // (uniform)
          cell[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(curC, j == kFastLoads - 1 ? voff_last : voff, j * kFastRowGroups * row_bytes, 0));
#pragma unroll
      for (int j = 0; j < kFastLoads; ++j)
        if (j * kFastRowGroups < wh && rg + j * kFastRowGroups < kFastRows) dst[j * kFastRowGroups * (kFastPitch / 2)] = cell[j];
    } else {
      // a window that reaches over the image border: coordinates clamped like the frame build's border code (rgbd_image.cpp:419-489),
      // so that the clamped central differences come out of the same subtraction.  Image width and window origin are even: a pair
      // lies entirely inside the image, entirely left of it (both cells = column 0) or entirely right of it (both = column w - 1).
      const int x = x0 + 2 * cxp;
      const int xl = min(max(x, 0), g.w - 2) * 8;
      f32x4 cell[kFastLoads];
#pragma unroll
      for (int j = 0; j < kFastLoads; ++j) {
        const int cy = rg + j * kFastRowGroups;
        const int y = min(max(y0 + cy, 0), g.h - 1);
        cell[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(curC, cy < wh ? y * row_bytes + xl : 0x7ffffff0, 0, 0));
      }
      if (x < 0) {
#pragma unroll
        for (int j = 0; j < kFastLoads; ++j) { cell[j].z = cell[j].x; cell[j].w = cell[j].y; }
      }
      if (x >= g.w) {
#pragma unroll
        for (int j = 0; j < kFastLoads; ++j) { cell[j].x = cell[j].z; cell[j].y = cell[j].w; }
      }
#pragma unroll
      for (int j = 0; j < kFastLoads; ++j)
        if (rg + j * kFastRowGroups < kFastRows) dst[j * kFastRowGroups * (kFastPitch / 2)] = cell[j];
    }
  }
}

// the twelve cells of a lane's 4 x 4 tap neighbourhood (corners left out).  CHECKED = false: every neighbourhood of the tile lies
// inside the window -- no per-lane test, no second source, and no vector-memory load whose counter the row would have to wait for.
// CHECKED: a tile whose projections spread beyond the window (a depth discontinuity under a large motion): lanes inside read the
// window, the others fetch their cells from memory, coordinates clamped like the window's -- correct for any motion.  The tap corner
// is not kept apart from the window index: it is projected again (tx, ty), the same instructions as in phase A.
template <bool CHECKED, int COMPAT>
__device__ __forceinline__ void fast_fetch_cells(const LevelGeom& g, const FastRcpSource& table, const float* KT, __amdgpu_buffer_rsrc_t curC, const float2* win,
                                                 const FastWindow& wnd, int neg_base, const FastRow& r, bool ok, float tx, float ty, f32x2 (&P)[4][4],
                                                 unsigned& n_fallback) {
  // (volatile: twelve ds_read_b64, two LDS cycles each; the compiler otherwise pairs them into ds_read2_b64, eight cycles a pair)
  if constexpr (!CHECKED) {
    const int addr = (r.idx << 3) + neg_base;
    FastLdsCellPtr q = (FastLdsCellPtr)(reinterpret_cast<const char*>(win) + (ok ? addr : 0));
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) P[rr][cc] = q[rr * kFastPitch + cc];
  } else {
    float pu, pv, pqz;
    fast_project<COMPAT>(g, table, KT, fmaf(KT[0], tx, KT[2]), fmaf(KT[4], tx, KT[6]), fmaf(KT[8], tx, KT[10]), r.z, tx, ty, pu, pv, pqz);
    const int u0 = int(pu), v0 = int(pv);
    const int cx = u0 - wnd.x0 - 1, cy = v0 - wnd.y0 - 1;
    const bool in_win = ok && unsigned(cx) <= unsigned(kFastCols - 4) && unsigned(cy) <= unsigned(kFastRows - 4);
    FastLdsCellPtr q = (FastLdsCellPtr)(win + (in_win ? cy * kFastPitch + cx : 0));
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        if ((rr == 1 || rr == 2) || (cc == 1 || cc == 2)) P[rr][cc] = q[rr * kFastPitch + cc];
    if (ok && !in_win) {
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
  }
}

// ---- the operand rows of a pixel row on the f16 matrix pipe -------------------------------------------------------------------------
// v = H + L (f16 high and low parts, gram_f16.h::split_pairs); G = H H^T + S + S^T with S = H L^T.  A 16-byte LDS store costs 8 array
// cycles whatever its active lanes (profiles/r03_pmc_utilisation.md), so all 64 lanes should store.  Two ways:
//   STORE 2 (variant 8, default): high parts and low parts take turns in the slab.  All 64 pixels' high blocks (32 B each: 2 KB) are
//     stored by their own lanes, read back transposed and KEPT as matrix operands (8 registers) for H H^T; then the low blocks
//     overwrite them and are read back for H L^T.  4 stores, 8 transposing reads, 4 matrix instructions per row, no cross-lane moves.
//     Layout: pixel p at 32 p bytes, nothing else; lane group g of a matrix operand takes pixels 4 g .. 4 g + 3 and 16 + 4 g .. 19 + 4 g of
//     its 32 (a transposing read covers 16 consecutive rows = 512 contiguous bytes).  Which pixel is which k index does not matter to a
//     sum over k as long as both operands agree.  (Measured, scripts/lds_conflicts.sh + scripts/ab_sweep.py: 16 bytes of padding behind
//     every eighth row, or the halves of every other 16 rows exchanged, give the same conflict count and the same time +-1 %.)
//   STORE 1 (variant 9): high and low blocks side by side (80-byte rows, gram_f16.h), 32 pixels at a time; v_permlane32_swap_b32 moves
//     half of every row to the idle half of the wavefront -- 8 swaps per row at 8.3 issue cycles each (scripts/ubench/issue_rate.hip).
// HI_J (round 5; levels of 150 000 pixels and more, LevelGeom::gram_hi_j): the twelve JACOBIAN components enter the matrix pipe as
// their f16 high parts alone, only the two residual components keep a low part -- 18 of a row's ~186 vector instructions less
// (finest level 2.041 -> 1.974 ms per 1024-pair launch, builds alternated on one box).  A component is then off by <= 2^-12 of itself,
// at random: the sums over N constraints by ~5 x 2^-12 / sqrt(N) -- measured against the oracle 1.5e-5 of |A| at N = 6 500 (a
// 160 x 120 level: NOT taken there), 2.7e-6 expected at the 190 000 constraints of a 640 x 480 level, where the measured distance to
// the oracle does not move (3.2e-5 / 2.1e-5 for A / b with one constraint flipped, either way); the residual components, whose
// products form b and the scale matrix, stay exact.
template <int STORE, bool HI_J>
__device__ __forceinline__ void fast_gram_row(float* my, int lane, const float (&comps)[14], f32x4& acc0, f32x4& acc1) {
  typedef volatile __attribute__((address_space(3))) u32x4* LdsQuadPtr;
  unsigned hh[7], ll[7];
  if constexpr (HI_J) {
    typedef float __attribute__((ext_vector_type(2))) f32pair;
    typedef _Float16 __attribute__((ext_vector_type(2))) f16pair;
#pragma unroll
    for (int k = 0; k < 7; ++k) hh[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32pair{comps[2 * k], comps[2 * k + 1]}, f16pair));
#pragma unroll
    for (int k = 0; k < 6; ++k) ll[k] = 0u;
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(comps[12]), "v"(hh[6]));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(comps[13]), "v"(hh[6]));
    ll[6] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32pair{ra, rb}, f16pair));
  } else {
    split_pairs(comps, hh, ll);
  }
  // the padding components (14, 15) may hold anything: they only reach rows / columns 14, 15 of the Gram matrix, which nobody reads
  unsigned pad;
  asm volatile("" : "=v"(pad));                           // (defined by nothing: no instruction, any register)
  if constexpr (STORE == 2) {
    char* base = reinterpret_cast<char*>(my);
    const int i = lane & 15, gq = lane >> 4;
    typedef __attribute__((address_space(3))) fp16x4* LdsTrPtr;
    LdsQuadPtr w0 = (LdsQuadPtr)(base + lane * 32);       // the lane's pixel: two 16-byte halves
    char* rd = base + gq * 128 + (i >> 2) * 32 + (i & 3) * 8;
    auto operand = [&](int block) {                       // pixels 32 block .. 32 block + 31 (k index) x 16 components
      const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LdsTrPtr)(rd + block * 1024));
      const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LdsTrPtr)(rd + block * 1024 + 512));
      return f16x8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    };
    w0[0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
    w0[1] = u32x4{hh[4], hh[5], hh[6], pad};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const f16x8 h0 = operand(0), h1 = operand(1);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, h0, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, h1, acc0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    w0[0] = u32x4{ll[0], ll[1], ll[2], ll[3]};
    w0[1] = u32x4{ll[4], ll[5], ll[6], pad};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const f16x8 l0 = operand(0), l1 = operand(1);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, l0, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, l1, acc1, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    const _Float16* img = reinterpret_cast<const _Float16*>(my);
    // E = components 0..7 (first 16 bytes of a pixel's hi / lo block), O = components 8..15 (second 16 bytes).  After the swaps
    // register set E holds, in lanes 0..31, E of pixel `lane` and, in lanes 32..63, O of pixel `lane - 32`: everything of the first
    // 32 pixels; set O likewise everything of pixels 32..63.
    unsigned hE[4] = {hh[0], hh[1], hh[2], hh[3]}, hO[4] = {hh[4], hh[5], hh[6], pad};
    unsigned lE[4] = {ll[0], ll[1], ll[2], ll[3]}, lO[4] = {ll[4], ll[5], ll[6], pad};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const fast_u32x2 a = __builtin_amdgcn_permlane32_swap(hE[j], hO[j], false, false);
      hE[j] = a[0]; hO[j] = a[1];
      const fast_u32x2 b = __builtin_amdgcn_permlane32_swap(lE[j], lO[j], false, false);
      lE[j] = b[0]; lO[j] = b[1];
    }
    LdsQuadPtr hw = (LdsQuadPtr)(reinterpret_cast<char*>(my) + (lane & 31) * (kHalfRow * 2) + (lane >> 5) * 16);
    hw[0] = u32x4{hE[0], hE[1], hE[2], hE[3]};
    hw[2] = u32x4{lE[0], lE[1], lE[2], lE[3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const f16x8 h = read_operand_f16(img, lane, 0), l = read_operand_f16(img + 16, lane, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    hw[0] = u32x4{hO[0], hO[1], hO[2], hO[3]};
    hw[2] = u32x4{lO[0], lO[1], lO[2], lO[3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      const f16x8 h = read_operand_f16(img, lane, 0), l = read_operand_f16(img + 16, lane, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// weights: sqrt(7 / (5 + r^T P r)) = c rsq(k + r^T P r); the first pass of a level (unit weights) rides the same code as 2 rsq(4 + 0)
// COMPAT (option "ref_compat"): w = 7 * _mm_rcp_ps(5 + r^T P r) with the host CPU's table (dense_tracking_impl.cpp:700), its square
// root through v_sqrt_f32; the first pass of a level has unit weights: w = 0 * rcp(5 + 0) + 1.
struct FastWeights {
  float P00, P11, P2x, wk, wc;
  float wm, wa;                                            // COMPAT: w = wm * rcp_host(arg) + wa
  __device__ __forceinline__ FastWeights(const float* P_prev, bool first) {   // PairState::P_prev, ::first
    P00 = first ? 0.0f : P_prev[0]; P11 = first ? 0.0f : P_prev[3]; P2x = first ? 0.0f : P_prev[1] + P_prev[2];
    wk = first ? 4.0f : 5.0f; wc = first ? 2.0f : 2.6457513110645906f;
    wm = first ? 0.0f : 7.0f; wa = first ? 1.0f : 0.0f;
  }
  __device__ __forceinline__ explicit FastWeights(const PairState& st) : FastWeights(st.P_prev, st.first != 0) {}
};

// Everything of a pixel row behind its twelve cells: blend, residual pair (stored for the log-likelihood pass: `resid` at vector offset
// off_v + scalar offset off_s), validity, weight, Jacobian at the untransformed point, Gram accumulation.  tx, ty, cx = 1 + tx^2:
// normalised coordinates of the lane's reference pixel.
// COMPACT (LevelGeom::compact): only a constraint's pair is stored, at the next free entry of the wavefront's slot (off_s: the slot).
// AUX: cache policy of the residual store (0; 16 = sc1, write-through: the pair's step runs in this very launch, solver_step.h)
template <int STORE, bool COMPACT, int COMPAT, bool HI_J, int AUX = 0>
__device__ __forceinline__ void fast_row_tail(const LevelGeom& g, const FastRcpSource& table, const f32x2 (&P)[4][4], const FastRow& r, unsigned long long ok_mask, float tx, float ty,
                                              float cx, const FastWeights& wt, __amdgpu_buffer_rsrc_t resid, int off_v, int off_s, float* my, int lane,
                                              f32x4& acc0, f32x4& acc1, int& n_valid) {
  // separable blend (rows first): E_j = row j of the neighbourhood at the tap's column position; intensity / depth, TWICE the
  // vertical and TWICE the horizontal central difference, each blended between rows 1 and 2.  A cell is an {I, Z} register pair and
  // both channels go through the same formula: packed f32 instructions (v_pk_add_f32 / v_pk_fma_f32), 24 instead of 38 instructions
  const f32x2 a1 = {r.a1, r.a1}, b1 = {r.b1, r.b1};
  const f32x2 e0 = fast_lerp2(P[0][1], P[0][2], a1), e1 = fast_lerp2(P[1][1], P[1][2], a1);
  const f32x2 e2 = fast_lerp2(P[2][1], P[2][2], a1), e3 = fast_lerp2(P[3][1], P[3][2], a1);
  const f32x2 cV = fast_lerp2(e1, e2, b1);
  const f32x2 cVy = fast_lerp2(e2 - e0, e3 - e1, b1);
  const f32x2 d1 = fast_lerp2(P[1][2] - P[1][0], P[1][3] - P[1][1], a1);
  const f32x2 d2 = fast_lerp2(P[2][2] - P[2][0], P[2][3] - P[2][1], a1);
  const f32x2 cVx = fast_lerp2(d1, d2, b1);
  const float cI = cV.x, cZ = cV.y, cIx = cVx.x, cZx = cVx.y, cIy = cVy.x, cZy = cVy.y;
  const float r0 = (cI - r.i) * (1.0f / 255.0f);             // dense_tracking.cpp:217-220
  const float r1 = cZ - r.qz;                                // reference depth := transformed z (dense_tracking_impl.cpp:269)
  const float dz = r.z - 0.4f;                               // occlusion threshold -20 (0.0012 + 0.0019 (z - 0.4)^2) (:122-128, Q5)
  const float thr = fmaf(dz * -0.038f, dz, -0.024f);
  // Q9: a hole in any of the twelve cells makes cZ (hence r1), cZx or cZy not-a-number; intensities are never holes
  const unsigned long long valid_mask = ok_mask & __builtin_amdgcn_ballot_w64(r1 > thr) & __builtin_amdgcn_ballot_w64(!__builtin_isunordered(cZx, cZy));
  const bool valid = __builtin_amdgcn_inverse_ballot_w64(valid_mask);
  if constexpr (COMPACT) {
    // a constraint's place: the wavefront's constraints so far (scalar) + those in the lanes below; the others store past the resource
    const int below = __builtin_amdgcn_mbcnt_hi(unsigned(valid_mask >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(valid_mask), 0));
    const f32x2 rr2 = {r0, r1};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fast_u32v2, rr2), resid, valid ? below * 8 : 0x7ffffff8, off_s + n_valid * 8, AUX);
  } else {
    const f32x2 rr2 = {valid ? r0 : __builtin_nanf(""), r1};  // (the log-likelihood pass tests the first component)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fast_u32v2, rr2), resid, off_v, off_s, AUX);
  }
  n_valid += __popcll(valid_mask);
  const float tq = fmaf(wt.P2x, r1, wt.P00 * r0);
  float sw_any;
  if constexpr (COMPAT != 0) {
    // (a lane without a constraint may hold anything in r0 / r1: the table index is masked to the table by construction, its weight is dropped)
    const float arg = fmaf(tq, r0, fmaf(wt.P11 * r1, r1, 5.0f));
    sw_any = __builtin_amdgcn_sqrtf(fmaf(wt.wm, fast_rcp_host<COMPAT>(table, arg), wt.wa));
  } else {
    sw_any = wt.wc * fast_rsqrt(fmaf(tq, r0, fmaf(wt.P11 * r1, r1, wt.wk)));
  }
  const float sw = valid ? sw_any : 0.0f;
  // gradient rows scaled by sqrt(w); a lane without a constraint has sw = 0 and every product in which one of its NaN terms meets
  // that zero is a LEGACY multiply (0 x anything = 0): no control flow on validity (an unfilled window cell may hold anything)
  const float gix = mul_legacy(sw * g.half_wi_x, cIx + r.gx), giy = mul_legacy(sw * g.half_wi_y, cIy + r.gy);
  const float gzx = mul_legacy(sw * g.half_fx, cZx), gzy = mul_legacy(sw * g.half_fy, cZy);
  const float iz = rcp_for_inline_asm(r.z);
  const float txy = tx * ty, cy = fmaf(ty, ty, 1.0f);
  const float sz = mul_legacy(sw, r.z);
  float c[14];
  c[0] = mul_legacy(gix, iz);
  c[1] = mul_legacy(giy, iz);
  c[2] = fmaf(-ty, c[1], -tx * c[0]);
  c[3] = fmaf(-giy, cy, -gix * txy);
  c[4] = fmaf(gix, cx, giy * txy);
  c[5] = fmaf(giy, tx, -gix * ty);
  c[6] = mul_legacy(gzx, iz);
  c[7] = mul_legacy(gzy, iz);
  c[8] = fmaf(-ty, c[7], fmaf(-tx, c[6], -sw));
  c[9] = fmaf(-ty, fmaf(gzx, tx, sz), -gzy * cy);            // -(gzy cy + gzx tx ty + sw y),  y = ty z
  c[10] = fmaf(tx, fmaf(gzy, ty, sz), gzx * cx);             //   gzx cx + gzy tx ty + sw x,   x = tx z
  c[11] = fmaf(gzy, tx, -gzx * ty);
  const float sr = sw * kResidualScale;
  c[12] = mul_legacy(sr, r0);
  c[13] = mul_legacy(sr, r1);
  fast_gram_row<STORE, HI_J>(my, lane, c, acc0, acc1);
}

// epilogue: G = H H^T + S + S^T summed over the four wavefronts by the 85 threads that own an accumulator (slab[w]: H H^T at [0, 256),
// S at [256, 512), entry row * 16 + col)
// WT: the row is stored write-through (agent-scope stores, sc1) -- see fast_row_tail
template <bool WT = false>
__device__ __forceinline__ void fast_epilogue(float (*slab)[kSlabFloatsF16], const int* counts, unsigned gram_entries, float* __restrict__ out_row,
                                              int* __restrict__ f16_range_flag) {
  auto put = [&](int i, float v) {
    if constexpr (WT) __hip_atomic_store(out_row + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out_row[i] = v;
  };
  const int kk = threadIdx.x;
  if (kk < kNumAcc) {
    auto G = [&](int e) {                                      // entry e of the tile's Gram matrix, residual scale removed
      const int et = (e & 15) * 16 + (e >> 4);
      float t = 0.0f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) t += (slab[w4][e] + slab[w4][256 + e]) + slab[w4][256 + et];
      const float s = ((e & 15) >= 12 ? 1.0f / kResidualScale : 1.0f) * ((e >> 4) >= 12 ? 1.0f / kResidualScale : 1.0f);
      return t * s;
    };
    float v;
    if (kk == kAccN) {
      v = float((counts[0] + counts[1]) + (counts[2] + counts[3]));
    } else {
      const int e1 = gram_entries & 0xff, e2 = gram_entries >> 8;
      v = G(e1);
      if (e2 != 0xff) v += G(e2);
      // a component beyond the f16 range became infinite in its high part (v_cvt_pk_f16_f32 rounds to nearest: overflow = infinity)
      // and every sum it enters is infinite or not-a-number: the caller repeats the work with the f32 Gram
      if (f16_range_flag && !(__builtin_fabsf(v) < __builtin_inff())) __hip_atomic_store(f16_range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    put(kk, v);
  } else if (kk < kNumAcc + 2) {
    // the wavefronts' counts, two per spare float of the row: where their packed residual pairs end (LevelGeom::compact)
    const int q = (kk - kNumAcc) * 2;
    put(kAccCounts + (kk - kNumAcc), float(counts[q] + 512 * counts[q + 1]));
  }
}

__device__ __forceinline__ void fast_count_fallbacks(unsigned long long* __restrict__ fallback_count, unsigned n_fallback, int lane) {
  const unsigned long long lanes = __ballot(n_fallback != 0);
  if (lanes) {
    unsigned t = n_fallback;
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    if (lane == 0) atomicAdd(fallback_count, (unsigned long long)t);
  }
}

// ===================================================================================================================================
// One 64 x 16 tile of one pair at one level: everything of k_sweep_fast behind "which pair, which tile, is the pair on this level".
// Shared by the launch path's kernel (align_fast.hip: one workgroup per tile and launch) and the fused coarse-level kernel
// (align_coarse.hip: one workgroup walks all tiles of its pair, iteration after iteration) -- the same instructions, hence the same bits.
// KT: float(K * estimate), uniform (scalar registers: PairState::KT in memory for the launch path, read out of LDS for the fused kernel);
// lds: the workgroup's window, operand slabs, bounding boxes and counts; the caller separates two tiles that use the same LDS by nothing:
// every reuse of a buffer lies behind one of the tile's own three barriers (phase A | window fill | phase C | epilogue).
// ===================================================================================================================================
struct FastLds {
  float (*slab)[kSlabFloatsF16];                        // [4]: one operand slab per wavefront
  float2* win;                                          // kFastCells cells
  int (*bbox)[2];                                       // [4][2]
  int* counts;                                          // [4]
};

template <int STORE, bool PARTIAL, bool COMPACT, int COMPAT, bool HI_J, bool WT = false>
__device__ __forceinline__ void fast_sweep_tile(const LevelGeom& g, const float* KT, const FastWeights& wt, const PairPtrs& pp, int pair, int tile,
                                                float* __restrict__ partials, float2* __restrict__ scratch, const FastLds& lds,
                                                const FastRcpSource& rcp_table, unsigned long long* __restrict__ fallback_count,
                                                int* __restrict__ f16_range_flag) {
  constexpr int RPW = kFastTileRows / 4;
  const int tiles = g.tiles_x * g.tiles_y;
  const int tile_x = tile % g.tiles_x, tile_y = tile / g.tiles_x;
  const int plane_bytes = g.w * g.h * 8;
  const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t curC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.curC), 0, plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t resid = COMPACT ? __builtin_amdgcn_make_buffer_rsrc(scratch + size_t(pair) * tiles * kCompactTileEntries, 0, tiles * kCompactTileEntries * 8, 0x00020000)
                                               : __builtin_amdgcn_make_buffer_rsrc(scratch + size_t(pair) * size_t(g.w) * g.h, 0, plane_bytes, 0x00020000);
  // (rows below the image -- a level whose height is no multiple of 16 -- go to a resource of no bytes: no branch around the store)
  const __amdgpu_buffer_rsrc_t resid_none = __builtin_amdgcn_make_buffer_rsrc(scratch, 0, 0, 0x00020000);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned gram_entries = kGramEntryTable.e[min(int(threadIdx.x), kNumAcc - 1)];
  // A level whose width is no multiple of 64 (160 x 120): the last tile column hangs over the right edge.  Its lanes beyond the image
  // read the last column again (the clamp the horizontal gradient wants there anyway), never count as projected (col_mask) and store
  // their residual past the end of the buffer resource, where stores are dropped.
  // (PARTIAL: an instantiation of its own -- widths that are multiples of 64 run the kernel without these three operations)
  const int u_r = PARTIAL ? min(tile_x * kTileW + lane, g.w - 1) : tile_x * kTileW + lane;
  const unsigned long long col_mask = PARTIAL ? __builtin_amdgcn_ballot_w64(tile_x * kTileW + lane < g.w) : ~0ull;
  const int row_bytes = g.w * 8;
  const float tx_u = g.tx[u_r];
  const float cx_u = fmaf(tx_u, tx_u, 1.0f);

  float (*slab)[kSlabFloatsF16] = lds.slab;
  float2* win = lds.win;
  int (*bbox)[2] = lds.bbox;
  int* counts = lds.counts;
  float* my = slab[wave];
  const int off_px = u_r * 8;
  const int off_store = !PARTIAL || tile_x * kTileW + lane < g.w ? off_px : plane_bytes;
  const int off_edge = (lane == 0 ? max(u_r - 1, 0) : lane == 63 ? min(u_r + 1, g.w - 1) : u_r) * 8 + 4;

  const int row0 = tile_y * kFastTileRows + wave;
  float ty_rows[RPW];
#pragma unroll
  for (int k = 0; k < RPW; ++k) ty_rows[k] = g.ty[min(row0 + k * 4, g.h - 1)];

  // ---- phase A: reference rows, projection, tap bounding box -------------------------------------------------------------------
  FastRow rs[RPW];
  unsigned long long ok_row[RPW];                              // lane masks, in scalar registers: validity never visits a vector register
  int umin = 0x7fff, vmin = 0x7fff, umax = 0, vmax = 0;        // over the lanes with a projection
  {
    // q = z (KT.col0 tx + KT.col1 ty + KT.col2) + KT.col3: the column's part of the bracket once per tile
    const float c0 = fmaf(KT[0], tx_u, KT[2]), c1 = fmaf(KT[4], tx_u, KT[6]), c2 = fmaf(KT[8], tx_u, KT[10]);
    const unsigned w2_bits = __builtin_bit_cast(unsigned, float(g.w - 2)), h2_bits = __builtin_bit_cast(unsigned, float(g.h - 2));
    float zv[RPW], iv[RPW], up[RPW], down[RPW], edge[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {                            // all loads of the four rows first: one round trip
      const int v = min(row0 + k * 4, g.h - 1);
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
      const int v_r = row0 + k * 4;
      const int ic = __builtin_bit_cast(int, iv[k]), ie = __builtin_bit_cast(int, edge[k]);
      const float right = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x130, 0xf, 0xf, false));   // wave_shl:1
      const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ie, ic, 0x138, 0xf, 0xf, false));    // wave_shr:1
      const float z = zv[k];
      float u, v, qz;
      fast_project<COMPAT>(g, rcp_table, KT, c0, c1, c2, z, tx_u, ty_rows[k], u, v, qz);
      // 0 <= u <= w - 2 on the integer image of the float: negative numbers and NaNs (sign or exponent bits) compare above every
      // non-negative bound (Q4; a hole's NaN depth fails here, Q19).  -0.0f fails too -- one float out of 2^32.
      const unsigned long long ok_mask = v_r < g.h ? __builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned, u) <= w2_bits) &
                                                         __builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned, v) <= h2_bits) & col_mask : 0ull;
      const bool ok = __builtin_amdgcn_inverse_ballot_w64(ok_mask);
      const int u0 = int(u), v0 = int(v);                      // (u, v >= 0 where it matters: truncation is the floor)
      rs[k].z = z;
      rs[k].i = iv[k];
      rs[k].gx = right - left;
      rs[k].gy = down[k] - up[k];
      rs[k].qz = qz;
      rs[k].a1 = __builtin_amdgcn_fractf(u);
      rs[k].b1 = __builtin_amdgcn_fractf(v);
      rs[k].idx = int(__umul24(v0, kFastPitch)) + u0;
      ok_row[k] = ok_mask;
      if (ok) {
        umin = min(umin, u0); umax = max(umax, u0);
        vmin = min(vmin, v0); vmax = max(vmax, v0);
      }
    }
  }
  fast_wave_min2max2_lane63(umin, vmin, umax, vmax);
  if (lane == 63) {
    bbox[wave][0] = umin | (vmin << 16); bbox[wave][1] = umax | (vmax << 16);
  }
  __syncthreads();
  const FastWindow wnd = fast_window_of(bbox);

  // ---- phase B: the window into LDS ----------------------------------------------------------------------------------------------
  fast_fill_window(g, curC, win, wnd, row_bytes);
  __syncthreads();

  // ---- phase C: taps from LDS, residual, weight, Jacobian, Gram accumulation ----------------------------------------------------
  const int neg_base = -8 * ((wnd.x0 + 1) + kFastPitch * (wnd.y0 + 1));   // LDS byte address of a lane's neighbourhood = 8 idx + neg_base
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;
  unsigned n_fallback = 0;
  auto sweep_row = [&](int k, auto checked_tag) __attribute__((always_inline)) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
    const int v_r = row0 + k * 4;
    f32x2 P[4][4];
    fast_fetch_cells<CHECKED, COMPAT>(g, rcp_table, KT, curC, win, wnd, neg_base, rs[k], __builtin_amdgcn_inverse_ballot_w64(ok_row[k]), tx_u, ty_rows[k], P, n_fallback);
    if constexpr (COMPACT)
      fast_row_tail<STORE, true, COMPAT, HI_J, WT ? 16 : 0>(g, rcp_table, P, rs[k], ok_row[k], tx_u, ty_rows[k], cx_u, wt, resid, 0, (tile * kCompactTileEntries + wave * kCompactWaveEntries) * 8, my, lane, acc0, acc1, n_valid);
    else
      fast_row_tail<STORE, false, COMPAT, HI_J, WT ? 16 : 0>(g, rcp_table, P, rs[k], ok_row[k], tx_u, ty_rows[k], cx_u, wt, v_r < g.h ? resid : resid_none, off_store, v_r * row_bytes, my, lane, acc0, acc1, n_valid);
  };
  if (wnd.all_in) {                                            // (uniform)
#pragma unroll
    for (int k = 0; k < RPW; ++k) sweep_row(k, std::false_type{});
  } else {
#pragma unroll
    for (int k = 0; k < RPW; ++k) sweep_row(k, std::true_type{});
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i];
    my[256 + ((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc1[i];
  }
  if (lane == 0) counts[wave] = n_valid;
  __syncthreads();
  fast_epilogue<WT>(slab, counts, gram_entries, partials + (size_t(pair) * tiles + tile) * kAccStride, f16_range_flag ? f16_range_flag + pair : nullptr);   // (one word per pair)
  if (fallback_count && !wnd.all_in) fast_count_fallbacks(fallback_count, n_fallback, lane);
}

}  // namespace dvo_hip
