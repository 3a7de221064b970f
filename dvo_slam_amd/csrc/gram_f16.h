// gram_f16.h -- the normal-equation accumulation of the sweeps on the f16 matrix pipe (schedule variant 7): device code shared by the
// window sweep (align_window.hip) and the gathering sweep (align_mfma.hip).
//
// The Gram matrix G = V^T V of the per-pixel vectors v = sqrt(w) [J0(6), J1(6), r0, r1] (least_squares.cpp:58-64 contracts exactly
// this with the 2x2 precision) is accumulated as  H H^T + H L^T + (H L^T)^T  with v = H + L split into f16 high and low parts: two
// v_mfma_f32_16x16x32_f16 per 32 pixels that run beside the vector ALU, where the f32 form (16 v_mfma_f32_16x16x4_f32 per 64 pixels)
// takes the vector ALU's issue slots.  The dropped L L^T term is 2^-22 of the result.
#pragma once
#include "sweep_parts.h"

namespace dvo_hip {

// The operand slab of a wavefront.  f32 Gram: the layout of sweep_parts.h (64 pixels, 4224 B).  f16 Gram: one 80-byte row per
// pixel -- 16 halfs "hi", 16 halfs "lo", 16 B of padding (conflict-free 16-byte stores: 80 i mod 128 are eight distinct 16-byte
// slots) -- for 32 pixels = one matrix instruction's worth (K = 32): the two half rows of a wavefront's row take turns (2560 B per
// wavefront instead of 5120: with the window that is 29.4 KB per workgroup, five workgroups per compute unit instead of four).
constexpr int kHalfRow = 40;                            // halfs per pixel row
constexpr int kSlabFloatsF16 = 32 * kHalfRow / 2;       // 640 floats = 2560 B (>= the 512 floats the Gram matrices need at the end)
constexpr float kResidualScale = 256.0f;                // the two residual components are lifted out of the f16 subnormal range

typedef _Float16 __attribute__((ext_vector_type(8))) f16x8;
typedef __fp16 __attribute__((__vector_size__(4 * sizeof(__fp16)))) fp16x4;
typedef __fp16 __attribute__((ext_vector_type(2))) fp16x2;
typedef unsigned __attribute__((ext_vector_type(4))) u32x4;

// v = hi + lo with hi, lo in f16; two components per call, packed as the two halves of a register.  hi is rounded toward zero (it
// saturates at the largest f16 instead of becoming infinite), lo = f16(v - hi) with the subtraction exact in f32:
// v_fma_mix_f32 reads the f16 half of `hi` directly and subtracts in f32 (the compiler emits a conversion and a subtraction
// unless it can fold a multiplication in); the two differences are packed with round toward zero as well.
// Range: a component beyond +-65504 has an INFINITE high part (round to nearest) and the sum no longer represents it; the epilogue
// finds the infinity (or the not-a-number it turns into) in H H^T and the batch is repeated with the f32 Gram (capi.hip, counter
// "f16_range_repeats").
// (seven pairs at a time, each step for all pairs before the next: an instruction never waits for the one right before it)
__device__ __forceinline__ void split_pairs(const float (&v)[14], unsigned (&hi)[7], unsigned (&lo)[7]) {
  // (round to nearest even -- v_cvt_pk_f16_f32, gfx950 -- for both parts: |v - hi - lo| <= 2^-22 |v|, four times closer than with
  // round toward zero, which matters where the stopping precision of a level is near the noise of the normal equations)
  typedef float __attribute__((ext_vector_type(2))) f32pair;
  typedef _Float16 __attribute__((ext_vector_type(2))) f16pair;
#pragma unroll
  for (int k = 0; k < 7; ++k) hi[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32pair{v[2 * k], v[2 * k + 1]}, f16pair));
  // (measured: v_fma_mixlo_f16 / v_fma_mixhi_f16 would write the rounded low parts straight into the halves of one register -- 14
  // instructions instead of 21 -- but the sweep runs 1.7 % SLOWER with them than with v_fma_mix_f32 + v_cvt_pkrtz: scripts/ab_sweep.py)
  float ra[7], rb[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra[k]) : "v"(v[2 * k]), "v"(hi[k]));
#pragma unroll
  for (int k = 0; k < 7; ++k) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb[k]) : "v"(v[2 * k + 1]), "v"(hi[k]));
#pragma unroll
  for (int k = 0; k < 7; ++k) lo[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32pair{ra[k], rb[k]}, f16pair));
}

// matrix operand of 32 pixels (k index) x 16 components from the pixel-major f16 image: lane (i = l & 15, g = l >> 4) gets eight
// k-values of component i (scripts/ubench/gram_f16.hip).  WHICH eight pixels is free -- the Gram sum runs over all of them, and both
// operands of a product are read with the same pattern: lane group g takes the pixel rows 8 g .. 8 g + 7 as (even rows, odd rows)
// rather than (first four, last four).  With 80-byte rows the eight 32-byte chunks a 32-lane group reads per instruction then tile the
// 64 banks exactly (rows 0, 2, 4, 6 and 8, 10, 12, 14 start at banks 0, 40, 16, 56 and 32, 8, 48, 24); rows 0..3 and 8..11 did not
// (the fourth row's chunk wraps onto the first's: SQ_LDS_BANK_CONFLICT, profiles/r03_pmc_utilisation.md).
__device__ __forceinline__ f16x8 read_operand_f16(const _Float16* img, int lane, int chunk32) {
  const int i = lane & 15, gq = lane >> 4;
  const _Float16* p0 = img + (chunk32 * 32 + 8 * gq + 2 * (i >> 2)) * kHalfRow + (i & 3) * 4;
  const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p0);
  const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(p0 + kHalfRow));
  return f16x8{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3], (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
}

// v_mul_legacy_f32: the multiply with 0 x anything = 0 (NaN and infinity included); the ordinary product otherwise.  (Inline assembly:
// this compiler has no builtin for it.)
__device__ __forceinline__ float mul_legacy(float a, float b) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// A reciprocal whose first consumer is inline assembly (mul_legacy above).  On gfx950 an instruction that reads the result of a
// transcendental instruction (v_rcp_f32, v_rsq_f32) needs a wait state between the two; the compiler inserts it for the instructions
// it knows, but it does not look into inline assembly -- and whenever its scheduler happened to place the reciprocal of the depth
// right in front of the first legacy product, that product read the register's OLD value (round 4: rows / columns 0 and 2 of the
// normal equations off by factors of 20 to 1000 in one build of a kernel and exact in its sibling; scripts/ubench/trans_hazard.hip
// reproduces it in six lines).  This is round 3's "wrong sums in the gathering sweep for a reason not found" (12-byte slab stores):
// that change moved the same two instructions together.  The wait states travel with the value here.
__device__ __forceinline__ float rcp_for_inline_asm(float x) {
  float r = fast_rcp(x);
  asm volatile("s_nop 1" : "+v"(r));
  return r;
}

// jacobian_rows_fast (pixel_math.h) for a lane that may be invalid: s = 0 there, and every product in which a term of that lane (NaN
// where the reference point or a tap is a hole) meets a zero is v_mul_legacy_f32 -- 0 x anything = 0.  For a valid lane the legacy
// multiply is the ordinary one.
__device__ __forceinline__ void jacobian_rows_masked(const PixelTerms& t, float s, float tx, float ty, float cx, float cy, float* J0, float* J1) {
  const float iz = rcp_for_inline_asm(t.Z);
  const float txy = tx * ty;
  const float gix = mul_legacy(s, t.gix), giy = mul_legacy(s, t.giy);
  const float gzx = mul_legacy(s, t.gzx), gzy = mul_legacy(s, t.gzy);
  J0[0] = mul_legacy(gix, iz);
  J0[1] = mul_legacy(giy, iz);
  J0[2] = fmaf(-ty, J0[1], -tx * J0[0]);
  J0[3] = fmaf(-giy, cy, -gix * txy);
  J0[4] = fmaf(gix, cx, giy * txy);
  J0[5] = fmaf(giy, tx, -gix * ty);
  J1[0] = mul_legacy(gzx, iz);
  J1[1] = mul_legacy(gzy, iz);
  J1[2] = fmaf(-ty, J1[1], fmaf(-tx, J1[0], -s));
  J1[3] = fmaf(-gzy, cy, fmaf(-gzx, txy, -mul_legacy(s, t.Y)));
  J1[4] = fmaf(gzx, cx, fmaf(gzy, txy, mul_legacy(s, t.X)));
  J1[5] = fmaf(gzy, tx, -gzx * ty);
}

// One pixel row of a wavefront: lane = pixel.  `sw` = sqrt of the pixel's weight, ZERO for a lane without a constraint (its terms may
// then hold NaN: see jacobian_rows_masked).  `my`: the wavefront's private slab (kSlabFloatsF16 floats).
__device__ __forceinline__ void gram_f16_row(float* my, int lane_c, const PixelTerms& o, float sw, float tx_u, float ty_p, float cx_u,
                                             f32x4& acc0, f32x4& acc1) {
      typedef volatile __attribute__((address_space(3))) u32x4* LdsQuadPtr;   // volatile: the stores stay where they are written
      LdsQuadPtr hw = (LdsQuadPtr)(reinterpret_cast<char*>(my) + (lane_c & 31) * (kHalfRow * 2));
      const bool low_half = lane_c < 32;
      float J0[6], J1[6];
      jacobian_rows_masked(o, sw, tx_u, ty_p, cx_u, fmaf(ty_p, ty_p, 1.0f), J0, J1);
      const float sr = sw * kResidualScale;
      unsigned hh[7], ll[7];
      const float comps[14] = {J0[0], J0[1], J0[2], J0[3], J0[4], J0[5], J1[0], J1[1], J1[2], J1[3], J1[4], J1[5], mul_legacy(sr, o.r0), mul_legacy(sr, o.r1)};
      split_pairs(comps, hh, ll);
      // (measured and dropped: 12-byte stores for the second quads over padding zeroed once per tile -- two register moves per row
      // less, -0.3 % in the window sweep, and WRONG sums in the gathering sweep on the device for a reason not found)
      const u32x4 h0 = {hh[0], hh[1], hh[2], hh[3]}, h1 = {hh[4], hh[5], hh[6], 0u};
      const u32x4 l0 = {ll[0], ll[1], ll[2], ll[3]}, l1 = {ll[4], ll[5], ll[6], 0u};
      if (low_half) { hw[0] = h0; hw[1] = h1; hw[2] = l0; hw[3] = l1; }
      const _Float16* img = reinterpret_cast<const _Float16*>(my);
      // pixels 0..31 of the row: the slab is private to the wavefront and LDS executes a wavefront's operations in order, so the
      // reads below follow the stores above, and the stores of the second half follow the reads -- only the compiler is fenced
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const f16x8 h = read_operand_f16(img, lane_c, 0), l = read_operand_f16(img + 16, lane_c, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc0, 0, 0, 0);      // H H^T
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, acc1, 0, 0, 0);      // S = H L^T ; G = H H^T + S + S^T
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (!low_half) { hw[0] = h0; hw[1] = h1; hw[2] = l0; hw[3] = l1; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        const f16x8 h = read_operand_f16(img, lane_c, 0), l = read_operand_f16(img + 16, lane_c, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, l, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
}

// The wavefront's Gram matrix, entry (row, col) at my[row * 16 + col] (the layout the f32 path leaves), from the two accumulators:
// G = H H^T + S + S^T, entry (row, col) of lane l, register i: row = (l >> 4) * 4 + i, col = l & 15; S^T through the slab.  The two
// residual components (12, 13) carry the factor kResidualScale.  f16_range_flag (may be null): see split_pairs.
__device__ __forceinline__ void gram_f16_finish(float* my, int lane, const f32x4& acc0, const f32x4& acc1, int* f16_range_flag) {
#pragma unroll
    for (int i = 0; i < 4; ++i) my[256 + ((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc1[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = lane & 15;
    if (f16_range_flag) {
      // a component beyond the f16 range became INFINITE in its high part (v_cvt_pk_f16_f32 rounds to nearest: overflow = infinity)
      // and puts infinity on the diagonal of H H^T, from where it spreads as infinity or not-a-number.  (Rounds 2-3 compared the
      // accumulated diagonal with one pixel's 65504^2 -- many moderately large components tripped it although nothing saturated.)
      const float largest = fmaxf(fmaxf(fabsf(acc0[0]), fabsf(acc0[1])), fmaxf(fabsf(acc0[2]), fabsf(acc0[3])));
      const bool out_of_range = !(largest < __builtin_inff()) || __builtin_isnan(acc0[0]) || __builtin_isnan(acc0[1]) || __builtin_isnan(acc0[2]) || __builtin_isnan(acc0[3]);
      if (__ballot(out_of_range) != 0 && lane == 0) __hip_atomic_store(f16_range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float cs = col >= 12 ? 1.0f / kResidualScale : 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lane >> 4) * 4 + i;
      const float rsc = row >= 12 ? 1.0f / kResidualScale : 1.0f;
      const float st = my[256 + col * 16 + row];
      my[row * 16 + col] = ((acc0[i] + acc1[i]) + st) * (cs * rsc);
    }
}

}  // namespace dvo_hip
