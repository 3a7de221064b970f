// align_common.h -- device helpers shared by the alignment kernels: global-address-space loaders and the
// DPP wavefront reduction.
#pragma once

#include "launch.h"
#include "pixel_math.h"
#include "reduce_scale.h"

namespace dvo_hip {

// Pointers loaded from a table in memory are generic to the compiler; these loaders re-type them as
// global (address space 1) so the taps compile to global_load_dwordx4 / dwordx2 instead of flat loads.
typedef float __attribute__((ext_vector_type(4))) vec4f;
typedef float __attribute__((ext_vector_type(2))) vec2f;
typedef const __attribute__((address_space(1))) vec4f* GlobalVec4;
typedef const __attribute__((address_space(1))) vec2f* GlobalVec2;

struct GlobalLoad4 {
  GlobalVec4 p;
  __device__ __forceinline__ float4 operator[](int i) const {
    const vec4f v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
  }
};
struct GlobalLoad2 {
  GlobalVec2 p;
  __device__ __forceinline__ float2 operator[](int i) const {
    const vec2f v = p[i];
    return make_float2(v.x, v.y);
  }
};

// ---- wavefront reduction of all accumulators with DPP (result valid in lane 63) -------------------------
// Six DPP stages: row_shr 1,2,4,8 leave each row-of-16 sum in its lane 15; row_bcast:15 folds rows
// 0->1 and 2->3; row_bcast:31 folds lane 31 into rows 2,3.  Written as inline assembly, five
// independent accumulators per statement, because (a) hipcc otherwise SLP-packs the adds of adjacent
// accumulators into v_pk_add_f32, which cannot carry a DPP modifier and costs three moves per add, and
// (b) a stage applied to five different registers needs one s_nop (VALU write -> DPP read hazard, 2
// wait states) per statement instead of one per add.
#define DVO_DPP5(ctrl)                                    \
  "s_nop 1\n\t"                                           \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n\t"                 \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n\t"                 \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n\t"                 \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n\t"                 \
  "v_add_f32_dpp %4, %4, %4 " ctrl

template <int STAGE>
__device__ __forceinline__ void dpp_stage5(float& a, float& b, float& c, float& d, float& e) {
  if constexpr (STAGE == 0)
    asm volatile(DVO_DPP5("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  else if constexpr (STAGE == 1)
    asm volatile(DVO_DPP5("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  else if constexpr (STAGE == 2)
    asm volatile(DVO_DPP5("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  else if constexpr (STAGE == 3)
    asm volatile(DVO_DPP5("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  else if constexpr (STAGE == 4)
    asm volatile(DVO_DPP5("row_bcast:15 row_mask:0xa bank_mask:0xf") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  else
    asm volatile(DVO_DPP5("row_bcast:31 row_mask:0xc bank_mask:0xf") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}

template <int STAGE, int N>
__device__ __forceinline__ void dpp_stage_all(float* acc) {
  static_assert(N % 5 == 0, "accumulators are reduced five at a time (pad the array)");
#pragma unroll
  for (int i = 0; i < N; i += 5) dpp_stage5<STAGE>(acc[i], acc[i + 1], acc[i + 2], acc[i + 3], acc[i + 4]);
}

// sums of N per-lane accumulators over the wavefront; valid in lane 63
template <int N>
__device__ __forceinline__ void wave_sum_all_to_lane63(float* acc) {
  dpp_stage_all<0, N>(acc);
  dpp_stage_all<1, N>(acc);
  dpp_stage_all<2, N>(acc);
  dpp_stage_all<3, N>(acc);
  dpp_stage_all<4, N>(acc);
  dpp_stage_all<5, N>(acc);
}

__device__ __forceinline__ double wave_sum_double(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}


// sum over a range of stored residual pairs of log(1 + 0.2 r^T P r) (dense_tracking_impl.cpp:413-422), two pixels per
// 16-B load.  The reference multiplies 50 terms between logs; here a lane multiplies ALL its terms and takes ONE log:
// after every eight factors the running product is renormalised with frexp (two instructions) and the exponent is
// summed separately, so the product can neither overflow (eight factors of up to 1e38 stay inside float64) nor lose precision.  `first_chunk`/`chunk_stride` in units
// of LOADS x blockDim pixel pairs; LOADS = 16-B loads in flight per lane between renormalisations (frexp only rescales by
// a power of two, so the value does not depend on LOADS -- only the latency hiding does).
template <int LOADS>
__device__ __forceinline__ double loglik_partial(const float2* __restrict__ res, int npx, const float* P, int first_chunk, int chunk_stride) {
  typedef const __attribute__((address_space(1))) vec4f* G4;
  const G4 r = (G4)res;
  const int npair2 = npx >> 1;
  double prod = 1.0;
  int exponent = 0;
  for (int base = first_chunk * kBlock * LOADS; base < npair2; base += chunk_stride * kBlock * LOADS) {
    vec4f rr[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      const int i = base + k * kBlock + threadIdx.x;
      const float nanv = __builtin_nanf("");
      const vec4f v = r[i < npair2 ? i : npair2 - 1];         // unconditional load, selected afterwards (no branch per load)
      rr[k] = i < npair2 ? v : vec4f{nanv, nanv, nanv, nanv};
    }
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      if (rr[k].x == rr[k].x) prod *= 1.0 + 0.2 * double(mahalanobis(rr[k].x, rr[k].y, P));
      if (rr[k].z == rr[k].z) prod *= 1.0 + 0.2 * double(mahalanobis(rr[k].z, rr[k].w, P));
      if ((k & 3) == 3 || k == LOADS - 1) {                  // eight factors at most between renormalisations: with a near-
        int e;                                                // singular scale (noise-free input) one factor can reach 1e13
        prod = frexp(prod, &e);
        exponent += e;
      }
    }
  }
  if ((npx & 1) && first_chunk == 0 && threadIdx.x == 0) {   // odd pixel count: last pixel
    const float2 rr = res[npx - 1];
    if (rr.x == rr.x) prod *= 1.0 + 0.2 * double(mahalanobis(rr.x, rr.y, P));
  }
  return log(prod) + double(exponent) * 0.6931471805599453094;
}

// The same sum over the PACKED residuals of a level (LevelGeom::compact): wavefront `wave_id` of `n_waves` takes the wavefront slots
// wave_id, wave_id + n_waves, ... of the pair (4 per tile, kCompactWaveEntries entries each of which the first count_q are there);
// SLOTS of them per round, two 16-B loads per lane and slot, the counts of the next round fetched while this one's pairs arrive.
// rows: the pair's partial rows (the counts); res: the pair's residual buffer.
template <int SLOTS>
__device__ __forceinline__ double loglik_partial_compact(const float2* __restrict__ res, const float* __restrict__ rows, int n_slots, const float* P,
                                                         int wave_id, int n_waves) {
  typedef const __attribute__((address_space(1))) float* G1;
  typedef float __attribute__((ext_vector_type(2))) f32x2;
  // (a buffer resource: the loads of a lane beyond its slot's count go to an offset outside of it -- zeros, no memory access, no branch)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(res), 0, n_slots * kCompactWaveEntries * 8, 0x00020000);
  const G1 counts = (G1)rows;
  const int lane = threadIdx.x & 63;
  double prod = 1.0;
  int exponent = 0;
  auto count_of = [&](int slot) -> float {                     // (both counts of the slot's half tile; the caller picks)
    const int sl = slot < n_slots ? slot : n_slots - 1;
    return counts[size_t(sl >> 2) * kAccStride + kAccCounts + ((sl >> 1) & 1)];
  };
  float packed[SLOTS];
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) packed[j] = count_of(wave_id + j * n_waves);
  constexpr int kChunks = kCompactWaveEntries / 64;            // a slot in chunks of one entry per lane: only the chunks that hold any are touched
  for (int s0 = wave_id; s0 < n_slots; s0 += SLOTS * n_waves) {
    f32x2 rr[SLOTS][kChunks];
    int cnt[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
      const int slot = s0 + j * n_waves;
      const int both = __builtin_amdgcn_readfirstlane(int(packed[j]));
      cnt[j] = slot < n_slots ? ((slot & 1) ? both >> 9 : both & 511) : 0;
#pragma unroll
      for (int m = 0; m < kChunks; ++m) {
        rr[j][m] = f32x2{0.0f, 0.0f};
        if (cnt[j] > 64 * m) {                                 // (uniform)
          const int i = lane + 64 * m;
          rr[j][m] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, i < cnt[j] ? i * 8 : 0x7ffffff8, slot * (kCompactWaveEntries * 8), 0));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) packed[j] = count_of(s0 + (SLOTS + j) * n_waves);
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
#pragma unroll
      for (int m = 0; m < kChunks; ++m)
        if (cnt[j] > 64 * m) {                                 // (uniform)
          const double f = 1.0 + 0.2 * double(mahalanobis(rr[j][m].x, rr[j][m].y, P));
          prod *= lane + 64 * m < cnt[j] ? f : 1.0;
        }
      int e;                                                   // four factors at most between renormalisations (loglik_partial: eight)
      prod = frexp(prod, &e);
      exponent += e;
    }
  }
  return log(prod) + double(exponent) * 0.6931471805599453094;
}

}  // namespace dvo_hip
