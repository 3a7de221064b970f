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

// The whole level by ONE workgroup of WAVES real wavefronts that play the 256 threads of a four-wavefront one (the solver step of a
// small level, solver_kernels.hip): thread t plays the threads t, t + 64 WAVES, ... -- every played thread multiplies the same factors
// in the same order as loglik_partial(res, npx, P, 0, 1) makes it, the loads of all of them go out together.  out[r]: the value of
// played thread threadIdx.x + 64 WAVES r.
template <int LOADS, int WAVES>
__device__ __forceinline__ void loglik_partial_played(const float2* __restrict__ res, int npx, const float* P, double* out) {
  typedef const __attribute__((address_space(1))) vec4f* G4;
  constexpr int kPlayed = kWavesPerBlock / WAVES;
  const G4 r = (G4)res;
  const int npair2 = npx >> 1;
  double prod[kPlayed];
  int exponent[kPlayed];
#pragma unroll
  for (int q = 0; q < kPlayed; ++q) { prod[q] = 1.0; exponent[q] = 0; }
  for (int base = 0; base < npair2; base += kBlock * LOADS) {
    vec4f rr[kPlayed][LOADS];
#pragma unroll
    for (int q = 0; q < kPlayed; ++q)
#pragma unroll
      for (int k = 0; k < LOADS; ++k) {
        const int i = base + k * kBlock + int(threadIdx.x) + 64 * WAVES * q;
        const float nanv = __builtin_nanf("");
        const vec4f v = r[i < npair2 ? i : npair2 - 1];
        rr[q][k] = i < npair2 ? v : vec4f{nanv, nanv, nanv, nanv};
      }
#pragma unroll
    for (int q = 0; q < kPlayed; ++q)
#pragma unroll
      for (int k = 0; k < LOADS; ++k) {
        if (rr[q][k].x == rr[q][k].x) prod[q] *= 1.0 + 0.2 * double(mahalanobis(rr[q][k].x, rr[q][k].y, P));
        if (rr[q][k].z == rr[q][k].z) prod[q] *= 1.0 + 0.2 * double(mahalanobis(rr[q][k].z, rr[q][k].w, P));
        if ((k & 3) == 3 || k == LOADS - 1) {
          int e;
          prod[q] = frexp(prod[q], &e);
          exponent[q] += e;
        }
      }
  }
  if ((npx & 1) && threadIdx.x == 0) {                       // odd pixel count: last pixel (played thread 0)
    const float2 rr = res[npx - 1];
    if (rr.x == rr.x) prod[0] *= 1.0 + 0.2 * double(mahalanobis(rr.x, rr.y, P));
  }
#pragma unroll
  for (int q = 0; q < kPlayed; ++q) out[q] = log(prod[q]) + double(exponent[q]) * 0.6931471805599453094;
}

// The same sum over the PACKED residuals of a level (LevelGeom::compact): wavefront `wave_id` of `n_waves` takes the wavefront slots
// wave_id, wave_id + n_waves, ... of the pair (4 per tile, kCompactWaveEntries entries each of which the first count_q are there);
// SLOTS of them per round, two 16-B loads per lane and slot, the counts of the next round fetched while this one's pairs arrive.
// rows: the pair's partial rows (the counts); res: the pair's residual buffer.
// PLAYED > 1: the wavefront plays the wavefronts wave_id, wave_id + played_stride, ... of the n_waves (see loglik_partial_played), each
// with its own product, the loads of all of them going out together; out[q]: the value of played wavefront q.
template <int SLOTS, int PLAYED>
__device__ __forceinline__ void loglik_partial_compact_played(const float2* __restrict__ res, const float* __restrict__ rows, int n_slots, const float* P,
                                                              int wave_id, int played_stride, int n_waves, double* out) {
  typedef const __attribute__((address_space(1))) float* G1;
  typedef float __attribute__((ext_vector_type(2))) f32x2;
  // (a buffer resource: the loads of a lane beyond its slot's count go to an offset outside of it -- zeros, no memory access, no branch)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(res), 0, n_slots * kCompactWaveEntries * 8, 0x00020000);
  const G1 counts = (G1)rows;
  const int lane = threadIdx.x & 63;
  double prod[PLAYED];
  int exponent[PLAYED];
#pragma unroll
  for (int q = 0; q < PLAYED; ++q) { prod[q] = 1.0; exponent[q] = 0; }
  auto count_of = [&](int slot) -> float {                     // (both counts of the slot's half tile; the caller picks)
    const int sl = slot < n_slots ? slot : n_slots - 1;
    return counts[size_t(sl >> 2) * kAccStride + kAccCounts + ((sl >> 1) & 1)];
  };
  float packed[PLAYED][SLOTS];
#pragma unroll
  for (int q = 0; q < PLAYED; ++q)
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) packed[q][j] = count_of(wave_id + q * played_stride + j * n_waves);
  constexpr int kChunks = kCompactWaveEntries / 64;            // a slot in chunks of one entry per lane: only the chunks that hold any are touched
  for (int s0 = wave_id; s0 < n_slots; s0 += SLOTS * n_waves) {   // (wave_id is the smallest played wavefront: its slots end last)
    f32x2 rr[PLAYED][SLOTS][kChunks];
    int cnt[PLAYED][SLOTS];
#pragma unroll
    for (int q = 0; q < PLAYED; ++q)
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) {
        const int slot = s0 + q * played_stride + j * n_waves;
        const int both = __builtin_amdgcn_readfirstlane(int(packed[q][j]));
        cnt[q][j] = slot < n_slots ? ((slot & 1) ? both >> 9 : both & 511) : 0;
#pragma unroll
        for (int m = 0; m < kChunks; ++m) {
          rr[q][j][m] = f32x2{0.0f, 0.0f};
          if (cnt[q][j] > 64 * m) {                            // (uniform)
            const int i = lane + 64 * m;
            rr[q][j][m] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, i < cnt[q][j] ? i * 8 : 0x7ffffff8, slot * (kCompactWaveEntries * 8), 0));
          }
        }
      }
#pragma unroll
    for (int q = 0; q < PLAYED; ++q)
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) packed[q][j] = count_of(s0 + q * played_stride + (SLOTS + j) * n_waves);
#pragma unroll
    for (int q = 0; q < PLAYED; ++q)
#pragma unroll
      for (int j = 0; j < SLOTS; ++j) {
#pragma unroll
        for (int m = 0; m < kChunks; ++m)
          if (cnt[q][j] > 64 * m) {                            // (uniform)
            const double f = 1.0 + 0.2 * double(mahalanobis(rr[q][j][m].x, rr[q][j][m].y, P));
            prod[q] *= lane + 64 * m < cnt[q][j] ? f : 1.0;
          }
        int e;                                                 // four factors at most between renormalisations (loglik_partial: eight)
        prod[q] = frexp(prod[q], &e);
        exponent[q] += e;
      }
  }
#pragma unroll
  for (int q = 0; q < PLAYED; ++q) out[q] = log(prod[q]) + double(exponent[q]) * 0.6931471805599453094;
}

template <int SLOTS>
__device__ __forceinline__ double loglik_partial_compact(const float2* __restrict__ res, const float* __restrict__ rows, int n_slots, const float* P,
                                                         int wave_id, int n_waves) {
  double out[1];
  loglik_partial_compact_played<SLOTS, 1>(res, rows, n_slots, P, wave_id, 0, n_waves, out);
  return out[0];
}

}  // namespace dvo_hip
