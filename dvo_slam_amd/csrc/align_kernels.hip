// align_kernels.hip -- the per-iteration streaming kernels of the photometric+geometric alignment.
//
// k_residual_reduce  passes 1, 2, 3 and 5 of dvo_core/src/dense_tracking.cpp:271-343 in ONE sweep over
//                    the level: SE(3) warp + projection (dense_tracking_impl.cpp:148-203), 4-tap
//                    bilinear sample of the six current-frame channels (:212-258), NaN / bounds /
//                    occlusion tests (:203,:261,:275), bivariate residual (:269-281), t-distribution
//                    weight (:640-644), per-pixel 2x6 Jacobian (dense_tracking.cpp:448-476, :333-340)
//                    and the reduction of n, sum w r r^T and the Gram sums of J^T W J / J^T W r
//                    (least_squares.cpp:58-64, math_sse.cpp:82-178).
// k_loglik           pass 4 (dense_tracking_impl.cpp:406-425): needs the precision P that is only known
//                    after the whole of pass 3, hence a second, light sweep over the 8-B residual pairs.
//
// Memory-bound per-pixel work (about 6 flop/B), so no MFMA: the design goals are coalesced 16-B
// streaming of the reference plane, 16-B/8-B gathers of the current planes that stay in L1/L2 thanks
// to 2-D tiles, XCD-aware tile->workgroup mapping, and a deterministic (atomic-free, fixed-order)
// wavefront-DPP -> LDS -> per-workgroup-partial reduction.
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

template <int STAGE>
__device__ __forceinline__ void dpp_stage_all(float* acc) {
  static_assert(kNumAcc % 5 == 0, "accumulators are reduced five at a time");
#pragma unroll
  for (int i = 0; i < kNumAcc; i += 5) dpp_stage5<STAGE>(acc[i], acc[i + 1], acc[i + 2], acc[i + 3], acc[i + 4]);
}

__device__ __forceinline__ void wave_sum_all_to_lane63(float* acc) {
  dpp_stage_all<0>(acc);
  dpp_stage_all<1>(acc);
  dpp_stage_all<2>(acc);
  dpp_stage_all<3>(acc);
  dpp_stage_all<4>(acc);
  dpp_stage_all<5>(acc);
}

__device__ __forceinline__ double wave_sum_double(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}

// FINEST only tags the instantiation that sweeps pyramid level 0 with a distinct symbol, so that a rocprofv3
// kernel trace reports the finest-level launches (the roofline kernel) separately from the coarser ones.
template <int RPW, bool FINEST>
__global__ __launch_bounds__(kBlock) void k_residual_reduce(const LevelGeom g, const PairPtrs* __restrict__ pairs,
                                                           const PairState* __restrict__ states, int n_pairs,
                                                           float* __restrict__ partials, float2* __restrict__ scratch,
                                                           int blocks_per_xcd) {
  // XCD-aware mapping: workgroup b runs on XCD b % 8 (observed dispatch order); give every XCD one
  // contiguous run of (pair, tile) work items so that a pair's current-frame planes -- re-read by
  // vertically adjacent tiles -- flow through a single L2.
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;

  const PairState& st = states[pair];
  if (!st.active) return;                            // wave-uniform: pair finished on this level
  const PairPtrs pp = pairs[pair];
  // pointers loaded from memory are generic; tell the compiler they are global so it emits global_load
  const GlobalLoad4 refR{(GlobalVec4)pp.refR}, curA{(GlobalVec4)pp.curA};
  const GlobalLoad2 curB{(GlobalVec2)pp.curB};

  float KT[12], Pp[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) Pp[i] = st.P_prev[i];
  const bool first = st.first != 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u_r = (tile % g.tiles_x) * kTileW + lane;
  const int row0 = (tile / g.tiles_x) * (kWavesPerBlock * RPW) + wave * RPW;
  const size_t pix_base = size_t(pair) * size_t(g.w) * g.h;

  float acc[kNumAcc];
#pragma unroll
  for (int i = 0; i < kNumAcc; ++i) acc[i] = 0.0f;

#pragma unroll 1
  for (int k = 0; k < RPW; ++k) {
    const int v_r = row0 + k;
    if (u_r >= g.w || v_r >= g.h) continue;
    const int idx = v_r * g.w + u_r;
    const float4 ref = refR[idx];                 // 64 lanes x 16 B = 1 KiB contiguous per wave
    PixelTerms t;
    const bool valid = pixel_residual(g, KT, curA, curB, ref, u_r, v_r, t);
    const float nanv = __builtin_nanf("");
    scratch[pix_base + idx] = valid ? make_float2(t.r0, t.r1) : make_float2(nanv, nanv);
    if (!valid) continue;

    // t-distribution weight with the PREVIOUS pass' precision (Q11); first pass on a level: w = 1
    const float w = first ? 1.0f : tdist_weight(t.r0, t.r1, Pp);
    accumulate_pixel(acc, t, w);
  }

  // stage 1: DPP reduction inside each wavefront; stage 2: the four wave results through LDS;
  // stage 3 (fixed block order, float64) happens in the per-pair solver kernel.  No float atomics.
  __shared__ float lds[kWavesPerBlock][kAccStride];
  wave_sum_all_to_lane63(acc);
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < kNumAcc; ++i) lds[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < kNumAcc) {
    const float s = (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
    partials[(size_t(pair) * tiles + tile) * kAccStride + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(kBlock) void k_loglik(const LevelGeom g, const PairState* __restrict__ states, int n_pairs,
                                                   const float* __restrict__ partials, const float2* __restrict__ scratch,
                                                   double* __restrict__ ll_partials, int blocks_per_pair) {
  const int pair = blockIdx.y;
  if (!states[pair].active) return;
  __shared__ double sh[4 * kBlock];
  float C[3], P[4];
  double s4[4];
  const int tiles = g.tiles_x * g.tiles_y;
  const int n = reduce_scale(partials, pair, tiles, sh, s4, C, P);
  __syncthreads();
  double total = 0.0;
  if (n >= 6) {
    const int npx = g.w * g.h;
    const float2* r = scratch + size_t(pair) * npx;
    // product of up to 8 terms per lane, one log per lane per chunk (the reference multiplies 50
    // terms between logs, dense_tracking_impl.cpp:413-422)
    for (int base = blockIdx.x * kBlock * 8; base < npx; base += blocks_per_pair * kBlock * 8) {
      double prod = 1.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = base + k * kBlock + threadIdx.x;
        if (i < npx) {
          const float2 rr = r[i];
          if (rr.x == rr.x) prod *= 1.0 + 0.2 * double(mahalanobis(rr.x, rr.y, P));
        }
      }
      total += log(prod);
    }
  }
  total = wave_sum_double(total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = total;
  __syncthreads();
  if (threadIdx.x == 0) ll_partials[size_t(pair) * blocks_per_pair + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <int RPW>
static void launch_rr(hipStream_t s, bool finest, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                      float* partials, float2* scratch) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  if (finest)
    k_residual_reduce<RPW, true><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd);
  else
    k_residual_reduce<RPW, false><<<dim3(per_xcd * 8), dim3(kBlock), 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd);
}

void launch_residual_reduce(hipStream_t s, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs,
                            const PairState* states, int n_pairs, float* partials, float2* scratch) {
  switch (rows_per_wave) {
    case 1: launch_rr<1>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 2: launch_rr<2>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 4: launch_rr<4>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 16: launch_rr<16>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    default: launch_rr<8>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
  }
}

void launch_loglik(hipStream_t s, const LevelGeom& g, const PairState* states, int n_pairs, const float* partials,
                   const float2* scratch, double* ll_partials, int blocks_per_pair) {
  k_loglik<<<dim3(blocks_per_pair, n_pairs), dim3(kBlock), 0, s>>>(g, states, n_pairs, partials, scratch, ll_partials, blocks_per_pair);
}

}  // namespace dvo_hip
