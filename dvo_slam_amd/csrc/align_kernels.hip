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
// This file holds the all-VALU schedule the north star sketches ("variant 0": coalesced 16-B streaming of the reference
// plane, 16-B/8-B gathers of the current planes that stay in L1/L2 thanks to 2-D tiles, XCD-aware tile->workgroup
// mapping, and a deterministic (atomic-free, fixed-order) wavefront-DPP -> LDS -> per-workgroup-partial reduction of the
// 85 accumulators), which stores the residual pairs and leaves the log-likelihood to k_loglik.  The default schedule
// (align_mfma.hip) produces the same outputs 1.3x faster; variant 0 is kept as the independent second implementation the
// parity tests compare it with.
#include "align_common.h"

namespace dvo_hip {

// FINEST only tags the instantiation that sweeps pyramid level 0 with a distinct symbol, so that a rocprofv3
// kernel trace reports the finest-level launches separately from the coarser ones.
template <int RPW, bool FINEST>
__global__ __launch_bounds__(kBlock) void k_residual_reduce(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd) {
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
  if (!st.active || st.level != g.level) return;     // wave-uniform: pair finished on this level (and maybe begun the next)
  const PairPtrs pp = pairs[pair];
  // pointers loaded from memory are generic; tell the compiler they are global so it emits global_load
  const GlobalLoad2 refR{(GlobalVec2)pp.refR};
  const GlobalLoad4 curA{(GlobalVec4)pp.curA};
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
  const float nanv = __builtin_nanf("");
  const bool col_ok = u_r < g.w;

  float acc[kNumAcc];
#pragma unroll
  for (int i = 0; i < kNumAcc; ++i) acc[i] = 0.0f;

  // 64 lanes x 16 B = 1 KiB contiguous per wave; rows beyond the image read as "not selected"
  auto load_ref = [&](int k) -> float4 {
    const int v_r = row0 + k;
    if (k < RPW && col_ok && v_r < g.h) {
      // the reference plane holds {Zsel, I}; the intensity gradient is the clamped central difference of the frame build
      // (pyramid_kernels.hip::derive_at), recomputed here from the four neighbours
      const int at = v_r * g.w + u_r;
      const float2 zi = refR[at];
      const float left = refR[at - (u_r > 0 ? 1 : 0)].y, right = refR[at + (u_r < g.w - 1 ? 1 : 0)].y;
      const float up = refR[at - (v_r > 0 ? g.w : 0)].y, down = refR[at + (v_r < g.h - 1 ? g.w : 0)].y;
      return make_float4(zi.x, zi.y, (right - left) * 0.5f, (down - up) * 0.5f);
    }
    return make_float4(nanv, 0.0f, 0.0f, 0.0f);
  };
  // blend, test, store the residual pair for the log-likelihood sweep, weight and accumulate
  auto finish_row = [&](int k, const float4 ref, const PixelProj& p, const PixelTaps& t) {
    const int v_r = row0 + k;
    if (!col_ok || v_r >= g.h) return;
    PixelTerms o;
    const bool valid = p.ok && pixel_finish(g, ref, p, t, o);
    scratch[pix_base + size_t(v_r) * g.w + u_r] = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);
    if (!valid) return;
    // t-distribution weight with the PREVIOUS pass' precision (Q11); first pass on a level: w = 1
    const float w = first ? 1.0f : g.rcp_table ? 7.0f * rcp_like_the_host(g.rcp_table, g.rcp_shift, 5.0f + mahalanobis(o.r0, o.r1, Pp))   // (option "ref_compat")
                                               : tdist_weight(o.r0, o.r1, Pp);
    accumulate_pixel(acc, o, w);
  };

#pragma unroll 1
  for (int k = 0; k < RPW; ++k) {
    const float4 ref = load_ref(k);
    const PixelProj p = pixel_project(g, KT, ref, col_ok ? u_r : 0, min(row0 + k, g.h - 1));
    PixelTaps t;
    if (p.ok) pixel_fetch(g, curA, curB, p, t);
    finish_row(k, ref, p, t);
  }

  // stage 1: DPP reduction inside each wavefront; stage 2: the four wave results through LDS;
  // stage 3 (fixed block order, float64) happens in the per-pair solver kernel.  No float atomics.
  __shared__ float lds[kWavesPerBlock][kAccStride];
  wave_sum_all_to_lane63<kNumAcc>(acc);
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

// LOADS: 16-B loads in flight per lane (the value of the sum does not depend on it beyond the grouping of the partial sums)
template <int LOADS>
__global__ __launch_bounds__(kBlock) void k_loglik(const LevelGeom g, const PairState* __restrict__ states, int n_pairs,
                                                   const float* __restrict__ partials, const float2* __restrict__ scratch,
                                                   double* __restrict__ ll_partials, int blocks_per_pair) {
  const int pair = pair_of_launch_index(g, blockIdx.y);       // (the slow lane of a batch: a list of pairs; the batch's own launches skip its pairs)
  if (pair < 0) return;
  if (!states[pair].active || states[pair].level != g.level) return;
  __shared__ double sh[16];
  __shared__ double sums[4];
  __shared__ float stage[kScaleStageFloats];
  float C[3], P[4];
  reduce_partials_scale(partials, pair, g.tiles_x * g.tiles_y, stage, sh, sums);
  const int n = scale_from_sums(sums, C, P);
  double total = 0.0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (n >= 6) {
    const int tiles = g.tiles_x * g.tiles_y;
    if (g.compact)                                            // (uniform) the packed residuals of the contracted window sweep
      total = loglik_partial_compact<4>(scratch + size_t(pair) * residual_entries(g), partials + size_t(pair) * tiles * kAccStride, tiles * 4, P,
                                        blockIdx.x * kWavesPerBlock + wave, blocks_per_pair * kWavesPerBlock);
    else
      total = loglik_partial<LOADS>(scratch + size_t(pair) * g.w * g.h, g.w * g.h, P, blockIdx.x, blocks_per_pair);
  }
  total = wave_sum_double(total);
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

// the sweep of this level has an instantiation that ends with the pairs' solver steps (launch_residual_reduce's own dispatch)
bool sweep_has_tail(int variant, int rows_per_wave, const LevelGeom& g) {
  if (g.small) return false;
  if (variant >= 8 && rows_per_wave == 4 && fast_sweep_supports(g)) return sweep_fast_has_tail(variant, g);
  if (variant >= 6 && rows_per_wave == 4 && window_sweep_supports(g)) return false;
  return variant >= 5 && mfma_sweep_has_tail(variant, rows_per_wave, g);
}

// (the kernels that read LevelGeom::pair_list: k_sweep_fast and k_residual_reduce_mfma -- what launch_residual_reduce picks below for
// the default schedule, variants 8 / 9, on every level but those of the opt-in small-level sweep and widths of 64 the exact window sweep takes)
bool sweep_takes_pair_list(int variant, int rows_per_wave, const LevelGeom& g) {
  if (variant < 8 || g.small) return false;
  if (rows_per_wave == 4 && fast_sweep_supports(g)) return true;
  return !(rows_per_wave == 4 && window_sweep_supports(g));
}

void launch_residual_reduce(hipStream_t s, int variant, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs,
                            const PairState* states, int n_pairs, float* partials, float2* scratch, unsigned long long* window_fallbacks, int* f16_range_flag,
                            const SolverStepArgs* tail) {
  if (g.small) {                                              // (the default schedule on a level small enough for LDS: align_small.hip)
    launch_sweep_small(s, rows_per_wave, g, pairs, states, n_pairs, partials, scratch, f16_range_flag);
    return;
  }
  if (variant >= 8 && rows_per_wave == 4 && fast_sweep_supports(g)) {
    launch_sweep_fast(s, variant, g, pairs, states, n_pairs, partials, scratch, window_fallbacks, f16_range_flag, tail);
    return;
  }
  if (variant >= 6 && rows_per_wave == 4 && window_sweep_supports(g)) {
    launch_sweep_window(s, variant >= 7, g, pairs, states, n_pairs, partials, scratch, window_fallbacks, f16_range_flag);
    return;
  }
  if (variant >= 5) {
    launch_residual_reduce_mfma(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch, variant >= 8 ? 2 : variant >= 7 ? 1 : 0, f16_range_flag, tail);
    return;
  }
  switch (rows_per_wave) {
    case 1: launch_rr<1>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 2: launch_rr<2>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 4: launch_rr<4>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    case 16: launch_rr<16>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
    default: launch_rr<8>(s, finest, g, pairs, states, n_pairs, partials, scratch); break;
  }
}

// The sweep's bytes without the sweep: every pixel's reference pair, current-frame quad and gradient pair are read in
// pixel order and an 8-byte pair is written where the sweep writes its residuals.  No gather, no arithmetic, no reduction:
// the time of this kernel is what the memory system needs for the sweep's algorithmic traffic (dvo_hip_time_stream_mix).
// WRITE = false: the read side alone (what the default sweep moves since it keeps the residual pairs in registers).
// WINDOW: the planes of the window sweep (align_window.hip) -- reference {Zsel, I} 8 B + current {I, Z} 8 B -- instead of the
// gathering sweep's 8 + 16 + 8 B.
template <bool WRITE, bool WINDOW>
__global__ __launch_bounds__(kBlock) void k_stream_mix(const PairPtrs* __restrict__ pairs, int n_px, float2* __restrict__ scratch, float* __restrict__ sink) {
  const PairPtrs pp = pairs[blockIdx.y];
  float2* out = scratch + size_t(blockIdx.y) * n_px;
  float fold = 0.0f;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n_px; i += gridDim.x * kBlock) {
    const float2 r = pp.refR[i];
    float4 a;
    float2 b;
    if constexpr (WINDOW) {
      const float2 c = pp.curC[i];
      a = make_float4(c.x, c.y, c.x, c.y);
      b = c;
    } else {
      a = pp.curA[i];
      b = pp.curB[i];
    }
    if constexpr (WRITE) out[i] = make_float2(r.x + a.y + b.x, r.y + a.z + b.y);
    else fold += (r.x + a.y + b.x) + (r.y + a.z + b.y) + (a.x + a.w);
  }
  if constexpr (!WRITE) {
    // keep the loads alive with one store per workgroup
    for (int off = 32; off > 0; off >>= 1) fold += __shfl_down(fold, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sink[blockIdx.y * gridDim.x + blockIdx.x], fold);
  }
}

void launch_stream_mix(hipStream_t s, const PairPtrs* pairs, int n_pairs, int n_px, float2* scratch, float* sink, bool window_planes) {
  const int per_pair = (n_px + kBlock * 8 - 1) / (kBlock * 8);
  const dim3 grid(per_pair, n_pairs), block(kBlock);
  if (window_planes) {
    if (scratch) k_stream_mix<true, true><<<grid, block, 0, s>>>(pairs, n_px, scratch, sink);
    else k_stream_mix<false, true><<<grid, block, 0, s>>>(pairs, n_px, scratch, sink);
  } else {
    if (scratch) k_stream_mix<true, false><<<grid, block, 0, s>>>(pairs, n_px, scratch, sink);
    else k_stream_mix<false, false><<<grid, block, 0, s>>>(pairs, n_px, scratch, sink);
  }
}

void launch_loglik(hipStream_t s, const LevelGeom& g, const PairState* states, int n_pairs, const float* partials,
                   const float2* scratch, double* ll_partials, int blocks_per_pair, bool one_schedule) {
  // few pairs: the sweep is a handful of dependent round trips per lane, more loads in flight shorten it (one pair 0.52 -> 0.50 ms);
  // a full batch is bandwidth-bound and runs 6 % slower with the larger chunks
  // (one_schedule: the grouping of the partial sums must not follow the batch size -- option "deterministic")
  if (n_pairs <= 16 && !one_schedule && !g.pair_list) k_loglik<8><<<dim3(blocks_per_pair, n_pairs), dim3(kBlock), 0, s>>>(g, states, n_pairs, partials, scratch, ll_partials, blocks_per_pair);
  else k_loglik<4><<<dim3(blocks_per_pair, n_pairs), dim3(kBlock), 0, s>>>(g, states, n_pairs, partials, scratch, ll_partials, blocks_per_pair);
}

}  // namespace dvo_hip
