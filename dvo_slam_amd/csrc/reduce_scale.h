// reduce_scale.h -- stage 3 of the deterministic reduction: a pair's per-tile float partials summed in a FIXED
// order in float64 by one 256-thread workgroup.
//
// Both the log-likelihood kernel and the solver kernel need the precision P = (S/(n-3))^-1 of the current
// pass (dense_tracking.cpp:295).  They call this one routine with the same thread count, so both obtain
// bit-identical n, S and P regardless of launch geometry elsewhere.  Order: wavefront w adds tiles
// w, w+4, w+8, ... (lane k owns accumulator k, lanes 0..23 also own accumulator 64+k: each row of the partial
// table is read as one coalesced 352-byte line), then the four wave sums are combined as (0+1)+(2+3).
#pragma once

#include "pixel_math.h"

namespace dvo_hip {

// Rows in flight per lane and round.  The sums are latency-bound, not bandwidth-bound: the partial rows were written
// by other CUs a kernel ago and every round is a ~2 us trip to HBM/MALL, so a round covers 4 wavefronts x 32 rows = 128
// tiles (measured: 32 tiles per round cost 100 us per iteration for a single pair's 1200 finest-level tiles).  Only the
// number of loads in flight changes with this constant, never the order of the additions.
constexpr int kReduceInFlight = 32;

// blockDim.x must be kBlock.  sh: kWavesPerBlock * kAccStride doubles of LDS.  sums: kAccStride doubles of LDS,
// valid for all threads after the call.
__device__ inline void reduce_partials(const float* __restrict__ partials, int pair, int tiles, double* sh, double* sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = partials + size_t(pair) * tiles * kAccStride;
  const bool hi = lane < kAccStride - 64;
  double a0 = 0.0, a1 = 0.0;
  // independent row loads in flight, then added in tile order (the summation order is part of the contract)
  for (int t0 = wave; t0 < tiles; t0 += kReduceInFlight * kWavesPerBlock) {
    float v0[kReduceInFlight], v1[kReduceInFlight];
    // branch-free: every load is issued (out-of-range rows re-read the last row), the selection happens at the add --
    // a conditional load costs a branch and a full s_waitcnt each, which serialises the round trips
#pragma unroll
    for (int j = 0; j < kReduceInFlight; ++j) {
      const int t = t0 + j * kWavesPerBlock;
      const float* row = base + size_t(t < tiles ? t : tiles - 1) * kAccStride;
      v0[j] = row[lane];
      v1[j] = row[hi ? 64 + lane : 64];
    }
#pragma unroll
    for (int j = 0; j < kReduceInFlight; ++j) {
      const bool in = t0 + j * kWavesPerBlock < tiles;
      a0 += in ? double(v0[j]) : 0.0;                       // x + 0.0 == x: the masked adds do not perturb the sum
      a1 += in ? double(v1[j]) : 0.0;
    }
  }
  sh[wave * kAccStride + lane] = a0;
  if (hi) sh[wave * kAccStride + 64 + lane] = a1;
  __syncthreads();
  if (threadIdx.x < kAccStride) {
    const int k = threadIdx.x;
    sums[k] = (sh[k] + sh[kAccStride + k]) + (sh[2 * kAccStride + k] + sh[3 * kAccStride + k]);
  }
  __syncthreads();
}

// The same reduction restricted to the four scale accumulators (n, S00, S01, S11), for kernels that only need P:
// lane k < 4 of wavefront w adds tiles w, w+4, ... in the same order as reduce_partials, so the result is
// bit-identical to sums[0..3] of the full reduction while reading 16 B instead of 352 B per tile.
__device__ inline void reduce_partials_scale(const float* __restrict__ partials, int pair, int tiles, double* sh, double* sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < 4) {
    const float* base = partials + size_t(pair) * tiles * kAccStride;
    double a0 = 0.0;
    for (int t0 = wave; t0 < tiles; t0 += kReduceInFlight * kWavesPerBlock) {
      float v0[kReduceInFlight];
#pragma unroll
      for (int j = 0; j < kReduceInFlight; ++j) {
        const int t = t0 + j * kWavesPerBlock;
        v0[j] = base[size_t(t < tiles ? t : tiles - 1) * kAccStride + lane];
      }
#pragma unroll
      for (int j = 0; j < kReduceInFlight; ++j) a0 += (t0 + j * kWavesPerBlock < tiles) ? double(v0[j]) : 0.0;
    }
    sh[wave * 4 + lane] = a0;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    sums[k] = (sh[k] + sh[4 + k]) + (sh[8 + k] + sh[12 + k]);
  }
  __syncthreads();
}

// n, C = S/(n-3) (float) and P = C^-1 from the reduced sums
__device__ inline int scale_from_sums(const double* sums, float* C, float* P) {
  const double d = sums[kAccN] - 3.0;
  scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, C, P);
  return int(sums[kAccN] + 0.5);
}

}  // namespace dvo_hip
