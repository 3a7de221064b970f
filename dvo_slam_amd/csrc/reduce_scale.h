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

// blockDim.x must be kBlock.  sh: kWavesPerBlock * kAccStride doubles of LDS.  sums: kAccStride doubles of LDS,
// valid for all threads after the call.
__device__ inline void reduce_partials(const float* __restrict__ partials, int pair, int tiles, double* sh, double* sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* row = partials + (size_t(pair) * tiles + wave) * kAccStride;
  double a0 = 0.0, a1 = 0.0;
  for (int t = wave; t < tiles; t += kWavesPerBlock, row += size_t(kWavesPerBlock) * kAccStride) {
    a0 += double(row[lane]);
    if (lane < kAccStride - 64) a1 += double(row[64 + lane]);
  }
  sh[wave * kAccStride + lane] = a0;
  if (lane < kAccStride - 64) sh[wave * kAccStride + 64 + lane] = a1;
  __syncthreads();
  if (threadIdx.x < kAccStride) {
    const int k = threadIdx.x;
    sums[k] = (sh[k] + sh[kAccStride + k]) + (sh[2 * kAccStride + k] + sh[3 * kAccStride + k]);
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
