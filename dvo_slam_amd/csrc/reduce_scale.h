// reduce_scale.h -- fixed-order float64 reduction of a pair's four scale accumulators over its tiles.
//
// Both the log-likelihood kernel and the solver kernel need the precision P = (S/(n-3))^-1 of the
// current pass (dense_tracking.cpp:295).  They call this one routine with the same thread count, so
// both obtain bit-identical n, S and P regardless of launch geometry elsewhere.
#pragma once

#include "pixel_math.h"

namespace dvo_hip {

// blockDim.x must be kBlock; sh holds 4*kBlock doubles.  All threads return the same values.
__device__ inline int reduce_scale(const float* __restrict__ partials, int pair, int tiles, double* sh,
                                   double* s4, float* C, float* P) {
  const int tid = threadIdx.x;
  double s[4] = {0, 0, 0, 0};
  for (int b = tid; b < tiles; b += kBlock) {
    const float* p = partials + (size_t(pair) * tiles + b) * kAccStride;
    s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3];
  }
  for (int k = 0; k < 4; ++k) sh[k * kBlock + tid] = s[k];
  __syncthreads();
  for (int stride = kBlock / 2; stride > 0; stride >>= 1) {
    if (tid < stride)
      for (int k = 0; k < 4; ++k) sh[k * kBlock + tid] += sh[k * kBlock + tid + stride];
    __syncthreads();
  }
  for (int k = 0; k < 4; ++k) s4[k] = sh[k * kBlock];
  const double d = s4[0] - 3.0;
  scale_to_precision(s4[1] / d, s4[2] / d, s4[3] / d, C, P);
  return int(s4[0] + 0.5);
}

}  // namespace dvo_hip
