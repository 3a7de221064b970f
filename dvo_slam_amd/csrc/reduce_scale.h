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
// WAVES: wavefronts the workgroup really has (4, or 2: the solver step of a small level, so that four of them fit a compute unit).
// The additions do not depend on it: a real wavefront plays the "virtual" wavefronts w, w + WAVES, ... of a four-wavefront
// workgroup, each with its own accumulators.
// IN_FLIGHT: row loads in flight per lane and round (registers against round trips; a caller whose levels have few tiles and whose
// registers are scarce -- the fused coarse-level kernel -- asks for fewer); like WAVES it does not touch the order of the additions.
template <int WAVES = kWavesPerBlock, int IN_FLIGHT = kReduceInFlight>
__device__ inline void reduce_partials(const float* __restrict__ partials, int pair, int tiles, double* sh, double* sums) {
  static_assert(WAVES == 2 || WAVES == 4, "two or four wavefronts");
  constexpr int kPlayed = kWavesPerBlock / WAVES;           // virtual wavefronts per real one
  constexpr int kInFlight = IN_FLIGHT / kPlayed;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = partials + size_t(pair) * tiles * kAccStride;
  const bool hi = lane < kAccStride - 64;
  double a0[kPlayed], a1[kPlayed];
#pragma unroll
  for (int r = 0; r < kPlayed; ++r) a0[r] = a1[r] = 0.0;
  // independent row loads in flight, then added in tile order (the summation order is part of the contract)
  for (int t0 = wave; t0 < tiles; t0 += kInFlight * kWavesPerBlock) {
    float v0[kPlayed][kInFlight], v1[kPlayed][kInFlight];
    // branch-free: every load is issued (out-of-range rows re-read the last row), the selection happens at the add --
    // a conditional load costs a branch and a full s_waitcnt each, which serialises the round trips
#pragma unroll
    for (int r = 0; r < kPlayed; ++r)
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const int t = t0 + r * WAVES + j * kWavesPerBlock;
        const float* row = base + size_t(t < tiles ? t : tiles - 1) * kAccStride;
        v0[r][j] = row[lane];
        v1[r][j] = row[hi ? 64 + lane : 64];
      }
#pragma unroll
    for (int r = 0; r < kPlayed; ++r)
#pragma unroll
      for (int j = 0; j < kInFlight; ++j) {
        const bool in = t0 + r * WAVES + j * kWavesPerBlock < tiles;
        a0[r] += in ? double(v0[r][j]) : 0.0;               // x + 0.0 == x: the masked adds do not perturb the sum
        a1[r] += in ? double(v1[r][j]) : 0.0;
      }
  }
#pragma unroll
  for (int r = 0; r < kPlayed; ++r) {
    sh[(wave + r * WAVES) * kAccStride + lane] = a0[r];
    if (hi) sh[(wave + r * WAVES) * kAccStride + 64 + lane] = a1[r];
  }
  __syncthreads();
  if (threadIdx.x < kAccStride) {
    const int k = threadIdx.x;
    sums[k] = (sh[k] + sh[kAccStride + k]) + (sh[2 * kAccStride + k] + sh[3 * kAccStride + k]);
  }
  __syncthreads();
}

// The same reduction restricted to the four scale accumulators (n, S00, S01, S11), for kernels that only need P.  The
// additions happen in the same order as in reduce_partials (wavefront w adds tiles w, w+4, ...; then (0+1)+(2+3)), so the
// result is bit-identical to sums[0..3] of the full reduction while reading 16 B instead of 352 B per tile.  All 64 lanes of
// a wavefront fetch (lane l: component l&3 of the wavefront's rows l>>2, l>>2 + 16, ...), up to kScaleRowsPerRound rows per
// round trip, into LDS; lanes 0..3 then add their component in row order.  (With only the four adding lanes fetching, a lone
// pair's 600 finest-level tiles cost five dependent round trips: 10 of the log-likelihood kernel's 17 us.)
constexpr int kScaleLoadsInFlight = 16;
constexpr int kScaleRowsPerRound = 16 * kScaleLoadsInFlight;      // per wavefront
constexpr int kScaleStageFloats = kWavesPerBlock * kScaleRowsPerRound * 4;

// stage: kScaleStageFloats floats of LDS; sh: 16 doubles; sums: 4 doubles, valid for all threads after the call
__device__ inline void reduce_partials_scale(const float* __restrict__ partials, int pair, int tiles, float* stage, double* sh, double* sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = partials + size_t(pair) * tiles * kAccStride;
  const int my_rows = tiles > wave ? (tiles - wave + kWavesPerBlock - 1) / kWavesPerBlock : 0;   // tiles wave, wave + 4, ...
  float* mine = stage + wave * kScaleRowsPerRound * 4;
  double a0 = 0.0;
  for (int r0 = 0; r0 < my_rows; r0 += kScaleRowsPerRound) {
    float v[kScaleLoadsInFlight];
#pragma unroll
    for (int j = 0; j < kScaleLoadsInFlight; ++j) {
      const int r = r0 + j * 16 + (lane >> 2);
      const int t = wave + (r < my_rows ? r : my_rows - 1) * kWavesPerBlock;
      v[j] = base[size_t(t) * kAccStride + (lane & 3)];
    }
#pragma unroll
    for (int j = 0; j < kScaleLoadsInFlight; ++j) mine[(j * 16 + (lane >> 2)) * 4 + (lane & 3)] = v[j];
    // the slab is private to the wavefront: LDS executes its operations in order, only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 4) {
      const int count = min(kScaleRowsPerRound, my_rows - r0);
      for (int r = 0; r < count; ++r) a0 += double(mine[r * 4 + lane]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 4) sh[wave * 4 + lane] = a0;
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
