// align_mfma.hip -- the fused residual / Jacobian / reduce sweep with the normal-equation accumulation on the
// matrix cores (schedule variant 5 of k_residual_reduce; same per-pixel arithmetic, same outputs).
//
// Why a matrix instruction in a "memory-bound per-pixel" kernel: measured on MI355X (profiles/r01_*), the
// all-VALU schedule is NOT memory-bound.  Its 85 per-lane accumulators cost ~125 FMAs per pixel plus a
// 600-instruction DPP reduction per wavefront, SQ_ACTIVE_INST_VALU shows the vector ALUs 78 % busy, and the
// kernel stops at 0.50 ms per 128-pair finest-level launch (39 % of the HBM roofline) with 150 VGPRs = 3
// wavefronts per SIMD.  The accumulation itself is a rank-k update  G += V^T V  of the 16x16 Gram matrix of the
// per-pixel vectors  v = sqrt(w) * [J0(6), J1(6), r0, r1, 0, 0]  (least_squares.cpp:58-64 contracts exactly this
// with the 2x2 precision) -- matrix math, four pixels (k = 4) per v_mfma_f32_16x16x4_f32.  That instruction is an
// exact f32 fmaf chain (no reduced precision), runs on the otherwise idle matrix pipe, keeps the whole Gram
// matrix in 4 accumulator registers per lane instead of 85, and leaves NO cross-lane reduction to do: the sum
// over the 64 pixels of a row happens inside the 16 MFMAs.  The per-pixel part (warp, bilinear taps, residual,
// weights, Jacobian: dense_tracking_impl.cpp:148-281, dense_tracking.cpp:448-476) is unchanged VALU code.
//
// Data movement per pixel row of a wavefront: each lane (= pixel) writes its 16-vector to LDS as four
// conflict-free ds_write_b128; the MFMA operand for pixel group g (lane l <- component l&15 of pixel 4g + l>>4)
// is one conflict-free ds_read_b32 at a constant offset.  A == B (the sqrt(w)-scaled vector on both sides).
// The LDS slab is private to the wavefront, so the row loop contains no barrier.
#include "mfma_sweep.h"
#include "solver_step.h"

namespace dvo_hip {

// FINEST: not used in the body -- it only gives the launches of pyramid level 0 a kernel name of their own in profiler traces.
// TAIL (round 6): the workgroup that completes the last tile of a pair runs the pair's Gauss-Newton step right here (solver_step.h; see
// k_sweep_fast) -- one launch per iteration instead of two on the levels whose log-likelihood pass fits the step.
// (with a tail the kernel is built for five workgroups per compute unit: the out-of-line step gets the kernel's register budget,
// and left alone -- 248 registers -- it would leave the sweep two)
template <int RPW, bool FINEST, bool LINEAR, int MODE, int TAIL>
__global__ __launch_bounds__(kBlock, TAIL ? 5 : 1) void k_residual_reduce_mfma(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd, int* __restrict__ f16_range_flag, const SolverStepArgs tail) {
  // XCD-aware (pair, tile) -> workgroup mapping, see k_residual_reduce
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int entry = item / tiles, tile = item - entry * tiles;
  const int pair = pair_of_launch_index(g, entry);           // (a level's overlapped tail: the launch covers a list of pairs)
  if (pair < 0) return;
  const PairState& st = states[pair];
  constexpr int kSlabBytes = kWavesPerBlock * mfma_slab_floats<MODE>() * 4;
  constexpr int kLdsFloats = TAIL && int(sizeof(SweepTailLds)) + 16 > kSlabBytes ? (int(sizeof(SweepTailLds)) + 16 + 3) / 4 : kSlabBytes / 4;   // (the step's LDS lies over the slabs)
  __shared__ __attribute__((aligned(16))) float slab_mem[kLdsFloats];
  __shared__ int counts[kWavesPerBlock];
  if constexpr (RPW <= 2) {
    // Short tiles (coarse levels, small batches): the prologue is most of a workgroup's life.  The pair's plane pointers
    // are requested together with the activity flag and pinned above the branch, so the scalar loads travel together
    // instead of as a chain of dependent round trips (the compiler otherwise sinks each to its first use).  Tall tiles run
    // measurably better with the lazy order.
    const PairPtrs early = pairs[pair];
    const int active = st.active && st.level == g.level;
    asm volatile("" ::"s"(early.refR), "s"(early.curA), "s"(early.curB), "s"(active));
    if (!active) {
      if constexpr (TAIL == 1) sweep_tail(g, tail, pair, reinterpret_cast<char*>(slab_mem));
      return;
    }
  } else {
    if (!st.active || st.level != g.level) {
      if constexpr (TAIL == 1) sweep_tail(g, tail, pair, reinterpret_cast<char*>(slab_mem));
      return;
    }
  }
  const PairPtrs pp = pairs[pair];
  DVO_TCLK_START(tc0);
  mfma_sweep_tile<RPW, LINEAR, MODE, TAIL != 0>(g, st.KT, st.P_prev, st.first != 0, pp, pair, tile, partials, scratch, slab_mem, counts, f16_range_flag);
  if constexpr (TAIL) {
    __syncthreads();                                           // (the fold has read every wavefront's slab)
    DVO_TCLK(0, tc0);                                          // the tile
    if constexpr (TAIL == 1) sweep_tail(g, tail, pair, reinterpret_cast<char*>(slab_mem));
    else sweep_tail_wide(g, tail, pair, reinterpret_cast<char*>(slab_mem));
  }
}

template <int RPW, int F16>
static void launch_m(hipStream_t s, bool finest, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                     float* partials, float2* scratch, int* f16_range_flag, const SolverStepArgs* tail) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(kBlock);
  const SolverStepArgs no_tail = {};
  if constexpr (F16 == 2 && RPW <= 8) {
    if (tail) {                                                // (the default schedule's gathering sweep, never the finest level: mfma_sweep_has_tail)
      if (tail->pair_sums) {
        if (g.linear) k_residual_reduce_mfma<RPW, false, true, 2, 2><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, *tail);
        else k_residual_reduce_mfma<RPW, false, false, 2, 2><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, *tail);
      } else {
        if (g.linear) k_residual_reduce_mfma<RPW, false, true, 2, 1><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, *tail);
        else k_residual_reduce_mfma<RPW, false, false, 2, 1><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, *tail);
      }
      return;
    }
  }
  if (g.linear) {
    if (finest) k_residual_reduce_mfma<RPW, true, true, F16, 0><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, no_tail);
    else k_residual_reduce_mfma<RPW, false, true, F16, 0><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, no_tail);
  } else {
    if (finest) k_residual_reduce_mfma<RPW, true, false, F16, 0><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, no_tail);
    else k_residual_reduce_mfma<RPW, false, false, F16, 0><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, f16_range_flag, no_tail);
  }
}

template <int F16>
static void launch_rpw(hipStream_t s, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs, const PairState* states,
                       int n_pairs, float* partials, float2* scratch, int* f16_range_flag, const SolverStepArgs* tail) {
  switch (rows_per_wave) {
    case 1: launch_m<1, F16>(s, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, tail); break;
    case 2: launch_m<2, F16>(s, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, tail); break;
    case 4: launch_m<4, F16>(s, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, tail); break;
    case 16: launch_m<16, F16>(s, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, nullptr); break;
    default: launch_m<8, F16>(s, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, tail); break;
  }
}

// the instantiations with a tail: the contracted gathering sweep (what variant 8 runs where the window sweep does not take the level),
// tiles of up to 8 rows per wavefront, no "ref_compat"
bool mfma_sweep_has_tail(int variant, int rows_per_wave, const LevelGeom& g) { return variant >= 8 && !g.rcp_table && rows_per_wave != 16; }

void launch_residual_reduce_mfma(hipStream_t s, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs,
                                 const PairState* states, int n_pairs, float* partials, float2* scratch, int mode, int* f16_range_flag,
                                 const SolverStepArgs* tail) {
  if (mode == 2 && g.rcp_table) mode = 1;                       // (option "ref_compat": the exact arithmetic with the table's reciprocal)
  if (mode == 2) launch_rpw<2>(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, tail);
  else if (mode == 1) launch_rpw<1>(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch, f16_range_flag, nullptr);
  else launch_rpw<0>(s, rows_per_wave, finest, g, pairs, states, n_pairs, partials, scratch, nullptr, nullptr);
}

#ifdef DVO_TAIL_CLOCKS
extern "C" int dvo_hip_debug_tail_clocks(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tail_clk), sizeof(g_tail_clk)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

}  // namespace dvo_hip
