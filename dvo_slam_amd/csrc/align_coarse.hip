// align_coarse.hip -- the coarse pyramid levels of a batch in ONE launch: a workgroup owns a pair from gn_level_begin to the
// termination of the level, level after level, with the solver step inline.
//
// The launch path (capi_schedule.inc::run_batch) spends two launches per Gauss-Newton iteration of a level -- the sweep and the solver step --
// for the WHOLE batch, iteration after iteration, until the last pair has left the level.  On the fine levels that is the right shape:
// a sweep is hundreds of microseconds of issue-bound work per launch.  On the coarse levels (160 x 120 and below) it is not: at 1024
// pairs a level-3 iteration is 50 us of sweep at 0.31 of the kernel's roofline and a 36-us solver launch, fifteen times, the later ones
// for a handful of pairs (profiles/r05_step_breakdown.txt: levels 2-3, their solver steps and tails are 3.7 of a step's 11.5 ms); at 128
// pairs the same chain is 24 dependent launches of 12-17 us.  In the reference one thread runs one match() from start to finish without
// meeting anybody (dvo_core/src/dense_tracking.cpp:200-357; the pairs are spread over threads, dvo_slam/src/keyframe_graph.cpp:576-593).
// This kernel does the same with a workgroup:
//
//   for level = first .. last:                          (gn_level_begin, dense_tracking.cpp:200-238)
//     repeat                                            (dense_tracking.cpp:240-357)
//       sweep    every tile of the level, one after the other, with the launch path's own tile functions: fast_sweep_tile (fast_sweep.h: the
//                current frame's window in LDS, contracted arithmetic, f16 Gram, packed residual pairs) where the window sweep takes the
//                level, mfma_sweep_tile (mfma_sweep.h: gathered taps) below; per-tile partial rows and residual pairs go to the same
//                buffers in the same layout
//       reduce   reduce_partials (stage 3 of the deterministic reduction, reduce_scale.h)
//       loglik   loglik_partial_compact / loglik_partial over the workgroup's own residual pairs (the solver step's fused form)
//       body     gn_step on lane 0 (solver_logic.h), the state in LDS for the whole run of levels
//     until the pair leaves the level
//
// Every stage is the launch path's device function with the launch path's data layout, so a pair's records are the launch path's BIT FOR
// BIT (tests/test_gpu_coarse.py) -- the template parameters that differ (loads in flight of the reduction, slots per round of the
// log-likelihood) change how many loads travel together, never the order of an addition or a product.  A workgroup never waits for
// another one: no residency requirement, no time-out, any batch size; pairs that need 4 iterations leave after 4 and the hardware
// dispatcher hands their compute unit to the next pair -- or to the background ingest of the caller's next batch.
//
// Resources: 256 threads, at most 128 registers (four workgroups per compute unit: the sweep was measured level at four and five; the
// serial float64 lane of gn_step spills ~40 registers under that cap, in the shadow of the other workgroups' sweeps), 34.6 KB of LDS
// (window 21.5 KB + operand slabs 10 KB + state, record and sums 3 KB; the reduction's staging area lives in the idle window).
#include "fast_sweep.h"
#include "mfma_sweep.h"
#include "solver_logic.h"

namespace dvo_hip {

namespace {

__device__ __forceinline__ int coarse_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float coarse_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

template <typename T>
__device__ __forceinline__ void coarse_copy(T* dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (int i = threadIdx.x; i < int(sizeof(T) / 4); i += kBlock) d[i] = s[i];
}

constexpr int kCoarseReduceInFlight = 8;              // a level of the fused kernel has at most a few dozen tiles: two rounds at most
constexpr int kCoarseLlSlots = 4, kCoarseLlLoads = 4; // (see the file header: the order of the products does not depend on them)

}  // namespace

template <int WG_PER_CU>
__global__ __launch_bounds__(kBlock, WG_PER_CU) void k_match_coarse(const CoarseArgs a) {
  __shared__ __attribute__((aligned(16))) float slab[kWavesPerBlock][kSlabFloatsF16];
  __shared__ __attribute__((aligned(16))) float2 win[kFastCells];
  __shared__ __attribute__((aligned(16))) int bbox[kWavesPerBlock][2];
  __shared__ int counts[kWavesPerBlock];
  __shared__ PairState st;
  __shared__ dvo_hip_level_stats lvl;
  __shared__ dvo_hip_iteration_stats rec;
  __shared__ double sums[kAccStride];
  __shared__ double ll_waves[kWavesPerBlock];
  __shared__ double Amat[36], bvec[6];
  __shared__ int information_ready, rec_index;
  static_assert(sizeof(win) >= sizeof(double) * kWavesPerBlock * kAccStride, "the reduction's staging area borrows the window");
  double* sh = reinterpret_cast<double*>(win);                 // (reduce_partials: the window is idle between two sweeps)

  const int pair = blockIdx.x;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.T_init) {
    if (tid == 0) gn_init_pair(st, a.prm, a.T_init + size_t(pair) * 16);
  } else {
    coarse_copy(&st, &a.states[pair]);
  }
  __syncthreads();
  const FastLds lds = {slab, win, bbox, counts};
  const FastRcpSource no_table = {__builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, 0x00020000), nullptr, 0};

  for (int level = a.first_level; level >= a.last_level; --level) {
    const LevelGeom& g = a.geom[level];
    const PairPtrs pp = a.pair_ptrs[size_t(level) * a.n_pairs + pair];
    const int level_slot = coarse_uniform(st.n_levels);        // the record gn_level_begin is about to open
    const bool have_level = level_slot < a.prm.cap_levels;
    SolverParams local = a.prm;                                // gn_* address levels[n_levels - 1] and iters[n_iters_total]:
    local.cap_levels = have_level ? level_slot + 1 : 0;        // biased so that those land in LDS
    local.record_prefilled = 1;
    __syncthreads();
    if (tid == 0) gn_level_begin(st, local, g, level, *pp.n_selected, &lvl - level_slot);
    __syncthreads();
    const int tiles = g.tiles_x * g.tiles_y;
    const bool window_level = !g.linear && g.compact;           // (uniform; what launch_residual_reduce sends to the contracted window sweep)
    // (the pair's OWN region of the two buffers, whatever level it is on: the pairs of a launch are on different levels at the same
    // time, and the launch path's [pair][tiles of the level] layout would let a pair on level 2 write over its neighbours on level 3.
    // The shared device functions address "pair 0" of the pre-offset pointers.)
    float* my_partials = a.partials + size_t(pair) * a.max_tiles * kAccStride;
    float2* my_residuals = a.scratch + size_t(pair) * a.max_entries;
    int* my_flag = a.f16_range_flag ? a.f16_range_flag + pair : nullptr;

    for (;;) {
      // ---- the sweep: the pose and the weights' precision out of LDS into scalar registers, then tile after tile ----------------
      float KT[12], Pp[4];
#pragma unroll
      for (int i = 0; i < 12; ++i) KT[i] = coarse_uniform(st.KT[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) Pp[i] = coarse_uniform(st.P_prev[i]);
      const bool first = coarse_uniform(st.first) != 0;
      if (window_level) {
        const FastWeights wt(Pp, first);
        if (g.w % kTileW != 0) {
#pragma unroll 1
          for (int tile = 0; tile < tiles; ++tile)
            fast_sweep_tile<2, true, true, 0, false>(g, KT, wt, pp, 0, tile, my_partials, my_residuals, lds, no_table, a.fallback_count, my_flag);
        } else {
#pragma unroll 1
          for (int tile = 0; tile < tiles; ++tile)
            fast_sweep_tile<2, false, true, 0, false>(g, KT, wt, pp, 0, tile, my_partials, my_residuals, lds, no_table, a.fallback_count, my_flag);
        }
      } else if (g.linear) {
#pragma unroll 1
        for (int tile = 0; tile < tiles; ++tile) {
          mfma_sweep_tile<kCoarseRowsPerWave, true, 2>(g, KT, Pp, first, pp, 0, tile, my_partials, my_residuals, &slab[0][0], counts, my_flag);
          __syncthreads();                                     // (the fold reads every wavefront's slab)
        }
      } else {
#pragma unroll 1
        for (int tile = 0; tile < tiles; ++tile) {
          mfma_sweep_tile<kCoarseRowsPerWave, false, 2>(g, KT, Pp, first, pp, 0, tile, my_partials, my_residuals, &slab[0][0], counts, my_flag);
          __syncthreads();
        }
      }
      __syncthreads();                                         // partial rows and residual pairs of every tile are in memory (one compute unit: one L1)

      // ---- the solver step (solver_kernels.hip::k_solver_step with the fused log-likelihood, four wavefronts) -------------------
      reduce_partials<kWavesPerBlock, kCoarseReduceInFlight>(my_partials, 0, tiles, sh, sums);
      for (int i = tid; i < int(sizeof(dvo_hip_iteration_stats) / 8); i += kBlock) reinterpret_cast<double*>(&rec)[i] = dvo_nan();
      if (tid < 42) {                                          // A = J^T W J, b = J^T W r: the Gram sums contracted with the pass' precision (GnAssist)
        const double d = sums[kAccN] - 3.0;
        float Cc[3], Pc[4];
        scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, Cc, Pc);
        const double p00 = double(Pc[0]), p01 = double(Pc[1]), p11 = double(Pc[3]);
        const int k = tid;
        if (k < 36) {
          const int i = k / 6, j = k - i * 6, lo = i < j ? i : j, hi = i < j ? j : i;
          const int o = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);
          Amat[k] = gn_contract(p00, p01, p11, sums[kAccJ00 + o], sums[kAccJ01 + o], sums[kAccJ11 + o]);
        } else {
          bvec[k - 36] = gn_contract(p00, p01, p11, sums[kAccB00 + k - 36], sums[kAccB01 + k - 36], sums[kAccB11 + k - 36]);
        }
      }
      {
        float C[3], P[4];
        const int n = scale_from_sums(sums, C, P);
        double t = 0.0;
        if (n >= 6) {
          if (g.compact) t = loglik_partial_compact<kCoarseLlSlots>(my_residuals, my_partials, tiles * 4, P, wave, kWavesPerBlock);
          else t = loglik_partial<kCoarseLlLoads>(my_residuals, g.w * g.h, P, 0, 1);
        }
        const double total = wave_sum_double(t);
        if ((tid & 63) == 0) ll_waves[wave] = total;
      }
      __syncthreads();
      if (tid == 0) {
        const double ll_sum = (ll_waves[0] + ll_waves[1]) + (ll_waves[2] + ll_waves[3]);
        rec_index = st.n_iters_total;
        SolverParams step = local;
        step.cap_iters = rec_index + 1;
        GnAssist assist = {Amat, bvec, 1, 0};
        gn_step(st, step, g, sums, ll_sum, &lvl - level_slot, &rec - rec_index, nullptr, &assist);
        information_ready = assist.information_ready;
      }
      __syncthreads();
      if (information_ready && tid < 36) rec.information[tid] = Amat[tid];   // (uniform; GnAssist::defer_information)
      __syncthreads();
      if (rec_index < a.prm.cap_iters) coarse_copy(a.iters + size_t(pair) * a.prm.cap_iters + rec_index, &rec);
      if (!coarse_uniform(st.active)) break;                   // the pair has left the level (gn_level_end)
    }
    __syncthreads();
    if (have_level) coarse_copy(a.levels + size_t(pair) * a.prm.cap_levels + level_slot, &lvl);
  }
  __syncthreads();
  coarse_copy(&a.states[pair], &st);
  if (a.results) {                                             // the match ends here: dense_tracking.cpp:368-373
    __threadfence();                                           // the level and iteration records this workgroup wrote are read back below
    __syncthreads();
    dvo_hip_result* result = reinterpret_cast<dvo_hip_result*>(win);   // composed in LDS (gn_finish reads its own output back)
    static_assert(sizeof(win) >= sizeof(dvo_hip_result), "the result is composed in the idle window");
    if (tid == 0) gn_finish(st, a.prm, a.levels + size_t(pair) * a.prm.cap_levels, a.iters + size_t(pair) * a.prm.cap_iters, result);
    __syncthreads();
    coarse_copy(a.results + pair, result);
    if (tid == 0) a.states[pair].finished = 1;
  }
}

// what the kernel is built for: the contracted window sweep with packed residual pairs (tile height 16), or the gathering sweep with the
// contracted arithmetic at kCoarseRowsPerWave rows per wavefront; no reciprocal table (option "ref_compat" stays on the launch path)
bool coarse_kernel_takes(const LevelGeom& g, bool window_level) {
  if (g.rcp_table || g.gram_hi_j || g.small) return false;   // (a small level's sweep reads plane C out of LDS, align_small.hip: no tile function of this kernel)
  if (window_level) return g.compact != 0 && !g.linear && fast_sweep_supports(g);
  const int th = kWavesPerBlock * kCoarseRowsPerWave;
  const int want_y = g.linear ? ((g.w * g.h + kTileW - 1) / kTileW + th - 1) / th : (g.h + th - 1) / th;
  return !g.compact && g.tiles_y == want_y && size_t(g.w) * g.h < (size_t(1) << 24);
}

hipError_t launch_match_coarse(hipStream_t s, const CoarseArgs& args, int workgroups_per_cu) {
  const dim3 grid(args.n_pairs), block(kBlock);
  if (workgroups_per_cu == 3) k_match_coarse<3><<<grid, block, 0, s>>>(args);
  else k_match_coarse<4><<<grid, block, 0, s>>>(args);
  return hipGetLastError();
}

}  // namespace dvo_hip
