// solver_kernels.hip -- per-pair Gauss-Newton bookkeeping on the device (one workgroup per frame pair).
//
// k_solver_step finishes the deterministic reduction (stage 3: the per-tile float partials are summed
// in tile order in float64), then lane 0 runs the reference's loop body after the residual sweep:
// precision, log-likelihood, accept/revert, normal-equation contraction, 6x6 solve, SE(3) update,
// termination (dvo_core/src/dense_tracking.cpp:273-363; logic in solver_logic.h).  Keeping this on the
// device removes the per-iteration D2H/H2D round trip a host-driven loop would need; the host only
// polls one integer (pairs still active) every few iterations.
#include "align_common.h"
#ifdef DVO_SOLVER_CLOCKS
// experiment build only: the stages of gn_step (solver_logic.h, DVO_GN_CLK) on thread 0 of workgroup 0, 100 MHz wall clock
namespace dvo_hip {
__device__ unsigned long long g_gn_clk[16];
__device__ unsigned long long g_gn_prev;
}
#define DVO_GN_CLK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); if ((i) > 0) atomicAdd(&dvo_hip::g_gn_clk[(i)], now_ - dvo_hip::g_gn_prev); else atomicAdd(&dvo_hip::g_gn_clk[0], 1ull); dvo_hip::g_gn_prev = now_; } } while (0)
#endif
#include "solver_logic.h"

namespace dvo_hip {

__global__ void k_init_pairs(PairState* states, int n_pairs, SolverParams prm, const double* __restrict__ T_init) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  gn_init_pair(states[p], prm, T_init + size_t(p) * 16);
}

// T_init != null: the first level of a match -- the pair is initialised here as well (one launch less per match)
__global__ void k_level_begin(PairState* states, int n_pairs, SolverParams prm, LevelGeom g, int level,
                              const PairPtrs* __restrict__ pairs, dvo_hip_level_stats* levels, const double* __restrict__ T_init) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  if (T_init) gn_init_pair(states[p], prm, T_init + size_t(p) * 16);
  gn_level_begin(states[p], prm, g, level, *pairs[p].n_selected, levels + size_t(p) * prm.cap_levels);
}

// words of a POD copied cooperatively between global memory and LDS
template <typename T>
__device__ __forceinline__ void coop_copy(T* dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (int i = threadIdx.x; i < int(sizeof(T) / 4); i += blockDim.x) d[i] = s[i];
}

// How the host learns that a level is finished without a copy, an event or a synchronisation: every workgroup of a step
// adds (1 | active << 32) to the step's tally; the one that completes the count stores "done | pairs still active" into
// the step's word of a pinned host array, which the host thread polls.  (The previous 4-byte D2H copy + event per poll
// cost ~10 us of idle GPU each -- 8 % of a 128-pair match.)
__device__ __forceinline__ void publish_step(unsigned long long* step_tally, int* host_status, int n_pairs, bool active) {
  const unsigned long long add = 1ull + (active ? (1ull << 32) : 0ull);
  const unsigned long long now = atomicAdd(step_tally, add) + add;
  if ((now & 0xffffffffull) == static_cast<unsigned long long>(n_pairs))
    __hip_atomic_store(host_status, int(now >> 32) | kStepDoneFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifdef DVO_SOLVER_CLOCKS
// experiment build only (scripts/ubench/solver_clocks.sh): where the time of a solver step goes, 100 MHz wall clock
// (one row of 16 per level class: 640, 320, 160 pixels wide and narrower; [15] of a row counts the steps)
__device__ unsigned long long g_solver_clk[64];
#define CLK_ROW (g.w >= 640 ? 0 : g.w >= 320 ? 16 : g.w >= 160 ? 32 : 48)
// (every 64th workgroup, so that a large batch shows what a step costs with the memory system under load, not only its first workgroup)
#define CLK(i) do { if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_solver_clk[CLK_ROW + (i)], now_ - clk_prev_); clk_prev_ = now_; } } while (0)
extern "C" int dvo_hip_debug_gn_clocks(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gn_clk), sizeof(g_gn_clk)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gn_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int dvo_hip_debug_solver_clocks(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_solver_clk), sizeof(g_solver_clk)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[64] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_solver_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define CLK(i) ((void)0)
#endif

// WAVES: 4, or 2 for the small levels of a batch that does not fit the device in one go with four -- the kernel holds 210 registers,
// two four-wavefront workgroups per compute unit = 512 pairs at a time, and what a workgroup does on a small level is mostly lane 0's
// serial float64 work.  A two-wavefront workgroup plays the four (reduce_scale.h, loglik_partial_played): the records are the same bits.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_solver_step(PairState* states, int n_pairs, SolverParams prm, LevelGeom g,
                                                        const float* __restrict__ partials,
                                                        const double* __restrict__ ll_partials, int ll_blocks_per_pair,
                                                        const float2* __restrict__ scratch_for_fused_ll,
                                                        dvo_hip_level_stats* levels, dvo_hip_iteration_stats* iters,
                                                        unsigned long long* step_tally, int* host_status, int level_slot_hint, NextLevel next) {
  const int pair = blockIdx.x;
#ifdef DVO_SOLVER_CLOCKS
  unsigned long long clk_prev_ = wall_clock64();
#endif
  // The state machine is one lane of serial float64 work; every global access it made used to be a dependent
  // ~1 us round trip.  Stage the pair's state, its level record and the new iteration record in LDS: loaded and
  // stored by all 256 lanes at once, touched by lane 0 at LDS latency.
  __shared__ PairState st;
  __shared__ dvo_hip_level_stats lvl;
  // [1]: the new iteration record; [0]: the one before it, fetched only for a pair whose result is written here (gn_finish reads either)
  __shared__ dvo_hip_iteration_stats recs[2];
  __shared__ dvo_hip_result res;
  dvo_hip_iteration_stats& rec = recs[1];
  __shared__ double sh[kWavesPerBlock * kAccStride];
  __shared__ double sums[kAccStride];
  __shared__ double ll_waves[kWavesPerBlock];
  __shared__ double ll_stage[32];
  __shared__ double Amat[36], bvec[6];         // J^T W J and J^T W r of this pass, contracted by 42 lanes (GnAssist)
  __shared__ int information_ready;
  __shared__ int rec_index;
  // ONE round trip for everything whose address does not depend on loaded data (round 4; with 1024 workgroups in flight a dependent
  // trip costs 4-5 us and the step was a chain of nine): the pair's state, its level record (the slot every live pair of the batch
  // is at -- a hint from the host, checked below), the partial rows, the log-likelihood partial sums of k_loglik.  A pair that turns
  // out to be finished has loaded them for nothing.
  constexpr int kThreads = WAVES * 64;
  constexpr int kStWords = int(sizeof(PairState) / 4), kLvlWords = int(sizeof(dvo_hip_level_stats) / 4);
  static_assert(sizeof(PairState) % 4 == 0 && sizeof(dvo_hip_level_stats) % 4 == 0, "copied by words");
  constexpr int kStPer = (kStWords + kThreads - 1) / kThreads, kLvlPer = (kLvlWords + kThreads - 1) / kThreads;
  unsigned st_w[kStPer], lvl_w[kLvlPer];
  const bool hint_ok = level_slot_hint >= 0 && level_slot_hint < prm.cap_levels;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&states[pair]);
#pragma unroll
    for (int k = 0; k < kStPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      st_w[k] = src[i < kStWords ? i : 0];
    }
    const unsigned* lsrc = reinterpret_cast<const unsigned*>(levels + size_t(pair) * prm.cap_levels + (hint_ok ? level_slot_hint : 0));
#pragma unroll
    for (int k = 0; k < kLvlPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      lvl_w[k] = prm.cap_levels > 0 ? lsrc[i < kLvlWords ? i : 0] : 0u;
    }
  }
  double ll_mine = 0.0;
  if (!scratch_for_fused_ll && threadIdx.x < 32) ll_mine = ll_partials[size_t(pair) * ll_blocks_per_pair + min(int(threadIdx.x), ll_blocks_per_pair - 1)];
  reduce_partials<WAVES>(partials, pair, g.tiles_x * g.tiles_y, sh, sums);   // same routine, same order as k_loglik: identical n, S, P
  {
    unsigned* dst = reinterpret_cast<unsigned*>(&st);
#pragma unroll
    for (int k = 0; k < kStPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      if (i < kStWords) dst[i] = st_w[k];
    }
    unsigned* ldst = reinterpret_cast<unsigned*>(&lvl);
#pragma unroll
    for (int k = 0; k < kLvlPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      if (i < kLvlWords) ldst[i] = lvl_w[k];
    }
    if (threadIdx.x < 32) ll_stage[threadIdx.x] = ll_mine;
    // what the serial lane would otherwise do one store at a time: every word of the new iteration record NaN (SolverParams::record_prefilled)
    static_assert(sizeof(dvo_hip_iteration_stats) % 8 == 0, "prefilled by 8-byte words");
    for (int i = threadIdx.x; i < int(sizeof(dvo_hip_iteration_stats) / 8); i += kThreads) reinterpret_cast<double*>(&rec)[i] = dvo_nan();
    // ... and the contraction of the Gram sums with the pass' precision (the sums are in place since reduce_partials' barrier): lane
    // i * 6 + j forms A(i, j) from the upper-triangle entry of (min, max), lanes 36..41 J^T W r
    if (threadIdx.x < 42) {
      const double d = sums[kAccN] - 3.0;
      float Cc[3], Pc[4];
      scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, Cc, Pc);
      const double p00 = double(Pc[0]), p01 = double(Pc[1]), p11 = double(Pc[3]);
      const int k = threadIdx.x;
      if (k < 36) {
        const int i = k / 6, j = k - i * 6, lo = i < j ? i : j, hi = i < j ? j : i;
        const int o = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);           // index of (lo, hi) in the row-major upper triangle
        Amat[k] = gn_contract(p00, p01, p11, sums[kAccJ00 + o], sums[kAccJ01 + o], sums[kAccJ11 + o]);
      } else {
        bvec[k - 36] = gn_contract(p00, p01, p11, sums[kAccB00 + k - 36], sums[kAccB01 + k - 36], sums[kAccB11 + k - 36]);
      }
    }
  }
  __syncthreads();
  if (!st.active || st.level != g.level) {   // uniform: the pair is not (or no longer) on this level
    if (threadIdx.x == 0) publish_step(step_tally, host_status, n_pairs, false);
    // The hand-over (NextLevel): a pair that ENDED this level in an earlier step begins the next level here -- or, on the last level,
    // has its result written -- in the shadow of the workgroups that still iterate: no launch between two levels, none behind the
    // last.  (Doing it in the very step that ends the level was measured and dropped: that workgroup is the launch's slowest.)
    if (st.level != g.level || st.finished) return;
    const int slot_now = st.n_levels - 1;
    const bool have_now = slot_now >= 0 && slot_now < prm.cap_levels;
    if (next.valid) {
      __shared__ dvo_hip_level_stats lvl_next;
      __shared__ int stored;
      if (threadIdx.x == 0) {
        const int n_selected_next = *next.pairs[pair].n_selected;
        LevelGeom gn = g;
        gn.fx = next.fx; gn.fy = next.fy; gn.ox = next.ox; gn.oy = next.oy;
        SolverParams begin = prm;
        begin.cap_levels = st.n_levels < prm.cap_levels ? st.n_levels + 1 : 0;
        stored = begin.cap_levels ? st.n_levels : -1;
        gn_level_begin(st, begin, gn, next.level, n_selected_next, &lvl_next - st.n_levels);   // (dense_tracking.cpp:200-238)
      }
      __syncthreads();
      coop_copy(&states[pair], &st);
      if (stored >= 0) coop_copy(levels + size_t(pair) * prm.cap_levels + stored, &lvl_next);
    } else if (next.results) {
      // dense_tracking.cpp:368-373: the last iteration with an increment is the level's last record or the one before it
      const int last = st.n_iters_total - 1;
      if (have_now && !(hint_ok && slot_now == level_slot_hint)) coop_copy(&lvl, levels + size_t(pair) * prm.cap_levels + slot_now);
      if (last >= 0 && last < prm.cap_iters) coop_copy(&recs[1], iters + size_t(pair) * prm.cap_iters + last);
      if (last >= 1 && last - 1 < prm.cap_iters) coop_copy(&recs[0], iters + size_t(pair) * prm.cap_iters + last - 1);
      __syncthreads();
      if (threadIdx.x == 0) {
        gn_finish(st, prm, &lvl - slot_now, &recs[1] - last, &res);
        st.finished = 1;
      }
      __syncthreads();
      coop_copy(next.results + pair, &res);
      if (threadIdx.x == 0) states[pair].finished = 1;
    }
    return;
  }
  CLK(0);
  const int level_slot = st.n_levels - 1;
  dvo_hip_level_stats* lvl_global = levels + size_t(pair) * prm.cap_levels + level_slot;
  const bool have_level = level_slot >= 0 && level_slot < prm.cap_levels;
  if (have_level && !(hint_ok && level_slot == level_slot_hint)) {   // (uniform; not expected: the levels of a batch begin together)
    coop_copy(&lvl, lvl_global);
    __syncthreads();
  }
  CLK(1);
  if (scratch_for_fused_ll) {
    // coarse levels: the log-likelihood sweep is small enough for this workgroup, which saves a launch per iteration
    float C[3], P[4];
    const int n = scale_from_sums(sums, C, P);
    constexpr int kPlayed = kWavesPerBlock / WAVES;          // wavefronts of a four-wavefront workgroup each real one plays
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double t[kPlayed];
#pragma unroll
    for (int q = 0; q < kPlayed; ++q) t[q] = 0.0;
    if (n >= 6) {
      const int tiles = g.tiles_x * g.tiles_y;
      if (g.compact) {                                        // (uniform) the packed residuals of the contracted window sweep
        if constexpr (WAVES == kWavesPerBlock)
          t[0] = loglik_partial_compact<16>(scratch_for_fused_ll + size_t(pair) * residual_entries(g), partials + size_t(pair) * tiles * kAccStride, tiles * 4, P,
                                           wave, kWavesPerBlock);
        else
          loglik_partial_compact_played<4, kPlayed>(scratch_for_fused_ll + size_t(pair) * residual_entries(g), partials + size_t(pair) * tiles * kAccStride,
                                                    tiles * 4, P, wave, WAVES, kWavesPerBlock, t);
      } else {
        if constexpr (WAVES == kWavesPerBlock) t[0] = loglik_partial<16>(scratch_for_fused_ll + size_t(pair) * g.w * g.h, g.w * g.h, P, 0, 1);
        else loglik_partial_played<8, WAVES>(scratch_for_fused_ll + size_t(pair) * g.w * g.h, g.w * g.h, P, t);
      }
    }
#pragma unroll
    for (int q = 0; q < kPlayed; ++q) {
      const double total = wave_sum_double(t[q]);
      if ((threadIdx.x & 63) == 0) ll_waves[wave + q * WAVES] = total;
    }
    __syncthreads();
  }
  CLK(2);
  if (threadIdx.x == 0) {
    double ll_sum = 0.0;
    if (scratch_for_fused_ll) {
      ll_sum = (ll_waves[0] + ll_waves[1]) + (ll_waves[2] + ll_waves[3]);
    } else {
      for (int b = 0; b < ll_blocks_per_pair; ++b) ll_sum += ll_stage[b];
    }
    rec_index = st.n_iters_total;
    // gn_step addresses levels[n_levels - 1] and iters[n_iters_total]: hand it pointers biased so that those land in LDS
    SolverParams local = prm;
    local.cap_levels = have_level ? level_slot + 1 : 0;
    local.cap_iters = rec_index + 1;
    local.record_prefilled = 1;
    GnAssist assist = {Amat, bvec, 1, 0};
    gn_step(st, local, g, sums, ll_sum, &lvl - level_slot, &rec - rec_index, nullptr, &assist);
    information_ready = assist.information_ready;
    CLK(3);
    publish_step(step_tally, host_status, n_pairs, st.active != 0);
    CLK(4);
  }
  __syncthreads();
  if (information_ready && threadIdx.x < 36) rec.information[threadIdx.x] = Amat[threadIdx.x];   // (uniform; GnAssist::defer_information)
  __syncthreads();
  coop_copy(&states[pair], &st);
  if (have_level) coop_copy(lvl_global, &lvl);
  if (rec_index < prm.cap_iters) coop_copy(iters + size_t(pair) * prm.cap_iters + rec_index, &rec);
  CLK(5);
#ifdef DVO_SOLVER_CLOCKS
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_solver_clk[CLK_ROW + 15], 1ull);
#endif
}

__global__ void k_finish(const PairState* states, int n_pairs, SolverParams prm, const dvo_hip_level_stats* levels,
                         const dvo_hip_iteration_stats* iters, dvo_hip_result* results) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  gn_finish(states[p], prm, levels + size_t(p) * prm.cap_levels, iters + size_t(p) * prm.cap_iters, results + p);
}

// ---- single-shot linearisation (parity entry point dvo_hip_level_iteration) ---------------------------
__global__ void k_set_fixed_state(PairState* states, LevelGeom g, const float* __restrict__ T34, const float* __restrict__ Pprev, int first) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  PairState& st = states[0];
  float T[12];
  for (int i = 0; i < 12; ++i) T[i] = T34[i];
  make_KT(g.fx, g.fy, g.ox, g.oy, T, st.KT);
  for (int i = 0; i < 4; ++i) st.P_prev[i] = Pprev[i];
  st.first = first;
  st.active = 1;
  st.level = g.level;
}

__global__ __launch_bounds__(kBlock) void k_single_shot_out(LevelGeom g, const float* __restrict__ partials,
                                                            const double* __restrict__ ll_partials, int ll_blocks_per_pair,
                                                            int n_selected, dvo_hip_iteration_out* out) {
  __shared__ double sh[kWavesPerBlock * kAccStride];
  __shared__ double sums[kAccStride];
  float C[3], P[4];
  reduce_partials(partials, 0, g.tiles_x * g.tiles_y, sh, sums);
  const int n = scale_from_sums(sums, C, P);
  const int tid = threadIdx.x;
  if (tid != 0) return;
  out->n = n;
  out->n_selected = n_selected;
  out->sum_w = 0.0;
  for (int i = 0; i < 3; ++i) out->scale_cov[i] = C[i];
  for (int i = 0; i < 4; ++i) out->precision[i] = P[i];
  double ll_sum = 0.0;
  for (int b = 0; b < ll_blocks_per_pair; ++b) ll_sum += ll_partials[b];
  const double det = double(P[0]) * double(P[3]) - double(P[1]) * double(P[2]);
  out->neg_loglik = -double(float(0.5 * double(n) * log(det) - 3.5 * ll_sum));   // float like the reference's (gn_step)
  const double p00 = double(P[0]), p01 = double(P[1]), p11 = double(P[3]);
  int o = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      const double a = p00 * sums[kAccJ00 + o] + p01 * sums[kAccJ01 + o] + p11 * sums[kAccJ11 + o];
      out->A[i * 6 + j] = a;
      out->A[j * 6 + i] = a;
      ++o;
    }
  for (int i = 0; i < 6; ++i) out->b[i] = -(p00 * sums[kAccB00 + i] + p01 * sums[kAccB01 + i] + p11 * sums[kAccB11 + i]);
}

void launch_init_pairs(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, const double* T_init) {
  k_init_pairs<<<dim3((n_pairs + 63) / 64), dim3(64), 0, s>>>(states, n_pairs, prm, T_init);
}

void launch_level_begin(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, LevelGeom g, int level,
                        const PairPtrs* pairs, dvo_hip_level_stats* levels, const double* T_init_or_null) {
  k_level_begin<<<dim3((n_pairs + 63) / 64), dim3(64), 0, s>>>(states, n_pairs, prm, g, level, pairs, levels, T_init_or_null);
}

void launch_solver_step(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, LevelGeom g,
                        const float* partials, const double* ll_partials, int ll_blocks_per_pair, const float2* scratch_for_fused_ll,
                        dvo_hip_level_stats* levels, dvo_hip_iteration_stats* iters, unsigned long long* step_tally, int* host_status, bool two_waves,
                        int level_slot_hint, const NextLevel* next_or_null) {
  NextLevel next;
  next.valid = 0; next.level = 0; next.fx = next.fy = next.ox = next.oy = 0.0f; next.pairs = nullptr; next.results = nullptr;
  if (next_or_null) next = *next_or_null;
  // (two wavefronts: see the kernel; a level of at most 32 tiles -- 160 x 120, 80 x 60 -- of a batch beyond two workgroups per compute unit)
  if (two_waves)
    k_solver_step<2><<<dim3(n_pairs), dim3(128), 0, s>>>(states, n_pairs, prm, g, partials, ll_partials, ll_blocks_per_pair,
                                                         scratch_for_fused_ll, levels, iters, step_tally, host_status, level_slot_hint, next);
  else
    k_solver_step<kWavesPerBlock><<<dim3(n_pairs), dim3(kBlock), 0, s>>>(states, n_pairs, prm, g, partials, ll_partials, ll_blocks_per_pair,
                                                                         scratch_for_fused_ll, levels, iters, step_tally, host_status, level_slot_hint, next);
}

void launch_finish(hipStream_t s, const PairState* states, int n_pairs, SolverParams prm,
                   const dvo_hip_level_stats* levels, const dvo_hip_iteration_stats* iters, dvo_hip_result* results) {
  k_finish<<<dim3((n_pairs + 63) / 64), dim3(64), 0, s>>>(states, n_pairs, prm, levels, iters, results);
}

__global__ void k_force_active(PairState* states, int n_pairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  states[p].active = 1;
  if (states[p].iteration > 0) states[p].first = 0;
}

void launch_force_active(hipStream_t s, PairState* states, int n_pairs) {
  k_force_active<<<dim3((n_pairs + 63) / 64), dim3(64), 0, s>>>(states, n_pairs);
}

void launch_set_fixed_state(hipStream_t s, PairState* states, LevelGeom g, const float* T34_dev, const float* Pprev_dev, int first) {
  k_set_fixed_state<<<dim3(1), dim3(64), 0, s>>>(states, g, T34_dev, Pprev_dev, first);
}

void launch_single_shot_out(hipStream_t s, LevelGeom g, const float* partials, const double* ll_partials, int ll_blocks_per_pair,
                            int n_selected, dvo_hip_iteration_out* out_dev) {
  k_single_shot_out<<<dim3(1), dim3(kBlock), 0, s>>>(g, partials, ll_partials, ll_blocks_per_pair, n_selected, out_dev);
}

}  // namespace dvo_hip
