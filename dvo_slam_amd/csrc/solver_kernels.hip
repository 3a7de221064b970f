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
#include "solver_step.h"

namespace dvo_hip {

__global__ void k_init_pairs(PairState* states, int n_pairs, SolverParams prm, const double* __restrict__ T_init) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  gn_init_pair(states[p], prm, T_init + size_t(p) * 16);
}

// T_init != null: the first level of a match -- the pair is initialised here as well (one launch less per match)
// flags != null (a batch with a slow lane, capi_schedule.inc::run_batch): which == 0 -- every pair but the flagged ones (the batch's own chain);
// which == 1 -- the flagged ones that have left level `from_level` (the slow lane's: its members arrive on different levels)
__global__ void k_level_begin(PairState* states, int n_pairs, SolverParams prm, LevelGeom g, int level,
                              const PairPtrs* __restrict__ pairs, dvo_hip_level_stats* levels, const double* __restrict__ T_init,
                              const unsigned char* __restrict__ flags, int which, int from_level) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  if (flags && int(flags[p] != 0) != which) return;
  if (from_level >= 0 && (states[p].level != from_level || states[p].active)) return;
  if (T_init) gn_init_pair(states[p], prm, T_init + size_t(p) * 16);
  gn_level_begin(states[p], prm, g, level, *pairs[p].n_selected, levels + size_t(p) * prm.cap_levels);
}

#ifdef DVO_SOLVER_CLOCKS
extern "C" int dvo_hip_debug_gn_clocks(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gn_clk), sizeof(g_gn_clk)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_gn_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// the step of one pair (solver_step.h::solver_step_body -- shared with the sweeps' tail, align_fast.hip / align_mfma.hip)
// MIN_WG: workgroups per compute unit the kernel is built for (1: the compiler's choice -- 168 registers = three four-wavefront workgroups
// since the pivoted 6 x 6 solve, a path of singular systems, is out of line (round 6: 184 before, two workgroups; 1024-pair step 11.6 ->
// 11.4 ms); 3: the same asked for by name -- option "solver_occupancy")
template <int WAVES, int MIN_WG = 1>
__global__ __launch_bounds__(WAVES * 64, MIN_WG) void k_solver_step(LevelGeom g, SolverStepArgs a) {
  __shared__ SolverLds L;
  const int pair = pair_of_launch_index(g, blockIdx.x);
  if (pair < 0) {                                             // (an empty entry of a straggler list: counted, nothing else)
    if (threadIdx.x == 0) publish_step(a.step_tally, a.host_status, a.n_pairs, false);
    return;
  }
  solver_step_body<WAVES>(L, g, a, pair);
}

// The stragglers of a level (round 6, the slow lane -- capi_schedule.inc::run_batch): the pairs still active on `level` get a flag byte, and EVERY
// flagged pair of the batch -- these and the ones of earlier levels -- goes into list[0 .. cap) in ascending order (-1 behind the last;
// the host's `cap` adds up the counts its polls saw, which only shrink).  One workgroup: a batch has a few thousand pairs at most, and
// the order must not depend on who arrives first.  (The state of a pair flagged earlier is not looked at: the slow lane is changing it.)
// list_only (the active-pair list of a level's last steps on the batch's own chain): nothing is flagged -- the pairs active on the level
// that are NOT flagged (not the lane's) go into the list.
__global__ __launch_bounds__(1024) void k_mark_stragglers(const PairState* __restrict__ states, int n_pairs, int level, unsigned char* __restrict__ flags,
                                                          int* __restrict__ list, int cap, int list_only) {
  __shared__ int wave_count[16];
  __shared__ int running;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < n_pairs; base += 1024) {
    const int p = base + int(threadIdx.x);
    const bool was = p < n_pairs && flags && flags[p] != 0;
    const bool on_level = p < n_pairs && !was && states[p].active != 0 && states[p].level == level;
    const bool is = list_only ? on_level : (was || on_level);
    if (!list_only && on_level) flags[p] = 1;
    const unsigned long long ballot = __ballot(is);
    if (lane == 0) wave_count[wave] = __popcll(ballot);
    __syncthreads();
    int before = running;
    for (int w2 = 0; w2 < wave; ++w2) before += wave_count[w2];
    const int pos = before + __popcll(ballot & ((1ull << lane) - 1ull));
    if (is && pos < cap) list[pos] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
      int total = 0;
      for (int w2 = 0; w2 < 16; ++w2) total += wave_count[w2];
      running += total;
    }
    __syncthreads();
  }
  for (int k = running + int(threadIdx.x); k < cap; k += 1024) list[k] = -1;
}

__global__ void k_clear_flags(unsigned char* flags, int n_pairs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_pairs) flags[p] = 0;
}

// the serial half of the step, one wavefront per pair: the sums and log-likelihood terms of the pass come from the sweep's tail
// (sweep_tail_wide, SolverStepArgs::pair_sums)
__global__ __launch_bounds__(64) void k_solver_serial(LevelGeom g, SolverStepArgs a) {
  __shared__ SolverLds L;
  solver_step_body<1, kReduceInFlight, 16, 16, true>(L, g, a, blockIdx.x);
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
                        const PairPtrs* pairs, dvo_hip_level_stats* levels, const double* T_init_or_null, const unsigned char* flags, int which,
                        int from_level) {
  k_level_begin<<<dim3((n_pairs + 63) / 64), dim3(64), 0, s>>>(states, n_pairs, prm, g, level, pairs, levels, T_init_or_null, flags, which, from_level);
}

void launch_mark_stragglers(hipStream_t s, const PairState* states, int n_pairs, int level, unsigned char* flags, int* list, int cap, bool list_only) {
  k_mark_stragglers<<<dim3(1), dim3(1024), 0, s>>>(states, n_pairs, level, flags, list, cap, list_only ? 1 : 0);
}

void launch_clear_flags(hipStream_t s, unsigned char* flags, int n_pairs) {
  k_clear_flags<<<dim3((n_pairs + 255) / 256), dim3(256), 0, s>>>(flags, n_pairs);
}

void launch_solver_serial(hipStream_t s, int n_pairs, LevelGeom g, const SolverStepArgs& a) {
  k_solver_serial<<<dim3(n_pairs), dim3(64), 0, s>>>(g, a);
}

int g_solver_occupancy = 0;   // (experiment: option "solver_occupancy")

void launch_solver_step(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, LevelGeom g,
                        const float* partials, const double* ll_partials, int ll_blocks_per_pair, const float2* scratch_for_fused_ll,
                        dvo_hip_level_stats* levels, dvo_hip_iteration_stats* iters, unsigned long long* step_tally, int* host_status, bool two_waves,
                        int level_slot_hint, const NextLevel* next_or_null) {
  const SolverStepArgs a = make_solver_step_args(states, n_pairs, prm, partials, ll_partials, ll_blocks_per_pair, scratch_for_fused_ll, levels, iters, step_tally,
                                                 host_status, level_slot_hint, next_or_null, nullptr);
  // (two wavefronts: see solver_step_body; a level of at most 32 tiles -- 160 x 120, 80 x 60 -- of a batch beyond two workgroups per compute unit)
  if (two_waves) k_solver_step<2><<<dim3(n_pairs), dim3(128), 0, s>>>(g, a);
  else if (g_solver_occupancy == 3) k_solver_step<kWavesPerBlock, 3><<<dim3(n_pairs), dim3(kBlock), 0, s>>>(g, a);
  else k_solver_step<kWavesPerBlock><<<dim3(n_pairs), dim3(kBlock), 0, s>>>(g, a);
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
