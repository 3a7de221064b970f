// solver_step.h -- the Gauss-Newton step of ONE pair by one workgroup: stage 3 of the deterministic reduction (per-tile float partials
// summed in tile order in float64), the log-likelihood of a small level, then lane 0 runs the reference's loop body behind the residual
// sweep -- precision, accept / revert, normal-equation contraction, 6 x 6 solve, SE(3) update, termination
// (dvo_core/src/dense_tracking.cpp:273-363; logic in solver_logic.h).
//
// Two callers, the same instructions, hence the same bits:
//   * k_solver_step (solver_kernels.hip): a launch of its own behind the sweep, one workgroup per pair;
//   * the sweeps' TAIL (align_fast.hip, align_mfma.hip; round 6): the workgroup that completes the LAST tile of a pair runs the pair's
//     step right there, in the sweep's launch -- one launch per Gauss-Newton iteration instead of two, and the step (a chain of memory
//     round trips and 2 us of serial float64 on one lane) runs in the shadow of the other pairs' tiles instead of holding the chip.
#pragma once
#include "align_common.h"
#include "solver_logic.h"

// DVO_TAIL_CLOCKS (experiment build of one translation unit, scripts/r6_tailclk.py on a library built with -DDVO_TAIL_CLOCKS): where the time of a sweep's tail goes -- 100 MHz wall
// clock, thread 0 of every workgroup (arrival) / of the pair's last workgroup (the rest); g_tail_clk[k] sums, g_tail_clk[8 + k] counts
#ifdef DVO_TAIL_CLOCKS
namespace dvo_hip { __device__ unsigned long long g_tail_clk[16]; }
#define DVO_TCLK(k, t_from) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&dvo_hip::g_tail_clk[(k)], now_ - (t_from)); atomicAdd(&dvo_hip::g_tail_clk[8 + (k)], 1ull); (t_from) = now_; } } while (0)
#define DVO_TCLK_START(t) unsigned long long t = wall_clock64()
#else
#define DVO_TCLK(k, t_from) ((void)0)
#define DVO_TCLK_START(t) ((void)0)
#endif

namespace dvo_hip {

// the step's LDS (k_solver_step declares one; the sweeps' tail lays it over their window, idle by then)
struct SolverLds {
  PairState st;
  dvo_hip_level_stats lvl;
  // [1]: the new iteration record; [0]: the one before it, fetched only for a pair whose result is written here (gn_finish reads either)
  dvo_hip_iteration_stats recs[2];
  dvo_hip_result res;
  dvo_hip_level_stats lvl_next;
  double sh[kWavesPerBlock * kAccStride];
  double sums[kAccStride];
  double ll_waves[kWavesPerBlock];
  double ll_stage[32];
  double Amat[36], bvec[6];              // J^T W J and J^T W r of this pass, contracted by 42 lanes (GnAssist)
  int information_ready;
  int rec_index;
  int stored;
};

// words of a POD copied cooperatively between global memory and LDS
template <int THREADS, typename T>
__device__ __forceinline__ void coop_copy(T* dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (int i = threadIdx.x; i < int(sizeof(T) / 4); i += THREADS) d[i] = s[i];
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

// WAVES: 4, or 2 for the small levels of a batch that does not fit the device in one go with four -- the kernel holds 210 registers,
// two four-wavefront workgroups per compute unit = 512 pairs at a time, and what a workgroup does on a small level is mostly lane 0's
// serial float64 work.  A two-wavefront workgroup plays the four (reduce_scale.h, loglik_partial_played): the records are the same bits.
// REDUCE_IN_FLIGHT / LL_SLOTS / LL_LOADS: loads in flight of the reduction and of the fused log-likelihood (registers against round
// trips; the order of the additions and products does not depend on them).
// PRE (round 6, k_solver_serial): the reduced sums and the four wavefronts' log-likelihood sums of the pass were formed in the sweep's
// tail (sweep_tail_wide) and are read from SolverStepArgs::pair_sums -- the step is its serial half, any number of wavefronts.
template <int WAVES, int REDUCE_IN_FLIGHT = kReduceInFlight, int LL_SLOTS = 16, int LL_LOADS = 16, bool PRE = false>
__device__ __forceinline__ void solver_step_body(SolverLds& L, const LevelGeom& g, const SolverStepArgs& a, int pair) {
  PairState& st = L.st;
  dvo_hip_level_stats& lvl = L.lvl;
  dvo_hip_iteration_stats& rec = L.recs[1];
  const SolverParams& prm = a.prm;
  // The state machine is one lane of serial float64 work; every global access it made used to be a dependent
  // ~1 us round trip.  Stage the pair's state, its level record and the new iteration record in LDS: loaded and
  // stored by all 256 lanes at once, touched by lane 0 at LDS latency.
  // ONE round trip for everything whose address does not depend on loaded data (round 4; with 1024 workgroups in flight a dependent
  // trip costs 4-5 us and the step was a chain of nine): the pair's state, its level record (the slot every live pair of the batch
  // is at -- a hint from the host, checked below), the partial rows, the log-likelihood partial sums of k_loglik.  A pair that turns
  // out to be finished has loaded them for nothing.
  constexpr int kThreads = WAVES * 64;
  constexpr int kStWords = int(sizeof(PairState) / 4), kLvlWords = int(sizeof(dvo_hip_level_stats) / 4);
  static_assert(sizeof(PairState) % 4 == 0 && sizeof(dvo_hip_level_stats) % 4 == 0, "copied by words");
  constexpr int kStPer = (kStWords + kThreads - 1) / kThreads, kLvlPer = (kLvlWords + kThreads - 1) / kThreads;
  unsigned st_w[kStPer], lvl_w[kLvlPer];
  const bool hint_ok = a.level_slot_hint >= 0 && a.level_slot_hint < prm.cap_levels;
  {
    const unsigned* src = reinterpret_cast<const unsigned*>(&a.states[pair]);
#pragma unroll
    for (int k = 0; k < kStPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      st_w[k] = src[i < kStWords ? i : 0];
    }
    const unsigned* lsrc = reinterpret_cast<const unsigned*>(a.levels + size_t(pair) * prm.cap_levels + (hint_ok ? a.level_slot_hint : 0));
#pragma unroll
    for (int k = 0; k < kLvlPer; ++k) {
      const int i = int(threadIdx.x) + k * kThreads;
      lvl_w[k] = prm.cap_levels > 0 ? lsrc[i < kLvlWords ? i : 0] : 0u;
    }
  }
  double ll_mine = 0.0;
  if constexpr (PRE) {
    const double* mine = a.pair_sums + size_t(pair) * kPairSumsStride;
    for (int i = threadIdx.x; i < kAccStride; i += kThreads) L.sums[i] = mine[i];
    if (threadIdx.x < kWavesPerBlock) L.ll_waves[threadIdx.x] = mine[kAccStride + threadIdx.x];
    __syncthreads();
  } else {
    if (!a.scratch_for_fused_ll && threadIdx.x < 32) ll_mine = a.ll_partials[size_t(pair) * a.ll_blocks_per_pair + min(int(threadIdx.x), a.ll_blocks_per_pair - 1)];
    reduce_partials<WAVES, REDUCE_IN_FLIGHT>(a.partials, pair, g.tiles_x * g.tiles_y, L.sh, L.sums);   // same routine, same order as k_loglik: identical n, S, P
  }
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
    if (threadIdx.x < 32) L.ll_stage[threadIdx.x] = ll_mine;
    // what the serial lane would otherwise do one store at a time: every word of the new iteration record NaN (SolverParams::record_prefilled)
    static_assert(sizeof(dvo_hip_iteration_stats) % 8 == 0, "prefilled by 8-byte words");
    for (int i = threadIdx.x; i < int(sizeof(dvo_hip_iteration_stats) / 8); i += kThreads) reinterpret_cast<double*>(&rec)[i] = dvo_nan();
    // ... and the contraction of the Gram sums with the pass' precision (the sums are in place since reduce_partials' barrier): lane
    // i * 6 + j forms A(i, j) from the upper-triangle entry of (min, max), lanes 36..41 J^T W r
    for (int k = threadIdx.x; k < 42; k += kThreads) {
      const double d = L.sums[kAccN] - 3.0;
      float Cc[3], Pc[4];
      scale_to_precision(L.sums[kAccS] / d, L.sums[kAccS + 1] / d, L.sums[kAccS + 2] / d, Cc, Pc);
      const double p00 = double(Pc[0]), p01 = double(Pc[1]), p11 = double(Pc[3]);
      if (k < 36) {
        const int i = k / 6, j = k - i * 6, lo = i < j ? i : j, hi = i < j ? j : i;
        const int o = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);           // index of (lo, hi) in the row-major upper triangle
        L.Amat[k] = gn_contract(p00, p01, p11, L.sums[kAccJ00 + o], L.sums[kAccJ01 + o], L.sums[kAccJ11 + o]);
      } else {
        L.bvec[k - 36] = gn_contract(p00, p01, p11, L.sums[kAccB00 + k - 36], L.sums[kAccB01 + k - 36], L.sums[kAccB11 + k - 36]);
      }
    }
  }
  __syncthreads();
  if (!st.active || st.level != g.level) {   // uniform: the pair is not (or no longer) on this level
    if (threadIdx.x == 0) publish_step(a.step_tally, a.host_status, a.n_pairs, false);
    // The hand-over (NextLevel): a pair that ENDED this level in an earlier step begins the next level here -- or, on the last level,
    // has its result written -- in the shadow of the workgroups that still iterate: no launch between two levels, none behind the
    // last.  (Doing it in the very step that ends the level was measured and dropped: that workgroup is the launch's slowest.)
    if (st.level != g.level || st.finished) return;
    const int slot_now = st.n_levels - 1;
    const bool have_now = slot_now >= 0 && slot_now < prm.cap_levels;
    if (a.next.valid) {
      if (threadIdx.x == 0) {
        const int n_selected_next = *a.next.pairs[pair].n_selected;
        LevelGeom gn = g;
        gn.fx = a.next.fx; gn.fy = a.next.fy; gn.ox = a.next.ox; gn.oy = a.next.oy;
        SolverParams begin = prm;
        begin.cap_levels = st.n_levels < prm.cap_levels ? st.n_levels + 1 : 0;
        L.stored = begin.cap_levels ? st.n_levels : -1;
        gn_level_begin(st, begin, gn, a.next.level, n_selected_next, &L.lvl_next - st.n_levels);   // (dense_tracking.cpp:200-238)
      }
      __syncthreads();
      coop_copy<kThreads>(&a.states[pair], &st);
      if (L.stored >= 0) coop_copy<kThreads>(a.levels + size_t(pair) * prm.cap_levels + L.stored, &L.lvl_next);
    } else if (a.next.results) {
      // dense_tracking.cpp:368-373: the last iteration with an increment is the level's last record or the one before it
      const int last = st.n_iters_total - 1;
      if (have_now && !(hint_ok && slot_now == a.level_slot_hint)) coop_copy<kThreads>(&lvl, a.levels + size_t(pair) * prm.cap_levels + slot_now);
      if (last >= 0 && last < prm.cap_iters) coop_copy<kThreads>(&L.recs[1], a.iters + size_t(pair) * prm.cap_iters + last);
      if (last >= 1 && last - 1 < prm.cap_iters) coop_copy<kThreads>(&L.recs[0], a.iters + size_t(pair) * prm.cap_iters + last - 1);
      __syncthreads();
      if (threadIdx.x == 0) {
        gn_finish(st, prm, &lvl - slot_now, &L.recs[1] - last, &L.res);
        st.finished = 1;
      }
      __syncthreads();
      coop_copy<kThreads>(a.next.results + pair, &L.res);
      if (threadIdx.x == 0) a.states[pair].finished = 1;
    }
    return;
  }
  const int level_slot = st.n_levels - 1;
  dvo_hip_level_stats* lvl_global = a.levels + size_t(pair) * prm.cap_levels + level_slot;
  const bool have_level = level_slot >= 0 && level_slot < prm.cap_levels;
  if (have_level && !(hint_ok && level_slot == a.level_slot_hint)) {   // (uniform; not expected: the levels of a batch begin together)
    coop_copy<kThreads>(&lvl, lvl_global);
    __syncthreads();
  }
  if constexpr (PRE) {
    // (the four wavefronts' sums are in place)
  } else if (a.scratch_for_fused_ll) {
    // coarse levels: the log-likelihood sweep is small enough for this workgroup, which saves a launch per iteration
    float C[3], P[4];
    const int n = scale_from_sums(L.sums, C, P);
    constexpr int kPlayed = kWavesPerBlock / WAVES;          // wavefronts of a four-wavefront workgroup each real one plays
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double t[kPlayed];
#pragma unroll
    for (int q = 0; q < kPlayed; ++q) t[q] = 0.0;
    if (n >= 6) {
      const int tiles = g.tiles_x * g.tiles_y;
      if (g.compact) {                                        // (uniform) the packed residuals of the contracted window sweep
        if constexpr (WAVES == kWavesPerBlock)
          t[0] = loglik_partial_compact<LL_SLOTS>(a.scratch_for_fused_ll + size_t(pair) * residual_entries(g), a.partials + size_t(pair) * tiles * kAccStride, tiles * 4, P,
                                                  wave, kWavesPerBlock);
        else
          loglik_partial_compact_played<4, kPlayed>(a.scratch_for_fused_ll + size_t(pair) * residual_entries(g), a.partials + size_t(pair) * tiles * kAccStride,
                                                    tiles * 4, P, wave, WAVES, kWavesPerBlock, t);
      } else {
        if constexpr (WAVES == kWavesPerBlock) t[0] = loglik_partial<LL_LOADS>(a.scratch_for_fused_ll + size_t(pair) * g.w * g.h, g.w * g.h, P, 0, 1);
        else loglik_partial_played<8, WAVES>(a.scratch_for_fused_ll + size_t(pair) * g.w * g.h, g.w * g.h, P, t);
      }
    }
#pragma unroll
    for (int q = 0; q < kPlayed; ++q) {
      const double total = wave_sum_double(t[q]);
      if ((threadIdx.x & 63) == 0) L.ll_waves[wave + q * WAVES] = total;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double ll_sum = 0.0;
    if (PRE || a.scratch_for_fused_ll) {
      ll_sum = (L.ll_waves[0] + L.ll_waves[1]) + (L.ll_waves[2] + L.ll_waves[3]);
    } else {
      for (int b = 0; b < a.ll_blocks_per_pair; ++b) ll_sum += L.ll_stage[b];
    }
    L.rec_index = st.n_iters_total;
    // gn_step addresses levels[n_levels - 1] and iters[n_iters_total]: hand it pointers biased so that those land in LDS
    SolverParams local = prm;
    local.cap_levels = have_level ? level_slot + 1 : 0;
    local.cap_iters = L.rec_index + 1;
    local.record_prefilled = 1;
    GnAssist assist = {L.Amat, L.bvec, 1, 0};
    gn_step(st, local, g, L.sums, ll_sum, &lvl - level_slot, &rec - L.rec_index, nullptr, &assist);
    L.information_ready = assist.information_ready;
    publish_step(a.step_tally, a.host_status, a.n_pairs, st.active != 0);
  }
  __syncthreads();
  if (L.information_ready)                                    // (uniform; GnAssist::defer_information)
    for (int i = threadIdx.x; i < 36; i += kThreads) rec.information[i] = L.Amat[i];
  __syncthreads();
  coop_copy<kThreads>(&a.states[pair], &st);
  if (have_level) coop_copy<kThreads>(lvl_global, &lvl);
  if (L.rec_index < prm.cap_iters) coop_copy<kThreads>(a.iters + size_t(pair) * prm.cap_iters + L.rec_index, &rec);
}

// ---- the sweeps' tail ---------------------------------------------------------------------------------------------------------------
// Every workgroup of a sweep launched with a tail -- also those of a pair that is not on the level: they have read the pair's state by
// then, which the step is about to change -- ends with sweep_tail_arrive.  The tile's partial row and residual pairs were stored
// write-through (sc1) and have left the wavefronts (s_waitcnt vmcnt(0) on every thread, then the workgroup barrier) before lane 0
// takes the pair's ticket; the workgroup that draws the last ticket of the pair resets the word, invalidates its compute unit's L1
// (agent-scope acquire: the other tiles' rows were written by other compute units; the per-XCD L2s are kept free of stale lines by the
// memory's probes) and runs the step.  Nobody waits for anybody: a workgroup either leaves or has everything it needs.
// `ticket_lds`: one int of LDS.  Returns true in the workgroup that has to run the step.
__device__ __forceinline__ bool sweep_tail_arrive(int* arrivals, int pair, int tiles, int* ticket_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(arrivals + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == tiles - 1) {
      __hip_atomic_store(arrivals + pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next launch finds it at zero)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *ticket_lds = ticket;
  }
  __syncthreads();
  return *ticket_lds == tiles - 1;
}

// the step itself, out of line: the sweep's own registers and schedule are what they are without a tail (the compiler allots this
// function the calling kernel's register budget and spills the serial lane's float64 state beyond it).  The level's geometry and the
// step's arguments -- kernel arguments of the sweep -- travel through LDS (SweepTailLds): a non-inlined function gets no kernel arguments.
struct SweepTailLds {
  SolverLds step;
  LevelGeom g;
  SolverStepArgs a;
};

__device__ __attribute__((noinline)) static void sweep_tail_step(__attribute__((address_space(3))) char* lds, int pair) {
  SweepTailLds& T = *reinterpret_cast<SweepTailLds*>((char*)lds);
  solver_step_body<kWavesPerBlock, 8, 4, 4>(T.step, T.g, T.a, pair);
}

// The WIDE half of the step in the tail (TAIL = 2): the last workgroup of a pair sums the pair's partial rows (stage 3 of the reduction)
// and, with the pass' precision, the log-likelihood terms of its residual pairs -- the step's memory round trips -- and leaves the
// kAccStride sums and the four wavefronts' log-likelihood sums in SolverStepArgs::pair_sums; the serial half (k_solver_serial, one
// wavefront per pair) follows in a launch of its own.  What this half needs fits the sweep's register budget.
__device__ __attribute__((noinline)) static void sweep_tail_wide_step(__attribute__((address_space(3))) char* lds, int pair) {
  SweepTailLds& T = *reinterpret_cast<SweepTailLds*>((char*)lds);
  const LevelGeom& g = T.g;
  const SolverStepArgs& a = T.a;
  const int tiles = g.tiles_x * g.tiles_y;
  DVO_TCLK_START(tc);
  reduce_partials<kWavesPerBlock, 8>(a.partials, pair, tiles, T.step.sh, T.step.sums);
  DVO_TCLK(3, tc);
  float C[3], P[4];
  const int n = scale_from_sums(T.step.sums, C, P);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double t = 0.0;
  if (n >= 6) {
    if (g.compact) t = loglik_partial_compact<4>(a.scratch_for_fused_ll + size_t(pair) * residual_entries(g), a.partials + size_t(pair) * tiles * kAccStride, tiles * 4, P, wave, kWavesPerBlock);
    else t = loglik_partial<4>(a.scratch_for_fused_ll + size_t(pair) * g.w * g.h, g.w * g.h, P, 0, 1);
  }
  const double total = wave_sum_double(t);
  DVO_TCLK(4, tc);
  double* out = a.pair_sums + size_t(pair) * kPairSumsStride;
  if ((threadIdx.x & 63) == 0) out[kAccStride + wave] = total;
  if (threadIdx.x < kAccStride) out[threadIdx.x] = T.step.sums[threadIdx.x];
  DVO_TCLK(5, tc);
}

// the whole tail behind a tile: arrive; the last workgroup of the pair stages the arguments and runs the step.  `lds`: the workgroup's
// idle LDS, at least sizeof(SweepTailLds) + 16 bytes, 16-byte aligned.
__device__ __forceinline__ void sweep_tail(const LevelGeom& g, const SolverStepArgs& a, int pair, char* lds) {
  SweepTailLds* T = reinterpret_cast<SweepTailLds*>(lds);
  int* ticket = reinterpret_cast<int*>(lds + sizeof(SweepTailLds));
  if (!sweep_tail_arrive(a.arrivals, pair, g.tiles_x * g.tiles_y, ticket)) return;
  if (threadIdx.x == 0) {
    T->g = g;
    T->a = a;
  }
  __syncthreads();
  sweep_tail_step((__attribute__((address_space(3))) char*)lds, pair);
}

// ... and with the wide half only (only the workgroups of pairs that ARE on the level come here: nothing of the pair's state changes)
__device__ __forceinline__ void sweep_tail_wide(const LevelGeom& g, const SolverStepArgs& a, int pair, char* lds) {
  SweepTailLds* T = reinterpret_cast<SweepTailLds*>(lds);
  int* ticket = reinterpret_cast<int*>(lds + sizeof(SweepTailLds));
  DVO_TCLK_START(tc);
  const bool last = sweep_tail_arrive(a.arrivals, pair, g.tiles_x * g.tiles_y, ticket);
  DVO_TCLK(1, tc);                                           // arrival: the stores' completion, two barriers, the ticket
  if (!last) return;
  if (threadIdx.x == 0) {
    T->g = g;
    T->a = a;
  }
  __syncthreads();
  DVO_TCLK(2, tc);                                           // arguments staged
  sweep_tail_wide_step((__attribute__((address_space(3))) char*)lds, pair);
  DVO_TCLK(6, tc);                                           // the whole wide half
}

}  // namespace dvo_hip
