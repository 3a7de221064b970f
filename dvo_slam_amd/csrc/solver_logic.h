// solver_logic.h -- the Gauss-Newton state machine of DenseTracker::match, one instance per frame pair.
//
// Restates the control flow of dvo_core/src/dense_tracking.cpp:131-376 (accept / revert / termination,
// level hand-over of the increment, statistics) as host/device inline functions.  On the GPU they
// are executed by lane 0 of the per-pair solver workgroup (solver_kernels.hip), so a whole
// coarse-to-fine alignment -- or a batch of them -- runs without the pose ever visiting the host.
// tests/ compiles the same header with the host compiler to check the logic against the oracle
// without a GPU (test-only).
#pragma once

#include <float.h>

#include "device_types.h"
#include "pixel_math.h"
#include "se3_device.h"

#ifndef DVO_GN_CLK
#define DVO_GN_CLK(i) ((void)0)       // experiment builds stamp the stages of gn_step (scripts/ubench/build_clocks.sh)
#endif

namespace dvo_hip {

DVO_HD double dvo_nan() { return __builtin_nan(""); }

DVO_HD double inf_norm6(const double* v) {
  double m = 0;
  for (int i = 0; i < 6; ++i) {
    if (v[i] != v[i]) return dvo_nan();
    const double a = fabs(v[i]);
    if (a > m) m = a;
  }
  return m;
}

// inc = exp(x); initial = inc^-1 * initial; estimate = inc * estimate; KT = float(K * estimate)
// (dense_tracking.cpp:259-263), in two halves: what the next residual sweep needs ...
DVO_HD void gn_advance_estimate(PairState& st, const LevelGeom& g) {
  se3_exp(st.x, st.inc);
  st.estimate_old = st.estimate;
  se3_mul(st.inc, st.estimate, st.estimate);
  float T[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = float(st.estimate.R[i * 3 + j]);
    T[i * 4 + 3] = float(st.estimate.t[i]);
  }
  make_KT(g.fx, g.fy, g.ox, g.oy, T, st.KT);
}
// ... and what it does not (the prior's `initial` only enters the next solve)
DVO_HD void gn_advance_initial(PairState& st) {
  SE3d inv;
  se3_inverse(st.inc, inv);
  st.initial_old = st.initial;
  se3_mul(inv, st.initial, st.initial);
}
DVO_HD void gn_begin_iteration(PairState& st, const LevelGeom& g) {
  gn_advance_estimate(st, g);
  gn_advance_initial(st);
}

// dense_tracking.cpp:137-150
DVO_HD void gn_init_pair(PairState& st, const SolverParams& prm, const double* T_init_row_major_4x4) {
  if (prm.use_initial_estimate) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) st.inc.R[i * 3 + j] = T_init_row_major_4x4[i * 4 + j];
      st.inc.t[i] = T_init_row_major_4x4[i * 4 + 3];
    }
  } else {
    se3_identity(st.inc);
  }
  st.initial = st.inc;
  st.initial_old = st.inc;
  se3_identity(st.estimate);
  se3_identity(st.estimate_old);
  for (int i = 0; i < 6; ++i) st.x[i] = 0;
  for (int i = 0; i < 36; ++i) st.A_last[i] = dvo_nan();
  st.error = DBL_MAX;
  st.last_error = DBL_MAX;
  st.level = prm.first_level;
  st.iteration = 0;
  st.active = 0;
  st.first = 1;
  st.n_iters_total = 0;
  st.n_levels = 0;
  st.level_first_iter = 0;
  st.finished = 0;
  for (int i = 0; i < 4; ++i) st.P_prev[i] = 0.0f;
  for (int i = 0; i < 12; ++i) st.KT[i] = 0.0f;
}

// dense_tracking.cpp:200-238 : per-level reset and "x = inc.log()" (Q12, Q21)
DVO_HD void gn_level_begin(PairState& st, const SolverParams& prm, const LevelGeom& g, int level, int n_selected,
                           dvo_hip_level_stats* levels) {
  st.level = level;
  st.iteration = 0;
  st.error = DBL_MAX;
  st.last_error = DBL_MAX;
  st.first = 1;
  st.active = 1;
  for (int i = 0; i < 4; ++i) st.P_prev[i] = 0.0f;
  st.level_first_iter = st.n_iters_total;
  if (st.n_levels < prm.cap_levels) {
    dvo_hip_level_stats& ls = levels[st.n_levels];
    ls.id = level;
    double maxp = double(prm.max_points_level0);
    for (int l = 0; l < level; ++l) maxp *= 0.25;          // point_selection.cpp:68-71
    ls.max_valid_pixels = int(maxp);
    ls.valid_pixels = n_selected;
    ls.termination = DVO_HIP_TERMINATION_UNSET;
    ls.n_iterations = 0;
    ls.first_iteration_index = st.n_iters_total;
  }
  st.n_levels += 1;
  se3_log(st.inc, st.x);
  gn_begin_iteration(st, g);
}

// dense_tracking.cpp:359-363 run after the do/while however it was left
DVO_HD void gn_level_end(PairState& st, const SolverParams& prm, dvo_hip_level_stats* levels) {
  if (st.n_levels - 1 < prm.cap_levels) {
    dvo_hip_level_stats& ls = levels[st.n_levels - 1];
    if (inf_norm6(st.x) <= prm.precision) ls.termination = DVO_HIP_INCREMENT_TOO_SMALL;
    if (st.iteration >= prm.max_iterations) ls.termination = DVO_HIP_ITERATIONS_EXCEEDED;
  }
  st.active = 0;
}

// The resident kernel (align_resident.hip) runs the loop body BEFORE the log-likelihood of the pass is known -- as if the pass were
// accepted -- and settles the accept / revert question one exchange later (gn_commit_loglik): everything below except the
// error bookkeeping is independent of the log-likelihood sum.
struct GnSpeculation {
  double half_n_logdet;               // 0.5 n log det P of the pass (the log-likelihood without its data term)
  int needs_loglik;                   // out: 0 = the pass ended the level without a log-likelihood (too few constraints)
  int replay_reject;                  // in: 1 = full form of a pass gn_commit_loglik has rejected (the decision is not taken twice)
  // The record of a pass is 51 doubles of stores for one lane.  A caller with idle lanes next to the solver lane takes the
  // bulk of it off the serial path:
  int record_prefilled;               // in: every 8-byte word of the record already holds NaN
  int defer_information;              // in: rec.information is left to the caller, who copies st.A_last when information_ready
  int information_ready;              // out
  // in: a word (shared with the wavefronts that sweep) that receives release_value, after a block-level fence, as soon as
  // everything the next residual sweep reads is in place (KT, P_prev, first, active -- or the end of the level); the rest of
  // the pass (records, A_last, the prior's `initial`) is written behind it
  volatile int* release_word;
  int release_value;
};

DVO_HD void gn_release_sweep(GnSpeculation* spec) {
  if (spec && spec->release_word) {
    DVO_FENCE_BLOCK();
    *spec->release_word = spec->release_value;
  }
}

// One pass of the loop body after the residual sweep: dense_tracking.cpp:273-357.
// sums = the kNumAcc accumulators reduced over all tiles; ll_sum = sum log(1 + 0.2 r^T P r).
// spec != null, replay_reject == 0: speculative form -- ll_sum is ignored, the pass is treated as accepted, st.error /
// st.last_error and rec.tdist_loglik are left for gn_commit_loglik.  replay_reject == 1: the full form, taking the revert path.
// What a caller with idle lanes next to the solver lane takes off the serial path (k_solver_step; the host emulation and the resident
// kernel pass none):
//   A, b: the contraction of the Gram sums with the pass' precision, formed with gn_contract (the same expression, the same bits) --
//     A the full symmetric 6 x 6 WITHOUT the prior's mu on its diagonal, b = J^T W r (not negated); gn_step then works in A itself
//     instead of PairState::A_last (which it leaves alone);
//   defer_information: the 36 doubles of rec.information are left to the caller, who copies A when information_ready comes back 1.
struct GnAssist {
  double* A;
  const double* b;
  int defer_information;
  int information_ready;
};

DVO_HD double gn_contract(double p00, double p01, double p11, double s00, double s01, double s11) { return p00 * s00 + p01 * s01 + p11 * s11; }

DVO_HD void gn_step(PairState& st, const SolverParams& prm, const LevelGeom& g, const double* sums, double ll_sum,
                    dvo_hip_level_stats* levels, dvo_hip_iteration_stats* iters, GnSpeculation* spec = nullptr, GnAssist* assist = nullptr) {
  if (assist) assist->information_ready = 0;
  const bool speculate = spec && !spec->replay_reject;
  if (spec) spec->needs_loglik = spec->information_ready = 0;
  if (!st.active) return;
  DVO_GN_CLK(0);
  dvo_hip_iteration_stats dummy;
  dvo_hip_iteration_stats& rec = (st.n_iters_total < prm.cap_iters) ? iters[st.n_iters_total] : dummy;
  st.n_iters_total += 1;
  if (st.n_levels - 1 < prm.cap_levels) levels[st.n_levels - 1].n_iterations += 1;

  const int n = int(sums[kAccN] + 0.5);
  rec.id = st.iteration;
  rec.valid_constraints = n;
  rec.tdist_mean[0] = rec.tdist_mean[1] = 0.0;              // Q8
  if (!(prm.record_prefilled || (spec && spec->record_prefilled))) {
    rec.tdist_loglik = dvo_nan();
    rec.prior_loglik = dvo_nan();
    for (int i = 0; i < 4; ++i) rec.tdist_precision[i] = dvo_nan();
    for (int i = 0; i < 6; ++i) rec.increment[i] = dvo_nan();
    for (int i = 0; i < 36; ++i) rec.information[i] = dvo_nan();
  }

  DVO_GN_CLK(1);                                             // record initialised
  if (n < 6) {                                               // :276-284
    st.initial = st.initial_old;
    st.estimate = st.estimate_old;
    if (st.n_levels - 1 < prm.cap_levels) levels[st.n_levels - 1].termination = DVO_HIP_TOO_FEW_CONSTRAINTS;
    gn_level_end(st, prm, levels);
    return;
  }

  float C[3], P[4];
  const double d = double(n) - 3.0;
  scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, C, P);   // :295
  const double det = double(P[0]) * double(P[3]) - double(P[1]) * double(P[2]);
  // the reference's log-likelihood is a float (computeCompleteDataLogLikelihood returns float into `float ll`, :297,
  // impl:406-425): the rounding decides `Error < LastError` at the noise floor, so it is part of the algorithm
  // (speculative form: the logarithm is only needed by the verdict one exchange later, it is taken behind the release)
  const double half_n_logdet = speculate ? 0.0 : 0.5 * double(n) * log(det);
  const double ll = double(float(half_n_logdet - 3.5 * ll_sum));
  if (!speculate) rec.tdist_loglik = -ll;
  for (int i = 0; i < 4; ++i) rec.tdist_precision[i] = double(P[i]);
  double li[6] = {0, 0, 0, 0, 0, 0};
  double sq = 0;
  if (prm.mu != 0.0) {                                       // with Mu = 0 the prior terms vanish; skip log(initial)
    se3_log(st.initial, li);
    for (int i = 0; i < 6; ++i) sq += li[i] * li[i];
  }
  rec.prior_loglik = prm.mu * sq;                            // :302
  DVO_GN_CLK(2);                                             // precision, log det, prior

  bool accept = true;
  if (speculate) {
    spec->needs_loglik = 1;
  } else {
    st.last_error = st.error;
    st.error = -ll;
    accept = st.error < st.last_error && !spec;              // :312
  }
  if (!accept) {
    st.initial = st.initial_old;
    st.estimate = st.estimate_old;
    if (st.n_levels - 1 < prm.cap_levels) levels[st.n_levels - 1].termination = DVO_HIP_LOGLIKELIHOOD_DECREASED;
    gn_level_end(st, prm, levels);
    return;
  }

  // contract the Gram sums with W = w P : A = J^T W J, b = -J^T W r (least_squares.cpp:58-64)
  const double p00 = double(P[0]), p01 = double(P[1]), p11 = double(P[3]);
  // (formed in place in st.A_last, which nothing reads between here and the end of the pass: on the device the state lives in LDS,
  // a local array of 36 doubles in scratch memory -- three round trips of the serial lane per pass)
  double* const A = assist ? assist->A : st.A_last;
  double b[6];
  if (assist) {
    for (int i = 0; i < 6; ++i) b[i] = -assist->b[i];
  } else {
    int o = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = i; j < 6; ++j) {
        const double a = gn_contract(p00, p01, p11, sums[kAccJ00 + o], sums[kAccJ01 + o], sums[kAccJ11 + o]);
        A[i * 6 + j] = a;
        A[j * 6 + i] = a;
        ++o;
      }
    for (int i = 0; i < 6; ++i) b[i] = -gn_contract(p00, p01, p11, sums[kAccB00 + i], sums[kAccB01 + i], sums[kAccB11 + i]);
  }
  for (int i = 0; i < 6; ++i) {                              // :345-346
    A[i * 6 + i] += prm.mu;
    b[i] += prm.mu * li[i];
  }
  DVO_GN_CLK(3);                                             // contraction
  solve6(A, b, st.x);                                        // :347
  DVO_GN_CLK(4);                                             // 6x6 solve
  st.iteration += 1;
  const double xn = inf_norm6(st.x);
  const bool go_on = xn > prm.precision && st.iteration < prm.max_iterations;   // :357
  if (go_on) {
    for (int i = 0; i < 4; ++i) st.P_prev[i] = P[i];         // next pass weights use this P (Q11)
    st.first = 0;
    gn_advance_estimate(st, g);
  } else {
    gn_level_end(st, prm, levels);
  }
  DVO_GN_CLK(5);                                             // exp, product, K T
  gn_release_sweep(spec);
  // behind the release: nothing below is read by the next residual sweep
  if (speculate) spec->half_n_logdet = 0.5 * double(n) * log(det);
  for (int i = 0; i < 6; ++i) rec.increment[i] = st.x[i];
  if (spec && spec->defer_information) {
    spec->information_ready = 1;
  } else if (assist && assist->defer_information) {
    assist->information_ready = 1;
  } else {
    for (int i = 0; i < 36; ++i) rec.information[i] = A[i];
  }
  if (go_on) gn_advance_initial(st);
  DVO_GN_CLK(6);                                             // record, A_last, inverse, product
}

// The accept test of a pass that gn_step ran speculatively: true = accepted (the error chain moves on, the record gets its
// log-likelihood); false = the caller restores the state from before that gn_step call and runs it again in full form, which
// then takes the revert path (dense_tracking.cpp:312-317).
DVO_HD bool gn_commit_loglik(PairState& st, const GnSpeculation& spec, double ll_sum, dvo_hip_iteration_stats& rec) {
  const double ll = double(float(spec.half_n_logdet - 3.5 * ll_sum));   // float like the reference's, see gn_step
  if (!(-ll < st.error)) return false;
  rec.tdist_loglik = -ll;
  st.last_error = st.error;
  st.error = -ll;
  return true;
}

// dense_tracking.cpp:368-373
DVO_HD void gn_finish(const PairState& st, const SolverParams& prm, const dvo_hip_level_stats* levels,
                      const dvo_hip_iteration_stats* iters, dvo_hip_result* out) {
  SE3d inv;
  se3_inverse(st.estimate, inv);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out->transformation[i * 4 + j] = inv.R[i * 3 + j];
    out->transformation[i * 4 + 3] = inv.t[i];
  }
  out->transformation[12] = out->transformation[13] = out->transformation[14] = 0.0;
  out->transformation[15] = 1.0;
  for (int i = 0; i < 36; ++i) out->information[i] = dvo_nan();
  out->loglik = dvo_nan();
  out->n_levels = st.n_levels;
  out->n_iterations_total = st.n_iters_total;
  out->entropy = dvo_nan();
  out->condition_number = dvo_nan();
  out->constraint_ratio = dvo_nan();
  out->constraint_ratio_accepted = 0.0;
  if (st.n_levels >= 1 && st.n_levels - 1 < prm.cap_levels) {
    const dvo_hip_level_stats& ls = levels[st.n_levels - 1];
    int idx = ls.n_iterations - 1;
    if (ls.termination == DVO_HIP_LOGLIKELIHOOD_DECREASED) idx -= 1;   // :369
    const int abs_idx = ls.first_iteration_index + idx;
    if (idx >= 0 && abs_idx < prm.cap_iters) {
      const dvo_hip_iteration_stats& it = iters[abs_idx];
      for (int i = 0; i < 36; ++i) out->information[i] = it.information[i] * 0.008 * 0.008;
      out->loglik = it.tdist_loglik + it.prior_loglik;
    }
    // keyframe-selection statistics (dvo_slam/src/keyframe_tracker.cpp:165-196, tracking_result_evaluation.cpp:52-55,
    // constraints/constraint_proposal_voter.cpp:136-140)
    const int last = ls.first_iteration_index + ls.n_iterations - 1;
    if (ls.n_iterations >= 1 && last < prm.cap_iters)
      out->constraint_ratio = double(iters[last].valid_constraints) / double(ls.valid_pixels);
    const bool two = ls.termination == DVO_HIP_LOGLIKELIHOOD_DECREASED || ls.termination == DVO_HIP_TOO_FEW_CONSTRAINTS;
    if (ls.n_iterations >= (two ? 2 : 1)) {                 // LevelStats::HasIterationWithIncrement, dense_tracking_config.cpp:138-150
      const int acc = ls.termination == DVO_HIP_LOGLIKELIHOOD_DECREASED ? last - 1 : last;
      if (acc < prm.cap_iters) out->constraint_ratio_accepted = double(iters[acc].valid_constraints) / double(ls.valid_pixels);
    }
  }
  out->entropy = log(sym6_determinant(out->information));   // NaN for a negative determinant like std::log
  if (prm.want_condition_number) {                          // ~20 us of serial float64 Jacobi sweeps: on request only
    double ev[6];
    sym6_eigenvalues(out->information, ev);
    double lo = ev[0], hi = ev[0], sum = ev[0];
    for (int i = 1; i < 6; ++i) {
      lo = ev[i] < lo ? ev[i] : lo;
      hi = ev[i] > hi ? ev[i] : hi;
      sum += ev[i];
    }
    out->condition_number = sum == sum ? fabs(hi / lo) : dvo_nan();
  }
}

}  // namespace dvo_hip
