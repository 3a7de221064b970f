// emul_device.cpp -- TEST-ONLY host emulation of the device code path.
//
// Compiles the very headers the gfx950 kernels are built from (pixel_math.h, solver_logic.h,
// se3_device.h) with the host C++ compiler and drives them pixel by pixel / pair by pair, so that the
// per-pixel arithmetic and the Gauss-Newton state machine can be checked against the oracle in the
// CPU-only test tier.  It is never linked into libdvo_hip.so and is not a fallback: the product has none.
#include <cmath>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../dvo_slam_amd/csrc/linear_walk.h"
#include "../../dvo_slam_amd/csrc/solver_logic.h"

using namespace dvo_hip;

extern "C" {

struct emul_level {
  int w, h;
  float fx, fy, ox, oy;
  const float* R;   // w*h*4 {Zsel, I, Idx, Idy}
  const float* A;   // w*h*4 {I, Z, Idx, Idy}
  const float* B;   // w*h*2 {Zdx, Zdy}
  int n_selected;
};

}  // extern "C"

namespace {

struct LevelCtx {
  LevelGeom g;
  std::vector<float> tx, ty;
  const float4* R;
  const float4* A;
  const float2* B;
};

void make_level(const emul_level& e, LevelCtx& c) {
  c.tx.resize(e.w);
  c.ty.resize(e.h);
  for (int x = 0; x < e.w; ++x) c.tx[x] = (float(x) - e.ox) / e.fx;
  for (int y = 0; y < e.h; ++y) c.ty[y] = (float(y) - e.oy) / e.fy;
  c.g.w = e.w; c.g.h = e.h;
  c.g.fx = e.fx; c.g.fy = e.fy; c.g.ox = e.ox; c.g.oy = e.oy;
  c.g.wi_x = 0.5f * e.fx / 255.0f; c.g.wi_y = 0.5f * e.fy / 255.0f;
  c.g.tx = c.tx.data(); c.g.ty = c.ty.data();
  c.g.pair_list = nullptr;
  c.g.skip_flags = nullptr;
  c.g.rcp_table = nullptr; c.g.rcp_shift = 0;               // (the emulation runs the default arithmetic)
  c.g.tiles_x = (e.w + kTileW - 1) / kTileW;
  c.g.tiles_y = e.h;
  c.g.linear = 0;
  c.R = reinterpret_cast<const float4*>(e.R);
  c.A = reinterpret_cast<const float4*>(e.A);
  c.B = reinterpret_cast<const float2*>(e.B);
}

// 0: the staged forms (pixel_residual, tdist_weight, accumulate_pixel); 1: what the matrix-core sweep runs -- the straight-line
// stages, sqrt(w)-scaled rows from jacobian_rows_fast, Gram accumulation of the 14-vector
int g_schedule = 0;

// G += v v^T of v = [J0(6), J1(6), r0, r1] (already scaled by sqrt(w)), written into the canonical accumulator layout
void accumulate_vector(float* acc, const float* J0, const float* J1, float r0, float r1) {
  float v[14];
  for (int i = 0; i < 6; ++i) { v[i] = J0[i]; v[6 + i] = J1[i]; }
  v[12] = r0; v[13] = r1;
  auto G = [&](int r, int c) { return v[r] * v[c]; };
  acc[kAccN] += 1.0f;
  acc[kAccS] += G(12, 12); acc[kAccS + 1] += G(12, 13); acc[kAccS + 2] += G(13, 13);
  int o = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j, ++o) {
      acc[kAccJ00 + o] += G(i, j);
      acc[kAccJ11 + o] += G(6 + i, 6 + j);
      acc[kAccJ01 + o] += G(i, 6 + j) + G(j, 6 + i);
    }
  for (int k = 0; k < 6; ++k) {
    acc[kAccB00 + k] += G(k, 12);
    acc[kAccB01 + k] += G(k, 13) + G(6 + k, 12);
    acc[kAccB11 + k] += G(6 + k, 13);
  }
}

bool flat_pixel(const LevelCtx& c, const float* KT, const float* Pp, bool first, int u, int v, float* acc, float* r) {
  const float4 ref = c.R[v * c.g.w + u];
  const float tx = c.g.tx[u], ty = c.g.ty[v];
  const PixelProj p = pixel_project_flat(c.g, KT, ref.x, tx, ty);
  if (!p.ok) return false;
  PixelTaps t;
  pixel_fetch(c.g, c.A, c.B, p, t);
  PixelTerms o;
  if (!pixel_finish_flat(c.g, ref, p, t, o)) return false;
  r[0] = o.r0; r[1] = o.r1;
  const float sw = first ? 1.0f : tdist_weight_sqrt_fast(o.r0, o.r1, Pp[0], Pp[1] + Pp[2], Pp[3]);
  float J0[6], J1[6];
  jacobian_rows_fast(o, sw, tx, ty, fmaf(tx, tx, 1.0f), fmaf(ty, ty, 1.0f), J0, J1);
  accumulate_vector(acc, J0, J1, sw * o.r0, sw * o.r1);
  return true;
}

// one sweep of the residual kernel + the log-likelihood sweep: float accumulation per row of 64 pixels
// (one "wavefront row"), float64 across rows, like the device's per-tile partials
void sweep(const LevelCtx& c, const float* KT, const float* Pp, bool first, double* sums, std::vector<float>& res) {
  const int w = c.g.w, h = c.g.h;
  res.assign(size_t(w) * h * 2, std::nanf(""));
  for (int i = 0; i < kNumAcc; ++i) sums[i] = 0.0;
  for (int v = 0; v < h; ++v)
    for (int u0 = 0; u0 < w; u0 += kTileW) {
      float acc[kNumAcc];
      for (int i = 0; i < kNumAcc; ++i) acc[i] = 0.0f;
      for (int u = u0; u < u0 + kTileW && u < w; ++u) {
        const int idx = v * w + u;
        if (g_schedule != 0) {
          flat_pixel(c, KT, Pp, first, u, v, acc, &res[2 * size_t(idx)]);
          continue;
        }
        PixelTerms t;
        if (!pixel_residual(c.g, KT, c.A, c.B, c.R[idx], u, v, t)) continue;
        res[2 * size_t(idx)] = t.r0;
        res[2 * size_t(idx) + 1] = t.r1;
        const float wgt = first ? 1.0f : tdist_weight(t.r0, t.r1, Pp);
        accumulate_pixel(acc, t, wgt);
      }
      for (int i = 0; i < kNumAcc; ++i) sums[i] += double(acc[i]);
    }
}

double loglik_sum(const std::vector<float>& res, const double* sums) {
  const int n = int(sums[kAccN] + 0.5);
  if (n < 6) return 0.0;
  float C[3], P[4];
  const double d = double(n) - 3.0;
  scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, C, P);
  double total = 0.0;
  for (size_t i = 0; i < res.size() / 2; ++i)
    if (res[2 * i] == res[2 * i]) total += std::log(1.0 + 0.2 * double(mahalanobis(res[2 * i], res[2 * i + 1], P)));
  return total;
}

}  // namespace

extern "C" {

void emul_set_schedule(int schedule) { g_schedule = schedule; }

int emul_level_iteration(const emul_level* L, const float T34[12], const float P_prev[4], int first,
                         dvo_hip_iteration_out* out, float* residuals) {
  LevelCtx c;
  make_level(*L, c);
  float KT[12];
  make_KT(c.g.fx, c.g.fy, c.g.ox, c.g.oy, T34, KT);
  double sums[kNumAcc];
  std::vector<float> res;
  sweep(c, KT, P_prev, first != 0, sums, res);
  const double ll_sum = loglik_sum(res, sums);
  std::memset(out, 0, sizeof(*out));
  const int n = int(sums[kAccN] + 0.5);
  out->n = n;
  out->n_selected = L->n_selected;
  if (residuals) std::memcpy(residuals, res.data(), res.size() * sizeof(float));
  if (n < 6) return 1;
  float C[3], P[4];
  const double d = double(n) - 3.0;
  scale_to_precision(sums[kAccS] / d, sums[kAccS + 1] / d, sums[kAccS + 2] / d, C, P);
  for (int i = 0; i < 3; ++i) out->scale_cov[i] = C[i];
  for (int i = 0; i < 4; ++i) out->precision[i] = P[i];
  const double det = double(P[0]) * double(P[3]) - double(P[1]) * double(P[2]);
  out->neg_loglik = -(0.5 * double(n) * std::log(det) - 3.5 * ll_sum);
  const double p00 = P[0], p01 = P[1], p11 = P[3];
  int o = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      const double a = p00 * sums[kAccJ00 + o] + p01 * sums[kAccJ01 + o] + p11 * sums[kAccJ11 + o];
      out->A[i * 6 + j] = a;
      out->A[j * 6 + i] = a;
      ++o;
    }
  for (int i = 0; i < 6; ++i) out->b[i] = -(p00 * sums[kAccB00 + i] + p01 * sums[kAccB01 + i] + p11 * sums[kAccB11 + i]);
  return 0;
}

// the batched driver of capi.hip for one pair, with the kernels replaced by the sweeps above
int emul_match(const emul_level* levels, const dvo_hip_config* cfg, dvo_hip_result* result,
               dvo_hip_level_stats* lstats, int cap_levels, dvo_hip_iteration_stats* istats, int cap_iters) {
  SolverParams prm;
  prm.max_iterations = cfg->max_iterations_per_level;
  prm.first_level = cfg->first_level;
  prm.last_level = cfg->last_level;
  prm.use_initial_estimate = cfg->use_initial_estimate;
  prm.precision = cfg->precision;
  prm.mu = cfg->mu;
  prm.cap_iters = cap_iters;
  prm.cap_levels = cap_levels;
  prm.max_points_level0 = levels[0].w * levels[0].h;
  prm.want_condition_number = 1;
  prm.record_prefilled = 0;
  PairState st;
  std::memset(&st, 0, sizeof(st));
  gn_init_pair(st, prm, result->transformation);
  for (int level = cfg->first_level; level >= cfg->last_level; --level) {
    LevelCtx c;
    make_level(levels[level], c);
    gn_level_begin(st, prm, c.g, level, levels[level].n_selected, lstats);
    while (st.active) {
      double sums[kNumAcc];
      std::vector<float> res;
      sweep(c, st.KT, st.P_prev, st.first != 0, sums, res);
      const double ll_sum = loglik_sum(res, sums);
      gn_step(st, prm, c.g, sums, ll_sum, lstats, istats);
    }
  }
  gn_finish(st, prm, lstats, istats, result);
  return 0;
}

// The same match with the control flow of the resident kernel (dvo_slam_amd/csrc/align_resident.hip, the block its solver wavefront
// runs after every exchange): the loop body runs SPECULATIVELY, before the log-likelihood of its pass is known; the verdict arrives
// one pass later (gn_commit_loglik); a rejection restores the state the pass started from (double-buffered snapshots, the error
// chain taken from the live state) and runs the loop body once more in full form, which takes the revert path.  The sweeps are
// the host sweeps above, so this driver must reproduce emul_match bit for bit: state, results, level and iteration records.
int emul_match_speculative(const emul_level* levels, const dvo_hip_config* cfg, dvo_hip_result* result,
                           dvo_hip_level_stats* lstats, int cap_levels, dvo_hip_iteration_stats* istats, int cap_iters) {
  SolverParams prm;
  prm.max_iterations = cfg->max_iterations_per_level;
  prm.first_level = cfg->first_level;
  prm.last_level = cfg->last_level;
  prm.use_initial_estimate = cfg->use_initial_estimate;
  prm.precision = cfg->precision;
  prm.mu = cfg->mu;
  prm.cap_iters = cap_iters;
  prm.cap_levels = cap_levels;
  prm.max_points_level0 = levels[0].w * levels[0].h;
  prm.want_condition_number = 1;
  prm.record_prefilled = 0;
  PairState st;
  std::memset(&st, 0, sizeof(st));
  gn_init_pair(st, prm, result->transformation);
  volatile int release_word = 0;
  for (int level = cfg->first_level; level >= cfg->last_level; --level) {
    LevelCtx c;
    make_level(levels[level], c);
    const int slot = st.n_levels;
    gn_level_begin(st, prm, c.g, level, levels[level].n_selected, lstats);
    PairState st_before[2];
    dvo_hip_level_stats lvl_before[2];
    st_before[0] = st;
    if (slot < cap_levels) lvl_before[0] = lstats[slot];
    double sums[2][kNumAcc];
    std::vector<float> res;
    GnSpeculation spec;
    std::memset(&spec, 0, sizeof(spec));
    bool pending = false;
    double ll_pending = 0.0;
    int rec_pending = -1;
    for (int pass = 0;; ++pass) {
      const int cur = pass & 1, prev = cur ^ 1;
      const bool do_sweep = st.active != 0;
      if (do_sweep) sweep(c, st.KT, st.P_prev, st.first != 0, sums[cur], res);     // at the speculatively advanced estimate
      bool restore = false, over = false;
      if (pending) {                                                                // the verdict on the pass before
        dvo_hip_iteration_stats dummy;
        dvo_hip_iteration_stats& rec = rec_pending < cap_iters ? istats[rec_pending] : dummy;
        restore = !gn_commit_loglik(st, spec, ll_pending, rec);
        pending = false;
      }
      if (restore) {
        const double error_now = st.error, last_error_now = st.last_error;
        st = st_before[prev];
        if (slot < cap_levels) lstats[slot] = lvl_before[prev];
        st.error = error_now;
        st.last_error = last_error_now;
        GnSpeculation replay = spec;
        replay.replay_reject = 1;
        gn_step(st, prm, c.g, sums[prev], ll_pending, lstats, istats, &replay);
        over = true;
      } else if (!do_sweep) {
        over = true;
      } else {
        rec_pending = st.n_iters_total;
        spec.replay_reject = 0;
        spec.record_prefilled = 0;
        spec.defer_information = 1;
        spec.release_word = &release_word;
        spec.release_value = pass + 1;
        gn_step(st, prm, c.g, sums[cur], 0.0, lstats, istats, &spec);
        if (spec.information_ready && rec_pending < cap_iters)
          for (int i = 0; i < 36; ++i) istats[rec_pending].information[i] = st.A_last[i];
        pending = spec.needs_loglik != 0;
        if (!pending) over = true;
        else ll_pending = loglik_sum(res, sums[cur]);                               // travels with the next exchange
      }
      if (over) break;
      st_before[prev] = st;                                                         // what the next pass starts from
      if (slot < cap_levels) lvl_before[prev] = lstats[slot];
    }
  }
  gn_finish(st, prm, lstats, istats, result);
  return 0;
}

// The exchange of the resident kernel (align_resident.hip: rows of 8-byte {value, sequence number} slots, relaxed stores, relaxed
// polling loads, rows double-buffered by the parity of the exchange count) with host threads in the place of workgroups: every
// thread does `exchanges` rounds of  write my row -> gather everybody's rows of this round -> check the sum , at its own, randomly
// disturbed pace.  Returns the number of rounds in which a thread gathered anything but the values of that very round (0 = the
// protocol never hands out stale or torn rows, however far the fastest thread runs ahead -- it cannot pass the round its slowest
// peer has not written yet).
int emul_exchange_stress(int group, int exchanges, int slots, unsigned seed) {
  std::vector<std::atomic<unsigned long long>> rows(size_t(group) * 2 * slots);
  for (auto& r : rows) r.store(0, std::memory_order_relaxed);
  std::atomic<int> bad(0);
  auto value = [](int t, int k, int s) { return unsigned(t * 2654435761u + k * 40503u + s * 97u); };
  auto worker = [&](int t) {
    unsigned rng = seed * 747796405u + unsigned(t) * 2891336453u + 1u;
    for (int k = 1; k <= exchanges; ++k) {
      rng = rng * 1664525u + 1013904223u;
      if ((rng >> 28) == 0) std::this_thread::sleep_for(std::chrono::microseconds((rng >> 8) & 63));   // a slow sweep now and then
      const int parity = k & 1;
      for (int s = 0; s < slots; ++s)
        rows[(size_t(t) * 2 + parity) * slots + s].store((static_cast<unsigned long long>(unsigned(k)) << 32) | value(t, k, s), std::memory_order_relaxed);
      for (int s = 0; s < slots; ++s) {
        unsigned long long sum = 0, want = 0;
        for (int j = 0; j < group; ++j) {
          unsigned long long v;
          for (;;) {
            v = rows[(size_t(j) * 2 + parity) * slots + s].load(std::memory_order_relaxed);
            if (unsigned(v >> 32) == unsigned(k)) break;
            std::this_thread::yield();                       // (more threads than cores: let the slow peer run)
          }
          sum += unsigned(v);
          want += value(j, k, s);
        }
        if (sum != want) bad.fetch_add(1);
      }
    }
  };
  std::vector<std::thread> threads;
  for (int t = 0; t < group; ++t) threads.emplace_back(worker, t);
  for (auto& th : threads) th.join();
  return bad.load();
}

// every flat pixel index of a w x h level through the division-free row / column computation of the linear walk: mismatches
int emul_locate_check(int w, int h) {
  const float inv_w = 1.0f / float(w);
  int bad = 0;
  for (int idx = 0; idx < w * h; ++idx) {
    int row, col;
    locate_pixel(idx, w, inv_w, row, col);
    bad += (row != idx / w || col != idx % w);
  }
  return bad;
}

// exposed for unit tests of the device SE(3) / solve code
void emul_se3_exp(const double x[6], double T[16]) {
  SE3d S;
  se3_exp(x, S);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = S.R[i * 3 + j];
    T[i * 4 + 3] = S.t[i];
  }
  T[12] = T[13] = T[14] = 0;
  T[15] = 1;
}

void emul_se3_log(const double T[16], double x[6]) {
  SE3d S;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) S.R[i * 3 + j] = T[i * 4 + j];
    S.t[i] = T[i * 4 + 3];
  }
  se3_log(S, x);
}

int emul_solve6(const double A[36], const double b[6], double x[6]) { return solve6(A, b, x) ? 0 : 1; }
void emul_sym6_eigenvalues(const double A[36], double ev[6]) { sym6_eigenvalues(A, ev); }

}  // extern "C"
