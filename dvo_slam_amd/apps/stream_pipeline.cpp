// stream_pipeline.cpp -- the steady state of a streaming consumer of the C-ABI, as one C function: re-ingest the next batch of
// raw planes in the roles its frames will play (build stream), then align the current batch (main stream).  This is the loop
// bench.py times; it is written against include/dvo_hip.h only (no access to the library's internals) and lives in its own
// shared object so that the benchmark harness pays one foreign call per step instead of a dozen plus array marshalling.
//
// Reference call pattern it stands for: dvo_benchmark/src/benchmark_slam.cpp:327-383 (load pair -> create pyramid -> track),
// with the proposals of dvo_slam/src/keyframe_graph.cpp:576-593 as the source of independent pairs.
#include <pthread.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dvo_hip.h"

extern "C" {

// One step on frame sets `next` (to be re-ingested: n reference frames from (grey_ref, raw_ref), n current frames from
// (grey_cur, raw_cur), device pointers) and `now` (to be aligned: results[n], identity initial guess).  The frames of `next_refs` are
// good for the reference role with this configuration's thresholds until they are ingested again (they keep no copy of their raw planes).
int dvo_stream_step(dvo_hip_context* ctx, int n, dvo_hip_frame* const* next_refs, dvo_hip_frame* const* next_curs,
                    const void* const* grey_ref, const void* const* raw_ref, const void* const* grey_cur, const void* const* raw_cur,
                    float depth_scale, dvo_hip_frame* const* now_refs, dvo_hip_frame* const* now_curs, const dvo_hip_config* cfg,
                    dvo_hip_result* results) {
  int rc = DVO_HIP_OK;
  // the ingest of the next batch is handed over first but carried out behind the first launches of the alignment (option
  // "defer_ingest": the host's share of it no longer delays the alignment's start)
  // (measured, scripts/r5_midsize.py, builds alternated on one box: 16 pairs 0.575 -> 0.553 ms per step, 128 pairs 1.863 -> 1.825; a
  // 1024-pair step is better off with the early ingest, 11.57 vs 11.74 ms -- it hides beside the latency-bound coarse levels, and half a
  // millisecond later there is less of them left)
  // (up to one pair per compute unit: the library's batch-size policy for the context's device, dvo_slam_amd/csrc/batch_policy.h)
  static const int defer_env = std::getenv("DVO_STREAM_DEFER_MAX") ? std::atoi(std::getenv("DVO_STREAM_DEFER_MAX")) : -1;
  long long defer_policy = 256;
  if (defer_env < 0) (void)dvo_hip_get_counter(ctx, "defer_ingest_max_pairs", &defer_policy);
  const int defer_max = defer_env >= 0 ? defer_env : int(defer_policy);
  // (per-call flags, not the context's options: the caller's own settings stay what they were, and another thread on the context
  // never sees a toggled option -- round-5 advisor finding)
  const bool deferring = n <= defer_max && next_refs && now_refs;
  const unsigned defer_flag = deferring ? DVO_HIP_INGEST_DEFER : 0u;
  if (next_refs) {
    // a frame this loop ingests as a reference is ingested again before it plays any other part: no copy of its raw planes (3 of
    // the 16 bytes per pixel a reference frame's ingest moves; 1024-pair step 11.48 -> 11.37 ms)
    rc = dvo_hip_frames_update_raw_device_as_ex(ctx, n, next_refs, grey_ref, raw_ref, depth_scale, DVO_HIP_ROLE_REFERENCE, cfg,
                                                defer_flag | DVO_HIP_INGEST_NO_RAW_COPY);
    if (rc == DVO_HIP_OK)
      rc = dvo_hip_frames_update_raw_device_as_ex(ctx, n, next_curs, grey_cur, raw_cur, depth_scale, DVO_HIP_ROLE_CURRENT, cfg, defer_flag);
  }
  if (rc == DVO_HIP_OK && now_refs) {
    static const double identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < n; ++i) std::memcpy(results[i].transformation, identity, sizeof(identity));   // Result is in/out
    rc = dvo_hip_match_batch(ctx, n, now_refs, now_curs, cfg, results, nullptr, 0, nullptr, 0);
  }
  if (deferring) {
    const int rc_off = dvo_hip_flush_deferred(ctx);              // (carries out whatever is still recorded)
    if (rc == DVO_HIP_OK) rc = rc_off;
  }
  return rc;
}

// The fixed-size record of an alignment that travels between the ranks of a multi-GPU job (dvo_slam_amd/parallel.py: RECORD = 32
// doubles): twist (v, omega) of the transform (closed-form log, small-angle safe) | upper triangle of the information matrix (21) |
// log-likelihood | flag (0) | padding (3).  Same formulas as parallel.twists_of / pack_records (tests/test_parallel.py compares them);
// here so that a step of the streaming loop hands its records to the all-gather without per-step array work in the host language.
void dvo_stream_pack_records(int n, const dvo_hip_result* results, double* records) {
  for (int i = 0; i < n; ++i) {
    const double* T = results[i].transformation;               // row-major 4 x 4
    double* rec = records + size_t(i) * 32;
    for (int k = 0; k < 32; ++k) rec[k] = 0.0;
    const double R[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
    const double t[3] = {T[3], T[7], T[11]};
    double c = ((R[0][0] + R[1][1] + R[2][2]) - 1.0) * 0.5;
    c = c < -1.0 ? -1.0 : c > 1.0 ? 1.0 : c;
    const double th = std::acos(c);
    const double axis[3] = {0.5 * (R[2][1] - R[1][2]), 0.5 * (R[0][2] - R[2][0]), 0.5 * (R[1][0] - R[0][1])};
    const double scale = th < 1e-6 ? 1.0 + th * th / 6.0 : th / std::sin(th);
    const double w[3] = {axis[0] * scale, axis[1] * scale, axis[2] * scale};
    const double O[3][3] = {{0.0, -w[2], w[1]}, {w[2], 0.0, -w[0]}, {-w[1], w[0], 0.0}};
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double cc = 1.0 / 12.0;
    if (!(th2 < 1e-10)) {
      const double a = std::sqrt(th2);
      cc = (1.0 - a * std::cos(a / 2) / (2 * std::sin(a / 2))) / th2;
    }
    for (int r = 0; r < 3; ++r) {
      double v = 0.0;
      for (int k = 0; k < 3; ++k) {
        double oo = 0.0;
        for (int m = 0; m < 3; ++m) oo += O[r][m] * O[m][k];
        v += ((r == k ? 1.0 : 0.0) - 0.5 * O[r][k] + cc * oo) * t[k];
      }
      rec[r] = v;
      rec[3 + r] = w[r];
    }
    int o = 6;
    for (int r = 0; r < 6; ++r)
      for (int k = r; k < 6; ++k) rec[o++] = results[i].information[r * 6 + k];
    rec[27] = results[i].loglik;
  }
}

// The same step fed from HOST memory (SURVEY.md 8d config 4 "incl. H2D of 2 planes per frame"): the raw planes of `next` are DMA-ed
// from pinned host memory on the upload stream, built on the build stream, while `now` is aligned on the main stream.
int dvo_stream_step_host(dvo_hip_context* ctx, int n, dvo_hip_frame* const* next_refs, dvo_hip_frame* const* next_curs,
                         const uint8_t* const* grey_ref, const uint16_t* const* raw_ref, const uint8_t* const* grey_cur,
                         const uint16_t* const* raw_cur, float depth_scale, dvo_hip_frame* const* now_refs, dvo_hip_frame* const* now_curs,
                         const dvo_hip_config* cfg, dvo_hip_result* results) {
  static const bool trace = std::getenv("DVO_STREAM_TRACE") != nullptr;   // where the host thread's time of a step goes (stderr)
  const auto t0 = std::chrono::steady_clock::now();
  int rc = DVO_HIP_OK;
  if (next_refs) {
    rc = dvo_hip_frames_update_raw_as_ex(ctx, n, next_refs, grey_ref, raw_ref, depth_scale, DVO_HIP_ROLE_REFERENCE, cfg, DVO_HIP_INGEST_NO_RAW_COPY);   // (as in dvo_stream_step)
    if (rc != DVO_HIP_OK) return rc;
    rc = dvo_hip_frames_update_raw_as_ex(ctx, n, next_curs, grey_cur, raw_cur, depth_scale, DVO_HIP_ROLE_CURRENT, cfg, 0u);
    if (rc != DVO_HIP_OK) return rc;
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (!now_refs) return rc;
  static const double identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int i = 0; i < n; ++i) std::memcpy(results[i].transformation, identity, sizeof(identity));
  rc = dvo_hip_match_batch(ctx, n, now_refs, now_curs, cfg, results, nullptr, 0, nullptr, 0);
  if (trace) {
    const auto t2 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "step_host: enqueue of the next batch %.2f ms, match %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  return rc;
}

// ---- several LANES on one GPU (round 6) ---------------------------------------------------------------------------------------------
// The steady state of a streaming consumer with enough independent pairs per step (two per compute unit and more): the pairs are dealt
// to G lanes -- lane l takes the pairs l, l + G, ... -- each with a context of its own on the SAME device (own streams, own scratch) and
// a host thread that runs dvo_stream_step on the lane's shard, step after step, WITHOUT waiting for the other lanes: they drift out of
// phase, and one lane's latency-bound phases (coarse sweeps, solver steps, the host's share of a step) pass beside another's sweeps.
// This is the reference's own model for independent pairs -- the workers of a tbb::parallel_reduce each run whole match() calls
// (dvo_slam/src/keyframe_graph.cpp:576-593) -- with the GPU's queues in place of the cores.  Measured (scripts/r6_groups.py, MI355X,
// 1024 pairs of 640 x 480 per step): 1 lane 11.5-11.7 ms per step, 2 lanes 10.9, 3 lanes 10.8-10.9; lanes that meet after every step
// keep about half of it.  A step's results are collected in submission order, up to `depth` steps behind the submission.
// (pthreads and malloc rather than <thread> / new: the object binds the C-ABI and libc only, tests/test_capi.py.)
struct dvo_stream_lane {
  dvo_hip_context* ctx;
  int n;
  dvo_hip_frame* const* refs[2];                 // the lane's two frame sets (double buffer), n frames each
  dvo_hip_frame* const* curs[2];
  const void* const* grey_ref; const void* const* raw_ref; const void* const* grey_cur; const void* const* raw_cur;   // n device pointers each
};

struct dvo_stream_lanes {
  int n_lanes, depth;
  float depth_scale;
  dvo_hip_config cfg;
  struct Lane {
    dvo_stream_lane d;
    dvo_stream_lanes* owner;
    pthread_t thread;
    dvo_hip_result* slots;                       // depth x n results
    long long done;                              // steps finished (guarded by owner->m)
    int rc;                                      // first failure of the lane
  }* lanes;
  pthread_mutex_t m;
  pthread_cond_t cv;
  long long submitted, collected;
  int quit;
};

static void* dvo_stream_lane_main(void* arg) {
  dvo_stream_lanes::Lane* lane = static_cast<dvo_stream_lanes::Lane*>(arg);
  dvo_stream_lanes* L = lane->owner;
  for (;;) {
    pthread_mutex_lock(&L->m);
    while (!L->quit && lane->done >= L->submitted) pthread_cond_wait(&L->cv, &L->m);
    if (L->quit) {
      pthread_mutex_unlock(&L->m);
      return nullptr;
    }
    const long long k = lane->done;
    pthread_mutex_unlock(&L->m);
    const int now = int(k & 1), nxt = int((k + 1) & 1);        // step k aligns set k % 2 and re-ingests the other for step k + 1
    const int rc = dvo_stream_step(lane->d.ctx, lane->d.n, lane->d.refs[nxt], lane->d.curs[nxt], lane->d.grey_ref, lane->d.raw_ref, lane->d.grey_cur,
                                   lane->d.raw_cur, L->depth_scale, lane->d.refs[now], lane->d.curs[now], &L->cfg,
                                   lane->slots + size_t(k % L->depth) * lane->d.n);
    pthread_mutex_lock(&L->m);
    if (rc != DVO_HIP_OK && lane->rc == DVO_HIP_OK) lane->rc = rc;
    lane->done = k + 1;
    pthread_cond_broadcast(&L->cv);
    pthread_mutex_unlock(&L->m);
  }
}

// depth: steps that may be submitted ahead of their collection (>= 1; 2 lets the lanes drift a step apart).  The frame set 0 of every
// lane is ingested here (the pipeline is primed); returns null on failure.
dvo_stream_lanes* dvo_stream_lanes_create(int n_lanes, const dvo_stream_lane* lanes, float depth_scale, const dvo_hip_config* cfg, int depth) {
  if (n_lanes < 1 || !lanes || !cfg || depth < 1) return nullptr;
  dvo_stream_lanes* L = static_cast<dvo_stream_lanes*>(std::calloc(1, sizeof(dvo_stream_lanes)));
  if (!L) return nullptr;
  L->n_lanes = n_lanes; L->depth = depth; L->depth_scale = depth_scale; L->cfg = *cfg;
  L->lanes = static_cast<dvo_stream_lanes::Lane*>(std::calloc(size_t(n_lanes), sizeof(dvo_stream_lanes::Lane)));
  pthread_mutex_init(&L->m, nullptr);
  pthread_cond_init(&L->cv, nullptr);
  bool ok = L->lanes != nullptr;
  for (int l = 0; ok && l < n_lanes; ++l) {
    dvo_stream_lanes::Lane& lane = L->lanes[l];
    lane.d = lanes[l];
    lane.owner = L;
    lane.slots = static_cast<dvo_hip_result*>(std::calloc(size_t(depth) * size_t(lanes[l].n), sizeof(dvo_hip_result)));
    ok = lane.slots != nullptr &&
         dvo_stream_step(lane.d.ctx, lane.d.n, lane.d.refs[0], lane.d.curs[0], lane.d.grey_ref, lane.d.raw_ref, lane.d.grey_cur, lane.d.raw_cur, depth_scale,
                         nullptr, nullptr, cfg, nullptr) == DVO_HIP_OK;
  }
  int started = 0;
  for (; ok && started < n_lanes; ++started) ok = pthread_create(&L->lanes[started].thread, nullptr, dvo_stream_lane_main, &L->lanes[started]) == 0;
  if (!ok) {
    pthread_mutex_lock(&L->m);
    L->quit = 1;
    pthread_cond_broadcast(&L->cv);
    pthread_mutex_unlock(&L->m);
    for (int l = 0; l < started - 1; ++l) pthread_join(L->lanes[l].thread, nullptr);
    for (int l = 0; L->lanes && l < n_lanes; ++l) std::free(L->lanes[l].slots);
    std::free(L->lanes);
    std::free(L);
    return nullptr;
  }
  return L;
}

// one more step for every lane; DVO_HIP_ERR_CAPACITY when `depth` steps are already waiting to be collected
int dvo_stream_lanes_submit(dvo_stream_lanes* L) {
  if (!L) return DVO_HIP_ERR_INVALID;
  pthread_mutex_lock(&L->m);
  const bool room = L->submitted - L->collected < L->depth;
  if (room) {
    L->submitted += 1;
    pthread_cond_broadcast(&L->cv);
  }
  pthread_mutex_unlock(&L->m);
  return room ? DVO_HIP_OK : DVO_HIP_ERR_CAPACITY;
}

// waits for the oldest submitted step on every lane; out[l + j * n_lanes] = result j of lane l (the order the pairs were dealt in).
// Returns the first failure of any lane's steps (the error text: dvo_hip_last_error of that lane's context).
int dvo_stream_lanes_collect(dvo_stream_lanes* L, dvo_hip_result* out) {
  if (!L || !out) return DVO_HIP_ERR_INVALID;
  pthread_mutex_lock(&L->m);
  if (L->collected >= L->submitted) {
    pthread_mutex_unlock(&L->m);
    return DVO_HIP_ERR_INVALID;                                 // nothing outstanding
  }
  const long long k = L->collected;
  int rc = DVO_HIP_OK;
  for (int l = 0; l < L->n_lanes; ++l) {
    while (L->lanes[l].done <= k) pthread_cond_wait(&L->cv, &L->m);
    if (rc == DVO_HIP_OK) rc = L->lanes[l].rc;
  }
  pthread_mutex_unlock(&L->m);                                  // (slot k % depth of a lane is not written again before `collected` moves)
  for (int l = 0; l < L->n_lanes; ++l) {
    const dvo_hip_result* src = L->lanes[l].slots + size_t(k % L->depth) * L->lanes[l].d.n;
    for (int j = 0; j < L->lanes[l].d.n; ++j) out[size_t(l) + size_t(j) * L->n_lanes] = src[j];
  }
  pthread_mutex_lock(&L->m);
  L->collected = k + 1;
  pthread_mutex_unlock(&L->m);
  return rc;
}

void dvo_stream_lanes_destroy(dvo_stream_lanes* L) {
  if (!L) return;
  pthread_mutex_lock(&L->m);
  L->quit = 1;
  pthread_cond_broadcast(&L->cv);
  pthread_mutex_unlock(&L->m);
  for (int l = 0; l < L->n_lanes; ++l) {
    pthread_join(L->lanes[l].thread, nullptr);
    std::free(L->lanes[l].slots);
  }
  pthread_mutex_destroy(&L->m);
  pthread_cond_destroy(&L->cv);
  std::free(L->lanes);
  std::free(L);
}

}  // extern "C"
