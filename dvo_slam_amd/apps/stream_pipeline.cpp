// stream_pipeline.cpp -- the steady state of a streaming consumer of the C-ABI, as one C function: re-ingest the next batch of
// raw planes in the roles its frames will play (build stream), then align the current batch (main stream).  This is the loop
// bench.py times; it is written against include/dvo_hip.h only (no access to the library's internals) and lives in its own
// shared object so that the benchmark harness pays one foreign call per step instead of a dozen plus array marshalling.
//
// Reference call pattern it stands for: dvo_benchmark/src/benchmark_slam.cpp:327-383 (load pair -> create pyramid -> track),
// with the proposals of dvo_slam/src/keyframe_graph.cpp:576-593 as the source of independent pairs.
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
  static const int defer_max = std::getenv("DVO_STREAM_DEFER_MAX") ? std::atoi(std::getenv("DVO_STREAM_DEFER_MAX")) : 256;
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

}  // extern "C"
