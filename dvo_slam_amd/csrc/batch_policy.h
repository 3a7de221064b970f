// batch_policy.h -- every batch-size threshold of the host driver (capi.hip), as a function of the device's compute units, in ONE place.
//
// The thresholds were measured on MI355X (256 compute units, 160 KB of LDS and 512 vector registers per SIMD each) over rounds 2-6
// (HISTORY.md); what they depend on is how many WORKGROUPS of a kernel the chip holds at once, so they are stated
// in pairs per compute unit and follow the device the context runs on -- on 256 compute units every function below returns the constant
// rounds 2-5 had pinned (tests/test_capi.py::test_batch_policy_reproduces_the_measured_thresholds).  Host code only, no HIP types.
#pragma once

namespace dvo_hip {

struct BatchPolicy {
  int cus;                                                     // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  explicit BatchPolicy(int compute_units) : cus(compute_units > 0 ? compute_units : 256) {}

  // ---- the resident (latency) kernel: every workgroup of a launch with groups must be on the device at once -- one 512-thread
  // workgroup of up to 256 registers per compute unit ----
  // whole match, results straight into pinned host memory: up to one pair per 16 compute units (16 pairs; each pair then gets 16
  // workgroups, the most the exchange pays for)
  int resident_direct_max_pairs() const { return cus / 16; }
  // coarse levels resident: as long as every pair gets at least two workgroups on half the chip (64 pairs)
  bool resident_takes_coarse_levels(int n) const { return n * 4 <= cus; }
  // first level alone, two workgroups per pair that leave an eighth of the chip to the background ingest: up to 7/16 pair per compute
  // unit (112 pairs); from an eighth of a pair per compute unit on (32 pairs) when the frames were streamed in without their taps
  bool resident_first_level_fits(int n) const { return n * 2 * 8 <= cus * 7; }
  bool taps_missing_prefers_first_level_only(int n) const { return n * 8 >= cus; }
  // a role-aware ingest of this many frames leaves the gathered taps out (plane C alone: a third of the bytes): an eighth of a frame
  // per compute unit (32 frames)
  bool ingest_skips_taps(int n_frames) const { return n_frames * 8 >= cus; }

  // ---- the launch path ----
  // short tiles (2 rows per wavefront) on the gathering levels: from one pair per compute unit on the batch fills the chip whatever the
  // tile (256 pairs)
  bool short_gather_tiles(int n) const { return n >= cus; }
  // smallest launch that counts as filling the chip when the tile height is chosen for a smaller batch: two workgroups per compute unit
  // for up to 8 pairs, four below a quarter pair per compute unit, eight above (512 / 1024 / 2048)
  int min_workgroups(int n) const { return n <= cus / 32 ? 2 * cus : n * 4 < cus ? 4 * cus : 8 * cus; }
  // the level hand-over inside the solver steps (no launch between levels, none at the end): up to one pair per compute unit (256) --
  // beyond that the step that carries it is the launch's slowest workgroup too often
  bool level_hand_over(int n) const { return n <= cus; }
  // the log-likelihood pass of a 320 x 240 level inside the solver step: from two pairs per compute unit on (512)
  bool fused_loglik_on_large_levels(int n) const { return n >= 2 * cus; }
  // workgroups per pair of a log-likelihood launch (each begins with a ~10 us reduction of the pair's scale sums): 32, 16 from 3/16
  // pair per compute unit on (48 pairs), 8 from one pair per compute unit on (256)
  int loglik_blocks(int n) const { return n >= cus ? 8 : n * 16 >= cus * 3 ? 16 : 32; }
  // steps of the first level enqueued ahead of a deferred ingest's host work: 7 from 3/4 pair per compute unit on (192), else 3
  int deferred_ingest_lead(int n) const { return n * 4 >= cus * 3 ? 7 : 3; }
  // k_finish as a launch of its own behind the last level: where there is no hand-over
  bool finish_launch(int n) const { return !level_hand_over(n); }
  // the solver step of the smallest levels in two-wavefront workgroups: a batch beyond two four-wavefront workgroups per compute unit (512)
  bool solver_two_waves(int n) const { return n > 2 * cus; }

  // workgroups per pair of a small level's sweep (align_small.hip: three 53-KB workgroups per compute unit): as many as give the batch whole
  // rounds of the chip's slots -- 3 from one pair per compute unit on (1024 pairs: four rounds), 6 at half a pair (128 pairs: one round) --
  // at most 8 (every workgroup copies the whole level into LDS first)
  int small_level_tiles(int n) const {
    const int t = (3 * cus + n / 2) / (n > 0 ? n : 1);
    return t < 3 ? 3 : t > 8 ? 8 : t;
  }

  // ---- a streaming caller (dvo_slam_amd/apps/stream_pipeline.cpp, bench.py) ----
  // the ingest of the next batch deferred behind the alignment's first launches: up to one pair per compute unit (256)
  int defer_ingest_max_pairs() const { return cus; }
  // the grid a background frame build is capped at: one workgroup per compute unit (256)
  int background_build_workgroups() const { return cus; }
};

}  // namespace dvo_hip
