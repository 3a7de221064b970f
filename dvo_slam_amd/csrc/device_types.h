// device_types.h -- PODs shared by the host side (capi.hip) and the gfx950 kernels.
#pragma once

#include <stdint.h>

#include "hd_compat.h"

#include "../../include/dvo_hip.h"

namespace dvo_hip {

constexpr int kMaxLevels = DVO_HIP_MAX_LEVELS;
constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kWavesPerBlock = 4;
constexpr int kTileW = 64;           // one wavefront spans 64 consecutive pixels of a row

// Accumulator layout of the fused residual/Jacobian/reduce kernel (all float):
//   0            n  (count of valid constraints)
//   1..3         S00 S01 S11          = sum w r r^T                    (scale, DT.cpp:295)
//   4..24        sum w J0_i J0_j  (i<=j, row-major upper triangle)
//   25..45       sum w J1_i J1_j
//   46..66       sum w (J0_i J1_j + J0_j J1_i)
//   67..72       sum w J0_k r0
//   73..78       sum w (J0_k r1 + J1_k r0)
//   79..84       sum w J1_k r1
// The 2x2 precision P is only known after the pass (it is the inverse of S/(n-3)), so the pass
// accumulates the P-independent Gram sums and the tiny per-pair solver kernel contracts them with P.
constexpr int kAccN = 0, kAccS = 1, kAccJ00 = 4, kAccJ11 = 25, kAccJ01 = 46, kAccB00 = 67, kAccB01 = 73, kAccB11 = 79;
constexpr int kNumAcc = 85;
constexpr int kAccStride = 88;       // padded row of the per-block partial table

struct LevelGeom {                    // identical for every pair of a batch (one camera)
  int w, h;
  float fx, fy, ox, oy;
  float wi_x, wi_y;                   // 0.5f * fx / 255.0f, 0.5f * fy / 255.0f (intensity-gradient weights)
  // half of wi_x, wi_y, fx, fy: the f16 schedule's gradient channels are plain differences (no central difference's 0.5) and the factor
  // rides here (there is no scalar float multiply on this part: formed in the kernel it is four vector instructions per pixel row)
  float half_wi_x, half_wi_y, half_fx, half_fy;
  const float* tx;                    // (u - ox)/fx per column, the reference's pointcloud_template_
  const float* ty;                    // (v - oy)/fy per row
  int tiles_x, tiles_y;
  // 1: the sweep kernel walks the level as ONE row of w*h pixels in 64-pixel segments (tiles_x = 1) instead of 64-column
  // tiles.  Used when w is not a multiple of 64: an 80-pixel-wide level otherwise runs every second wavefront-row with 16
  // of its 64 lanes (37 % of the issue slots wasted), a 160-wide one with 32 (17 %).
  int linear;
  // Reference-compatible arithmetic (option "ref_compat"; null = off): the reference's projection and weights multiply with
  // _mm_rcp_ps (dense_tracking_impl.cpp:192, :700), an approximate reciprocal whose value is a lookup on the leading mantissa bits of
  // its operand -- different on different CPUs.  The table is dumped from the host CPU's own instruction when the option is switched
  // on: rcp(1.m x 2^e) = rcp_table[m >> rcp_shift] x 2^-e (pixel_math.h::rcp_like_the_host).
  const float* rcp_table;
  int rcp_shift;
  // 1: behind the 2^(23 - rcp_shift) floats of the table lie as many 16-bit entries, (bits(value) - 0x3f000000) >> 8 -- every value is in
  // (0.5, 1] and the instruction leaves the low mantissa bits zero (checked when the table is dumped), at most 8 KB: the contracted
  // window sweep keeps that copy in LDS (align_fast.hip, COMPAT 2)
  int rcp_packed;
  // 1: the sweep stores only the residual pairs of CONSTRAINTS (roughly every second pixel), packed: tile t of the level owns 1024
  // entries of the pair's residual buffer, wavefront q of the tile the entries [256 q, 256 q + count_q), in the order it met them;
  // the four counts ride in the two spare floats of the tile's partial row (kAccCounts).  The log-likelihood pass -- the only reader --
  // then moves half the bytes.  Contracted window sweep only (align_fast.hip); 0 = one pair per pixel at its pixel's place, NaN
  // where there is no constraint (every other sweep, and whenever the caller wants the residuals by pixel).
  int compact;
  // 1: a SMALL level of the default schedule (align_small.hip): the whole current level staged in LDS, walked linearly
  int small;
  // 1: the contracted window sweep forms its Gram operands from the f16 HIGH parts of the twelve Jacobian components alone (the two
  // residual components keep high + low parts): levels of 150 000 pixels and more, where the rounding of a component (2^-12, at random)
  // averages out over the constraints -- align_fast.hip, fast_gram_row; option "gram_lo_parts" 1 keeps every low part
  int gram_hi_j;
  // the pyramid level this geometry belongs to: a launch on the launch-per-step path works on the pairs that are ACTIVE ON THIS LEVEL
  // (PairState::active && PairState::level == level) -- a pair that has left the level may already have begun the next one
  // (k_solver_step, NextLevel) and waits there for the rest of the batch
  int level;
  // non-null (round 6, the slow lane of a batch -- capi_schedule.inc::run_batch): the launch covers the n_pairs ENTRIES of
  // this list instead of pairs 0 .. n_pairs - 1 -- entry k names the pair its workgroups work on, -1 = none (they leave at once).  Partial
  // rows, residual pairs, states and records stay where the pair's index puts them: a pair's arithmetic does not know about the list.
  const int* pair_list;
  // non-null: a flag byte per pair -- the launch leaves the flagged pairs alone (they are the slow lane's: its launches name them in
  // pair_list, the batch's own launches skip them)
  const unsigned char* skip_flags;
};

// the pair a launch's index `k` stands for (LevelGeom::pair_list, ::skip_flags); negative: none
DVO_HD int pair_of_launch_index(const LevelGeom& g, int k) {
  const int pair = g.pair_list ? g.pair_list[k] : k;
  return pair >= 0 && g.skip_flags && g.skip_flags[pair] ? -1 : pair;
}

// What a solver step needs to hand the pairs that have LEFT its level over to the next one without a launch between the levels
// (round 5): a pair that is no longer active when the step begins -- it ended the level in an earlier step -- begins the next level
// there, in the shadow of the workgroups that still iterate (dense_tracking.cpp:200-238); a pair that has left the LAST level gets its
// result written (dense_tracking.cpp:368-373).  The step the host enqueues ahead of its poll does it for the pairs that ended last.
// valid == 0 and results == null: nothing of the kind (the parity and measurement entry points).
struct PairPtrs;
struct NextLevel {
  int valid;
  int level;
  float fx, fy, ox, oy;                // intrinsics of that level (gn_level_begin forms K T with them)
  const PairPtrs* pairs;               // that level's plane table (the selection count of the pair's reference frame)
  dvo_hip_result* results;             // non-null (with valid == 0): the match ends on this level
};

// slot of the tile's partial row that holds count_0 + 512 count_1 (the next one: count_2 + 512 count_3), as exact floats
constexpr int kAccCounts = kNumAcc;
constexpr int kCompactTileEntries = 1024, kCompactWaveEntries = 256;

// float2 entries of one pair's residual buffer at this level
__host__ __device__ inline size_t residual_entries(const LevelGeom& g) {
  return g.compact ? size_t(g.tiles_x) * size_t(g.tiles_y) * kCompactTileEntries : size_t(g.w) * size_t(g.h);
}

struct PairPtrs {                     // device planes of one pair at one level
  const float2* refR;                 // {Z (NaN = not selected), I}              8 B / pixel, streamed; the intensity gradient
                                      // the Jacobian needs is the central difference of I, recomputed from the neighbours
  const float4* curA;                 // {I, Z, Idx, Idy}                        16 B / pixel, gathered
  const float2* curB;                 // {Zdx, Zdy}                               8 B / pixel, gathered
  const int* n_selected;              // device counter: selected reference pixels at this level
  const float2* curC;                 // {I, Z}                                   8 B / pixel: the plane the window sweep stages in LDS
                                      // (align_window.hip); null when the frame has none at this level (the sweep then reads A.xy)
};

// The current-frame role of a frame level comes in two flavours: the gathered taps A + B (gathering sweep, resident kernel, plane
// downloads) and the 8-byte {I, Z} plane C the window sweep stages in LDS.  A level is built in the flavour(s) its consumer asks for;
// either is derived from the other on demand, bit-identically.
constexpr int kCurAB = 1, kCurC = 2;

struct FrameBuildPtrs {                // one frame of a batched pyramid build
  const uint8_t* grey;                 // raw planes (device), may be null for the float ingest path
  const uint16_t* raw;
  uint8_t* keep_grey;                  // where k_build_from_raw leaves a copy of the raw planes (the frame's own staging area), or null
  uint16_t* keep_raw;
  float* I[kMaxLevels];
  float* Z[kMaxLevels];
  float4* A[kMaxLevels];
  float2* B[kMaxLevels];
  float2* R[kMaxLevels];
  float2* C[kMaxLevels];               // {I, Z} of a current frame (null: not kept at this level)
  int* sel_count;                      // one counter per level
};

// several pyramid levels of one camera for a launch that covers them all (k_derive_levels): 64 x 16 tiles, level l's tiles of a frame
// are tile0[l] .. tile0[l + 1] - 1
struct LevelSpan {
  int l0, l1;
  int w[kMaxLevels], h[kMaxLevels], flavor[kMaxLevels];
  int tile0[kMaxLevels + 1];
};

struct SE3d {
  double R[9];
  double t[3];
};

struct PairState {
  SE3d inc, initial, initial_old, estimate, estimate_old;
  double x[6];
  double error, last_error;
  double A_last[36];
  float KT[12];                       // float(K * estimate) for the residual kernel
  float P_prev[4];                    // precision used for the weights of this pass
  int level;                          // level this pair is working on
  int iteration;                      // iterations completed on this level
  int active;                         // 1 while the pair still iterates on `level`
  int first;                          // first pass on the level: weights = 1
  int n_iters_total;                  // iteration records written so far
  int n_levels;                       // level records written so far
  int level_first_iter;               // index of the first iteration record of the current level
  int finished;                       // the pair's result has been written (k_solver_step, NextLevel::results)
};

// a step's word in the pinned status array: kStepDoneFlag | number of pairs still active on the level
constexpr int kStepDoneFlag = 0x40000000;

struct SolverParams {
  int max_iterations;
  int first_level, last_level;
  int use_initial_estimate;
  double precision;
  double mu;
  int cap_iters;                      // per-pair capacity of the iteration record array
  int cap_levels;
  int max_points_level0;
  int want_condition_number;          // gn_finish also computes the eigenvalue range of the information matrix
  // set by a caller of gn_step that has idle lanes next to the solver lane (k_solver_step; 0 from the host): every 8-byte word of the
  // new iteration record already holds NaN -- 51 stores of one lane off the serial path
  int record_prefilled;
};

// what a step needs besides the level's geometry (kernel arguments of both callers)
struct SolverStepArgs {
  PairState* states;
  int n_pairs;
  SolverParams prm;
  const float* partials;
  const double* ll_partials;             // sums of k_loglik's workgroups (when the log-likelihood ran in a launch of its own)
  int ll_blocks_per_pair;
  const float2* scratch_for_fused_ll;    // non-null: the log-likelihood sweep runs inside the step (small levels)
  dvo_hip_level_stats* levels;
  dvo_hip_iteration_stats* iters;
  unsigned long long* step_tally;        // publish_step
  int* host_status;
  int level_slot_hint;
  NextLevel next;
  int* arrivals;                         // the sweeps' tail: one word per pair, the tiles of the pair that are through (self-resetting)
  // the WIDE half of the step in the sweep's tail (reduction + log-likelihood), the serial half in a one-wavefront launch behind it:
  // kPairSumsStride doubles per pair -- the kAccStride reduced sums, then the four wavefronts' log-likelihood sums
  double* pair_sums;
};
constexpr int kPairSumsStride = 96;

__host__ inline SolverStepArgs make_solver_step_args(PairState* states, int n_pairs, const SolverParams& prm, const float* partials, const double* ll_partials,
                                            int ll_blocks_per_pair, const float2* scratch_for_fused_ll, dvo_hip_level_stats* levels,
                                            dvo_hip_iteration_stats* iters, unsigned long long* step_tally, int* host_status, int level_slot_hint,
                                            const NextLevel* next_or_null, int* arrivals, double* pair_sums = nullptr) {
  SolverStepArgs a;
  a.states = states; a.n_pairs = n_pairs; a.prm = prm; a.partials = partials; a.ll_partials = ll_partials; a.ll_blocks_per_pair = ll_blocks_per_pair;
  a.scratch_for_fused_ll = scratch_for_fused_ll; a.levels = levels; a.iters = iters; a.step_tally = step_tally; a.host_status = host_status;
  a.level_slot_hint = level_slot_hint;
  a.next.valid = 0; a.next.level = 0; a.next.fx = a.next.fy = a.next.ox = a.next.oy = 0.0f; a.next.pairs = nullptr; a.next.results = nullptr;
  if (next_or_null) a.next = *next_or_null;
  a.arrivals = arrivals;
  a.pair_sums = pair_sums;
  return a;
}

// ---- the resident match kernel (align_resident.hip): one group of workgroups owns a pair for a whole run of levels ----
constexpr int kResidentWaves = 8;                    // wavefronts per workgroup (512 threads: the solver lane needs > 128 registers)
constexpr int kResidentSweepers = kResidentWaves - 1;  // wavefront 0 is the workgroup's solver
constexpr int kResidentRowsLds = 3;                  // residual pairs of a wavefront's first segments stay in LDS (the rest: scratch).
                                                     // Not more: the kernel is faster with less than 64 KB of LDS per workgroup (12 rows =
                                                     // 91 KB: the front end's 0.17-ms match took 0.178 ms), and the coarse levels, where the
                                                     // passes are, give a wavefront one to three segments
constexpr int kResidentBlock = kResidentWaves * 64;
constexpr int kResidentMaxGroup = 64;                // workgroups per pair at most (power of two)
constexpr int kResidentRing = 4;                     // exchange rows in flight per workgroup (a power of two; align_resident.hip, "Flow control")
constexpr int kResidentSlots = 96;                   // 8-byte {value, sequence} slots of an exchange row: the 85 accumulators, ...
constexpr int kResidentSlotLl = 88;                  // ... and the two halves of the workgroup's float64 log-likelihood sum

constexpr int kResidentInline = 4;                   // pairs whose plane pointers and initial guesses travel in the kernel arguments
constexpr int kResidentFlagNoQuietPoll = 2;           // measurement: everybody polls every slot from the start
constexpr int kResidentFlagWithhold = 4;              // test hook: workgroup 1 of every group withholds its rows (its peers time out)

struct ResidentArgs {
  LevelGeom geom[kMaxLevels];                         // per absolute level
  const PairPtrs* pair_ptrs;                          // device [levels][n_pairs]
  PairState* states;
  dvo_hip_level_stats* levels;
  dvo_hip_iteration_stats* iters;
  const double* T_init;                               // non-null: the pairs are initialised here (the launch starts the match)
  float2* scratch;                                    // residual pairs, n_pairs x pixels of the level
  unsigned long long* exchange;                       // [n_pairs][group][kResidentRing][kResidentSlots] + one heartbeat word per workgroup, used when group > 1
  int* error_word;                                    // pinned host word: set when a group timed out
  SolverParams prm;
  int n_pairs, group;                                 // group: workgroups per pair (power of two)
  int first_level, last_level;                        // the levels this launch runs, coarse to fine
  unsigned sequence_base;                             // exchange sequence numbers of this launch start above it
  dvo_hip_result* results;                            // non-null: the launch ends the match and writes the results (gn_finish);
                                                      // may be pinned host memory
  // n_pairs <= kResidentInline: no table uploads in front of the launch
  PairPtrs inline_ptrs[kMaxLevels * kResidentInline]; // [level][pair]
  double inline_T[kResidentInline * 16];
  int use_inline;
  // non-null (with results): the pair's level and iteration records are copied here (pinned host memory, same layout as `levels` /
  // `iters`) and *done_word is incremented (system scope) when everything of the pair has been written
  dvo_hip_level_stats* host_levels;
  dvo_hip_iteration_stats* host_iters;
  int* done_word;
  int flags;                                          // kResidentFlag*
};

// ---- the fused coarse-level kernel (align_coarse.hip): ONE workgroup runs the coarse levels of a pair from gn_level_begin to the level's
// termination -- sweep over all tiles, stage 3 of the reduction, log-likelihood, loop body -- with the device functions the launch path's
// kernels are made of (fast_sweep_tile, mfma_sweep_tile, reduce_partials, loglik_partial*, gn_step): the records are the launch path's
// bit for bit.  No workgroup waits for another: no residency requirement, any batch size.
constexpr int kCoarseRowsPerWave = 2;                // tile height of the gathering sweep inside the fused kernel (levels the window sweep does not take)
struct CoarseArgs {
  LevelGeom geom[kMaxLevels];                         // per absolute level (rows_per_wave 4 on window levels, kCoarseRowsPerWave elsewhere)
  const PairPtrs* pair_ptrs;                          // device [levels][n_pairs]
  PairState* states;
  dvo_hip_level_stats* levels;
  dvo_hip_iteration_stats* iters;
  const double* T_init;                               // non-null: the pairs are initialised here (the launch starts the match)
  float* partials;                                    // per-tile partial rows, [pair][max_tiles][kAccStride]: a pair's own region on every level
  float2* scratch;                                    // residual pairs, [pair][max_entries]
  int max_tiles;                                      // largest tile count among the launch's levels
  size_t max_entries;                                 // largest residual_entries among them
  unsigned long long* fallback_count;                 // window sweep: lanes whose taps were fetched from memory (may be null)
  int* f16_range_flag;                                // one word per pair (pinned host memory; may be null)
  SolverParams prm;
  int n_pairs;
  int first_level, last_level;                        // the levels this launch runs, coarse to fine
  dvo_hip_result* results;                            // non-null: the launch ends the match and writes the results (gn_finish)
};

}  // namespace dvo_hip
