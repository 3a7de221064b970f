// launch.h -- host-callable launchers of the gfx950 kernels (one per .hip translation unit).
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"

namespace dvo_hip {

// pyramid_kernels.hip
// raw ingest + pyramid levels 1..3 in one pass (levels beyond the fourth: launch_pyr_down)
void launch_pyr_down(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h);
// max_workgroups > 0 caps the grid (the kernels walk the tiles with a grid stride): background build next to an alignment
// level 0 in role `role` (-1 none, 0 current, 1 reference) + pyramid levels 1..3 straight from raw planes; `wide`: 4-pixel aligned rows
// cur_flavor (role 0): which planes of the current role are written, kCurAB | kCurC (device_types.h)
// c_levels: bit l set = pyramid level l (1..3) also gets the current role's {I, Z} plane C in the same pass (ingest_strips.hip only:
// ingest_strips_supports tells the caller whether the pass will honour it)
void launch_build_from_raw(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, float scale, int w0, int h0, int levels, int role, bool wide,
                           float ithr, float dthr, int max_workgroups, int cur_flavor = kCurAB, int c_levels = 0);
// ingest_strips.hip: the role planes of one level from the float planes I / Z in strips (even widths); role 1: counters zeroed before
bool derive_strips_supports(int w);
void launch_derive_strips(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int role, float ithr, float dthr,
                          int max_workgroups, int cur_flavor);
// ingest_strips.hip: the same pass with one 128 x 8 strip per wavefront, registers only (even widths, aligned planes)
bool ingest_strips_supports(int w0, bool wide);
void launch_ingest_strips(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, float scale, int w0, int h0, int levels, int role,
                          float ithr, float dthr, int max_workgroups, int cur_flavor, int c_levels);
void launch_derive_current(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int max_workgroups, int cur_flavor = kCurAB);
// mode 0: A + B from C; 1: C from A; 2: R + selection count from C (the level's counters are zeroed first)
void launch_from_current_plane(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, int mode, float ithr, float dthr,
                               int max_workgroups);
void launch_derive_reference(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, int level, int w, int h, float ithr, float dthr,
                             int max_workgroups);
// the role planes (role 0: current, flavours per level in span.flavor; role 1: reference) of the levels span.l0 .. span.l1 in one launch
void launch_derive_levels(hipStream_t s, const FrameBuildPtrs* tbl, int n_frames, const LevelSpan& span, int role, float ithr, float dthr,
                          int max_workgroups);
void launch_select_pack(hipStream_t s, const float4* A, const float2* B, int n, float ithr, float dthr, float2* R, int* count, uint8_t* mask);
void launch_unpack_plane(hipStream_t s, const float4* A, const float2* B, int n, int plane, float* out);

// align_kernels.hip / align_mfma.hip
// variant 5 (default): Gram accumulation on the matrix cores (align_mfma.hip); variant 0: the all-VALU schedule with the DPP + LDS
// two-stage reduction (align_kernels.hip).  Same outputs.
void launch_residual_reduce(hipStream_t s, int variant, int rows_per_wave, bool finest_level, const LevelGeom& g, const PairPtrs* pairs,
                            const PairState* states, int n_pairs, float* partials, float2* scratch, unsigned long long* window_fallbacks = nullptr,
                            int* f16_range_flag = nullptr, const SolverStepArgs* tail = nullptr);
// `tail` (non-null; only where sweep_has_tail says so): the workgroup that completes the last tile of a pair runs the pair's solver step
// in the sweep's launch (solver_step.h) -- no launch_solver_step behind it
bool sweep_has_tail(int variant, int rows_per_wave, const LevelGeom& g);
bool sweep_fast_has_tail(int variant, const LevelGeom& g);
bool mfma_sweep_has_tail(int variant, int rows_per_wave, const LevelGeom& g);
// mode 0: f32 Gram (the f32 matrix instruction); 1: Gram accumulation on the f16 matrix pipe (gram_f16.h; variant 7); 2: 1 with the contracted
// per-pixel arithmetic of align_fast.hip (variants 8 / 9 on the levels the window sweep does not take)
void launch_residual_reduce_mfma(hipStream_t s, int rows_per_wave, bool finest_level, const LevelGeom& g, const PairPtrs* pairs,
                                 const PairState* states, int n_pairs, float* partials, float2* scratch, int mode = 0,
                                 int* f16_range_flag = nullptr, const SolverStepArgs* tail = nullptr);
// align_window.hip: variants 6 (f32 Gram) and 7 (f16 hi/lo Gram) -- the current frame's {I, Z} window staged in LDS; tiled levels whose
// width is a multiple of 64 only (window_sweep_supports), tile height 16 (rows_per_wave 4).  fallback_count (may be null): lanes
// whose taps fell outside the staged window and were fetched from memory.  f16_range_flag (may be null; pinned host memory): set
// to 1 by a workgroup of variant 7 whose Jacobian components left the f16 range -- the caller repeats the work with variant 6.
bool window_sweep_supports(const LevelGeom& g);
void launch_sweep_window(hipStream_t s, bool f16, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                         float* partials, float2* scratch, unsigned long long* fallback_count, int* f16_range_flag = nullptr);
// align_fast.hip: variants 8 / 9 -- the window sweep with contracted per-pixel arithmetic (same function, rounding differences of a few
// ulp against variants 6 / 7; window 84 x 28 cells at a pitch of 96; 8: high and low operand parts take turns in the slab, 9: side by side, moved with v_permlane32_swap).
bool fast_sweep_takes_width(int w);   // (64-column tiles; a width that is no multiple of 64 leaves the last tile column partly empty)
bool fast_sweep_supports(const LevelGeom& g);
void launch_sweep_fast(hipStream_t s, int variant, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                       float* partials, float2* scratch, unsigned long long* fallback_count, int* f16_range_flag = nullptr,
                       const SolverStepArgs* tail = nullptr);
// align_small.hip: a level small enough for the whole current plane C to live in LDS ((w + 2) x (h + 2) cells of 8 B <= 43 KB, even width:
// 80 x 60, 40 x 30 ...), walked linearly, rows_per_wave segments per wavefront and workgroup; contracted arithmetic, f16 Gram, residual pairs by pixel
bool small_sweep_takes(int w, int h);
void launch_sweep_small(hipStream_t s, int rows_per_wave, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                        float* partials, float2* scratch, int* f16_range_flag = nullptr);
// scratch == null: read-only (one float per workgroup goes to `sink`, which must hold a float per (8 * 256)-pixel block)
// window_planes: the planes the window sweep reads (reference 8 B + current {I, Z} 8 B) instead of the gathering sweep's 8 + 16 + 8 B
void launch_stream_mix(hipStream_t s, const PairPtrs* pairs, int n_pairs, int n_px, float2* scratch, float* sink, bool window_planes = false);
void launch_loglik(hipStream_t s, const LevelGeom& g, const PairState* states, int n_pairs, const float* partials,
                   const float2* scratch, double* ll_partials, int blocks_per_pair, bool one_schedule = false);

// align_resident.hip: levels first_level..last_level of every pair in one launch.  The n_pairs * group workgroups must fit the
// device at once when group > 1 (one per compute unit); `cooperative` launches them through hipLaunchCooperativeKernel.
hipError_t launch_match_resident(hipStream_t s, const ResidentArgs& args, bool cooperative);

// align_coarse.hip: levels first_level..last_level of every pair in one launch, one workgroup per pair (no workgroup waits for another).
// workgroups_per_cu: 4 (128 registers; default) or 3 (168)
bool coarse_kernel_takes(const LevelGeom& g, bool window_level);
hipError_t launch_match_coarse(hipStream_t s, const CoarseArgs& args, int workgroups_per_cu);

// solver_kernels.hip
extern int g_solver_occupancy;   // experiment (option "solver_occupancy"): 3 / 4 = the four-wavefront solver step built for that many workgroups per compute unit
void launch_init_pairs(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, const double* T_init);
// flags (may be null: every pair): which == 0 -- every pair whose flag byte is zero, which == 1 -- the flagged ones; from_level >= 0:
// of those, the pairs that have left that level
void launch_level_begin(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, LevelGeom g, int level,
                        const PairPtrs* pairs, dvo_hip_level_stats* levels, const double* T_init_or_null = nullptr,
                        const unsigned char* flags = nullptr, int which = 0, int from_level = -1);
// the slow lane of a batch (capi_schedule.inc::run_batch): the pairs still active on `level` get their flag byte set, and the indices of all flagged
// pairs go into list[0 .. cap) in ascending order, -1 behind the last; LevelGeom::pair_list then makes a launch cover the list, ::skip_flags
// makes one leave the flagged pairs alone.  The flags of a batch start at zero (launch_clear_flags).
// list_only: nothing is flagged; the list gets the unflagged pairs active on the level (flags may be null) -- the active-pair list of a level's last steps
void launch_mark_stragglers(hipStream_t s, const PairState* states, int n_pairs, int level, unsigned char* flags, int* list, int cap, bool list_only = false);
void launch_clear_flags(hipStream_t s, unsigned char* flags, int n_pairs);
// ... which the sweep of the level must understand (the contracted window sweep and the gathering sweep of the default schedule do)
bool sweep_takes_pair_list(int variant, int rows_per_wave, const LevelGeom& g);
void launch_solver_step(hipStream_t s, PairState* states, int n_pairs, SolverParams prm, LevelGeom g,
                        const float* partials, const double* ll_partials, int ll_blocks_per_pair, const float2* scratch_for_fused_ll,
                        dvo_hip_level_stats* levels, dvo_hip_iteration_stats* iters, unsigned long long* step_tally,
                        int* host_status, bool two_waves = false, int level_slot_hint = -1, const NextLevel* next_or_null = nullptr);
// the serial half of a step whose wide half ran in the sweep's tail (a.pair_sums)
void launch_solver_serial(hipStream_t s, int n_pairs, LevelGeom g, const SolverStepArgs& a);
void launch_finish(hipStream_t s, const PairState* states, int n_pairs, SolverParams prm,
                   const dvo_hip_level_stats* levels, const dvo_hip_iteration_stats* iters, dvo_hip_result* results);
// single-shot linearisation for parity tests: fixed T34 / P_prev, no state machine
// measurement: every pair back on its level with the weights on (after warm-up iterations)
void launch_force_active(hipStream_t s, PairState* states, int n_pairs);
void launch_set_fixed_state(hipStream_t s, PairState* states, LevelGeom g, const float* T34_dev, const float* Pprev_dev, int first);
void launch_single_shot_out(hipStream_t s, LevelGeom g, const float* partials, const double* ll_partials, int ll_blocks_per_pair,
                            int n_selected, dvo_hip_iteration_out* out_dev);

}  // namespace dvo_hip
