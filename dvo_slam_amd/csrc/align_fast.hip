// align_fast.hip -- schedule variants 8 / 9 of the sweep: the window sweep of align_window.hip (current frame's {I, Z} window
// staged in LDS, Gram accumulation on the f16 matrix pipe) with CONTRACTED per-pixel arithmetic.
//
// Variants 6 / 7 spend about a third of their vector instructions on reproducing the residuals of the oracle's MATH mode bit for bit
// (dense_tracking_impl.cpp:148-281 without fused multiply-adds, the reference's operation order, correctly rounded divisions) -- a
// stricter bar than the float tolerance the alignment is specified to, and stricter than what MATH itself keeps against the
// reference (3e-5 in the final twist).  The kernel is bound by vector-instruction issue (profiles/r03_pmc_utilisation.md), so these
// variants compute THE SAME FUNCTION with the fewest instructions instead:
//
//   * projection  q = K T (tx z, ty z, z, 1) factored as  z (KT.col0 tx + KT.col1 ty + KT.col2) + KT.col3  -- the bracket is a
//     constant of the column plus a constant of the row: 6 fused multiply-adds per pixel instead of 23 operations
//   * u = qx rcp(qz) (v_rcp_f32, 1 ulp) instead of the correctly rounded quotient (13 instructions); bounds test on the integer
//     image of the float (0 <= u <= w - 2 is one unsigned comparison); floor / fraction through v_cvt / v_fract
//   * bilinear blend and the four gradient channels in separable lerp form (48 instead of 56 operations, no tap weights)
//   * the validity of a pixel travels as a lane mask in scalar registers, not as a sentinel in a vector register
//   * window pitch 96 cells = 768 B = 0 banks mod 64: the tap reads of a wavefront whose warped row steps to the next window row no
//     longer collide (23 % of the LDS cycles of variant 7); 84 x 28 cells in use
//   * the operand rows are written by all 64 lanes -- a 16-byte LDS store costs 8 array cycles whatever its active lanes: high and low
//     parts take turns in the slab (fast_gram_row), 4 instead of 8 stores per pixel row
//   * the window fill of a tile whose window lies inside the image needs no clamping: one address per thread, scalar row offsets
//   * the epilogue leaves H H^T and S to the 85 reducing threads instead of symmetrising per wavefront
//
// Measured and dropped in round 4 (scripts/ab_sweep.py, builds alternated on one box; base 2.18-2.24 ms per 1024-pair launch):
//   * the scheduler that maximises instruction-level parallelism (align_window.hip's flag): 4 registers spilled, 2.48 ms;
//   * a uniform branch around rows without a projection or without a constraint (4-16 % of the bench's rows): 2.45 ms -- the four rows
//     of a wavefront are one basic block and every branch is a scheduling barrier;
//   * level constants and per-row scalars copied into vector registers (an instruction with a scalar operand issues in ~4.5 cycles,
//     with vector operands in ~3: scripts/ubench/issue_rate.hip): +2 %, the copies pin the schedule;
//   * the window requested from a GUESSED origin before phase A (what a per-tile depth range from the frame build would allow: one memory
//     round trip per tile instead of two): -1.3 % only, not worth seven writers of the reference plane;
//   * four workgroups per compute unit with 128 registers each: level with five at the default scheduler, 2.35 ms with max-ilp.
// (Also measured and dropped: packing the tile's SELECTED reference pixels -- Zsel not NaN -- into a list in LDS first and sweeping
// ceil(n / 64) instead of 16 wavefront rows, what the reference's PointSelection does once per key frame.  Correct, same residuals;
// 2.45 ms instead of 2.18 per 1024-pair launch: 81 % of the bench's reference pixels are selected -- the constraints that are lost
// are lost to holes of the CURRENT frame, known only after the taps -- so a tile keeps 13-14 of its 16 rows and the list costs more
// than they save.)
//
// What is NOT changed: which pixels are tested, the NaN / occlusion semantics (Q4, Q5, Q9, Q19), the weights, the Jacobian at the
// untransformed point (Q10), the accumulator layout, the deterministic reduction order.  Differences to variant 7 are rounding
// differences of a few ulp in u, v and the blends (tests/test_gpu_parity.py::test_contracted_sweep_against_the_exact_one states and
// checks the bounds); variants 6 / 7 stay in the tree as the bit-exact anchors.
#include "fast_sweep.h"
#include "solver_step.h"

namespace dvo_hip {

// ===================================================================================================================================
// variants 8 / 9: every pixel of the tile, one wavefront row = one image row of 64 pixels
// ===================================================================================================================================
// COMPAT (option "ref_compat"): 0 = off; 1 = the host's reciprocal table through memory; 2 = its 16-bit copy in LDS (four workgroups
// per compute unit instead of five)
// TAIL (round 6; levels whose log-likelihood pass fits the solver step): the workgroup that completes the last tile of a pair runs the
// pair's Gauss-Newton step right here (solver_step.h) -- one launch per iteration instead of two.  Every workgroup of such a launch
// ends in sweep_tail, also those of a pair that is not on the level; partial rows and residual pairs are stored write-through.
// TAIL 2: the WIDE half of the step only (reduction + log-likelihood, sweep_tail_wide); the serial half follows in k_solver_serial.
template <int STORE, bool PARTIAL, bool COMPACT, int COMPAT, bool HI_J, int TAIL>
__global__ __launch_bounds__(256, COMPAT == 2 ? 4 : 5) void k_sweep_fast(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* __restrict__ partials, float2* __restrict__ scratch, int blocks_per_xcd, unsigned long long* __restrict__ fallback_count, int* __restrict__ f16_range_flag,
    const SolverStepArgs tail) {
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int entry = item / tiles, tile = item - entry * tiles;
  const int pair = pair_of_launch_index(g, entry);           // (a level's overlapped tail: the launch covers a list of pairs)
  if (pair < 0) return;
  const PairState& st = states[pair];
  __shared__ __attribute__((aligned(16))) float slab[4][kSlabFloatsF16];
  __shared__ __attribute__((aligned(16))) float2 win[kFastCells];
  __shared__ __attribute__((aligned(16))) int bbox[4][2];
  __shared__ int counts[4];
  static_assert(TAIL == 0 || sizeof(SweepTailLds) + 16 <= sizeof(win), "the step's LDS lies over the window");
  if (!st.active || st.level != g.level) {                  // (not on this level: finished it, and maybe begun the next)
    if constexpr (TAIL == 1) sweep_tail(g, tail, pair, reinterpret_cast<char*>(win));
    return;
  }
  const PairPtrs pp = pairs[pair];
  // (option "ref_compat": the host CPU's reciprocal table, 2^(23 - shift) floats; a resource of no bytes otherwise, never read)
  __shared__ __attribute__((aligned(16))) unsigned short rcp_lds[COMPAT == 2 ? 4096 : 8];
  const FastRcpSource rcp_table = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(COMPAT ? g.rcp_table : nullptr), 0, COMPAT ? (4 << (23 - g.rcp_shift)) : 0, 0x00020000),
                                   rcp_lds, g.rcp_shift};
  if constexpr (COMPAT == 2) {
    // the table's 16-bit copy (behind its floats) into LDS: 2^(23 - shift) entries of 2 B, 16 bytes per thread and round; the loads
    // travel with the reference rows' below, the barrier in front of the first lookup is this phase's only addition
    const int entries = 1 << (23 - g.rcp_shift);
    const __amdgpu_buffer_rsrc_t packed = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.rcp_table) + entries, 0, entries * 2, 0x00020000);
    f32x4 chunk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) chunk[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(packed, (int(threadIdx.x) + j * 256) * 16, 0, 0));
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if ((int(threadIdx.x) + j * 256) * 8 < entries) reinterpret_cast<f32x4*>(rcp_lds)[threadIdx.x + j * 256] = chunk[j];
    __syncthreads();
  }
  const FastLds lds = {slab, win, bbox, counts};
  const FastWeights wt(st);
  fast_sweep_tile<STORE, PARTIAL, COMPACT, COMPAT, HI_J, TAIL != 0>(g, st.KT, wt, pp, pair, tile, partials, scratch, lds, rcp_table, fallback_count, f16_range_flag);
  if constexpr (TAIL == 1) sweep_tail(g, tail, pair, reinterpret_cast<char*>(win));
  if constexpr (TAIL == 2) sweep_tail_wide(g, tail, pair, reinterpret_cast<char*>(win));
}

// the instantiations with a tail: the default schedule (variant 8, every low part of the Gram operands, no "ref_compat")
bool sweep_fast_has_tail(int variant, const LevelGeom& g) { return variant == 8 && !g.rcp_table && !g.gram_hi_j && fast_sweep_supports(g); }

bool fast_sweep_takes_width(int w) { return w >= kFastCols && w % 2 == 0; }

bool fast_sweep_supports(const LevelGeom& g) {
  // (a pair's planes and its residual buffer are addressed through 32-bit buffer offsets: 2 GB each at most -- a 16 384 x 16 384 level)
  const long long plane_bytes = 8ll * g.w * g.h, packed_bytes = 8ll * kCompactTileEntries * g.tiles_x * g.tiles_y;
  return !g.linear && fast_sweep_takes_width(g.w) && g.w < 32768 && g.h < 32768 && plane_bytes < (1ll << 31) && packed_bytes < (1ll << 31);
}

void launch_sweep_fast(hipStream_t s, int variant, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                       float* partials, float2* scratch, unsigned long long* fallback_count, int* f16_range_flag, const SolverStepArgs* tail) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(256);
  const bool partial = g.w % kTileW != 0, compact = g.compact != 0;
  const SolverStepArgs no_tail = {};
  if (tail) {                                                // (the default schedule only: sweep_fast_has_tail)
    auto with = [&](auto partial_tag, auto compact_tag) {
      constexpr bool kP = decltype(partial_tag)::value, kC = decltype(compact_tag)::value;
      if (tail->pair_sums) k_sweep_fast<2, kP, kC, 0, false, 2><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag, *tail);
      else k_sweep_fast<2, kP, kC, 0, false, 1><<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag, *tail);
    };
    if (compact) { if (partial) with(std::true_type{}, std::true_type{}); else with(std::false_type{}, std::true_type{}); }
    else { if (partial) with(std::true_type{}, std::false_type{}); else with(std::false_type{}, std::false_type{}); }
    return;
  }
  auto go = [&](auto store_tag, auto partial_tag, auto compact_tag, auto compat_tag) {
    constexpr int kStore = decltype(store_tag)::value;
    if (g.gram_hi_j && kStore == 2)
      k_sweep_fast<2, decltype(partial_tag)::value, decltype(compact_tag)::value, decltype(compat_tag)::value, true, 0>
          <<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag, no_tail);
    else
      k_sweep_fast<kStore, decltype(partial_tag)::value, decltype(compact_tag)::value, decltype(compat_tag)::value, false, 0>
          <<<grid, block, 0, s>>>(g, pairs, states, n_pairs, partials, scratch, per_xcd, fallback_count, f16_range_flag, no_tail);
  };
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using T = std::true_type;
  using F = std::false_type;
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  // Option "ref_compat" VALUE 1 -> template COMPAT 2 (the table's 16-bit copy in LDS) when the table packs, else COMPAT 1; option VALUE 2
  // -> COMPAT 1 (the table through memory) always: the numbers run the other way round on purpose of history, not of meaning.  Under
  // "ref_compat" the operand store is STORE 2 whatever the variant: "variant" 9 (STORE 1, a measurement schedule) is the same as 8 there.
  if (g.rcp_table && g.rcp_packed) {                           // option "ref_compat": the host's reciprocal table in projection and weights, from LDS
    if (compact) { if (partial) go(S2{}, T{}, T{}, C2{}); else go(S2{}, F{}, T{}, C2{}); }
    else { if (partial) go(S2{}, T{}, F{}, C2{}); else go(S2{}, F{}, F{}, C2{}); }
  } else if (g.rcp_table) {                                    // ... a table that does not fit a 16-bit copy: from memory
    if (compact) { if (partial) go(S2{}, T{}, T{}, C1{}); else go(S2{}, F{}, T{}, C1{}); }
    else { if (partial) go(S2{}, T{}, F{}, C1{}); else go(S2{}, F{}, F{}, C1{}); }
  } else if (variant == 8) {
    if (compact) { if (partial) go(S2{}, T{}, T{}, C0{}); else go(S2{}, F{}, T{}, C0{}); }
    else { if (partial) go(S2{}, T{}, F{}, C0{}); else go(S2{}, F{}, F{}, C0{}); }
  } else {
    if (compact) { if (partial) go(S1{}, T{}, T{}, C0{}); else go(S1{}, F{}, T{}, C0{}); }
    else { if (partial) go(S1{}, T{}, F{}, C0{}); else go(S1{}, F{}, F{}, C0{}); }
  }
}

}  // namespace dvo_hip
