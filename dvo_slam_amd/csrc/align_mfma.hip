// align_mfma.hip -- the fused residual / Jacobian / reduce / log-likelihood sweep (the default schedule, "variant 5").
//
// One launch does the reference's passes 1-5 (dense_tracking.cpp:271-343) for every pair of a batch:
//
//   per pixel   warp, bilinear taps, residual, validity tests, t-distribution weight, 2x6 Jacobian  (VALU code,
//               dense_tracking_impl.cpp:148-281, dense_tracking.cpp:448-476)
//   per tile    n, sum w r r^T and the P-independent Gram sums of J^T W J / J^T W r, accumulated on the matrix cores
//   per pair    the 2x2 precision P = (sum w r r^T / (n - 3))^-1 is only known when ALL tiles of the pair are through
//               (dense_tracking.cpp:295).  The last tile workgroup of a pair to arrive reduces the pair's scale sums in
//               the fixed tile order and publishes P; every tile workgroup of the pair then evaluates its share of the
//               log-likelihood sum  sum log(1 + 0.2 r^T P r)  (dense_tracking_impl.cpp:406-425) from the residual pairs
//               it STILL HOLDS IN REGISTERS.
//
// Round 1 wrote the residual pair of every pixel to HBM (8 B per pixel on top of the 40 B read: 1.24x the algorithmic
// traffic, and a read/write mix this part streams at 4.8 instead of 6.3 TB/s) and swept it again in a second kernel once P
// was known.  Both the write and the second launch are gone.
//
// Why a matrix instruction in a "memory-bound per-pixel" kernel: measured on MI355X (profiles/r01_*), the all-VALU schedule
// is NOT memory-bound.  Its 85 per-lane accumulators cost ~125 FMAs per pixel plus a 600-instruction DPP reduction per
// wavefront and 150 VGPRs (3 wavefronts per SIMD).  The accumulation itself is a rank-k update  G += V^T V  of the 16x16
// Gram matrix of the per-pixel vectors  v = sqrt(w) * [J0(6), J1(6), r0, r1, 0, 0]  (least_squares.cpp:58-64 contracts
// exactly this with the 2x2 precision) -- matrix math, four pixels (k = 4) per v_mfma_f32_16x16x4_f32.  That instruction is
// an exact f32 fmaf chain (no reduced precision), runs on the otherwise idle matrix pipe, keeps the whole Gram matrix in 4
// accumulator registers per lane instead of 85, and leaves NO cross-lane reduction to do.
//
// Data movement per pixel row of a wavefront: each lane (= pixel) writes its 16-vector to LDS as four conflict-free
// ds_write_b128; the MFMA operand for pixel group g (lane l <- component l&15 of pixel 4g + l>>4) is one conflict-free
// ds_read_b32 at a constant offset.  A == B (the sqrt(w)-scaled vector on both sides).  The LDS slab is private to the
// wavefront, so the row loop contains no barrier.
//
// The per-pair hand-off (MI355X: 8 XCDs with private L2s, per-CU L1s that other CUs' stores never refresh) follows the
// write-through recipe of the CDNA4 guide: every shared word is stored and loaded with agent-scope (sc1) accesses, the
// storing wavefront drains its stores (s_waitcnt vmcnt(0)) before the arrival ticket / the flag is touched, ONE lane polls
// ONE word with s_sleep in between, every spin is bounded and reports through a sticky error word.  A pair's tiles are
// dealt to consecutive workgroups of one XCD (150 at the finest level of a 640x480 pair, 192+ resident per XCD), so the
// waiting workgroups of a pair never keep its remaining tiles from being dispatched; should a dispatcher ever violate that,
// the bounded spin turns the hang into an error code, never into a wrong number.
#include <cstdlib>

#include "align_common.h"

namespace dvo_hip {

typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef __attribute__((address_space(1))) unsigned* GlobalU32;

constexpr int kQuadStride = 264;                 // floats per component quad: 64 pixels x 4 + 8 skew (bank-conflict-free reads)
constexpr int kSlabFloats = 4 * kQuadStride;     // per-wavefront LDS slab (4224 B)
static_assert(kWavesPerBlock * kSlabFloats >= kScaleStageFloats, "the last arriver stages the pair's scale sums in the slabs");

constexpr unsigned kSpinLimit = 1u << 22;        // polls (~1 us each) before a waiting workgroup gives up

__device__ __forceinline__ unsigned load_agent(const unsigned* p) {
  return __hip_atomic_load((GlobalU32)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(unsigned* p, unsigned v) {
  __hip_atomic_store((GlobalU32)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// reduce_partials_scale (reduce_scale.h) over rows that other workgroups of THIS launch have just written: the same
// additions in the same order (hence the same n, S and P the solver kernel derives a launch later from the same rows), the
// loads agent-scope so that they are served from behind the per-XCD L2s.
__device__ inline void reduce_partials_scale_coherent(const float* partials, int pair, int tiles, float* stage, double* sh, double* sums) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned* base = reinterpret_cast<const unsigned*>(partials + size_t(pair) * tiles * kAccStride);
  const int my_rows = tiles > wave ? (tiles - wave + kWavesPerBlock - 1) / kWavesPerBlock : 0;
  float* mine = stage + wave * kScaleRowsPerRound * 4;
  double a0 = 0.0;
  for (int r0 = 0; r0 < my_rows; r0 += kScaleRowsPerRound) {
    unsigned v[kScaleLoadsInFlight];
#pragma unroll
    for (int j = 0; j < kScaleLoadsInFlight; ++j) {
      const int r = r0 + j * 16 + (lane >> 2);
      const int t = wave + (r < my_rows ? r : my_rows - 1) * kWavesPerBlock;
      v[j] = load_agent(base + size_t(t) * kAccStride + (lane & 3));
    }
#pragma unroll
    for (int j = 0; j < kScaleLoadsInFlight; ++j) mine[(j * 16 + (lane >> 2)) * 4 + (lane & 3)] = __uint_as_float(v[j]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 4) {
      const int count = min(kScaleRowsPerRound, my_rows - r0);
      for (int r = 0; r < count; ++r) a0 += double(mine[r * 4 + lane]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < 4) sh[wave * 4 + lane] = a0;
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    sums[k] = (sh[k] + sh[4 + k]) + (sh[8 + k] + sh[12 + k]);
  }
  __syncthreads();
}

// LINEAR: the level is walked as one row of w*h pixels in 64-pixel segments (LevelGeom::linear) -- same per-pixel arithmetic,
// the pixel coordinates come from a division instead of the tile position.
// MODE 0: the log-likelihood sums are evaluated in this launch (hand-off of P, residual pairs in registers); nothing but the
//         partial rows and one float64 per tile is written.
// MODE 1: as 0, and the residual pairs are ALSO left in `scratch` (parity entry point / error image; never on the match path).
// MODE 2: no hand-off: the residual pairs go to `scratch` and k_loglik sweeps them a second time.  For levels with more tiles
//         per pair than are resident at a time (the hand-off needs every tile of a pair in flight together).
template <int RPW, bool FINEST, bool LINEAR, int MODE>
__global__ __launch_bounds__(kBlock) void k_residual_reduce_mfma(
    const LevelGeom g, const PairPtrs* __restrict__ pairs, const PairState* __restrict__ states, int n_pairs,
    float* partials, PairSync* sync, double* __restrict__ ll_partials, int ll_stride, float2* __restrict__ scratch,
    unsigned* error_word, int blocks_per_xcd, int exp_skip) {
  // XCD-aware (pair, tile) -> workgroup mapping: workgroup b runs on XCD b % 8 (observed dispatch order; speed only), every XCD
  // gets one contiguous run of (pair, tile) items so that a pair's planes flow through a single L2
  const int tiles = g.tiles_x * g.tiles_y;
  const int total = tiles * n_pairs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int item = xcd * blocks_per_xcd + slot;
  if (item >= total) return;
  const int pair = item / tiles, tile = item - pair * tiles;
  const PairState& st = states[pair];
  if constexpr (RPW <= 2) {
    // Short tiles (coarse levels, small batches): the prologue is most of a workgroup's life.  The pair's plane pointers
    // are requested together with the activity flag and pinned above the branch, so the scalar loads travel together
    // instead of as a chain of dependent round trips (the compiler otherwise sinks each to its first use).  Tall tiles run
    // measurably better with the lazy order.
    const PairPtrs early = pairs[pair];
    const int active = st.active;
    asm volatile("" ::"s"(early.refR), "s"(early.curA), "s"(early.curB), "s"(active));
    if (!active) return;
  } else {
    if (!st.active) return;
  }
  const PairPtrs pp = pairs[pair];
  float KT[12], Pp[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) Pp[i] = st.P_prev[i];
  const bool first = st.first != 0;
  const GlobalLoad4 refR{(GlobalVec4)pp.refR}, curA{(GlobalVec4)pp.curA};
  const GlobalLoad2 curB{(GlobalVec2)pp.curB};

  // the wavefront index is uniform: keeping it (and every row index derived from it) in scalar registers moves the row
  // bounds test, the row offsets and the ty table load from the vector ALU to the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // tiled: lane = column u_r of the tile, rows row0, row0 + 4, ...;  linear: 64-pixel segments row0, row0 + 4, ... of the
  // flattened level, lane = offset in the segment
  const int u_r = LINEAR ? lane : (tile % g.tiles_x) * kTileW + lane;
  // wavefront w sweeps rows w, w+4, w+8, ... of the tile: the four waves work on ADJACENT rows at the same time, so the
  // lower tap row of one wave is the upper tap row of the next and is served by the CU's L1 instead of a second L2 request
  const int row0 = (tile / g.tiles_x) * (kWavesPerBlock * RPW) + wave;
  const float nanv = __builtin_nanf("");
  const bool col_ok = LINEAR || u_r < g.w;
  const int n_px = g.w * g.h;
  const float inv_w = 1.0f / float(g.w);
  const float tx_u = LINEAR ? 0.0f : g.tx[col_ok ? u_r : 0];  // column term of the back-projection: constant over the rows
  const float cx_u = fmaf(tx_u, tx_u, 1.0f);
  const float P2x = Pp[1] + Pp[2];

  __shared__ __attribute__((aligned(16))) float slab[kWavesPerBlock][kSlabFloats];
  __shared__ int counts[kWavesPerBlock];
  __shared__ double sh[16];
  __shared__ double sums[4];
  __shared__ float sh_P[4];
  __shared__ unsigned sh_word;
  __shared__ int sh_n;
  float* my = slab[wave];
  // write side: component quad q of pixel `lane` at my[q*kQuadStride + lane*4 .. +3]
  f32x4* wr = reinterpret_cast<f32x4*>(my + lane * 4);
  // read side: MFMA operand lane l <- component c = l&15 of pixel 4g + (l>>4); g enters as a constant offset of 16 floats
  const float* rd = my + ((lane >> 2) & 3) * kQuadStride + (lane >> 4) * 4 + (lane & 3);

  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  int n_valid = 0;
  float hr0[RPW], hr1[RPW];                                   // this lane's residual pairs (NaN = no constraint), kept for the log-likelihood

  // The row is straight-line code with two divergent regions (tap fetch; constraint / no constraint): no nested early exits
  // and no per-exit default values (pixel_math.h, the *_flat stages).  The reference row of iteration k+1 is requested before
  // row k is processed: one of the two dependent memory round trips of a row (reference pixel -> projected tap addresses) is
  // off the critical path.  The rows are unrolled (the residual pairs live in registers, which cannot be indexed at run time).
  const float P00 = Pp[0], P11 = Pp[3];
  const int u_c = LINEAR ? lane : min(u_r, g.w - 1);
  auto load_ref = [&](int v_r) {                              // clamped: rows / segments past the end are masked by in_image
    const int idx = LINEAR ? min(v_r * kTileW + lane, n_px - 1) : min(v_r, g.h - 1) * g.w + u_c;
    return refR[idx];                                         // tiled: 64 lanes x 16 B = 1 KiB contiguous per wave
  };
  auto sweep_row = [&](int v_r, const float4 ref, float& keep0, float& keep1) __attribute__((always_inline)) {
    bool in_image;
    size_t pix;                                               // index of this lane's pixel in the level
    float tx_p, ty_p, cx;
    if constexpr (LINEAR) {
      const int idx = v_r * kTileW + lane;
      in_image = idx < n_px;
      pix = size_t(idx);
      const int pc = in_image ? idx : 0;
      int row = int(float(pc) * inv_w);                       // idx < 2^24: one float multiply lands within one row of the quotient
      int col = pc - row * g.w;
      if (col < 0) { col += g.w; row -= 1; }
      if (col >= g.w) { col -= g.w; row += 1; }
      tx_p = g.tx[col];
      ty_p = g.ty[row];
      cx = fmaf(tx_p, tx_p, 1.0f);
    } else {
      in_image = col_ok && v_r < g.h;
      pix = size_t(v_r) * g.w + u_r;                          // scalar row offset + lane
      tx_p = tx_u;
      ty_p = g.ty[min(v_r, g.h - 1)];
      cx = cx_u;
    }
    const PixelProj p = pixel_project_flat(g, KT, in_image ? ref.x : nanv, tx_p, ty_p);
    PixelTaps t;
    if (exp_skip & 4) {                                       // EXPERIMENT
      t.A00 = t.A10 = t.A01 = t.A11 = ref;
      t.B00 = t.B10 = t.B01 = t.B11 = make_float2(ref.z, ref.w);
    } else
    if (p.ok) pixel_fetch(g, curA, curB, p, t);               // lanes without a usable projection are masked out of `valid`
    PixelTerms o;
    const bool valid = pixel_finish_flat(g, ref, p, t, o) && p.ok;
    n_valid += __popcll(__ballot(valid));                     // exact count on the scalar unit
    if constexpr (MODE != 2) {
      keep0 = valid ? o.r0 : nanv;
      keep1 = o.r1;
    }
    if constexpr (MODE == 1 || MODE == 2) {
      if (in_image && !(exp_skip & 16)) scratch[size_t(pair) * size_t(n_px) + pix] = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);
    }
    if (exp_skip & 2) {                                       // EXPERIMENT: no accumulation at all
      acc0[0] += o.r0 + o.gix + o.giy + o.gzx + o.gzy;
      return;
    }
    if (valid && (exp_skip & 8)) {                            // EXPERIMENT: no weights / Jacobian arithmetic
      wr[0] = f32x4{o.gix, o.giy, o.gzx, o.gzy};
      wr[kQuadStride / 4] = f32x4{o.r0, o.r1, o.X, o.Y};
      wr[2 * (kQuadStride / 4)] = f32x4{o.Z, o.r1, o.X, o.Y};
      wr[3 * (kQuadStride / 4)] = f32x4{o.r0, o.r1, 0.0f, 0.0f};
    } else
    if (valid) {
      // t-distribution weight with the PREVIOUS pass' precision (Q11); first pass on a level: w = 1.  sqrt(w) is folded into
      // the four gradient factors of the Jacobian rows.
      const float sw = first ? 1.0f : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);
      float J0[6], J1[6];
      jacobian_rows_fast(o, sw, tx_p, ty_p, cx, fmaf(ty_p, ty_p, 1.0f), J0, J1);
      wr[0] = f32x4{J0[0], J0[1], J0[2], J0[3]};
      wr[kQuadStride / 4] = f32x4{J0[4], J0[5], J1[0], J1[1]};
      wr[2 * (kQuadStride / 4)] = f32x4{J1[2], J1[3], J1[4], J1[5]};
      wr[3 * (kQuadStride / 4)] = f32x4{sw * o.r0, sw * o.r1, 0.0f, 0.0f};
    } else {
      // a pixel without a constraint contributes a zero vector: four stores of one zero quad instead of clearing the
      // fourteen component registers on every row
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      wr[0] = zero;
      wr[kQuadStride / 4] = zero;
      wr[2 * (kQuadStride / 4)] = zero;
      wr[3 * (kQuadStride / 4)] = zero;
    }
    // the slab is private to this wavefront and LDS executes a wavefront's operations in order: only the compiler has to
    // be kept from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!(exp_skip & 1))
#pragma unroll
    for (int grp = 0; grp < 16; grp += 2) {                   // 4 pixels per MFMA, two independent accumulator chains
      const float a0 = rd[grp * 16], a1 = rd[grp * 16 + 16];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  float4 ref_cur = load_ref(row0);
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int v_r = row0 + k * kWavesPerBlock;                // scalar: image row (tiled) or segment (linear)
    float4 ref_next = ref_cur;
    if (k + 1 < RPW) ref_next = load_ref(v_r + kWavesPerBlock);
    sweep_row(v_r, ref_cur, hr0[k], hr1[k]);
    ref_cur = ref_next;
  }

  // lane l, register i holds G[row (l>>4)*4 + i][col l&15] of this wavefront's rows
#pragma unroll
  // (the wavefront is done with its slab: its Gram matrix goes into the first 256 floats)
  for (int i = 0; i < 4; ++i) my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
  if (lane == 0) counts[wave] = n_valid;
  __syncthreads();
  // fold the four wavefront Gram matrices into the canonical partial row (device_types.h); vector layout:
  // components 0..5 = J0, 6..11 = J1, 12 = r0, 13 = r1
  const int k = threadIdx.x;
  if (k < kNumAcc) {
    auto G = [&](int r, int c) {
      const int e = r * 16 + c;
      return (slab[0][e] + slab[1][e]) + (slab[2][e] + slab[3][e]);
    };
    float v;
    if (k == kAccN) v = float((counts[0] + counts[1]) + (counts[2] + counts[3]));
    else if (k == kAccS) v = G(12, 12);
    else if (k == kAccS + 1) v = G(12, 13);
    else if (k == kAccS + 2) v = G(13, 13);
    else if (k < kAccB00) {
      const int blockId = (k - kAccJ00) / 21;                // 0: J0J0, 1: J1J1, 2: J0J1 symmetrised
      int o = (k - kAccJ00) % 21, i = 0;
      while (o >= 6 - i) { o -= 6 - i; ++i; }                // upper-triangular row-major index -> (i, j)
      const int j = i + o;
      if (blockId == 0) v = G(i, j);
      else if (blockId == 1) v = G(6 + i, 6 + j);
      else v = G(i, 6 + j) + G(j, 6 + i);
    } else if (k < kAccB01) v = G(k - kAccB00, 12);
    else if (k < kAccB11) v = G(k - kAccB01, 13) + G(6 + (k - kAccB01), 12);
    else v = G(6 + (k - kAccB11), 13);
    float* row = partials + (size_t(pair) * tiles + tile) * kAccStride;
    // the four scale sums are read by another workgroup of this launch (the pair's last arriver): write-through stores
    if (MODE != 2 && k < 4) store_agent(reinterpret_cast<unsigned*>(row + k), __float_as_uint(v));
    else row[k] = v;
  }
  if constexpr (MODE == 2) return;
  if constexpr (MODE == 3) {                                  // EXPERIMENT (timing only): the log-likelihood arithmetic without the hand-off
    double prod = 1.0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const double f = 1.0 + 0.2 * double(mahalanobis(hr0[r], hr1[r], Pp));
      prod *= hr0[r] == hr0[r] ? f : 1.0;
    }
    double t = wave_sum_double(log(prod));
    if (lane == 0) sh[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) ll_partials[size_t(pair) * ll_stride + tile] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    return;
  }

  // ---- per-pair hand-off: arrive, (last arriver: publish P), wait for P ----------------------------------------------------
  PairSync* sy = sync + pair;
  if (wave == 0) {                                            // wavefront 0 holds the four write-through stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) sh_word = __hip_atomic_fetch_add((GlobalU32)&sy->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const unsigned ticket = sh_word;
  const unsigned gen = ticket / unsigned(tiles) + 1u;         // the generation (= sweep of this level) this arrival belongs to
  const bool last = ticket % unsigned(tiles) == unsigned(tiles) - 1u;
  if (last) {                                                 // workgroup-uniform
    reduce_partials_scale_coherent(partials, pair, tiles, &slab[0][0], sh, sums);
    if (threadIdx.x == 0) {
      float C[3], P[4];
      const int n = scale_from_sums(sums, C, P);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        sh_P[i] = P[i];
        store_agent(reinterpret_cast<unsigned*>(&sy->P[i]), __float_as_uint(P[i]));
      }
      sh_n = n;
      store_agent(reinterpret_cast<unsigned*>(&sy->n), unsigned(n));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the payload has left before the flag is stored
      store_agent(&sy->flag, gen);
    }
  } else if (threadIdx.x == 0) {
    unsigned spins = 0;
    bool ok = true;
    while (load_agent(&sy->flag) < gen) {                     // ONE lane polls ONE word
      __builtin_amdgcn_s_sleep(8);
      if (++spins > kSpinLimit) { ok = false; break; }
    }
    asm volatile("" ::: "memory");
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sh_P[i] = __uint_as_float(load_agent(reinterpret_cast<const unsigned*>(&sy->P[i])));
      sh_n = int(load_agent(reinterpret_cast<const unsigned*>(&sy->n)));
    } else {                                                  // never a wrong number: NaN precision + sticky error word
#pragma unroll
      for (int i = 0; i < 4; ++i) sh_P[i] = nanv;
      sh_n = 6;
      __hip_atomic_fetch_or((GlobalU32)error_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();

  // ---- this tile's share of sum log(1 + 0.2 r^T P r)  (dense_tracking_impl.cpp:413-422) ------------------------------------
  // A lane multiplies the factors of its RPW pixels in float64 (renormalised with frexp every eight factors, so the product
  // can neither overflow nor lose precision) and takes ONE log; wavefront sums by shuffle, then (w0 + w1) + (w2 + w3).
  float P[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) P[i] = sh_P[i];
  double t = 0.0;
  if (sh_n >= 6) {
    double prod = 1.0;
    int exponent = 0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const double f = 1.0 + 0.2 * double(mahalanobis(hr0[r], hr1[r], P));
      prod *= hr0[r] == hr0[r] ? f : 1.0;
      if ((r & 7) == 7 && r + 1 < RPW) {
        int e;
        prod = frexp(prod, &e);
        exponent += e;
      }
    }
    t = log(prod) + double(exponent) * 0.6931471805599453094;
  }
  t = wave_sum_double(t);
  if (lane == 0) sh[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) ll_partials[size_t(pair) * ll_stride + tile] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <int RPW>
static void launch_m(hipStream_t s, int ll_mode, bool finest, const LevelGeom& g, const PairPtrs* pairs, const PairState* states, int n_pairs,
                     float* partials, PairSync* sync, double* ll_partials, int ll_stride, float2* scratch, unsigned* error_word) {
  const int total = g.tiles_x * g.tiles_y * n_pairs;
  const int per_xcd = (total + 7) / 8;
  const dim3 grid(per_xcd * 8), block(kBlock);
  static const int exp_lds = getenv("DVO_EXP_LDS") ? atoi(getenv("DVO_EXP_LDS")) : 0;      // EXPERIMENT: occupancy limiter
  static const int exp_mode3 = getenv("DVO_EXP_MODE3") ? atoi(getenv("DVO_EXP_MODE3")) : 0;
  static const int exp_skip = getenv("DVO_EXP_SKIP") ? atoi(getenv("DVO_EXP_SKIP")) : 0;
#define DVO_LAUNCH_M(FIN, LIN, MODE) \
  k_residual_reduce_mfma<RPW, FIN, LIN, MODE><<<grid, block, exp_lds, s>>>(g, pairs, states, n_pairs, partials, sync, ll_partials, ll_stride, scratch, error_word, per_xcd, exp_skip)
  if (exp_mode3 && finest && !g.linear) {
    DVO_LAUNCH_M(true, false, 3);
  } else if (ll_mode == kLlSecondSweep) {
    if (g.linear) DVO_LAUNCH_M(false, true, 2);
    else if (finest) DVO_LAUNCH_M(true, false, 2);
    else DVO_LAUNCH_M(false, false, 2);
  } else if (scratch) {                                       // parity / error-image entry points only
    if (g.linear) DVO_LAUNCH_M(false, true, 1);
    else DVO_LAUNCH_M(false, false, 1);
  } else if (g.linear) {
    if (finest) DVO_LAUNCH_M(true, true, 0);
    else DVO_LAUNCH_M(false, true, 0);
  } else {
    if (finest) DVO_LAUNCH_M(true, false, 0);
    else DVO_LAUNCH_M(false, false, 0);
  }
#undef DVO_LAUNCH_M
}

void launch_residual_reduce_mfma(hipStream_t s, int ll_mode, int rows_per_wave, bool finest, const LevelGeom& g, const PairPtrs* pairs,
                                 const PairState* states, int n_pairs, float* partials, PairSync* sync, double* ll_partials,
                                 int ll_stride, float2* scratch, unsigned* error_word) {
  switch (rows_per_wave) {
    case 1: launch_m<1>(s, ll_mode, finest, g, pairs, states, n_pairs, partials, sync, ll_partials, ll_stride, scratch, error_word); break;
    case 2: launch_m<2>(s, ll_mode, finest, g, pairs, states, n_pairs, partials, sync, ll_partials, ll_stride, scratch, error_word); break;
    case 4: launch_m<4>(s, ll_mode, finest, g, pairs, states, n_pairs, partials, sync, ll_partials, ll_stride, scratch, error_word); break;
    default: launch_m<8>(s, ll_mode, finest, g, pairs, states, n_pairs, partials, sync, ll_partials, ll_stride, scratch, error_word); break;
  }
}

}  // namespace dvo_hip
