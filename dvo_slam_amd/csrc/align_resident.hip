// align_resident.hip -- a whole coarse-to-fine alignment (or its coarse levels) in ONE launch: the latency path.
//
// The launch path (capi_schedule.inc::run_batch) spends one to three launches per Gauss-Newton iteration; for a lone pair, or the two pairs
// LocalTracker aligns per frame (dvo_slam/src/local_tracker.cpp:180-184), nearly all of a match is launch floor (5 us per
// dependent launch on this queue) and memory round trips of few-microsecond kernels (profiles/r02_d_single_pair_timeline.txt:
// 66 launches, 0.5 ms).  Here a pair is owned by a GROUP of G resident workgroups of 8 wavefronts for the whole match
// ("The latency path"):
//
//   sweep      wavefronts 1..7 of every workgroup sweep their 64-pixel segments of the level (the per-pixel arithmetic and the Gram
//              accumulation on the matrix cores of align_mfma.hip; residual pairs stay in LDS, or in scratch read back by the SAME
//              wavefront only); the workgroup folds its wavefronts into one canonical partial row;
//   exchange   ONE per pass: the rows travel between the workgroups of the group (below) together with the partial log-likelihood
//              sums of the pass BEFORE; every workgroup adds them in the same order in float64 and derives the same scale /
//              precision P (dense_tracking.cpp:295);
//   loop body  wavefront 0 (no segments) of EVERY workgroup runs the reference's loop body (solver_logic.h::gn_step: 6x6 solve,
//              SE(3) update, termination) on its own LDS copy of the pair's state -- redundantly and identically, so the new pose
//              needs no broadcast.  It runs speculatively: the pass is treated as accepted, the sweeping wavefronts are released
//              the moment the next estimate is in place (the rest of the body is written behind the release), they sum
//              log(1 + 0.2 r^T P r) over their residuals next to it, and the accept / revert question is settled one exchange
//              later (gn_commit_loglik); a rejection restores the state the pass started from and runs the body in full form, which
//              takes the reference's revert path.  tests/test_emul_device.py runs this control flow on the host against the plain
//              loop: the same bits;
//   levels     follow each other inside the kernel (gn_level_begin); the host is not involved until the launch ends, and for small
//              batches not even then: results and statistics go to pinned host memory and a done word (capi_schedule.inc, direct path).
//
// The exchange is the "LL" protocol of collective libraries: a row is 8-byte slots {value, sequence number}, written with relaxed
// agent-scope stores and polled with relaxed agent-scope loads (sc1: through the non-coherent per-XCD L2s).  An aligned 8-byte
// access is single-copy atomic, so a slot whose sequence number is current carries current data -- no flag, no fence and ONE round
// trip when the data is there.  Rows travel in a ring of four buffers; heartbeat words keep a workgroup from running more than three
// exchange ahead of its peers.  All workgroups of a launch with G > 1 must be on the device together (one per compute unit; the
// host sizes G accordingly and lets such launches take turns); polling is bounded: a group that waits in vain raises the error
// word and leaves, and the host repeats the batch on the launch path.  With G = 1 (large batches, coarse levels) there is no
// exchange and no residency requirement at all.
#include "align_common.h"
#ifdef DVO_RESIDENT_CLOCKS
namespace dvo_hip {
__device__ unsigned long long g_resident_clk[32];
__device__ unsigned long long g_gn_prev;
}
#define DVO_GN_CLK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = wall_clock64(); if (i) dvo_hip::g_resident_clk[16 + (i)] += now_ - dvo_hip::g_gn_prev; dvo_hip::g_gn_prev = now_; } } while (0)
#endif
#include "solver_logic.h"
#include "linear_walk.h"
#include "sweep_parts.h"

namespace dvo_hip {

namespace {

constexpr int kSpinLimit = 1 << 18;              // polls before a group gives up (a fraction of a second)

typedef TapPlanes Taps;

struct RefSeg {                                   // one 64-pixel segment of the reference plane: {Zsel, I} and the four neighbours of I
  float z, i, left, right, up, down;
};

__device__ __forceinline__ unsigned long long slot_pack(unsigned value_bits, unsigned seq) {
  return (static_cast<unsigned long long>(seq) << 32) | value_bits;
}
__device__ __forceinline__ void slot_store(unsigned long long* p, unsigned value_bits, unsigned seq) {
  __hip_atomic_store(p, slot_pack(value_bits, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One slot of every row of a group in ONE round trip: the caller's lane polls rows q, q + kGatherLanes, ... (at most
// kResidentMaxGroup / kGatherLanes loads in flight) until all carry sequence number `seq`, and hands them to `consume` in row order.
// false: timed out.
constexpr int kGatherLanes = 4;
constexpr int kGatherRows = kResidentMaxGroup / kGatherLanes;
template <typename F>
__device__ __forceinline__ bool slot_gather(const unsigned long long* group_rows, int G, int q, int parity, int slot, unsigned seq, F&& consume) {
  unsigned long long v[kGatherRows];
  int spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int r = 0; r < kGatherRows; ++r) {
      const int j = q + r * kGatherLanes;
      if (j < G) {
        v[r] = __hip_atomic_load(group_rows + (size_t(j) * kResidentRing + parity) * kResidentSlots + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok &= unsigned(v[r] >> 32) == seq;
      }
    }
    if (ok) break;
    if (++spins > kSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int r = 0; r < kGatherRows; ++r)
    if (q + r * kGatherLanes < G) consume(unsigned(v[r]));
  return true;
}

// values every lane reads from LDS or from a table are the same in all lanes, but the compiler cannot know: say so, so that the
// loops and branches they steer stay scalar and the plane pointers stay in scalar registers (buffer resources)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ const T* uniform(const T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

constexpr int kQuietPollGroup = 16;              // groups with more active workgroups than this watch one slot per row first

// a POD copied inside LDS by one wavefront (its lanes' LDS operations execute in order: no barrier)
template <typename T>
__device__ __forceinline__ void wave_copy(T* dst, const T* src, int lane) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < int(sizeof(T) / 4); i += 64) d[i] = s[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ... and out of LDS into global memory
template <typename T>
__device__ __forceinline__ void wave_copy_out(T* dst, const T* src, int lane) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (int i = lane; i < int(sizeof(T) / 4); i += 64) d[i] = s[i];
}

template <typename T>
__device__ __forceinline__ void coop_copy(T* dst, const T* src) {
  static_assert(sizeof(T) % 4 == 0, "POD must be a multiple of 4 bytes");
  const unsigned* s = reinterpret_cast<const unsigned*>(src);
  unsigned* d = reinterpret_cast<unsigned*>(dst);
  for (int i = threadIdx.x; i < int(sizeof(T) / 4); i += blockDim.x) d[i] = s[i];
}

}  // namespace

#ifdef DVO_RESIDENT_CLOCKS
// experiment build only (scripts/ubench/resident_clocks.py): where an iteration's time goes, 100 MHz wall clock, workgroup 0
#define CLK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = wall_clock64(); g_resident_clk[i] += now_ - clk_prev_; clk_prev_ = now_; } } while (0)
extern "C" int dvo_hip_debug_resident_clocks(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_resident_clk), sizeof(g_resident_clk)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_resident_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define CLK(i) ((void)0)
#endif

// COMPAT: the reference's x * rcp(z) in projection and weights with the host CPU's reciprocal table (option "ref_compat",
// LevelGeom::rcp_table) -- the arithmetic of the launch path's sweeps in that mode
template <bool COMPAT>
__global__ __launch_bounds__(kResidentBlock) void k_match_resident(const ResidentArgs a) {
  extern __shared__ __attribute__((aligned(16))) float slab_mem[];          // one slab of kSlabFloats per SWEEPING wavefront
  __shared__ PairState st, st_before[2];                                    // st_before[p & 1]: the state pass p's loop body started from
  __shared__ dvo_hip_level_stats lvl, lvl_before[2];
  __shared__ dvo_hip_iteration_stats rec[2];                                // the records of the pass in flight and the one before
  __shared__ double sums[2][kAccStride];                                    // likewise its reduced accumulators
  __shared__ double sums_q[kGatherLanes][kAccStride];
  __shared__ float2 res_lds[kResidentSweepers][kResidentRowsLds][kTileW];   // the residual pairs of a sweeping wavefront's first segments
  __shared__ double ll_group[kResidentMaxGroup];
  __shared__ double ll_waves[kResidentWaves];
  __shared__ GnSpeculation speculation;
  __shared__ int counts[kResidentWaves];
  __shared__ int rec_slot[2];                                               // record index of rec[i], -1: nothing to write out
  __shared__ int bail, pending, level_over, kt_ready;

  const int G = a.group, tid = threadIdx.x;
  const int pair = blockIdx.x / G, wg = blockIdx.x - pair * G;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wavefront 0 of a workgroup is its solver: it holds no segments, so that the loop body starts the moment the exchange is through
  // while the other seven sum the log-likelihood of their residuals next to it
  const bool sweeper = wave > 0;
  const int gw = wg * kResidentSweepers + (wave - 1), W = G * kResidentSweepers;   // this sweeping wavefront among the group's
  unsigned long long* group_rows = a.exchange + size_t(pair) * G * kResidentRing * kResidentSlots;
  unsigned long long* my_rows = group_rows + size_t(wg) * kResidentRing * kResidentSlots;
  // one word per workgroup behind the rows of all pairs: the number of the exchange the workgroup has begun (flow control, below)
  unsigned* group_beats = reinterpret_cast<unsigned*>(a.exchange + size_t(a.n_pairs) * G * kResidentRing * kResidentSlots) + size_t(pair) * G;
  unsigned seq = a.sequence_base;

#ifdef DVO_RESIDENT_CLOCKS
  unsigned long long clk_prev_ = wall_clock64();
#endif
  if (tid == 0) { bail = 0; pending = 0; rec_slot[0] = rec_slot[1] = -1; }
  if (tid < kResidentWaves) ll_waves[tid] = 0.0;
  if (a.use_inline) {
    if (tid == 0) gn_init_pair(st, a.prm, a.inline_T + size_t(pair) * 16);
  } else if (a.T_init) {
    if (tid == 0) gn_init_pair(st, a.prm, a.T_init + size_t(pair) * 16);
  } else {
    coop_copy(&st, &a.states[pair]);
  }
  __syncthreads();

  // the Gram entries that make up accumulator `tid` of the canonical partial row, looked up once
  int fold_e1 = -1, fold_e2 = -1;
  if (tid > kAccN && tid < kNumAcc) gram_entries_of_accumulator(tid, fold_e1, fold_e2);

  float* my = slab_mem + (sweeper ? wave - 1 : 0) * kSlabFloats;              // (wavefront 0 never touches it)
  f32x4* wr = reinterpret_cast<f32x4*>(my + lane * 4);
  const float* rd = my + ((lane >> 2) & 3) * kQuadStride + (lane >> 4) * 4 + (lane & 3);
  const float nanv = __builtin_nanf("");

  for (int level = a.first_level; level >= a.last_level; --level) {
    const LevelGeom g = a.geom[level];
    PairPtrs pp = a.use_inline ? a.inline_ptrs[level * a.n_pairs + pair] : a.pair_ptrs[size_t(level) * a.n_pairs + pair];
    pp.refR = uniform(pp.refR);
    pp.curA = uniform(pp.curA);
    pp.curB = uniform(pp.curB);
    pp.n_selected = uniform(pp.n_selected);
    const int level_slot = uniform(st.n_levels);                              // the record gn_level_begin is about to open
    const bool have_level = level_slot < a.prm.cap_levels;
    dvo_hip_level_stats* lvl_global = a.levels + size_t(pair) * a.prm.cap_levels + level_slot;
    SolverParams local = a.prm;                                               // gn_* address levels[n_levels - 1] and
    local.cap_levels = have_level ? level_slot + 1 : 0;                       // iters[n_iters_total]: biased so that those are in LDS
    __syncthreads();
    if (tid == 0) {
      gn_level_begin(st, local, g, level, *pp.n_selected, &lvl - level_slot);
      level_over = 0;
      kt_ready = 0;
      rec_slot[0] = rec_slot[1] = -1;
    }
    // (the barrier that publishes the level's first estimate comes after the wavefronts have asked for their first reference
    //  segment: that round trip runs under lane 0's se3 log / exp)
    if (wave == 0) {                                           // the state the first pass starts from (see `restore` below)
      wave_copy(&st_before[0], &st, lane);
      wave_copy(&lvl_before[0], &lvl, lane);
    }

    const int n_px = g.w * g.h, n_seg = (n_px + kTileW - 1) / kTileW;
    // workgroups beyond the level's segments have nothing to sweep: their rows are zero and are neither written nor read
    const int G_act = min(G, (n_seg + kResidentSweepers - 1) / kResidentSweepers);
    const float inv_w = 1.0f / float(g.w);
    const __amdgpu_buffer_rsrc_t refR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.refR), 0, n_px * 8, 0x00020000);
    Taps taps;
    taps.A = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(pp.curA), 0, n_px * 16, 0x00020000);
    taps.B = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(pp.curB), 0, n_px * 8, 0x00020000);
    taps.rowA = g.w * 16;
    taps.rowB = g.w * 8;
    float2* residuals = a.scratch + size_t(pair) * n_px;

    auto locate = [&](int idx, int& row, int& col) { locate_pixel(idx, g.w, inv_w, row, col); };
    auto load_i = [&](int pixel) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(refR, pixel * 8 + 4, 0, 0)); };
    auto load_seg = [&](int seg) {
      RefSeg r;
      const int idx = min(seg * kTileW + lane, n_px - 1);
      int row, col;
      locate(idx, row, col);
      const f32x2 zi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(refR, idx * 8, 0, 0));
      r.z = zi.x; r.i = zi.y;
      r.left = load_i(idx - (col > 0 ? 1 : 0));               // clamped like the reference's derivative code, rgbd_image.cpp:419-489
      r.right = load_i(idx + (col < g.w - 1 ? 1 : 0));
      r.up = load_i(idx - (row > 0 ? g.w : 0));
      r.down = load_i(idx + (row < g.h - 1 ? g.w : 0));
      return r;
    };
    // the wavefront's first reference segment does not change over the iterations of a level: it stays in registers
    RefSeg seg0 = load_seg(min(max(gw, 0), n_seg - 1));
    __syncthreads();
    CLK(0);                                                    // level begin (and the kernel's prologue)

    int pass = 0;                                              // exchanges of this level so far (selects the double buffers)
    for (;;) {
      const int cur = pass & 1, prev = cur ^ 1;
      const bool do_sweep = uniform(st.active) != 0;          // 0: the level is over but for the verdict on its last pass
      // ---- sweep: this wavefront's segments gw, gw + W, ... at the (speculatively advanced) estimate ---------------------------
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      int n_valid = 0;
      if (do_sweep && sweeper && gw < n_seg) {
        float KT[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) KT[i] = st.KT[i];
        const float P00 = st.P_prev[0], P2x = st.P_prev[1] + st.P_prev[2], P11 = st.P_prev[3];
        const bool first = st.first != 0;
        RefSeg now = seg0;
        int held = 0;                                         // segments of this wavefront so far (scalar)
#pragma unroll 1
        for (int seg = gw; seg < n_seg; seg += W) {
          RefSeg nxt = now;
          if (seg + W < n_seg) nxt = load_seg(seg + W);       // requested before this segment is processed
          const float4 ref = make_float4(now.z, now.i, (now.right - now.left) * 0.5f, (now.down - now.up) * 0.5f);
          const int idx = seg * kTileW + lane;
          const bool in_image = idx < n_px;
          int row, col;
          locate(in_image ? idx : 0, row, col);
          const float tx_p = g.tx[col], ty_p = g.ty[row];
          const PixelProj p = pixel_project_flat<COMPAT>(g, KT, in_image ? ref.x : nanv, tx_p, ty_p);
          PixelTaps t;
          if (p.ok) taps.fetch(p.base, t);
          PixelTerms o;
          const bool valid = pixel_finish_flat(g, ref, p, t, o) && p.ok;
          n_valid += __popcll(__ballot(valid));
          {
            const float2 rr = valid ? make_float2(o.r0, o.r1) : make_float2(nanv, nanv);   // NaN: no constraint (and past the end)
            if (held < kResidentRowsLds) res_lds[wave - 1][held][lane] = rr;
            else if (in_image) residuals[idx] = rr;
            ++held;
          }
          if (valid) {
            const float sw = first ? 1.0f : COMPAT ? tdist_weight_sqrt_compat(g.rcp_table, g.rcp_shift, o.r0, o.r1, st.P_prev)
                                                   : tdist_weight_sqrt_fast(o.r0, o.r1, P00, P2x, P11);   // previous pass' precision (Q11)
            float J0[6], J1[6];
            jacobian_rows_fast(o, sw, tx_p, ty_p, fmaf(tx_p, tx_p, 1.0f), fmaf(ty_p, ty_p, 1.0f), J0, J1);
            wr[0] = f32x4{J0[0], J0[1], J0[2], J0[3]};
            wr[kQuadStride / 4] = f32x4{J0[4], J0[5], J1[0], J1[1]};
            wr[2 * (kQuadStride / 4)] = f32x4{J1[2], J1[3], J1[4], J1[5]};
            wr[3 * (kQuadStride / 4)] = f32x4{sw * o.r0, sw * o.r1, 0.0f, 0.0f};
          } else {
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            wr[0] = zero;
            wr[kQuadStride / 4] = zero;
            wr[2 * (kQuadStride / 4)] = zero;
            wr[3 * (kQuadStride / 4)] = zero;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int grp = 0; grp < 16; grp += 2) {
            const float a0 = rd[grp * 16], a1 = rd[grp * 16 + 16];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc1, 0, 0, 0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          now = nxt;
        }
      }
      CLK(1);                                                  // sweep (thread 0's wavefront)
      // ---- fold the workgroup's wavefronts into the canonical partial row (device_types.h) -----------------------------------
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (sweeper) my[((lane >> 4) * 4 + i) * 16 + (lane & 15)] = acc0[i] + acc1[i];
      if (lane == 0) counts[wave] = n_valid;
      __syncthreads();
      float row_value = 0.0f;
      if (tid < kNumAcc) {
        auto Gm = [&](int e) {
          float t = 0.0f;
#pragma unroll
          for (int wv = 0; wv < kResidentSweepers; ++wv) t += slab_mem[wv * kSlabFloats + e];
          return t;
        };
        if (tid == kAccN) {
          int c = 0;
#pragma unroll
          for (int wv = 1; wv < kResidentWaves; ++wv) c += counts[wv];
          row_value = float(c);
        } else {
          row_value = Gm(fold_e1);
          if (fold_e2 >= 0) row_value += Gm(fold_e2);
        }
      }
      CLK(2);                                                  // waiting for the other wavefronts + fold
      // ---- the exchange: partial rows of this pass and log-likelihood sums of the pass before it, one round trip ----------------
      if (G > 1) {
        seq += 1;
        const int parity = seq & (kResidentRing - 1);             // the ring buffer this exchange travels in
        unsigned long long* mine = my_rows + parity * kResidentSlots;
        // Flow control: a workgroup without segments on this level (wg >= G_act) contributes no row, but it consumes every exchange
        // like the others (all of them run the loop body redundantly).  Every workgroup announces the exchange it begins in its
        // HEARTBEAT word (G consecutive 4-byte words per group: two cache lines, read by 64 otherwise idle lanes in the same round
        // trip as the rows), and nobody completes exchange s before everybody has begun exchange s - 2.  The rows travel in a ring
        // of kResidentRing = 4 buffers: the rows of exchange s overwrite those of s - 4, which everybody has finished reading --
        // the writer completed s - 1, so all had begun s - 3.  The heartbeats asked for are two exchanges old: the check never
        // waits unless a workgroup really lags (started late: the device shared with another launch).  (Without flow control the
        // sweeping workgroups could run any number of exchanges ahead of an idle one, which then polled for a sequence number that
        // had been overwritten, timed out after a quarter of a second and sent the batch to the launch path -- the "one launch in
        // seven" of two launches sharing the device.  Measured on the way: a zero ROW per idle workgroup, waited for every pass:
        // +17 % on a single pair's latency; heartbeat words waited for every pass with two buffers: +9 %.)
        if (tid == kNumAcc + 1) __hip_atomic_store(group_beats + wg, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wg < G_act && !((a.flags & kResidentFlagWithhold) && wg == 1)) {
          if (tid < kNumAcc) slot_store(mine + tid, __float_as_uint(row_value), seq);
          if (tid == kNumAcc) {                                // the log-likelihood sum of the pass before (its wavefronts' parts)
            double ll_wg = 0.0;
#pragma unroll
            for (int wv = 1; wv < kResidentWaves; ++wv) ll_wg += ll_waves[wv];
            const unsigned long long bits = __double_as_longlong(ll_wg);
            slot_store(mine + kResidentSlotLl, unsigned(bits), seq);
            slot_store(mine + kResidentSlotLl + 1, unsigned(bits >> 32), seq);
          }
        }
        const int k = tid & 127, q = tid >> 7;                // lane group q of slot k adds rows q, q + 4, ...
        if (G_act > kQuietPollGroup && !(a.flags & kResidentFlagNoQuietPoll)) {
          // many rows: a failed poll of everything by everybody (G x 85 x G loads) drowns the fabric the sweeps of the slower
          // workgroups need.  One wavefront watches one slot per row until the rows have been started, then everybody reads.
          if (wave == 0 && lane < G_act) {
            int spins = 0;
            while (unsigned(__hip_atomic_load(group_rows + (size_t(lane) * kResidentRing + parity) * kResidentSlots + kAccN, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT) >> 32) != seq) {
              if (++spins > kSpinLimit) { bail = 1; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          __syncthreads();
        }
        if (k < kNumAcc) {
          double t = 0.0;
          if (!slot_gather(group_rows, G_act, q, parity, k, seq, [&](unsigned bits) { t += double(__uint_as_float(bits)); })) bail = 1;
          sums_q[q][k] = t;
        } else if (k >= 96 && q >= 2) {                       // heartbeats: lane j waits until workgroup j has begun exchange seq - 2
          const int j = (k - 96) + 32 * (q - 2);
          if (j < G) {
            int spins = 0;
            while (int(__hip_atomic_load(group_beats + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < -2) {
              if (++spins > kSpinLimit) { bail = 1; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
        } else if (k >= 96 && q < 2) {                        // the log-likelihood sum of workgroup j: both halves
          const int j = (k - 96) + 32 * q;
          if (j < G_act) {
            const unsigned long long* src = group_rows + (size_t(j) * kResidentRing + parity) * kResidentSlots + kResidentSlotLl;
            unsigned long long lo = 0, hi = 0;
            int spins = 0;
            for (;;) {
              lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (unsigned(lo >> 32) == seq && unsigned(hi >> 32) == seq) break;
              if (++spins > kSpinLimit) { bail = 1; break; }
              __builtin_amdgcn_s_sleep(1);
            }
            ll_group[j] = __longlong_as_double((hi << 32) | (lo & 0xffffffffull));
          }
        }
        __syncthreads();
        if (tid < kNumAcc) sums[cur][tid] = (sums_q[0][tid] + sums_q[1][tid]) + (sums_q[2][tid] + sums_q[3][tid]);
      } else {
        if (tid < kNumAcc) sums[cur][tid] = double(row_value);
        if (tid == 0) {
          double mine = 0.0;
#pragma unroll
          for (int wv = 1; wv < kResidentWaves; ++wv) mine += ll_waves[wv];
          ll_group[0] = mine;
        }
      }
      if (tid >= kNumAcc && tid < kAccStride) sums[cur][tid] = 0.0;
      __syncthreads();
      if (uniform(bail)) break;
      CLK(3);                                                  // exchange
      // ---- wavefront 0: the reference's loop body, redundantly (and identically) in every workgroup of the group; wavefronts
      //      1..7 meanwhile: the log-likelihood of THIS pass over their own residuals (dense_tracking_impl.cpp:406-425), which
      //      travels with the next exchange.  They do not wait for the whole loop body: gn_step publishes `ticket` in kt_ready
      //      the moment the next sweep's inputs (KT, P_prev, first, active) are in place and writes records, A_last and the
      //      prior's `initial` behind that, while the sweep is already running. -------------------------------------------------------
      const int ticket = pass + 1;
      bool over;
      if (wave == 0) {
        int restore = 0;
        if (uniform(pending)) {                                // the verdict on the pass before this one
          double ll_sum = lane < G_act ? ll_group[lane] : 0.0;  // a fixed tree: the same bits in every workgroup of the group
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) ll_sum += __shfl_xor(ll_sum, off, 64);
          if (lane == 0) {
            if (!gn_commit_loglik(st, speculation, ll_sum, rec[prev])) restore = 1;
            pending = 0;
            speculation.half_n_logdet = restore ? ll_sum : 0.0;   // (kept for the replay below)
          }
        }
        restore = uniform(restore);
        // One call site of the loop body for both of its uses (the float64 solver is inlined once):
        //   restore: the pass before was rejected -- back to the state it started from and once more in full form, which takes the
        //            revert path and ends the level (dense_tracking.cpp:312-317); the sweep just done at the estimate it would have
        //            produced is dropped;
        //   else:    this pass, speculatively (the level may already be over: then only the verdict was due).
        if (restore) {
          // st_before was taken when the pass before had run its loop body, i.e. before the verdict on the pass before THAT moved the
          // error chain on; a failed verdict leaves the chain alone, so the live values are the ones that belong to the restored state
          const double error_now = st.error, last_error_now = st.last_error;
          wave_copy(&st, &st_before[prev], lane);               // (the rejected pass is the one before this: parity `prev`)
          wave_copy(&lvl, &lvl_before[prev], lane);
          if (lane == 0) { st.error = error_now; st.last_error = last_error_now; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const int which = restore ? prev : cur;
        if (restore || do_sweep) {                             // the record's 51 NaNs by the whole wavefront, not by the solver lane
          unsigned long long* words = reinterpret_cast<unsigned long long*>(&rec[which]);
          for (int i = lane; i < int(sizeof(dvo_hip_iteration_stats) / 8); i += 64) words[i] = 0x7ff8000000000000ull;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
          if (restore || do_sweep) {
            const double ll_sum = restore ? speculation.half_n_logdet : 0.0;
            SolverParams prm = local;
            prm.cap_iters = st.n_iters_total + 1;
            if (!restore) rec_slot[cur] = st.n_iters_total;
            speculation.replay_reject = restore;
            speculation.record_prefilled = 1;
            speculation.defer_information = 1;
            speculation.release_word = &kt_ready;
            speculation.release_value = ticket;
            gn_step(st, prm, g, sums[which], ll_sum, &lvl - level_slot, &rec[which] - st.n_iters_total, &speculation);
            pending = restore ? 0 : speculation.needs_loglik;
            if (restore || !pending) level_over = 1;           // (not pending: too few constraints, over without a log-likelihood)
          } else {
            level_over = 1;                                    // the last pass was accepted and had ended the level itself
            speculation.information_ready = 0;
          }
          if (kt_ready != ticket) {                            // the paths that end the level did not get as far as the release
            __threadfence_block();
            kt_ready = ticket;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (uniform(speculation.information_ready)) {          // rec.information = A_last, 72 words by the wavefront
          const unsigned* src = reinterpret_cast<const unsigned*>(st.A_last);
          unsigned* dst = reinterpret_cast<unsigned*>(rec[which].information);
          for (int i = lane; i < 72; i += 64) dst[i] = src[i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        over = uniform(level_over) != 0;
        // the state the NEXT pass starts from, kept for the case that its verdict is a rejection (off the sweepers' path: they are
        // released)
        if (!over) {
          wave_copy(&st_before[prev], &st, lane);               // (the next pass has the parity of the one before this)
          wave_copy(&lvl_before[prev], &lvl, lane);
        }
        // the record of the pass before is final now; so is this pass' if the level ended without waiting for a log-likelihood
        if (wg == 0) {
          const int done_prev = uniform(rec_slot[prev]);
          if (done_prev >= 0 && done_prev < a.prm.cap_iters) wave_copy_out(a.iters + size_t(pair) * a.prm.cap_iters + done_prev, &rec[prev], lane);
          const int done_cur = uniform(rec_slot[cur]);
          if (over && !uniform(pending) && do_sweep && done_cur >= 0 && done_cur < a.prm.cap_iters)
            wave_copy_out(a.iters + size_t(pair) * a.prm.cap_iters + done_cur, &rec[cur], lane);
        }
        CLK(6);                                                // solver
#ifdef DVO_RESIDENT_CLOCKS
        if (tid == 0 && blockIdx.x == 0) g_resident_clk[15] += 1;
#endif
      } else {
        if (do_sweep) {
          float C[3], P[4];
          const int n = scale_from_sums(sums[cur], C, P);
          double ll = 0.0;
          if (n >= 6 && gw < n_seg) {
            double prod = 1.0;
            int exponent = 0, factors = 0, held = 0;
            for (int seg = gw; seg < n_seg; seg += W, ++held) {
              const int idx = seg * kTileW + lane;
              float2 r = make_float2(nanv, nanv);
              if (held < kResidentRowsLds) r = res_lds[wave - 1][held][lane];
              else if (idx < n_px) r = residuals[idx];
              if (r.x == r.x) prod *= 1.0 + 0.2 * double(mahalanobis(r.x, r.y, P));
              if (++factors == 8) {                           // eight factors at most between renormalisations (align_common.h)
                int e;
                prod = frexp(prod, &e);
                exponent += e;
                factors = 0;
              }
            }
            ll = log(prod) + double(exponent) * 0.6931471805599453094;
          }
          ll = wave_sum_double(ll);
          if (lane == 0) ll_waves[wave] = ll;
        }
        while (*static_cast<volatile int*>(&kt_ready) != ticket) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        over = *static_cast<volatile int*>(&level_over) != 0;
      }
      if (over) break;
      pass += 1;
    }
    if (uniform(bail)) break;
    __syncthreads();
    if (wg == 0 && have_level) coop_copy(lvl_global, &lvl);
  }
  if (uniform(bail)) {
    // the batch is repeated on the launch path: leave the pair in a defined, inactive state for whatever is still enqueued behind
    // this launch (the launch-path levels of a mixed run read it before the host has seen the error word)
    if (wg == 0 && tid == 0) {
      a.states[pair].active = 0;
      a.states[pair].n_levels = 0;
      a.states[pair].n_iters_total = 0;
    }
    if (tid == 0) __hip_atomic_store(a.error_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __syncthreads();
  if (wg == 0) {
    coop_copy(&a.states[pair], &st);
    if (a.results) {                                          // the match ends here: dense_tracking.cpp:368-373
      __threadfence();                                         // the level and iteration records this workgroup wrote
      __syncthreads();
      const dvo_hip_level_stats* my_levels = a.levels + size_t(pair) * a.prm.cap_levels;
      const dvo_hip_iteration_stats* my_iters = a.iters + size_t(pair) * a.prm.cap_iters;
      // composed in LDS (gn_finish reads its own output back) and written out by the workgroup: a.results may be host memory
      __shared__ dvo_hip_result result;
      if (tid == 0) gn_finish(st, a.prm, my_levels, my_iters, &result);
      __syncthreads();
      coop_copy(a.results + pair, &result);
      if (a.done_word) {
        if (a.host_levels) {                                   // the statistics the caller asked for, straight into its (pinned) arrays
          const int nl = min(uniform(st.n_levels), a.prm.cap_levels), ni = min(uniform(st.n_iters_total), a.prm.cap_iters);
          const unsigned* src = reinterpret_cast<const unsigned*>(my_levels);
          unsigned* dst = reinterpret_cast<unsigned*>(a.host_levels + size_t(pair) * a.prm.cap_levels);
          for (int i = tid; i < nl * int(sizeof(dvo_hip_level_stats) / 4); i += kResidentBlock) dst[i] = src[i];
          src = reinterpret_cast<const unsigned*>(my_iters);
          dst = reinterpret_cast<unsigned*>(a.host_iters + size_t(pair) * a.prm.cap_iters);
          for (int i = tid; i < ni * int(sizeof(dvo_hip_iteration_stats) / 4); i += kResidentBlock) dst[i] = src[i];
        }
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.done_word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  CLK(7);
}

size_t resident_dynamic_lds() { return size_t(kResidentSweepers) * kSlabFloats * sizeof(float); }

hipError_t launch_match_resident(hipStream_t s, const ResidentArgs& args, bool cooperative) {
  static bool configured[64] = {};                            // the attribute is per device (callers hold their context's lock;
  const size_t lds = resident_dynamic_lds();                  //  two contexts racing here set the same value twice)
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  if (device >= 64 || !configured[device]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_resident<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_resident<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    if (device < 64) configured[device] = true;
  }
  const bool compat = args.geom[args.first_level].rcp_table != nullptr;   // (option "ref_compat": every level's geometry carries the table)
  const dim3 grid(args.n_pairs * args.group), block(kResidentBlock);
  if (args.group > 1 && cooperative) {
    ResidentArgs copy = args;
    void* params[] = {&copy};
    return hipLaunchCooperativeKernel(compat ? reinterpret_cast<const void*>(k_match_resident<true>) : reinterpret_cast<const void*>(k_match_resident<false>),
                                      grid, block, params, unsigned(lds), s);
  }
  if (compat) k_match_resident<true><<<grid, block, lds, s>>>(args);
  else k_match_resident<false><<<grid, block, lds, s>>>(args);
  return hipGetLastError();
}

}  // namespace dvo_hip
