// capi.hip -- host side of libdvo_hip.so: contexts, device-resident frame pyramids and the batched
// coarse-to-fine Gauss-Newton driver behind the C-ABI of include/dvo_hip.h.
//
// The host never touches pixels or poses: it uploads two raw planes per frame, enqueues kernels on the
// context's HIP stream and polls one integer ("pairs still iterating") every few iterations.
#include <hip/hip_runtime.h>
#ifdef DVO_WITH_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
#endif
#include <xmmintrin.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dvo_hip.h"
#include "device_types.h"
#include "batch_policy.h"
#include "launch.h"

using namespace dvo_hip;

namespace {

std::string g_create_error;

// Bumped whenever device memory goes back to the allocator: what the small-table cache (PinnedRing) knows about the contents of
// device addresses is only good until an address can have been handed out again.
std::atomic<unsigned long long> g_free_epoch{1};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    if (p) {
      (void)hipFree(p);
      g_free_epoch.fetch_add(1, std::memory_order_relaxed);
    }
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
      g_free_epoch.fetch_add(1, std::memory_order_relaxed);
    }
    p = nullptr;
    bytes = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct PinnedBuf {               // host memory the device can write (mapped, coherent), grown on demand
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t n) {
    if (n <= bytes) return hipSuccess;
    release();
    hipError_t e = hipHostMalloc(&p, n, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) bytes = n;
    else p = nullptr;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct CameraGeom {             // RgbdCameraPyramid: per-level size, intrinsics and the point-cloud template
  int w0 = 0, h0 = 0, levels = 0;
  float K0[4] = {0, 0, 0, 0};
  int w[kMaxLevels], h[kMaxLevels];
  float K[kMaxLevels][4];
  float* tx[kMaxLevels];
  float* ty[kMaxLevels];
  DevBuf tables;
};

struct FrameLevel {
  int w = 0, h = 0;
  float* I = nullptr;
  float* Z = nullptr;
  float4* A = nullptr;
  float2* B = nullptr;
  float2* R = nullptr;
  float2* C = nullptr;         // {I, Z} of a current frame for the window sweep (null: not kept at this level)
  int cur_have = 0;            // flavours of the current-frame role that are built: kCurAB (A, B) | kCurC (C)
  bool selected = false;       // R / count built for (ithr, dthr) (reference role)
  float ithr = 0, dthr = 0;
};

}  // namespace

struct dvo_hip_frame {
  int levels = 0;
  const CameraGeom* cam = nullptr;
  FrameLevel lv[kMaxLevels];
  DevBuf pool;
  int* sel_count = nullptr;    // device, one int per level
  unsigned long long built_seq = 0;   // ticket of the last build-stream work that wrote this frame (0 = none pending)
  int deferred = 0;                   // named by a recorded, not yet executed ingest (option "defer_ingest")
  // Frames ingested from raw sensor planes have no float I / Z planes at level 0 (k_build_from_raw writes the role planes
  // straight from the raw data).  What level 0 can later be derived from: the current-role planes A + B if they exist (they
  // hold everything), else the 3-B copy of the raw planes in the frame's staging area.
  bool raw0 = false;
  bool raw_copy = false;
  float depth_scale = 0.0f;
};

// Small host -> device transfers (pointer tables, initial guesses) go through slots of pinned memory: from a pageable
// source the runtime stages the bytes itself and may hold the calling thread while it does; from a pinned slot the upload is a
// plain asynchronous copy and the host goes on enqueueing.  (No measurable difference in the benchmark loop, whose host side
// is dominated by the caller; kept because it takes a host-side wait out of the enqueue path.)  A slot is reused only after
// the copy that read it has completed (one event per slot).
// Small tables go from the pinned ring to device memory with a copy KERNEL, not with a copy command: copy commands of every stream
// share the DMA engine, where a few hundred bytes of table queue behind whatever is in flight -- behind 1.9 GB of raw planes when a
// caller streams 1024 pairs per step from host memory (the match then started 33 ms late, every step).
__global__ void k_copy_table(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, size_t n8) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n8) dst[i] = src[i];
}

// Waiting for a stream on the paths where the wait is part of a match's or a frame's latency: polling keeps the thread on its core
// instead of parking it in the driver; a wait that lasts longer than 20 ms falls back to the sleeping kind.
hipError_t sync_stream(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return hipSuccess;
    if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();   // "still running" is not an error to keep
    if (q != hipErrorNotReady) return q;
    if ((spins & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipStreamSynchronize(s);
  }
}

struct PinnedRing {
  static const int kSlots = 48;
  char* base = nullptr;
  size_t slot_bytes = 0;
  hipEvent_t done[kSlots] = {};
  bool in_flight[kSlots] = {};
  unsigned next = 0;

  // A streaming caller hands over the same frame sets step after step: the tables of a step (plane pointers of the frames to build,
  // of the pairs to align, identity initial guesses) are then the very bytes the device already holds at the very address.  The ring
  // remembers what it last sent to an address (a host copy) and sends nothing when it is asked for the same bytes again, on the same
  // stream, with no device memory freed in between (round 5: a 128-pair step of bench.py spent ~150 us of host time on ten such
  // uploads before its first alignment kernel was enqueued, and two dependent copy kernels at the head of the chain).
  struct Sent {
    void* dst = nullptr;
    hipStream_t stream = nullptr;
    unsigned long long epoch = 0, used = 0;
    std::vector<char> bytes;
  };
  static const int kSent = 24;
  Sent sent[kSent];
  unsigned long long clock = 0;
  long long skipped = 0;            // uploads answered from the cache (counter "table_uploads_skipped")
  bool cache = true;

  hipError_t upload(hipStream_t stream, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    Sent* slot = nullptr;
    if (cache) {
      const unsigned long long epoch = g_free_epoch.load(std::memory_order_relaxed);
      Sent* oldest = &sent[0];
      for (Sent& c : sent) {
        if (c.dst == dst) { slot = &c; break; }
        if (c.used < oldest->used) oldest = &c;
      }
      if (slot && slot->stream == stream && slot->epoch == epoch && slot->bytes.size() == bytes && std::memcmp(slot->bytes.data(), src, bytes) == 0) {
        slot->used = ++clock;
        skipped += 1;
        return hipSuccess;
      }
      if (!slot) slot = oldest;
      slot->dst = nullptr;                                       // (valid again once the copy below is enqueued)
      // Whatever else the cache believes about bytes of [dst, dst + bytes) is about to be overwritten: the slices of one table buffer
      // (ensure_roles) lie at offsets that depend on the batch size, so uploads of different batches overlap without sharing a start
      // address (round-5 advisor finding: batches of 3, 2, 3 pairs with a re-ingest between them left the 3-pair slice's entry alive
      // over the 2-pair slices' bytes, and the third upload was skipped).
      const char* lo = static_cast<const char*>(dst);
      for (Sent& c : sent)
        if (c.dst && static_cast<const char*>(c.dst) < lo + bytes && lo < static_cast<const char*>(c.dst) + c.bytes.size()) c.dst = nullptr;
    }
    const hipError_t e_up = upload_now(stream, dst, src, bytes);
    if (slot && e_up == hipSuccess) {
      slot->dst = dst;
      slot->stream = stream;
      slot->epoch = g_free_epoch.load(std::memory_order_relaxed);
      slot->used = ++clock;
      slot->bytes.assign(static_cast<const char*>(src), static_cast<const char*>(src) + bytes);
    }
    return e_up;
  }

  // which of `k` device buffers already holds exactly these bytes (sent on this stream, nothing freed since); -1: none
  int holder(DevBuf* candidates, int k, hipStream_t stream, const void* src, size_t bytes) const {
    if (!cache) return -1;
    const unsigned long long epoch = g_free_epoch.load(std::memory_order_relaxed);
    for (int i = 0; i < k; ++i)
      for (const Sent& c : sent)
        if (c.dst && c.dst == candidates[i].p && c.stream == stream && c.epoch == epoch && c.bytes.size() == bytes && std::memcmp(c.bytes.data(), src, bytes) == 0)
          return i;
    return -1;
  }

  void forget(const void* dst) {                                  // somebody else wrote to this address
    for (Sent& c : sent)
      if (c.dst == dst) c.dst = nullptr;
  }

  hipError_t upload_now(hipStream_t stream, void* dst, const void* src, size_t bytes) {
    if (bytes > slot_bytes) {                      // grow: wait for every copy in flight, then one new block
      hipError_t e = drain();
      if (e != hipSuccess) return e;
      if (base) (void)hipHostFree(base);
      base = nullptr;
      slot_bytes = 0;
      size_t want = 64 * 1024;
      while (want < bytes) want *= 2;
      e = hipHostMalloc(reinterpret_cast<void**>(&base), want * kSlots, hipHostMallocMapped);
      if (e != hipSuccess) return e;
      slot_bytes = want;
    }
    const unsigned k = next++ % kSlots;
    hipError_t e = hipSuccess;
    if (!done[k]) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
    if (e == hipSuccess && in_flight[k]) e = hipEventSynchronize(done[k]);
    if (e != hipSuccess) return e;
    std::memcpy(base + slot_bytes * k, src, bytes);
    if (bytes % 8 == 0 && reinterpret_cast<uintptr_t>(dst) % 8 == 0) {     // (every table here is pointers or doubles)
      const size_t n8 = bytes / 8;
      k_copy_table<<<dim3(unsigned((n8 + 255) / 256)), dim3(256), 0, stream>>>(
          static_cast<unsigned long long*>(dst), reinterpret_cast<const unsigned long long*>(base + slot_bytes * k), n8);
      e = hipGetLastError();
    } else {
      e = hipMemcpyAsync(dst, base + slot_bytes * k, bytes, hipMemcpyHostToDevice, stream);
    }
    if (e == hipSuccess) e = hipEventRecord(done[k], stream);
    in_flight[k] = e == hipSuccess;
    return e;
  }
  hipError_t drain() {
    for (int k = 0; k < kSlots; ++k)
      if (in_flight[k]) {
        const hipError_t e = hipEventSynchronize(done[k]);
        if (e != hipSuccess) return e;
        in_flight[k] = false;
      }
    return hipSuccess;
  }
  void release() {
    (void)drain();
    for (hipEvent_t& ev : done)
      if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
    if (base) (void)hipHostFree(base);
    base = nullptr;
    slot_bytes = 0;
  }
};

// The batch workspace: one HIP stream, device scratch and the pinned poll words of the Gauss-Newton loop.
// (Splitting a batch into concurrently iterating pair groups on several streams was measured and dropped: the coarse
// levels are bound by the host's launch rate, which more streams only divide -- profiles/r01_d_groups.txt.)
struct Workspace {
  hipStream_t stream = nullptr;
  DevBuf states, partials, scratch, ll_partials, lvl_stats, it_stats, results, t_init, counters, pair_sums;
  // the pairs' plane-pointer table, in one of a few buffers picked by the batch's first frames: a caller that alternates between two
  // or three frame sets finds each set's table where it left it (PinnedRing's cache of what was sent where)
  static const int kTableSlots = 4;
  DevBuf pair_ptrs[kTableSlots];
  unsigned pair_ptrs_next = 0;
  DevBuf win_fallbacks;          // one 64-bit counter: lanes of the window sweep whose taps were fetched from memory (align_window.hip)
  int* f16_range_flag = nullptr; // pinned, one word per pair of the batch: raised by a workgroup of the f16 Gram schedule whose Jacobian left the f16 range (gram_f16.h)
  size_t f16_range_words = 0;
  PinnedRing* tables = nullptr;  // the context's ring for small uploads
  int* host_status = nullptr;    // pinned: one word per Gauss-Newton step of a batch, written by the device (k_solver_step)
  size_t host_status_words = 0;
  std::string err;
  bool created = false;
  // resident match kernel: the exchange rows of the workgroup groups and the sequence numbers used so far
  DevBuf exchange;
  unsigned resident_sequence = 0;
  long long exchange_shape = -1;   // pairs x group of the launch the exchange buffer was last cleared for
  // ... and, when it runs a whole match of a small batch, where it leaves results and statistics: pinned host memory the host
  // thread reads as soon as the kernel has counted the pairs done (no copy command, no stream synchronisation)
  PinnedBuf direct_results, direct_levels, direct_iters, direct_done;
  bool needs_drain = false;        // a batch ended early: the stream is drained before buffers are reused
  bool device_may_lag = true;      // the last batch returned without waiting for the stream (the resident kernel's direct path)
  int resident_error_word = 0;     // index of the current resident launch's error word in host_status (a ring, see kResidentErrorWords)
  unsigned resident_launch_counter = 0;
  // the slow lane of a batch (run_batch, option "overlap_tails"): the stragglers of the levels run on a stream of their own beside the
  // batch's chain, with partial rows, residual pairs and log-likelihood sums of their own (two levels index those by different tile counts)
  hipStream_t tail_stream = nullptr;
  hipEvent_t tail_split = nullptr, tail_end = nullptr;
  DevBuf tail_partials, tail_scratch, tail_ll, tail_flags, tail_list;
};

// a helper thread of the concurrent pair groups (dvo_hip_context::opt_batch_groups) and the slice of the caller's batch it aligns
struct GroupWorker {
  dvo_hip_context* twin = nullptr;
  std::thread thread;
  std::mutex m;
  std::condition_variable cv;
  bool has_job = false, done = false, quit = false;
  // the job: a slice of the caller's batch
  int n = 0;
  dvo_hip_frame* const* refs = nullptr;
  dvo_hip_frame* const* curs = nullptr;
  const dvo_hip_config* cfg = nullptr;
  dvo_hip_result* results = nullptr;
  dvo_hip_level_stats* levels = nullptr;
  dvo_hip_iteration_stats* iters = nullptr;
  int cap_levels = 0, cap_iters = 0;
  hipEvent_t after = nullptr;                                  // the twin's stream waits for it (the role planes are ready)
  int rc = DVO_HIP_OK;
};


struct dvo_hip_context {
  // The reference hands one current pyramid to two trackers on two threads (dvo_slam/src/local_tracker.cpp:180-184) and runs
  // thread-local validators over shared keyframes (keyframe_graph.cpp:576-593): calls on one context from several host threads
  // are serialised here (recursive: dvo_hip_match -> dvo_hip_match_batch).  Contexts never share a lock.
  std::recursive_mutex mutex;
  int device = 0;
  hipStream_t stream = nullptr;    // == ws[0].stream
  std::string err;
  int opt_rows_per_wave = 0;
  int opt_iters_per_sync = 0;
  // Rendezvous of single-pair matches (dvo_hip_match): the reference's LocalTracker aligns every new image against two reference
  // frames from two threads at once (tbb::parallel_invoke, dvo_slam/src/local_tracker.cpp:180-184).  Two such calls with the same
  // current frame and the same configuration leave as ONE two-pair batch (one resident launch: 0.19 ms for both instead of 2 x 0.18
  // one after the other).  A caller waits for a partner only on a context where concurrent callers have been seen.
  struct MatchRequest {
    dvo_hip_frame* reference; dvo_hip_frame* current; const dvo_hip_config* cfg; dvo_hip_result* result;
    dvo_hip_level_stats* levels; int cap_levels; dvo_hip_iteration_stats* iters; int cap_iters;
    std::atomic<int> state{0};       // 0: waiting for a partner, 1: taken by one, 2: done (rc valid)
    int rc = 0;
  };
  std::mutex rendezvous_mutex;
  MatchRequest* rendezvous_waiting = nullptr;
  dvo_hip_frame* rendezvous_busy_current = nullptr;   // the current frame of the single-pair match that is running right now (null: none)
  int rendezvous_expect = 0;          // > 0: a lone caller waits (briefly) for a partner; refreshed whenever two callers met or collided
  int opt_rendezvous = 1;
  long long rendezvous_pairs = 0;     // two-pair batches formed (counter "rendezvous_pairs")
  long long f16_range_repeats = 0; // batches repeated with the f32 Gram because a Jacobian left the f16 range (counter "f16_range_repeats")
  long long strip_ingests = 0;     // frames ingested by the strip kernel (ingest_strips.hip), counter "strip_ingests"
  // Option "defer_ingest": a batched re-ingest (dvo_hip_frames_update_raw_device_as) is only recorded, and carried out by the next
  // dvo_hip_match_batch right behind the first launches of its first level (or by whatever entry point comes first).  A streaming
  // caller re-ingests the next batch and then aligns the current one: enqueueing the ingest first keeps the alignment's stream idle
  // for the ~0.1 ms (128 pairs) to ~0.5 ms (1024 pairs) of host time it takes -- this way the host does that work while the device
  // is already on the coarsest level.
  struct DeferredIngest {
    std::vector<dvo_hip_frame*> frames;
    std::vector<const void*> grey, raw;
    float depth_scale;
    int role;
    dvo_hip_config cfg;
    bool keep_raw_copy;
  };
  std::vector<DeferredIngest> deferred;
  int opt_defer_ingest = 0;
  int opt_defer_ingest_pixels = 0; // a recorded ingest waits until the launch chain is on a level of at least this many pixels (0: the chain's first launches)
  int opt_keep_raw_copy = 1;       // 0: a frame ingested straight into the reference role keeps no copy of its raw planes (option "keep_raw_copy")
  long long deferred_ingests = 0;  // ingests carried out behind the first launches of a match (counter "deferred_ingests")
  int opt_build_workgroups = 0;    // cap on the workgroups of a build-stream kernel (0 = one per tile): background builds
  int opt_tail_speculation = 0;    // 1: always enqueue the step ahead of the poll, also on the tail of a level whose empty step is costly (measurement)
  int opt_solver_waves = 0;        // wavefronts of a solver-step workgroup: 0 = by level and batch size, 2, 4
  int opt_ll_blocks = 0;           // workgroups per pair of the log-likelihood pass (0 = by batch size)
  int opt_compact_residuals = 1;   // the contracted window sweep stores only the residual pairs of constraints, packed (LevelGeom::compact)
  int opt_gram_lo_parts = 1;       // 1 (default since round 6): every Gram operand keeps its f16 low part on every level; 0: LevelGeom::gram_hi_j on the large ones
  int opt_min_workgroups = 0;      // tile-height heuristic: smallest launch that still counts as filling the chip
  int opt_condition_number = 0;    // results carry |lambda_max / lambda_min| of the information matrix
  int opt_fused_ll_pixels = 0;     // largest level (pixels) whose log-likelihood sweep runs inside the solver workgroup (0 = by batch size)
  float last_sel_ithr = 0.0f, last_sel_dthr = 0.0f;   // selection thresholds of the last match on this context
  int f32_gram_hold = 0;           // batches that still run with the f32 Gram after one left the f16 range (run_batch)
  long long warmup_wait_us = 0;    // longest of the waits dvo_hip_context_create made on the context's streams
  int opt_variant = 8;             // schedule of the sweep: 8 = current-frame window staged in LDS, contracted arithmetic + f16 hi/lo Gram on the matrix pipe where the level allows (7: residuals bit-identical to the oracle's)
                                   // (width a multiple of 64), else 5 = gathering sweep with the f32 Gram on the matrix cores
  // the resident match kernel (align_resident.hip): -1 = levels whose sweep is short enough for the groups that fit (default),
  // 0 = never (launches per iteration only), 1 = every level
  int opt_resident = -1;
  int opt_resident_rows = 0;       // "short enough": segments per wavefront and iteration at most (0 = kResidentRowsDefault)
  int opt_resident_group = 0;      // workgroups per pair (0 = as many as fit, up to kResidentMaxGroup)
  int resident_timeouts = 0;       // batches that had to be repeated because a group timed out (see run_batch)
  long long resident_launches = 0;
  long long resident_levels = 0;   // pyramid levels those launches ran (counter "resident_levels")
  // where the host thread's time of dvo_hip_match_batch goes (ns, accumulated): before the first launch of the batch, enqueueing,
  // waiting for the device, after the device is done
  long long host_ns[4] = {0, 0, 0, 0};
  long long host_batches = 0;
  std::chrono::steady_clock::time_point batch_entry;
  int opt_resident_flags = 0;      // kResidentFlag* (measurement and test hooks)
  // 1: on the levels whose log-likelihood pass fits the solver step, the step runs in the sweep's launch (solver_step.h).  Default 0:
  // measured slower in round 6 (profiles/r06_sweep_tail.txt) -- the step's serial float64 lane needs ~180 registers, the sweep it rides
  // in is built for 96 (five workgroups per compute unit), and the spilled step takes 40-50 us instead of 15
  int opt_sweep_tail = 0;
  long long tail_steps = 0;        // Gauss-Newton steps enqueued as ONE launch (sweep with a tail)
  // The overlapped tail of a level (round 6, run_batch): once at most opt_overlap_fraction-th of a large batch's pairs is still on a level, the
  // others begin the next level and the stragglers finish theirs beside it, on a stream of their own -- each pair leaves its level
  // independently, like the reference's match() calls do (dense_tracking.cpp:357).  0: off; 1: on (levels whose log-likelihood pass
  // runs inside the solver step, batches beyond the solver steps' hand-over).
  int opt_overlap_tails = 0;
  int opt_overlap_fraction = 8;
  // Active-pair lists (round 6): where an empty step is expensive (kCostlyEmptyStepWorkgroups) and at most 1/8 of the pairs is left on the
  // level, the steps that follow are launched over a LIST of those pairs -- tiles x active workgroups instead of tiles x pairs that
  // all but a few leave at once.  1: on.
  int opt_tail_lists = 0;
  long long listed_steps = 0;      // steps launched over an active-pair list (counter "listed_steps")
  long long overlapped_tails = 0;  // levels whose tail ran beside the next level (counter "overlapped_tails")
  long long overlapped_steps = 0;  // Gauss-Newton steps enqueued on the tail stream (counter "overlapped_steps")
  long long tail_drains = 0;       // batches that ended with a slow lane (counter "tail_drains") ...
  long long tail_wait_ns = 0;      // ... and how long the host waited for it behind the chain's last step (counter "tail_wait_us")
  // Concurrent pair groups (round 6): a large batch is aligned as two or three sub-batches at once -- the caller's thread runs the first
  // on this context, helper threads the others on TWIN contexts (same device, own stream, own scratch), like the reference spreads
  // independent match() calls over the workers of a tbb::parallel_reduce (dvo_slam/src/keyframe_graph.cpp:576-593).  Option
  // "batch_groups": 0 / 1 = one group (the default: see batch_groups_of for what was measured), 2 .. 4 = that many.
  int opt_batch_groups = 0;
  bool is_twin = false;            // a helper's context: never splits, never owns frames
  std::vector<GroupWorker*> group_workers;
  hipEvent_t roles_ready = nullptr;   // recorded behind ensure_batch_roles: the twins' streams wait for it
  long long grouped_batches = 0;
  // 1: levels small enough for LDS run align_small.hip under the default schedule; 0 (default): the gathering sweep.  Measured level: an
  // 80 x 60 launch of 1024 pairs 45-50 us against 46-48, of 128 pairs 10.2-11.3 against 9.0-9.5 -- three 53-KB workgroups per compute unit
  // hide a row's latencies no better than the gathering sweep's eight wavefronts per SIMD hide its taps' 
  int opt_small_sweep = 0;
  int opt_small_tiles = 0;         // its workgroups per pair (0 = BatchPolicy::small_level_tiles)
  int opt_coarse = 0;              // the fused coarse-level kernel (align_coarse.hip): 0 = off (default: measured and lost), 1 = whenever the levels admit it
  int opt_coarse_pixels = 0;       // levels of up to this many pixels run in it (0 = kCoarseMaxPixels)
  int opt_coarse_wgs = 0;          // its workgroups per compute unit: 0 / 4 (128 registers) or 3 (168)
  long long coarse_launches = 0, coarse_levels = 0;
  int opt_deterministic = 0;       // a pair's record does not depend on the batch it is aligned in (see dvo_hip.h, option "deterministic")
  int opt_ref_compat = 0;          // projection and weights multiply with the HOST CPU's _mm_rcp_ps like the reference does
  DevBuf rcp_table;                // ... from this table, dumped from the instruction itself when the option is first switched on
  int rcp_shift = 0;
  int rcp_packed = 0;              // a 16-bit copy of the table lies behind it (LevelGeom::rcp_packed)
  int opt_resident_cooperative = 0; // launch groups through hipLaunchCooperativeKernel (a separate hardware queue: +0.1 ms per launch)
  int compute_units = 0;
  std::vector<CameraGeom*> cameras;
  // Device blocks of destroyed frames, kept for the next frame of the same size: a tracking loop creates and destroys one frame per
  // image, and hipMalloc + hipFree (which waits for the device) cost more than building the frame (0.36 vs 0.13 ms at 640x480).
  // Reuse is safe without waiting: what still reads a destroyed frame can only be build-stream work queued before the next
  // frame's build (same stream, in order); main-stream readers are matches, and those have returned.
  struct PooledBlock { void* p; size_t bytes; };
  std::vector<PooledBlock> frame_pool;
  size_t frame_pool_bytes = 0;
  static constexpr size_t kFramePoolMaxBytes = size_t(1) << 30;
  static constexpr size_t kFramePoolMaxBlocks = 64;
  Workspace ws[1];
  DevBuf misc, role_tbl_cur, role_tbl_ref, prep_tbl_cur, prep_tbl_ref;
  static const int kTableSlots = 4;
  DevBuf build_tbl[kTableSlots];   // (a few, picked by the list's first frame: see Workspace::pair_ptrs)
  DevBuf* build_tbl_cur = nullptr; // the one that holds the table of build_tbl_frames
  unsigned build_tbl_next = 0;
  PinnedRing tables;
  // build_tbl holds, in build-stream order, the table of exactly these frames (frames_build): the per-level launches of an
  // eager prepare of the same list reuse it instead of uploading the same bytes again
  std::vector<dvo_hip_frame*> build_tbl_frames;
  // Frame construction (ingest, pyramid, eagerly prepared role planes) runs on its own stream so that the next batch of
  // frames can be built while the current batch is being aligned: the build is bandwidth-bound, the coarse pyramid levels
  // of an alignment are latency-bound, and the two overlap.  Every build call takes a ticket and records an event; an
  // alignment makes the main stream wait for the newest ticket among ITS frames only.
  hipStream_t build_stream = nullptr;
  static const int kBuildRing = 16;
  hipEvent_t build_events[kBuildRing] = {};
  unsigned long long build_seq = 0;          // last ticket issued
  unsigned long long main_waited_seq = 0;    // newest ticket the main stream already waits behind
  // Host -> device transfers of raw planes (dvo_hip_frames_update_raw) have a stream of their own, so that the DMA of batch
  // k+2 runs while batch k+1 is being built and batch k aligned.
  // The planes land in one of kUploadRing contiguous device buffers (not in the frames' own staging areas), so that host planes
  // that are adjacent in memory move in ONE transfer: a 0.9 MB copy per frame reaches ~30 GB/s, a whole batch per copy the link rate.
  hipStream_t upload_stream = nullptr;
  hipEvent_t upload_done = nullptr;
  static const int kUploadRing = 3;
  DevBuf upload_buf[kUploadRing];
  unsigned long long upload_buf_seq[kUploadRing] = {};   // ticket of the build that reads the buffer's current contents
  unsigned upload_next = 0;
  unsigned long long upload_waited_seq = 0;  // newest build ticket the upload stream already waits behind
};

namespace {
int flush_deferred(dvo_hip_context* ctx);
}

// (every entry point that works on frames or streams begins with this, under the context's lock: nothing overtakes a recorded ingest)
#define DVO_FLUSH_DEFERRED(ctx)                                     \
  do {                                                              \
    if ((ctx) && !(ctx)->deferred.empty()) {                        \
      const int rc_deferred__ = flush_deferred(ctx);                \
      if (rc_deferred__ != DVO_HIP_OK) return rc_deferred__;        \
    }                                                               \
  } while (0)

namespace {

// roctx ranges with the phase names of the reference's own (commented-out) stopwatches inside match()
// (dvo_core/src/dense_tracking.cpp:154-158, 222, 246, 309, 325, 349): "prep" = per-level set-up, "err" = passes 1-4 (the sweep and
// the log-likelihood pass), "linsys" = pass 5 + solve (the solver step); plus "build" for the frame construction.  They bracket
// the ENQUEUE of a phase on the host (rocprofv3 --marker-trace shows them next to the kernel trace); free when no tool listens.
// (marker ranges are a profiling aid: a ROCm installation without rocprofiler-sdk builds the library with WITH_ROCTX=0, Makefile)
struct Range {
#ifdef DVO_WITH_ROCTX
  explicit Range(const char* name) { roctxRangePushA(name); }
  ~Range() { roctxRangePop(); }
#else
  explicit Range(const char*) {}
#endif
};

#define DVO_HIP_TRY(ctx, expr)                                                                   \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                            \
      return DVO_HIP_ERR_HIP;                                                                     \
    }                                                                                             \
  } while (0)

#define DVO_WS_TRY(ws, expr)                                                                     \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (ws).err = std::string(#expr) + ": " + hipGetErrorString(e__);                              \
      return DVO_HIP_ERR_HIP;                                                                     \
    }                                                                                             \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int workspace_create(dvo_hip_context* ctx, int g) {
  Workspace& w = ctx->ws[g];
  if (w.created) return DVO_HIP_OK;
  // the alignment stream outranks the build stream: its short, dependent kernels must not queue behind the wide
  // elementwise kernels of a concurrent frame build
  int prio_least = 0, prio_greatest = 0;
  DVO_HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  DVO_HIP_TRY(ctx, hipStreamCreateWithPriority(&w.stream, hipStreamNonBlocking, prio_greatest));
  w.created = true;
  return DVO_HIP_OK;
}

void workspace_destroy(Workspace& w) {
  if (!w.created) return;
  (void)hipStreamSynchronize(w.stream);
  for (DevBuf& b : w.pair_ptrs) b.release();
  for (DevBuf* b : {&w.states, &w.partials, &w.scratch, &w.ll_partials, &w.lvl_stats, &w.it_stats, &w.results,
                    &w.t_init, &w.counters, &w.exchange, &w.win_fallbacks, &w.pair_sums, &w.tail_partials, &w.tail_scratch, &w.tail_ll, &w.tail_flags, &w.tail_list})
    b->release();
  if (w.tail_stream) {
    (void)hipStreamSynchronize(w.tail_stream);
    (void)hipStreamDestroy(w.tail_stream);
    (void)hipEventDestroy(w.tail_split);
    (void)hipEventDestroy(w.tail_end);
    w.tail_stream = nullptr;
  }
  if (w.host_status) (void)hipHostFree(w.host_status);
  w.host_status = nullptr;
  w.host_status_words = 0;
  if (w.f16_range_flag) (void)hipHostFree(w.f16_range_flag);
  w.f16_range_flag = nullptr;
  w.f16_range_words = 0;
  for (PinnedBuf* b : {&w.direct_results, &w.direct_levels, &w.direct_iters, &w.direct_done}) b->release();
  (void)hipStreamDestroy(w.stream);
  w.created = false;
}

int fail(dvo_hip_context* ctx, int code, const char* msg) {
  if (ctx) ctx->err = msg;
  return code;
}

// ticket + event for work just enqueued on the build stream; stamps the frames it wrote
int stamp_build(dvo_hip_context* ctx, int n, dvo_hip_frame* const* frames) {
  const unsigned long long seq = ++ctx->build_seq;
  DVO_HIP_TRY(ctx, hipEventRecord(ctx->build_events[seq % dvo_hip_context::kBuildRing], ctx->build_stream));
  for (int i = 0; i < n; ++i) frames[i]->built_seq = seq;
  return DVO_HIP_OK;
}

// make `stream` (the main stream, or the upload stream about to overwrite a transfer buffer) wait for build ticket `need`
int wait_for_ticket(dvo_hip_context* ctx, unsigned long long need, bool upload) {
  unsigned long long& waited = upload ? ctx->upload_waited_seq : ctx->main_waited_seq;
  hipStream_t stream = upload ? ctx->upload_stream : ctx->stream;
  if (need <= waited) return DVO_HIP_OK;
  // tickets older than the ring have had their event re-recorded for a newer ticket of the same stream: waiting for the
  // oldest live one still orders us after `need`
  const unsigned long long oldest_live = ctx->build_seq >= dvo_hip_context::kBuildRing ? ctx->build_seq - dvo_hip_context::kBuildRing + 1 : 1;
  const unsigned long long use = need < oldest_live ? oldest_live : need;
  DVO_HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->build_events[use % dvo_hip_context::kBuildRing], 0));
  waited = use;
  return DVO_HIP_OK;
}

// the main stream must not touch these frames before the build-stream work that produced them is done
int wait_for_build(dvo_hip_context* ctx, int n, dvo_hip_frame* const* frames) {
  unsigned long long need = 0;
  for (int i = 0; i < n; ++i)
    if (frames[i] && frames[i]->built_seq > need) need = frames[i]->built_seq;
  return wait_for_ticket(ctx, need, /*upload=*/false);
}

// The error word of a resident launch: one of a ring of words of Workspace::host_status indexed by the launch counter (the per-step
// words start behind the ring).  A workgroup of an EARLIER launch that gives up late -- the host returns from the direct path as soon
// as every pair is done, not when every workgroup has left -- raises its own launch's word, not the one the next batch has just reset.
constexpr int kResidentErrorWords = 8;

const int kLlBlocksPerPair = 32;
const size_t kCostlyEmptyStepWorkgroups = 131072;   // (see run_batch: from here on the step ahead of the poll is held back on a level's tail)
const int kFusedLoglikMaxPixels = 160 * 120;      // any batch
const int kFusedLoglikMaxPixelsBatch = 320 * 240;  // batches of 512 pairs and more (run_batch)

// RgbdCameraPyramid::build (rgbd_image.cpp:283-296) + RgbdCamera ctor template (:186-204)
int get_camera(dvo_hip_context* ctx, int w, int h, const float K[4], int levels, const CameraGeom** out) {
  // one geometry per (size, intrinsics), with the tables of every level the size admits: frames of one camera share it
  // whatever level count each was created with (a batch only needs FirstLevel + 1 levels on each pyramid)
  for (CameraGeom* c : ctx->cameras)
    if (c->w0 == w && c->h0 == h && std::memcmp(c->K0, K, 16) == 0) {
      *out = c;
      return DVO_HIP_OK;
    }
  int all = 1;
  while (all < kMaxLevels && (w >> all) >= 2 && (h >> all) >= 2) ++all;
  levels = all > levels ? all : levels;
  CameraGeom* c = new CameraGeom();
  c->w0 = w; c->h0 = h; c->levels = levels;
  std::memcpy(c->K0, K, 16);
  size_t total = 0;
  for (int l = 0; l < levels; ++l) {
    c->w[l] = l == 0 ? w : c->w[l - 1] / 2;
    c->h[l] = l == 0 ? h : c->h[l - 1] / 2;
    for (int k = 0; k < 4; ++k) c->K[l][k] = l == 0 ? K[k] : c->K[l - 1][k] * 0.5f;   // IntrinsicMatrix::scale(0.5f), Q17
    total += align_up(size_t(c->w[l]) * 4, 256) + align_up(size_t(c->h[l]) * 4, 256);
  }
  hipError_t e = c->tables.reserve(total);
  if (e != hipSuccess) {
    delete c;
    ctx->err = std::string("hipMalloc(camera tables): ") + hipGetErrorString(e);
    return DVO_HIP_ERR_HIP;
  }
  std::vector<float> host;
  char* base = c->tables.as<char>();
  size_t off = 0;
  for (int l = 0; l < levels; ++l) {
    const float fx = c->K[l][0], fy = c->K[l][1], ox = c->K[l][2], oy = c->K[l][3];
    host.resize(size_t(c->w[l]) + c->h[l]);
    for (int x = 0; x < c->w[l]; ++x) host[x] = (float(x) - ox) / fx;            // rgbd_image.cpp:198
    for (int y = 0; y < c->h[l]; ++y) host[c->w[l] + y] = (float(y) - oy) / fy;  // rgbd_image.cpp:199
    c->tx[l] = reinterpret_cast<float*>(base + off);
    off += align_up(size_t(c->w[l]) * 4, 256);
    c->ty[l] = reinterpret_cast<float*>(base + off);
    off += align_up(size_t(c->h[l]) * 4, 256);
    e = hipMemcpy(c->tx[l], host.data(), size_t(c->w[l]) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->ty[l], host.data() + c->w[l], size_t(c->h[l]) * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      c->tables.release();
      delete c;
      ctx->err = std::string("hipMemcpy(camera tables): ") + hipGetErrorString(e);
      return DVO_HIP_ERR_HIP;
    }
  }
  ctx->cameras.push_back(c);
  *out = c;
  return DVO_HIP_OK;
}

// tiles of the sweep kernel for one pair (see LevelGeom::linear)
void level_tiles(int w, int h, int rows_per_wave, bool linear, int* tiles_x, int* tiles_y) {
  const int th = kWavesPerBlock * rows_per_wave;
  if (linear) {
    const int segments = (w * h + kTileW - 1) / kTileW;
    *tiles_x = 1;
    *tiles_y = (segments + th - 1) / th;
  } else {
    *tiles_x = (w + kTileW - 1) / kTileW;
    *tiles_y = (h + th - 1) / th;
  }
}

// the contracted window sweep (align_fast.hip, variants 8 / 9) also takes widths that are no multiple of its 64 columns (160 x 120)
bool level_uses_fast_window(const dvo_hip_context* ctx, int w, int h) {
  return ctx->opt_variant >= 8 && fast_sweep_takes_width(w) && w < 32768 && h < 32768;   // (option "ref_compat" included: the COMPAT instantiations)
}

// the sweep with the whole current level in LDS (align_small.hip): the default schedule's levels that the window sweeps do not take
bool level_uses_small(const dvo_hip_context* ctx, int w, int h);

bool level_is_linear(const dvo_hip_context* ctx, int w) { return ctx->opt_variant >= 5 && w % kTileW != 0 && !level_uses_fast_window(ctx, w, 4); }

// the sweep that stages the current frame's window in LDS (align_window.hip, variants 6 / 7; align_fast.hip) handles this level; its tile is 64 x 16
bool level_uses_window(const dvo_hip_context* ctx, int w, int h) {
  return (ctx->opt_variant >= 6 && w % kTileW == 0 && w < 32768 && h < 32768) || level_uses_fast_window(ctx, w, h);
}

bool level_uses_small(const dvo_hip_context* ctx, int w, int h) {
  return ctx->opt_small_sweep && ctx->opt_variant >= 8 && !ctx->opt_ref_compat && !level_uses_window(ctx, w, h) && level_is_linear(ctx, w) && small_sweep_takes(w, h);
}

LevelGeom make_geom(const dvo_hip_context* ctx, const CameraGeom* cam, int level, int rows_per_wave) {
  LevelGeom g;
  g.w = cam->w[level]; g.h = cam->h[level];
  g.fx = cam->K[level][0]; g.fy = cam->K[level][1]; g.ox = cam->K[level][2]; g.oy = cam->K[level][3];
  g.wi_x = 0.5f * g.fx / 255.0f; g.wi_y = 0.5f * g.fy / 255.0f;
  g.half_wi_x = 0.5f * g.wi_x; g.half_wi_y = 0.5f * g.wi_y; g.half_fx = 0.5f * g.fx; g.half_fy = 0.5f * g.fy;
  g.tx = cam->tx[level]; g.ty = cam->ty[level];
  g.level = level;
  g.pair_list = nullptr;
  g.skip_flags = nullptr;
  g.linear = level_is_linear(ctx, g.w) ? 1 : 0;
  level_tiles(g.w, g.h, rows_per_wave, g.linear != 0, &g.tiles_x, &g.tiles_y);
  g.rcp_table = ctx->opt_ref_compat ? ctx->rcp_table.as<float>() : nullptr;
  g.rcp_shift = ctx->rcp_shift;
  g.rcp_packed = ctx->opt_ref_compat == 1 ? ctx->rcp_packed : 0;   // (2: the table through memory, the path of a table that does not pack)
  // (not under "ref_compat": a run that is compared with the reference's own numbers keeps every low part -- round-5 advisor finding)
  g.gram_hi_j = !ctx->opt_gram_lo_parts && !ctx->opt_deterministic && !ctx->opt_ref_compat && ctx->opt_variant == 8 && size_t(g.w) * g.h >= 150000 ? 1 : 0;
  g.small = level_uses_small(ctx, g.w, g.h) ? 1 : 0;
  g.compact = ctx->opt_compact_residuals && ctx->opt_variant >= 8 && rows_per_wave == 4 && fast_sweep_supports(g) ? 1 : 0;   // (launch_residual_reduce's test)
  return g;
}

// rows of 64 pixels each wavefront sweeps: large tiles amortise the 85-value wave reduction, small tiles
// keep all 256 CUs busy when the batch is small
int pick_rows_per_wave(const dvo_hip_context* ctx, const CameraGeom* cam, int level, int n_pairs) {
  if (level_uses_window(ctx, cam->w[level], cam->h[level])) return 4;
  if (level_uses_small(ctx, cam->w[level], cam->h[level]) && ctx->opt_rows_per_wave == 0) {
    // segments per wavefront such that a pair gets BatchPolicy::small_level_tiles workgroups
    const int tiles = ctx->opt_small_tiles > 0 ? ctx->opt_small_tiles : BatchPolicy(ctx->compute_units).small_level_tiles(n_pairs);
    const int segments = (cam->w[level] * cam->h[level] + kTileW - 1) / kTileW;
    const int rows = (segments + kWavesPerBlock * tiles - 1) / (kWavesPerBlock * tiles);
    return rows < 1 ? 1 : rows;
  }
  if (ctx->opt_deterministic) return ctx->opt_rows_per_wave > 0 ? ctx->opt_rows_per_wave : 4;   // one tile height whatever the batch
  if (ctx->opt_rows_per_wave > 0) return ctx->opt_rows_per_wave;
  // A large batch fills the device whatever the tile: short tiles (2 rows per wavefront: the schedule with the pinned prologue) run the
  // coarse levels' sweeps 6-11 % faster than tall ones since the f16 Gram (scripts/ab_sweep.py: 1024 pairs, 160x120 0.200 -> 0.187 ms,
  // 80x60 0.056 -> 0.050 ms; bench step 14.22 -> 13.89 ms)
  if (BatchPolicy(ctx->compute_units).short_gather_tiles(n_pairs) && ctx->opt_variant >= 7) return 2;
  const int candidates[4] = {8, 4, 2, 1};   // measured (profiles/r01_c_tile_sweep.txt): 8 rows is at or near the optimum on every level
  // The tallest tile that still yields this many workgroups.  Fewer, taller tiles also mean fewer partial rows for the
  // bookkeeping kernel, which matters most when there are few pairs (whole-match timings: profiles/r01_f_tile_heuristic.txt).
  const size_t enough = ctx->opt_min_workgroups > 0 ? size_t(ctx->opt_min_workgroups) : size_t(BatchPolicy(ctx->compute_units).min_workgroups(n_pairs));
  for (int r : candidates) {
    int tx, ty;
    level_tiles(cam->w[level], cam->h[level], r, level_is_linear(ctx, cam->w[level]), &tx, &ty);
    if (size_t(tx) * ty * n_pairs >= enough) return r;
  }
  return 1;
}

// (whatever the schedule variant of the moment: a frame outlives option changes)
static bool width_may_use_window(int w) { return w % kTileW == 0 || fast_sweep_takes_width(w); }
// (... or the small-level sweep: both read plane C)
static bool level_may_read_plane_c(int w, int h) { return width_may_use_window(w) || small_sweep_takes(w, h); }

// device layout of a frame: [raw staging][per level: I Z A B R][sel counts]
int frame_alloc(dvo_hip_context* ctx, int w, int h, const float K[4], int levels, dvo_hip_frame** out, size_t* raw_off) {
  if (!ctx) return DVO_HIP_ERR_INVALID;
  if (w < 4 || h < 4 || levels < 1 || levels > kMaxLevels || (w >> (levels - 1)) < 2 || (h >> (levels - 1)) < 2)
    return fail(ctx, DVO_HIP_ERR_INVALID, "frame_create: bad width/height/levels");
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const CameraGeom* cam = nullptr;
  int rc = get_camera(ctx, w, h, K, levels, &cam);
  if (rc != DVO_HIP_OK) return rc;
  dvo_hip_frame* f = new dvo_hip_frame();
  f->levels = levels;
  f->cam = cam;
  size_t total = align_up(size_t(w) * h * 3, 256);   // u8 grey + u16 depth staging
  *raw_off = 0;
  size_t offs[kMaxLevels][6];
  for (int l = 0; l < levels; ++l) {
    const size_t n = size_t(cam->w[l]) * cam->h[l];
    // (C = {I, Z} of a current frame, the plane the window sweep stages in LDS: levels that sweep can handle)
    const size_t sz[6] = {n * 4, n * 4, n * 16, n * 8, n * 8, level_may_read_plane_c(cam->w[l], cam->h[l]) ? n * 8 : 0};
    for (int k = 0; k < 6; ++k) {
      offs[l][k] = total;
      total += align_up(sz[k], 256);
    }
  }
  const size_t cnt_off = total;
  total += 256;
  hipError_t e = hipSuccess;
  for (size_t k = 0; k < ctx->frame_pool.size(); ++k)
    if (ctx->frame_pool[k].bytes == total) {                 // a block of a destroyed frame of this very layout
      f->pool.p = ctx->frame_pool[k].p;
      f->pool.bytes = total;
      ctx->frame_pool_bytes -= total;
      ctx->frame_pool.erase(ctx->frame_pool.begin() + long(k));
      break;
    }
  if (!f->pool.p) e = f->pool.reserve(total);
  if (e != hipSuccess) {
    delete f;
    ctx->err = std::string("hipMalloc(frame): ") + hipGetErrorString(e);
    return DVO_HIP_ERR_HIP;
  }
  char* base = f->pool.as<char>();
  for (int l = 0; l < levels; ++l) {
    FrameLevel& L = f->lv[l];
    L.w = cam->w[l]; L.h = cam->h[l];
    L.I = reinterpret_cast<float*>(base + offs[l][0]);
    L.Z = reinterpret_cast<float*>(base + offs[l][1]);
    L.A = reinterpret_cast<float4*>(base + offs[l][2]);
    L.B = reinterpret_cast<float2*>(base + offs[l][3]);
    L.R = reinterpret_cast<float2*>(base + offs[l][4]);
    L.C = level_may_read_plane_c(cam->w[l], cam->h[l]) ? reinterpret_cast<float2*>(base + offs[l][5]) : nullptr;
  }
  f->sel_count = reinterpret_cast<int*>(base + cnt_off);
  *out = f;
  return DVO_HIP_OK;
}

void fill_build_ptrs(dvo_hip_frame* f, FrameBuildPtrs& p) {
  p.grey = nullptr;
  p.raw = nullptr;
  p.keep_grey = nullptr;
  p.keep_raw = nullptr;
  for (int l = 0; l < f->levels; ++l) {
    p.I[l] = f->lv[l].I; p.Z[l] = f->lv[l].Z; p.A[l] = f->lv[l].A; p.B[l] = f->lv[l].B; p.R[l] = f->lv[l].R; p.C[l] = f->lv[l].C;
  }
  for (int l = f->levels; l < kMaxLevels; ++l) p.C[l] = nullptr;
  p.sel_count = f->sel_count;
}

// the frame's own staging area: [u16 depth][u8 grey], see frame_alloc
uint16_t* staging_depth(dvo_hip_frame* f) { return f->pool.as<uint16_t>(); }
uint8_t* staging_grey(dvo_hip_frame* f) { return f->pool.as<uint8_t>() + size_t(f->lv[0].w) * f->lv[0].h * 2; }

bool aligned_to(const void* p, size_t a) { return reinterpret_cast<uintptr_t>(p) % a == 0; }

// RgbdImagePyramid::build (rgbd_image.cpp:156-172) for n frames of one camera.  From float planes (grey == null: level 0 is
// already in place): the pyr-down chain, one launch per level for the whole batch; derived planes are built lazily per role
// (ensure_roles), like the reference's buildAccelerationStructure / PointSelection caches.  From raw planes: one fused pass
// (k_build_from_raw) that also writes level 0 in role `role` (-1: not known yet, 0: current, 1: reference with the given
// thresholds) and leaves a copy of the raw planes in the frame unless the current-role planes make it redundant.
// flavours of the current role a consumer of `n_frames` freshly built frames will most likely ask for at `level`: where the window
// sweep handles the level, a batch too large for the resident kernel only ever reads the 8-byte plane C (a third of the bytes to
// write); a small one may run the level resident (taps A + B) or on the launch path (C): both.  Whatever is missing at match time is
// derived then (ensure_roles).
int eager_current_flavor(const dvo_hip_context* ctx, const CameraGeom* cam, int level, int n_frames) {
  // a small level (align_small.hip reads plane C): a batch that may still run this level in the resident kernel (the gathered taps) gets
  // both flavours -- 38 KB and 115 KB per frame at 80 x 60 -- a larger one plane C alone
  if (level_uses_small(ctx, cam->w[level], cam->h[level]))
    return BatchPolicy(ctx->compute_units).resident_first_level_fits(n_frames) ? (kCurAB | kCurC) : kCurC;
  if (!level_uses_window(ctx, cam->w[level], cam->h[level])) return kCurAB;
  // (round 5: from an eighth as many frames as compute units on, plane C alone -- the taps cost three times the bytes to write, and a
  // streaming step of 32 / 48 / 64 pairs that re-ingests them every step is 0.885 / 0.98 / 1.24 -> 0.783 / 0.93 / 1.09 ms without them and
  // with the first level alone resident: plan_resident looks at what the frames hold)
  return BatchPolicy(ctx->compute_units).ingest_skips_taps(n_frames) ? kCurC : (kCurAB | kCurC);
}

int frames_build(dvo_hip_context* ctx, int n, dvo_hip_frame* const* frames, const void* const* grey, const void* const* raw,
                 float depth_scale, int role = -1, float ithr = 0.0f, float dthr = 0.0f, bool keep_raw_copy = true) {
  Range range("build");
  const CameraGeom* cam = frames[0]->cam;
  const int levels = frames[0]->levels;
  std::vector<FrameBuildPtrs> host(n);
  bool wide = cam->w[0] % 4 == 0;
  const int flavor0 = eager_current_flavor(ctx, cam, 0, n);
  for (int i = 0; i < n; ++i) {
    dvo_hip_frame* f = frames[i];
    if (f->cam != cam || f->levels != levels) return fail(ctx, DVO_HIP_ERR_INVALID, "frames of one build batch must share camera and levels");
    fill_build_ptrs(f, host[i]);
    for (int l = 0; l < levels; ++l) {   // new pixels: every cached role plane is stale (PointSelection::setRgbdImagePyramid)
      f->lv[l].cur_have = 0;
      f->lv[l].selected = false;
    }
    f->raw0 = grey != nullptr;
    f->raw_copy = false;
    f->depth_scale = depth_scale;
    if (!grey) continue;
    host[i].grey = static_cast<const uint8_t*>(grey[i]);
    host[i].raw = static_cast<const uint16_t*>(raw[i]);
    const bool in_place = host[i].raw == staging_depth(f) && host[i].grey == staging_grey(f);
    if (in_place) {
      f->raw_copy = true;
    } else if (role < 0 || (role == 1 && keep_raw_copy)) {
      host[i].keep_grey = staging_grey(f);
      host[i].keep_raw = staging_depth(f);
      f->raw_copy = true;
    }
    wide = wide && aligned_to(host[i].grey, 4) && aligned_to(host[i].raw, 8) && aligned_to(staging_grey(f), 4);
    if (role == 0) f->lv[0].cur_have = flavor0;
    if (role == 1) { f->lv[0].selected = true; f->lv[0].ithr = ithr; f->lv[0].dthr = dthr; }
  }
  hipStream_t bs = ctx->build_stream;
  // (one of a few buffers: the one that already holds this very table -- a streaming caller re-ingests the same frame sets from the same
  // planes step after step -- else the next in turn)
  int slot = ctx->tables.holder(ctx->build_tbl, dvo_hip_context::kTableSlots, bs, host.data(), size_t(n) * sizeof(FrameBuildPtrs));
  if (slot < 0) slot = int(ctx->build_tbl_next++ % dvo_hip_context::kTableSlots);
  DevBuf& build_tbl = ctx->build_tbl[slot];
  DVO_HIP_TRY(ctx, build_tbl.reserve(size_t(n) * sizeof(FrameBuildPtrs)));
  DVO_HIP_TRY(ctx, ctx->tables.upload(bs, build_tbl.p, host.data(), size_t(n) * sizeof(FrameBuildPtrs)));
  const FrameBuildPtrs* tbl = build_tbl.as<FrameBuildPtrs>();
  ctx->build_tbl_frames.assign(frames, frames + n);
  ctx->build_tbl_cur = &build_tbl;
  int built = 1;                                       // float ingest: level 0 is already in place
  if (grey) {
    // current frames: the {I, Z} plane of the pyramid levels the window sweep will read comes out of the same pass (no neighbours
    // needed), where the strip ingest runs (ingest_strips.hip)
    built = levels < 4 ? levels : 4;
    int c_levels = 0;
    if (role == 0 && ingest_strips_supports(cam->w[0], wide)) {
      for (int l = 1; l < built; ++l)
        if (eager_current_flavor(ctx, cam, l, n) & kCurC) c_levels |= 1 << l;
      for (int i = 0; i < n; ++i)
        for (int l = 1; l < built; ++l)
          if ((c_levels >> l & 1) && frames[i]->lv[l].C) frames[i]->lv[l].cur_have |= kCurC;
    }
    launch_build_from_raw(bs, tbl, n, depth_scale, cam->w[0], cam->h[0], levels, role, wide, ithr, dthr, ctx->opt_build_workgroups, flavor0, c_levels);
    if (ingest_strips_supports(cam->w[0], wide)) ctx->strip_ingests += n;
  }
  for (int l = built; l < levels; ++l) launch_pyr_down(bs, tbl, n, l, cam->w[l - 1], cam->h[l - 1]);
  DVO_HIP_TRY(ctx, hipGetLastError());
  return stamp_build(ctx, n, frames);
}

// Build the missing role planes of a set of frames for levels [l0, l1]: role 0 = current (flavours `cur_want[level]`, kCurAB | kCurC;
// null: the taps A + B), role 1 = reference (R + selection count for the given thresholds).  One launch per level (and source) for
// all frames that need it.
// `eager`: on the build stream (dvo_hip_frames_prepare), otherwise on the main stream right before the planes are used.
int ensure_roles(dvo_hip_context* ctx, int n, dvo_hip_frame* const* frames, int role, int l0, int l1, float ithr, float dthr, bool eager = false,
                 const int* cur_want = nullptr) {
  const CameraGeom* cam = frames[0]->cam;
  const size_t slice = size_t(n) * sizeof(FrameBuildPtrs);
  DevBuf& table = eager ? (role == 0 ? ctx->prep_tbl_cur : ctx->prep_tbl_ref) : (role == 0 ? ctx->role_tbl_cur : ctx->role_tbl_ref);
  hipStream_t stream = eager ? ctx->build_stream : ctx->stream;
  const int cap = eager ? ctx->opt_build_workgroups : 0;   // planes needed right now are built at full width
  // Every frame is checked before the state of any is touched: a frame ingested straight into a role without a copy of its raw planes
  // (option "keep_raw_copy" 0) has nothing its level 0 could be derived from in another role, and the marks below -- planes "built" --
  // are set while the launches are still being gathered (round-5 advisor finding: the error used to leave earlier frames of the list
  // marked as built without their launches, and a retry aligned against stale planes).
  if (l0 == 0)
    for (int i = 0; i < n; ++i) {
      const dvo_hip_frame* f = frames[i];
      const FrameLevel& L = f->lv[0];
      if (!f->raw0 || L.cur_have != 0 || f->raw_copy) continue;
      const int want0 = role == 0 ? (cur_want ? cur_want[0] : kCurAB) & (L.C ? (kCurAB | kCurC) : kCurAB) : 0;
      const bool need = role == 0 ? want0 != 0 : !(L.selected && L.ithr == ithr && L.dthr == dthr);
      if (need) return fail(ctx, DVO_HIP_ERR_INVALID, "frame has neither sampling planes nor a raw copy at level 0");
    }
  bool launched = false;
  int uploads = 0;                                           // table slices used so far (each launch reads its own)
  auto upload = [&](const std::vector<FrameBuildPtrs>& host, const FrameBuildPtrs** tbl, bool plane_pointers_only = true) -> int {
    if (plane_pointers_only && eager && ctx->build_tbl_cur && int(host.size()) == n && ctx->build_tbl_frames.size() == size_t(n) &&
        std::equal(frames, frames + n, ctx->build_tbl_frames.begin())) {
      *tbl = ctx->build_tbl_cur->as<FrameBuildPtrs>();      // the ingest of these very frames left their table on this stream
      return DVO_HIP_OK;
    }
    constexpr int kSlices = 4 * kMaxLevels;
    if (uploads == kSlices) {                                // (never in practice: a slice per level and source)
      DVO_HIP_TRY(ctx, hipStreamSynchronize(stream));
      uploads = 0;
    }
    DVO_HIP_TRY(ctx, table.reserve(slice * kSlices));
    FrameBuildPtrs* up = reinterpret_cast<FrameBuildPtrs*>(table.as<char>() + slice * uploads++);
    DVO_HIP_TRY(ctx, ctx->tables.upload(stream, up, host.data(), host.size() * sizeof(FrameBuildPtrs)));
    *tbl = up;
    return DVO_HIP_OK;
  };
  // Several levels, every frame missing the same planes on each of them, all to be derived from the float planes I / Z (a camera
  // frame that has just been built): ONE launch for all levels (k_derive_levels) instead of a table upload, a counter reset and a
  // launch per level.
  if (l1 > l0) {
    LevelSpan span;
    span.l0 = l0; span.l1 = l1;
    bool uniform = n <= 64;                                                             // (the check below is quadratic; large batches come through the ingest.
    // Round 6: the size test used to FOLLOW the double loop -- 524 288 comparisons per role of a 1024-pair batch, 0.4 ms of the host
    // thread in front of every streaming step's first launch)
    for (int i = 0; i < n && uniform; ++i)
      for (int j = 0; j < i && uniform; ++j) uniform = frames[i] != frames[j];        // (a frame listed twice: the general path skips its second visit)
    int tiles = 0;
    for (int l = l0; l <= l1 && uniform; ++l) {
      const int want = role == 0 ? (cur_want ? cur_want[l] : kCurAB) & (frames[0]->lv[l].C ? (kCurAB | kCurC) : kCurAB) : 0;
      int miss_all = -1;
      for (int i = 0; i < n && uniform; ++i) {
        dvo_hip_frame* f = frames[i];
        const FrameLevel& L = f->lv[l];
        const int miss = role == 0 ? want & ~L.cur_have : 0;
        const bool need = role == 0 ? miss != 0 : !(L.selected && L.ithr == ithr && L.dthr == dthr);
        uniform = need && !(l == 0 && f->raw0) && (role == 1 || L.cur_have == 0) && (miss_all < 0 || miss == miss_all);
        miss_all = miss;
      }
      span.w[l] = cam->w[l]; span.h[l] = cam->h[l]; span.flavor[l] = miss_all;
      span.tile0[l] = tiles;
      tiles += ((cam->w[l] + 63) / 64) * ((cam->h[l] + 15) / 16);
    }
    span.tile0[l1 + 1] = tiles;
    if (uniform) {
      std::vector<FrameBuildPtrs> host(n);
      for (int i = 0; i < n; ++i) fill_build_ptrs(frames[i], host[i]);
      const FrameBuildPtrs* tbl = nullptr;
      const int rc = upload(host, &tbl);
      if (rc != DVO_HIP_OK) return rc;
      launch_derive_levels(stream, tbl, n, span, role, ithr, dthr, cap);
      for (int i = 0; i < n; ++i)
        for (int l = l0; l <= l1; ++l) {
          FrameLevel& L = frames[i]->lv[l];
          if (role == 0) L.cur_have |= span.flavor[l];
          else { L.selected = true; L.ithr = ithr; L.dthr = dthr; }
        }
      if (eager) {
        const int rc2 = stamp_build(ctx, n, frames);
        if (rc2 != DVO_HIP_OK) return rc2;
      }
      DVO_HIP_TRY(ctx, hipGetLastError());
      return DVO_HIP_OK;
    }
  }
  for (int l = l0; l <= l1; ++l) {
    const int want = role == 0 ? (cur_want ? cur_want[l] : kCurAB) & (frames[0]->lv[l].C ? (kCurAB | kCurC) : kCurAB) : 0;
    // sources, per frame: float planes I / Z (levels >= 1, and level 0 of frames created from float planes); at level 0 of a frame
    // ingested from raw planes: the other flavour of the current role, else the frame's copy of its raw planes
    std::vector<FrameBuildPtrs> from_planes[4], from_raw[4], ab_from_c, c_from_a, ref_from_c;
    std::vector<dvo_hip_frame*> ref_from_ab;
    float raw_scale = 0.0f;
    bool deferred = false;                                   // frames of another depth scale than the launch gathered so far
    for (int i = 0; i < n; ++i) {
      dvo_hip_frame* f = frames[i];
      FrameLevel& L = f->lv[l];
      const int miss = role == 0 ? want & ~L.cur_have : 0;
      const bool need = role == 0 ? miss != 0 : !(L.selected && L.ithr == ithr && L.dthr == dthr);
      if (!need) continue;   // also skips the second visit of a frame that is listed twice
      FrameBuildPtrs p;
      fill_build_ptrs(f, p);
      if (l == 0 && f->raw0) {
        if (role == 0 && (L.cur_have & kCurC)) {
          ab_from_c.push_back(p);
        } else if (role == 0 && (L.cur_have & kCurAB)) {
          c_from_a.push_back(p);
        } else if (role == 1 && (L.cur_have & kCurAB)) {
          ref_from_ab.push_back(f);
        } else if (role == 1 && (L.cur_have & kCurC)) {
          ref_from_c.push_back(p);
        } else if (f->raw_copy && (raw_scale == 0.0f || f->depth_scale == raw_scale)) {
          raw_scale = f->depth_scale;
          p.grey = staging_grey(f);
          p.raw = staging_depth(f);
          from_raw[miss].push_back(p);
        } else if (f->raw_copy) {
          deferred = true;     // another depth scale than the frames gathered so far: picked up by the pass below
          continue;
        } else {
          return fail(ctx, DVO_HIP_ERR_INVALID, "frame has neither sampling planes nor a raw copy at level 0");
        }
      } else {
        from_planes[miss].push_back(p);
      }
      if (role == 0) L.cur_have |= miss;
      else { L.selected = true; L.ithr = ithr; L.dthr = dthr; }
    }
    const FrameBuildPtrs* tbl = nullptr;
    for (int miss = 0; miss < 4; ++miss) {
      if (!from_planes[miss].empty()) {
        int rc = upload(from_planes[miss], &tbl);
        if (rc != DVO_HIP_OK) return rc;
        if (role == 0) launch_derive_current(stream, tbl, int(from_planes[miss].size()), l, cam->w[l], cam->h[l], cap, miss);
        else launch_derive_reference(stream, tbl, int(from_planes[miss].size()), l, cam->w[l], cam->h[l], ithr, dthr, cap);
        launched = true;
      }
      if (!from_raw[miss].empty()) {
        int rc = upload(from_raw[miss], &tbl, /*plane_pointers_only=*/false);
        if (rc != DVO_HIP_OK) return rc;
        launch_build_from_raw(stream, tbl, int(from_raw[miss].size()), raw_scale, cam->w[0], cam->h[0], /*levels=*/1, role, cam->w[0] % 4 == 0, ithr, dthr, cap, miss);
        launched = true;
      }
    }
    const struct { std::vector<FrameBuildPtrs>* list; int mode; } conversions[3] = {{&ab_from_c, 0}, {&c_from_a, 1}, {&ref_from_c, 2}};
    for (const auto& c : conversions) {
      if (c.list->empty()) continue;
      int rc = upload(*c.list, &tbl);
      if (rc != DVO_HIP_OK) return rc;
      launch_from_current_plane(stream, tbl, int(c.list->size()), l, cam->w[l], cam->h[l], c.mode, ithr, dthr, cap);
      launched = true;
    }
    for (dvo_hip_frame* f : ref_from_ab) {   // PointSelection over a frame that has been a current frame so far
      FrameLevel& L = f->lv[0];
      DVO_HIP_TRY(ctx, hipMemsetAsync(f->sel_count, 0, sizeof(int), stream));
      launch_select_pack(stream, L.A, L.B, L.w * L.h, ithr, dthr, L.R, f->sel_count, nullptr);
      launched = true;
    }
    if (deferred) {   // frames of a second depth scale (one kernel launch takes one scale): rare, one more pass each
      if (eager && launched) {
        const int rc = stamp_build(ctx, n, frames);
        if (rc != DVO_HIP_OK) return rc;
      }
      DVO_HIP_TRY(ctx, hipStreamSynchronize(stream));   // the table slices are reused
      return ensure_roles(ctx, n, frames, role, l0, l1, ithr, dthr, eager, cur_want);
    }
  }
  if (eager && launched) {
    const int rc = stamp_build(ctx, n, frames);
    if (rc != DVO_HIP_OK) return rc;
  }
  DVO_HIP_TRY(ctx, hipGetLastError());
  return DVO_HIP_OK;
}

struct BatchPlan {
  int n = 0, nlev = 0, cap_levels = 0, cap_iters = 0;
  SolverParams prm;
  const CameraGeom* cam = nullptr;
  std::vector<int> rpw;          // per absolute level
  std::vector<LevelGeom> geom;   // per absolute level
  PairPtrs* pair_ptrs = nullptr; // device [levels][n] (null when the table only travels in kernel arguments)
  std::vector<PairPtrs> host_ptrs;   // the same table on the host
  int coarse_levels = 0;         // leading levels (first_level, first_level - 1, ...) the fused coarse-level kernel runs (plan_coarse)
};

int validate_batch(dvo_hip_context* ctx, int n, dvo_hip_frame* const* refs, dvo_hip_frame* const* curs, const dvo_hip_config* cfg) {
  if (!ctx || n < 1 || !refs || !curs || !cfg) return fail(ctx, DVO_HIP_ERR_INVALID, "match: null argument");
  if (cfg->first_level < cfg->last_level || cfg->last_level < 0 || cfg->first_level >= kMaxLevels)   // Config::IsSane, DT.cpp:74
    return fail(ctx, DVO_HIP_ERR_INVALID, "match: need 0 <= last_level <= first_level < DVO_HIP_MAX_LEVELS");
  if (cfg->max_iterations_per_level < 1) return fail(ctx, DVO_HIP_ERR_INVALID, "match: max_iterations_per_level < 1");
  const int need_levels = cfg->first_level + 1;   // Config::getNumLevels
  const CameraGeom* cam = refs[0] ? refs[0]->cam : nullptr;
  for (int i = 0; i < n; ++i) {
    if (!refs[i] || !curs[i]) return fail(ctx, DVO_HIP_ERR_INVALID, "match: null frame");
    if (refs[i]->levels < need_levels || curs[i]->levels < need_levels)
      return fail(ctx, DVO_HIP_ERR_INVALID, "match: frame pyramid has fewer levels than first_level + 1");
    if (refs[i]->cam != cam || curs[i]->cam != cam)
      return fail(ctx, DVO_HIP_ERR_INVALID, "match: all frames of a batch must share size and intrinsics");
  }
  return DVO_HIP_OK;
}

void make_plan(const dvo_hip_context* ctx, const CameraGeom* cam, const dvo_hip_config* cfg, int n, BatchPlan& bp) {
  const int need_levels = cfg->first_level + 1;
  bp.n = n;
  bp.cam = cam;
  bp.nlev = cfg->first_level - cfg->last_level + 1;
  bp.cap_levels = bp.nlev;
  bp.cap_iters = bp.nlev * cfg->max_iterations_per_level;
  bp.prm.max_iterations = cfg->max_iterations_per_level;
  bp.prm.first_level = cfg->first_level;
  bp.prm.last_level = cfg->last_level;
  bp.prm.use_initial_estimate = cfg->use_initial_estimate;
  bp.prm.precision = cfg->precision;
  bp.prm.mu = cfg->mu;
  bp.prm.cap_iters = bp.cap_iters;
  bp.prm.cap_levels = bp.cap_levels;
  bp.prm.max_points_level0 = cam->w0 * cam->h0;
  bp.prm.want_condition_number = ctx->opt_condition_number;
  bp.prm.record_prefilled = 0;
  bp.rpw.assign(need_levels, 1);
  bp.geom.resize(need_levels);
  for (int l = cfg->last_level; l <= cfg->first_level; ++l) {
    bp.rpw[l] = pick_rows_per_wave(ctx, cam, l, n);
    bp.geom[l] = make_geom(ctx, cam, l, bp.rpw[l]);
  }
}

// buildAccelerationStructure for the current frames, PointSelection::select for the reference frames (both cached per
// frame and level); enqueued on the context's main stream
// `launch_path_only`: the caller runs every level on the launch-per-step path (the parity / measurement entry points)
int resident_levels_of(const dvo_hip_context* ctx, const dvo_hip_config* cfg, const CameraGeom* cam, int n, bool taps_missing);
bool window_taps_missing(const dvo_hip_context* ctx, const dvo_hip_config* cfg, int n, dvo_hip_frame* const* curs);

int ensure_batch_roles(dvo_hip_context* ctx, int n, dvo_hip_frame* const* refs, dvo_hip_frame* const* curs, const dvo_hip_config* cfg,
                       bool launch_path_only = false) {
  Range range("build");
  ctx->last_sel_ithr = cfg->intensity_derivative_threshold;   // (what a speculative reference preparation will assume, prepare_roles)
  ctx->last_sel_dthr = cfg->depth_derivative_threshold;
  int rc = wait_for_build(ctx, n, refs);
  if (rc == DVO_HIP_OK) rc = wait_for_build(ctx, n, curs);
  if (rc != DVO_HIP_OK) return rc;
  // the flavour of the current role each level is read in: the resident kernel and the gathering sweep read the taps A + B, the
  // window sweep the 8-byte plane C
  const CameraGeom* cam = curs[0]->cam;
  const int resident = launch_path_only ? 0 : resident_levels_of(ctx, cfg, cam, n, window_taps_missing(ctx, cfg, n, curs));
  int want[kMaxLevels];
  for (int l = 0; l < kMaxLevels; ++l)
    want[l] = l > cfg->first_level - resident || l >= cam->levels || !(level_uses_window(ctx, cam->w[l], cam->h[l]) || level_uses_small(ctx, cam->w[l], cam->h[l])) ? kCurAB : kCurC;
  rc = ensure_roles(ctx, n, curs, 0, cfg->last_level, cfg->first_level, 0.0f, 0.0f, /*eager=*/false, want);
  if (rc == DVO_HIP_OK)
    rc = ensure_roles(ctx, n, refs, 1, cfg->last_level, cfg->first_level, cfg->intensity_derivative_threshold, cfg->depth_derivative_threshold);
  return rc;
}

// Device scratch for n pairs + the per-level pointer tables
int prepare_buffers(Workspace& w, const dvo_hip_config* cfg, dvo_hip_frame* const* refs, dvo_hip_frame* const* curs, BatchPlan& bp,
                    bool upload_table = true) {
  const int n = bp.n, need_levels = cfg->first_level + 1;
  size_t max_tiles = 1;
  for (int l = cfg->last_level; l <= cfg->first_level; ++l) max_tiles = std::max(max_tiles, size_t(bp.geom[l].tiles_x) * bp.geom[l].tiles_y);
  size_t npx = 0;                                             // residual entries per pair: the largest level's (packed ones own whole tiles)
  for (int l = cfg->last_level; l <= cfg->first_level; ++l) npx = std::max(npx, residual_entries(bp.geom[l]));
  DVO_WS_TRY(w, w.states.reserve(size_t(n) * sizeof(PairState)));

  DVO_WS_TRY(w, w.partials.reserve(size_t(n) * max_tiles * kAccStride * sizeof(float)));
  DVO_WS_TRY(w, w.scratch.reserve(size_t(n) * npx * sizeof(float2)));
  DVO_WS_TRY(w, w.ll_partials.reserve(size_t(n) * kLlBlocksPerPair * sizeof(double)));
  DVO_WS_TRY(w, w.lvl_stats.reserve(size_t(n) * bp.cap_levels * sizeof(dvo_hip_level_stats)));
  DVO_WS_TRY(w, w.it_stats.reserve(size_t(n) * bp.cap_iters * sizeof(dvo_hip_iteration_stats)));
  DVO_WS_TRY(w, w.results.reserve(size_t(n) * sizeof(dvo_hip_result)));
  DVO_WS_TRY(w, w.pair_sums.reserve(size_t(n) * kPairSumsStride * sizeof(double)));
  DVO_WS_TRY(w, w.t_init.reserve(size_t(n) * 16 * sizeof(double)));
  // per-step tallies, and behind them one arrival word per pair (the sweeps' tail, solver_step.h): cleared together at the start of a batch
  DVO_WS_TRY(w, w.counters.reserve(size_t(bp.cap_iters + 8 + kResidentErrorWords) * sizeof(unsigned long long) + align_up(size_t(n) * sizeof(int), 8)));
  if (!w.win_fallbacks.p) {
    DVO_WS_TRY(w, w.win_fallbacks.reserve(64));
    DVO_WS_TRY(w, hipMemsetAsync(w.win_fallbacks.p, 0, 64, w.stream));
  }
  if (w.f16_range_words < size_t(n)) {
    // (the words are only ever written by sweeps of a batch that is over by the time the next one is prepared: a larger array can
    // replace the old one here once the stream has drained)
    if (w.f16_range_flag) {
      DVO_WS_TRY(w, hipStreamSynchronize(w.stream));
      (void)hipHostFree(w.f16_range_flag);
      w.f16_range_flag = nullptr;
      w.f16_range_words = 0;
    }
    const size_t words = std::max<size_t>(64, size_t(n));
    DVO_WS_TRY(w, hipHostMalloc(reinterpret_cast<void**>(&w.f16_range_flag), words * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(w.f16_range_flag, 0, words * sizeof(int));
    w.f16_range_words = words;
  }
  std::vector<PairPtrs>& host = bp.host_ptrs;
  host.assign(size_t(n) * need_levels, PairPtrs());
  for (int l = cfg->last_level; l <= cfg->first_level; ++l)
    for (int i = 0; i < n; ++i) {
      PairPtrs& p = host[size_t(l) * n + i];
      p.refR = refs[i]->lv[l].R;
      p.curA = curs[i]->lv[l].A;
      p.curB = curs[i]->lv[l].B;
      p.n_selected = refs[i]->sel_count + l;
      p.curC = curs[i]->lv[l].C;
    }
  bp.pair_ptrs = nullptr;
  if (upload_table) {
    int slot = w.tables->holder(w.pair_ptrs, Workspace::kTableSlots, w.stream, host.data(), host.size() * sizeof(PairPtrs));
    if (slot < 0) slot = int(w.pair_ptrs_next++ % Workspace::kTableSlots);
    DevBuf& pair_ptrs = w.pair_ptrs[slot];
    DVO_WS_TRY(w, pair_ptrs.reserve(size_t(n) * need_levels * sizeof(PairPtrs)));
    DVO_WS_TRY(w, w.tables->upload(w.stream, pair_ptrs.p, host.data(), host.size() * sizeof(PairPtrs)));
    bp.pair_ptrs = pair_ptrs.as<PairPtrs>();
  }
  return DVO_HIP_OK;
}

// Spin on the pinned status word the device writes when the last workgroup of a step is through (publish_step in
// solver_kernels.hip).  A host-memory poll sees the word ~2 us after the store; an event synchronisation took ~10 us.
int step_wait_check(Workspace& w, std::chrono::steady_clock::time_point t0) {
  const hipError_t q = hipStreamQuery(w.stream);
  if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();   // "still running" is not an error to keep
  if (q != hipSuccess && q != hipErrorNotReady) {
    w.err = std::string("match: stream failed while waiting for a Gauss-Newton step: ") + hipGetErrorString(q);
    return DVO_HIP_ERR_HIP;
  }
  if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
    w.err = "match: timed out waiting for a Gauss-Newton step";
    return DVO_HIP_ERR_HIP;
  }
  return DVO_HIP_OK;
}

int wait_for_step(Workspace& w, int step, int* active) {
  volatile int* word = w.host_status + step;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    const int v = *word;
    if (v & kStepDoneFlag) {
      *active = v & ~kStepDoneFlag;
      return DVO_HIP_OK;
    }
    if ((spins & 0xfffff) == 0) {                            // every ~1M polls: has the stream died, or are we stuck?
      const int rc = step_wait_check(w, t0);
      if (rc != DVO_HIP_OK) return rc;
    }
  }
}

// The direct path of the resident kernel: spin on the pinned word in which the kernel counts the pairs whose results (and statistics)
// are complete in pinned host memory; a group that timed out raises the error word instead (the caller repeats the batch).
int wait_for_direct(Workspace& w, int n_pairs) {
  volatile int* done = w.direct_done.as<int>();
  volatile int* error_word = w.host_status + w.resident_error_word;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if (*error_word != 0) return DVO_HIP_OK;                 // (looked at first: an error raised in this launch is reported by this launch)
    if (*done >= n_pairs) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return DVO_HIP_OK;
    }
    if ((spins & 0xfffff) == 0) {
      const hipError_t q = hipStreamQuery(w.stream);
      if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
      if (q == hipSuccess && *done < n_pairs && *error_word == 0) {   // the kernel is gone and has not reported: it never ran
        w.err = "match: the resident kernel ended without reporting its pairs";
        return DVO_HIP_ERR_HIP;
      }
      if (q != hipSuccess && q != hipErrorNotReady) {
        w.err = std::string("match: stream failed while the resident kernel ran: ") + hipGetErrorString(q);
        return DVO_HIP_ERR_HIP;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
        w.err = "match: timed out waiting for the resident kernel";
        return DVO_HIP_ERR_HIP;
      }
    }
  }
}

// pinned status words the device writes (publish_step): coherent + mapped, so that a system-scope store is visible to the
// polling host thread whatever HIP_HOST_COHERENT says
int ensure_host_status(Workspace& w, size_t n_steps) {
  if (w.host_status_words < n_steps) {
    if (w.host_status) (void)hipHostFree(w.host_status);
    w.host_status = nullptr;
    w.host_status_words = 0;
    DVO_WS_TRY(w, hipHostMalloc(reinterpret_cast<void**>(&w.host_status), n_steps * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    w.host_status_words = n_steps;
  }
  return DVO_HIP_OK;
}

// ---- the resident kernel: which levels, how many workgroups per pair ------------------------------------------------------------
// Workgroups of a group wait for each other, so all groups in flight on a device must be on it together.  Contexts of one process
// (one per host thread, like the reference's one DenseTracker per TBB worker, keyframe_graph.cpp:576-593) share a per-device budget
// of compute units: launches with groups run SIDE BY SIDE as long as their workgroups fit the device together (one 512-thread
// workgroup per compute unit), and wait for each other beyond that.  (Round 2 let only one such launch run at a time: side by side,
// one launch in seven had timed out.  The cause was the exchange's missing flow control towards idle workgroups -- fixed in
// align_resident.hip, "Flow control" -- which a second launch on the device, delaying some workgroups' start, merely exposed.)
// Another PROCESS on the same device is not seen here; there a group can time out, and the batch is repeated on the launch path.
struct ResidentBudget {
  std::mutex m;
  std::condition_variable cv;
  int in_flight = 0;
  static ResidentBudget& of(int device) {
    static ResidentBudget budgets[64];
    return budgets[device >= 0 && device < 64 ? device : 63];
  }
  struct Hold {
    ResidentBudget* b = nullptr;
    int n = 0;
    Hold() = default;
    Hold(const Hold&) = delete;
    Hold& operator=(const Hold&) = delete;
    void take(ResidentBudget& budget, int workgroups, int capacity) {
      std::unique_lock<std::mutex> lock(budget.m);
      budget.cv.wait(lock, [&] { return budget.in_flight == 0 || budget.in_flight + workgroups <= capacity; });
      budget.in_flight += workgroups;
      b = &budget;
      n = workgroups;
    }
    ~Hold() {
      if (!b) return;
      {
        std::lock_guard<std::mutex> lock(b->m);
        b->in_flight -= n;
      }
      b->cv.notify_all();
    }
  };
};

constexpr int kResidentRowsDefault = 24;      // segments per wavefront and iteration up to which a level runs resident


struct ResidentPlan {
  int group = 1;                               // workgroups per pair
  int levels = 0;                              // leading levels (first_level, first_level - 1, ...) that run resident
  bool direct = false;                         // the whole match of a small batch: results and statistics land in pinned host memory
};

constexpr size_t kResidentDirectStatsBytes = size_t(64) << 20;

// `taps_missing`: current frames of the batch hold plane C but not the taps A + B on the level below the first one (they were ingested
// straight into their role in a batch of an eighth as many frames as compute units or more, eager_current_flavor): the resident kernel
// would have to have the taps derived for it first
bool window_taps_missing(const dvo_hip_context* ctx, const dvo_hip_config* cfg, int n, dvo_hip_frame* const* curs) {
  const int level = cfg->first_level - 1;
  if (level < cfg->last_level || level >= curs[0]->levels || !level_uses_window(ctx, curs[0]->cam->w[level], curs[0]->cam->h[level])) return false;
  for (int i = 0; i < n; ++i) {
    const int have = curs[i]->lv[level].cur_have;
    if ((have & kCurC) && !(have & kCurAB)) return true;
  }
  return false;
}

ResidentPlan plan_resident(const dvo_hip_context* ctx, const dvo_hip_config* cfg, const BatchPlan& bp, bool taps_missing) {
  ResidentPlan rp;
  if (ctx->opt_resident == 0 || ctx->opt_deterministic) return rp;   // (deterministic: one path whatever the batch size, and that is the launch path)
  const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
  // more pairs than compute units: the workgroups would run in shifts, and the launch path, which gives every phase the whole chip,
  // is as fast (measured: 256 pairs -3 %, 512 pairs +1.6 % against it).  (Not with a pinned group size: the caller asks for
  // records that do not depend on the batch size, so the choice of path must not either.)
  // (round 3: with the f16 Gram and the short tiles the launch path is level with or ahead of ONE workgroup per pair as well -- 192 / 256
  // pairs: screening stage 0.278 / 0.306 vs 0.293 / 0.341 ms, levels 3 -> 1 0.82 / 0.95 vs 0.87 / 1.01, full match equal; the resident
  // kernel is kept for batches that get at least two workgroups per pair -- with launches limited to half the chip that is
  // pairs <= compute units / 4.  Measured at the end of round 3, 96 / 128 pairs with ONE workgroup each: full match 1.63 / 1.93 ms
  // resident against 1.61 / 1.90 on the launch path, levels 3 -> 1 0.60 / 0.67 against 0.57 / 0.65, and a streaming step of 128 pairs
  // beside its background ingest 2.50 against 2.41 ms; at 64 pairs (two workgroups each) resident still wins, 1.19 against 1.26.)
  // (round 5: up to 7/16 as many pairs as compute units the FIRST level alone -- 18 of a streaming step's 34 iterations in BASELINE
  // config 4, each a sweep and a solver launch of 11 + 12 us -- is still quicker resident with two workgroups per pair, which leave
  // an eighth of the chip to the background ingest: streaming step of 72 / 80 / 96 / 112 pairs 1.215 / 1.283 / 1.440 / 1.634 ->
  // 1.172 / 1.232 / 1.378 / 1.587 ms.  128 pairs: two workgroups each 1.808 against 1.776, one 1.748 on one box and level on two
  // others -- left on the launch path; with the second level resident as well 1.59 at 96 pairs.  Only a level the launch path
  // reads through the taps too: its planes exist.)
  // (the same from an eighth of the compute units on when the batch came from a streaming ingest that left the taps out -- see
  // eager_current_flavor; frames that hold them, or nothing yet, get the coarse levels resident as before: a match of 64 prepared pairs
  // 0.96 ms against 1.02 with the first level alone)
  const BatchPolicy policy(ctx->compute_units);
  const bool first_level_only = ctx->opt_resident < 0 && ctx->opt_resident_group == 0 &&
                                (!policy.resident_takes_coarse_levels(bp.n) || (policy.taps_missing_prefers_first_level_only(bp.n) && taps_missing));
  if (first_level_only && (!policy.resident_first_level_fits(bp.n) || level_uses_window(ctx, bp.cam->w[cfg->first_level], bp.cam->h[cfg->first_level]))) return rp;
  // all workgroups of a launch with groups must be resident at once: one workgroup (8 wavefronts, up to 256 registers) per compute unit
  int group = 1;
  while (group * 2 <= kResidentMaxGroup && bp.n * group * 2 <= cus) group *= 2;
  if (ctx->opt_resident_group > 0) group = std::min(group, ctx->opt_resident_group);
  if (first_level_only) {                                       // the groups leave an eighth of the chip free
    group = 1;
    while (group * 2 <= kResidentMaxGroup && bp.n * group * 2 * 8 <= cus * 7) group *= 2;
  }
  rp.group = group;
  const int rows_max = ctx->opt_resident_rows > 0 ? ctx->opt_resident_rows : kResidentRowsDefault;
  for (int level = cfg->first_level; level >= cfg->last_level; --level) {
    const int segments = (bp.cam->w[level] * bp.cam->h[level] + kTileW - 1) / kTileW;
    const int rows = (segments + group * kResidentSweepers - 1) / (group * kResidentSweepers);
    if (ctx->opt_resident != 1 && rows > rows_max) break;
    if (first_level_only && rp.levels == 1) break;
    if (size_t(bp.cam->w[level]) * bp.cam->h[level] >= (size_t(1) << 24)) break;   // the kernel locates a pixel with one float multiply
    rp.levels += 1;
  }
  static const bool trace_plan = std::getenv("DVO_HIP_TRACE_PLAN") != nullptr;     // (stderr: which plan a batch got)
  if (trace_plan)
    std::fprintf(stderr, "plan_resident: %d pairs, taps missing %d -> %s, %d workgroup(s) per pair, %d level(s) resident\n", bp.n, int(taps_missing),
                 first_level_only ? "first level only" : "coarse levels", rp.group, rp.levels);
  rp.direct = rp.levels == cfg->first_level - cfg->last_level + 1 && bp.n <= policy.resident_direct_max_pairs() &&
              size_t(bp.n) * bp.cap_iters * sizeof(dvo_hip_iteration_stats) <= kResidentDirectStatsBytes;   // (pinned, if asked for)
  return rp;
}

// ---- the fused coarse-level kernel (align_coarse.hip): which levels -----------------------------------------------------------------
// One workgroup per pair runs the leading levels of the match to their termination in ONE launch (sweep over all tiles, reduction,
// log-likelihood, loop body), with the launch path's device functions on the launch path's data layout: the records are the launch
// path's bit for bit.  The levels: from the first one down, as long as the level is small (kCoarseMaxPixels) and the kernel has an
// instantiation for it (coarse_kernel_takes).  OPT-IN (option "coarse" 1): measured in round 6 against the launch path at 128 / 256 /
// 512 / 1024 pairs per step and slower at every size (1.80 -> 3.08, 3.16 -> 4.43, 6.0 -> 7.2, 11.5 -> 12.5 ms per streaming step): a
// workgroup walks the 10 + 34 tiles of levels 3 and 2 one after the other -- 60 / 200 us per iteration where the launch path, which
// spreads the tiles of an iteration over the chip, needs 25 / 40 -- and with as many pairs as workgroup slots (1024 = 256 compute units x
// 4) the launch lasts as long as its slowest pair (15 + 9 iterations against a mean of 8 + 5).  It pays where pairs outnumber the slots
// several times over; kept for that case and as the bit-exact cross-check of the launch path's hand-over logic.
constexpr int kCoarseMaxPixels = 160 * 120;

void plan_coarse(const dvo_hip_context* ctx, const dvo_hip_config* cfg, BatchPlan& bp, const ResidentPlan& rp) {
  bp.coarse_levels = 0;
  if (ctx->opt_coarse == 0 || rp.levels > 0 || ctx->opt_variant != 8) return;
  const int max_pixels = ctx->opt_coarse_pixels > 0 ? ctx->opt_coarse_pixels : kCoarseMaxPixels;
  for (int level = cfg->first_level; level >= cfg->last_level; --level) {
    if (bp.cam->w[level] * bp.cam->h[level] > max_pixels) break;
    const bool window_level = level_uses_window(ctx, bp.cam->w[level], bp.cam->h[level]);
    const int rpw = window_level ? 4 : kCoarseRowsPerWave;
    if (!window_level && ctx->opt_rows_per_wave > 0 && ctx->opt_rows_per_wave != rpw) break;   // (a tile height asked for by name stays)
    const LevelGeom g = make_geom(ctx, bp.cam, level, rpw);
    if (!coarse_kernel_takes(g, window_level)) break;
    bp.rpw[level] = rpw;
    bp.geom[level] = g;
    bp.coarse_levels += 1;
  }
}

// which of the two one-launch kernels takes the leading levels of a batch, if any
ResidentPlan plan_paths(const dvo_hip_context* ctx, const dvo_hip_config* cfg, BatchPlan& bp, bool taps_missing) {
  ResidentPlan rp = plan_resident(ctx, cfg, bp, taps_missing);
  if (ctx->opt_coarse == 1) {                                  // (option "coarse" 1: the fused coarse-level kernel wherever the levels admit it)
    const ResidentPlan none;
    plan_coarse(ctx, cfg, bp, none);
    if (bp.coarse_levels > 0) rp = none;
  } else {
    plan_coarse(ctx, cfg, bp, rp);
  }
  return rp;
}

int resident_levels_of(const dvo_hip_context* ctx, const dvo_hip_config* cfg, const CameraGeom* cam, int n, bool taps_missing) {
  BatchPlan bp;
  make_plan(ctx, cam, cfg, n, bp);
  return plan_paths(ctx, cfg, bp, taps_missing).levels;
}

int run_resident(dvo_hip_context* ctx, Workspace& w, const dvo_hip_config* cfg, const BatchPlan& bp, const ResidentPlan& rp,
                 const double* tinit, bool want_stats) {
  hipStream_t s = w.stream;
  ResidentArgs args;
  std::memset(&args, 0, sizeof(args));
  for (int l = cfg->last_level; l <= cfg->first_level; ++l) args.geom[l] = bp.geom[l];
  args.pair_ptrs = bp.pair_ptrs;
  args.states = w.states.as<PairState>();
  args.levels = w.lvl_stats.as<dvo_hip_level_stats>();
  args.iters = w.it_stats.as<dvo_hip_iteration_stats>();
  args.T_init = w.t_init.as<double>();
  args.scratch = w.scratch.as<float2>();
  args.error_word = w.host_status + w.resident_error_word;
  args.prm = bp.prm;
  args.n_pairs = bp.n;
  args.group = rp.group;
  args.first_level = cfg->first_level;
  args.last_level = cfg->first_level - rp.levels + 1;
  args.flags = ctx->opt_resident_flags;
  args.results = args.last_level == cfg->last_level ? w.results.as<dvo_hip_result>() : nullptr;
  if (bp.n <= kResidentInline) {                               // no table in device memory, no copy commands in front of the launch
    args.use_inline = 1;
    for (int l = cfg->last_level; l <= cfg->first_level; ++l)
      for (int i = 0; i < bp.n; ++i) args.inline_ptrs[l * bp.n + i] = bp.host_ptrs[size_t(l) * bp.n + i];
    std::memcpy(args.inline_T, tinit, size_t(bp.n) * 16 * sizeof(double));
  }
  if (rp.direct) {
    DVO_WS_TRY(w, w.direct_results.reserve(size_t(bp.n) * sizeof(dvo_hip_result)));
    DVO_WS_TRY(w, w.direct_done.reserve(64));
    args.results = w.direct_results.as<dvo_hip_result>();
    args.done_word = w.direct_done.as<int>();
    *static_cast<volatile int*>(args.done_word) = 0;
    if (want_stats) {
      DVO_WS_TRY(w, w.direct_levels.reserve(size_t(bp.n) * bp.cap_levels * sizeof(dvo_hip_level_stats)));
      DVO_WS_TRY(w, w.direct_iters.reserve(size_t(bp.n) * bp.cap_iters * sizeof(dvo_hip_iteration_stats)));
      args.host_levels = w.direct_levels.as<dvo_hip_level_stats>();
      args.host_iters = w.direct_iters.as<dvo_hip_iteration_stats>();
    }
  }
  ctx->resident_launches += 1;
  ctx->resident_levels += rp.levels;
  if (rp.group > 1) {
    // the rows of every group, and behind them one heartbeat word per workgroup (align_resident.hip, "Flow control")
    const size_t bytes = size_t(bp.n) * rp.group * kResidentRing * kResidentSlots * sizeof(unsigned long long) + align_up(size_t(bp.n) * rp.group * sizeof(unsigned), 256);
    const bool grown = bytes > w.exchange.bytes;
    DVO_WS_TRY(w, w.exchange.reserve(bytes));
    // sequence numbers never repeat between launches; when they would wrap (or the rows are new / in doubt) the rows are cleared
    const unsigned long long need64 = 2ull * unsigned(rp.levels) * unsigned(cfg->max_iterations_per_level) + 4ull;
    const unsigned need = need64 < 0x40000000ull ? unsigned(need64) : 0x40000000u;   // (more exchanges than that do not happen)
    // (... or the batch has another shape than the last one: the heartbeat words live BEHIND the rows, at an offset that follows pairs x
    // group, and a word that was a row slot of the previous launch -- the float half of a slot, 0x3f800000, reads as a beat far ahead --
    // would let a workgroup skip the wait the heartbeat exists for)
    const long long shape = (long long)bp.n * rp.group;
    if (grown || shape != w.exchange_shape || need64 >= 0x40000000ull || w.resident_sequence > 0xffffffffu - need - 1u) {
      DVO_WS_TRY(w, hipMemsetAsync(w.exchange.p, 0, w.exchange.bytes, s));
      w.resident_sequence = 0;
      w.exchange_shape = shape;
    }
    args.exchange = w.exchange.as<unsigned long long>();
    args.sequence_base = w.resident_sequence;
    w.resident_sequence += need;
  }
  DVO_WS_TRY(w, launch_match_resident(s, args, ctx->opt_resident_cooperative != 0));
  return DVO_HIP_OK;
}

int run_coarse(dvo_hip_context* ctx, Workspace& w, const dvo_hip_config* cfg, const BatchPlan& bp) {
  CoarseArgs args;
  std::memset(&args, 0, sizeof(args));
  for (int l = cfg->last_level; l <= cfg->first_level; ++l) args.geom[l] = bp.geom[l];
  args.pair_ptrs = bp.pair_ptrs;
  args.states = w.states.as<PairState>();
  args.levels = w.lvl_stats.as<dvo_hip_level_stats>();
  args.iters = w.it_stats.as<dvo_hip_iteration_stats>();
  args.T_init = w.t_init.as<double>();
  args.partials = w.partials.as<float>();
  args.scratch = w.scratch.as<float2>();
  args.fallback_count = w.win_fallbacks.as<unsigned long long>();
  args.f16_range_flag = w.f16_range_flag;
  args.prm = bp.prm;
  args.n_pairs = bp.n;
  args.first_level = cfg->first_level;
  args.last_level = cfg->first_level - bp.coarse_levels + 1;
  for (int l = args.last_level; l <= args.first_level; ++l) {   // (prepare_buffers sized both buffers for the largest level of the match)
    args.max_tiles = std::max(args.max_tiles, bp.geom[l].tiles_x * bp.geom[l].tiles_y);
    args.max_entries = std::max(args.max_entries, residual_entries(bp.geom[l]));
  }
  args.results = args.last_level == cfg->last_level ? w.results.as<dvo_hip_result>() : nullptr;
  ctx->coarse_launches += 1;
  ctx->coarse_levels += bp.coarse_levels;
  DVO_WS_TRY(w, launch_match_coarse(w.stream, args, ctx->opt_coarse_wgs == 3 ? 3 : 4));
  return DVO_HIP_OK;
}

std::mutex g_rare_path_mutex;   // (see run_batch: the repeat of pairs that left the f16 range)

constexpr int kF32GramHoldBatches = 32;   // after a batch left the f16 range of the Gram operands: this many batches go straight to the f32 Gram

hipEvent_t g_trace_ev[2] = {nullptr, nullptr};   // DVO_HIP_TRACE_SLOW: device time stamps around a batch's preparation

// The coarse-to-fine Gauss-Newton driver of a batch (dense_tracking.cpp:131-376 for every pair at once).
int run_batch(dvo_hip_context* ctx, int n, dvo_hip_frame* const* refs, dvo_hip_frame* const* curs, const dvo_hip_config* cfg,
              dvo_hip_result* results, dvo_hip_level_stats* levels, int cap_levels, dvo_hip_iteration_stats* iters, int cap_iters) {
  Workspace& w = ctx->ws[0];
  hipStream_t s = w.stream;
  // DVO_HIP_TRACE_SLOW=<ms>: where the host thread spent the time before the first launch of a batch whose preparation took longer
  static const double trace_slow_ms = std::getenv("DVO_HIP_TRACE_SLOW") ? std::atof(std::getenv("DVO_HIP_TRACE_SLOW")) : 0.0;
  double mark_ms[5] = {0, 0, 0, 0, 0};
  auto mark = [&](int k) { if (trace_slow_ms > 0.0) mark_ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->batch_entry).count(); };
  mark(0);                                                     // roles ensured (dvo_hip_match_batch)
  if (trace_slow_ms > 0.0) {
    if (!g_trace_ev[0]) { (void)hipEventCreate(&g_trace_ev[0]); (void)hipEventCreate(&g_trace_ev[1]); }
    (void)hipEventRecord(g_trace_ev[0], s);
  }
  // (the Gram schedule of the batch -- f16 high + low operand pairs or f32 -- was decided by the caller, effective_variant_scope; what
  // this function changes for a repeat is undone when it returns)
  struct VariantScope {
    dvo_hip_context* c; int keep;
    ~VariantScope() { c->opt_variant = keep; }
  } variant_scope{ctx, ctx->opt_variant};
  BatchPlan bp;
  make_plan(ctx, refs[0]->cam, cfg, n, bp);
  const BatchPolicy policy(ctx->compute_units);               // every batch-size threshold below: batch_policy.h
  // The coarse levels -- for a small batch every level -- run inside ONE launch (align_resident.hip): no launch per iteration, no
  // host poll, the pairs of a batch do not wait for each other.
  const ResidentPlan rp = plan_paths(ctx, cfg, bp, window_taps_missing(ctx, cfg, n, curs));
  const bool tables_inline = rp.direct && n <= kResidentInline;   // plane pointers and initial guesses travel as kernel arguments
  int rc = prepare_buffers(w, cfg, refs, curs, bp, !tables_inline);
  if (rc != DVO_HIP_OK) return rc;
  mark(1);

  // initial guesses (Result.Transformation is in/out, dense_tracking.cpp:137-147)
  std::vector<double> tinit(size_t(n) * 16);
  for (int i = 0; i < n; ++i) std::memcpy(&tinit[size_t(i) * 16], results[i].transformation, 16 * sizeof(double));
  if (!tables_inline) DVO_WS_TRY(w, w.tables->upload(s, w.t_init.p, tinit.data(), tinit.size() * sizeof(double)));
  // per-step tallies (device) and status words (pinned host memory the device writes, see publish_step)
  const size_t main_steps = size_t(bp.cap_iters) + 8 + kResidentErrorWords;
  // The slow lane of a batch (option "overlap_tails"): see in front of the level loop below.  Batches whose levels are begun by launches
  // (beyond the solver steps' hand-over), the plain launch chain (no step in the sweep's tail), more than one level on this path.
  bool overlap_batch = ctx->opt_overlap_tails != 0 && !policy.level_hand_over(n) && ctx->opt_sweep_tail == 0 && rp.levels == 0 && bp.coarse_levels == 0 &&
                       cfg->first_level > cfg->last_level;
  for (int l = cfg->last_level; l <= cfg->first_level; ++l)   // (the lane passes through every level: each one's sweep must take a list of pairs)
    overlap_batch = overlap_batch && sweep_takes_pair_list(ctx->opt_variant, bp.rpw[l], bp.geom[l]);
  const size_t tail_steps_cap = overlap_batch ? size_t(bp.cap_iters) + 12 * size_t(bp.nlev) + 8 : 0;   // (per level: the passes of its slowest pair, and up to eight steps enqueued ahead of what is known)
  const size_t n_steps = main_steps + tail_steps_cap;         // (the tail's status words and tallies lie behind the main chain's)
  if (overlap_batch) {
    size_t t_tiles = 1, t_entries = 0;
    for (int l = cfg->last_level; l <= cfg->first_level; ++l) {
      t_tiles = std::max(t_tiles, size_t(bp.geom[l].tiles_x) * bp.geom[l].tiles_y);
      t_entries = std::max(t_entries, residual_entries(bp.geom[l]));
    }
    DVO_WS_TRY(w, w.tail_partials.reserve(size_t(n) * t_tiles * kAccStride * sizeof(float)));
    DVO_WS_TRY(w, w.tail_scratch.reserve(size_t(n) * t_entries * sizeof(float2)));
    DVO_WS_TRY(w, w.tail_flags.reserve(align_up(size_t(n), 256)));
    DVO_WS_TRY(w, w.tail_ll.reserve(size_t(n) * kLlBlocksPerPair * sizeof(double)));

    DVO_WS_TRY(w, w.counters.reserve(n_steps * sizeof(unsigned long long) + align_up(size_t(n) * sizeof(int), 8)));
    if (!w.tail_stream) {
      int prio_least = 0, prio_greatest = 0;
      DVO_WS_TRY(w, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
      DVO_WS_TRY(w, hipStreamCreateWithPriority(&w.tail_stream, hipStreamNonBlocking, prio_greatest));
      DVO_WS_TRY(w, hipEventCreateWithFlags(&w.tail_split, hipEventDisableTiming));
      DVO_WS_TRY(w, hipEventCreateWithFlags(&w.tail_end, hipEventDisableTiming));
    }
  }
  // (the lane's two lists -- the one in use and the one being made -- and the chain's own active-pair list)
  if (overlap_batch || ctx->opt_tail_lists) DVO_WS_TRY(w, w.tail_list.reserve(3 * align_up(size_t(n), 64) * sizeof(int)));
  // a batch that ended early (time-out, HIP error) may have left steps queued: nothing of it may still be running when the
  // status words and tallies are reset
  if (w.needs_drain) {
    if (w.tail_stream) DVO_WS_TRY(w, hipStreamSynchronize(w.tail_stream));
    DVO_WS_TRY(w, sync_stream(s));
    // ... and the f16 range words it may have raised are nobody's business any more (they are cleared where they are read, at a batch's
    // regular end: left standing, the next batch would repeat unrelated pairs at the same indices -- round-5 advisor finding)
    if (w.f16_range_flag) std::memset(w.f16_range_flag, 0, w.f16_range_words * sizeof(int));
  }
  w.needs_drain = true;                                        // until this batch has come to its regular end
  mark(2);
  rc = ensure_host_status(w, n_steps);
  if (rc != DVO_HIP_OK) return rc;
  mark(3);
  w.resident_error_word = int(w.resident_launch_counter++ % kResidentErrorWords);
  if (rp.direct) {
    w.host_status[w.resident_error_word] = 0;
  } else {
    // (the previous batch may have ended on the host's side before the device was through with the step words: the direct path)
    // (only then: a batch that ended with a stream wait left nothing behind, and a wait for the table uploads just enqueued has been
    // seen to return 14-24 ms after the device was done with them -- 19 microseconds by its own time stamps: r03, the validator's
    // 128-pair batches in the first process on a box, never under the profiler)
    if (trace_slow_ms > 0.0) (void)hipEventRecord(g_trace_ev[1], s);
    if (w.device_may_lag) DVO_WS_TRY(w, sync_stream(s));
    w.device_may_lag = false;
    if (trace_slow_ms > 0.0) {
      mark(4);
      float gpu_ms = -1.0f;
      if (mark_ms[4] - mark_ms[3] > trace_slow_ms && hipEventElapsedTime(&gpu_ms, g_trace_ev[0], g_trace_ev[1]) == hipSuccess)
        std::fprintf(stderr, "dvo_hip: the stream wait took %.3f ms on the host; on the device %.3f ms passed between the batch's first and last command so far\n",
                     mark_ms[4] - mark_ms[3], gpu_ms);
    }
    std::memset(w.host_status, 0, n_steps * sizeof(int));
    DVO_WS_TRY(w, hipMemsetAsync(w.counters.p, 0, n_steps * sizeof(unsigned long long) + align_up(size_t(n) * sizeof(int), 8), s));
  }

  PairState* states = w.states.as<PairState>();
  dvo_hip_level_stats* d_levels = w.lvl_stats.as<dvo_hip_level_stats>();
  dvo_hip_iteration_stats* d_iters = w.it_stats.as<dvo_hip_iteration_stats>();
  float* partials = w.partials.as<float>();
  float2* scratch = w.scratch.as<float2>();
  double* ll_partials = w.ll_partials.as<double>();
  unsigned long long* tallies = w.counters.as<unsigned long long>();
  int* arrivals = reinterpret_cast<int*>(tallies + n_steps);
  const int per_level = cfg->max_iterations_per_level;
  const int per_sync = ctx->opt_iters_per_sync > 0 ? ctx->opt_iters_per_sync : 1;
  int step = kResidentErrorWords;                           // the first status words belong to the resident kernel's launches
  mark(4);
  if (trace_slow_ms > 0.0 && mark_ms[4] > trace_slow_ms)
    std::fprintf(stderr, "dvo_hip: slow preparation of a %d-pair batch: roles %.3f, plan + buffers + tables %.3f, initial guesses + drain %.3f, status words %.3f, "
                 "stream wait + reset %.3f ms\n", n, mark_ms[0], mark_ms[1] - mark_ms[0], mark_ms[2] - mark_ms[1], mark_ms[3] - mark_ms[2], mark_ms[4] - mark_ms[3]);
  const auto t_launch = std::chrono::steady_clock::now();
  int level_from = cfg->first_level;
  const bool want_stats = (levels && cap_levels > 0) || (iters && cap_iters > 0);
  ResidentBudget::Hold compute_units;                          // released when this batch is through with the device
  if (rp.levels > 0 && rp.group > 1)
    compute_units.take(ResidentBudget::of(ctx->device), n * rp.group, ctx->compute_units > 0 ? ctx->compute_units : 256);
  if (rp.levels > 0) {
    Range range("resident");
    rc = run_resident(ctx, w, cfg, bp, rp, tinit.data(), want_stats);
    if (rc != DVO_HIP_OK) return rc;
    if (!ctx->deferred.empty()) {                              // (see the launch path below)
      const int rc_deferred = flush_deferred(ctx);
      if (rc_deferred != DVO_HIP_OK) {
        w.err = ctx->err;
        return rc_deferred;
      }
    }
    level_from = cfg->first_level - rp.levels;
  }
  if (bp.coarse_levels > 0) {
    // the coarse levels of every pair in ONE launch, a workgroup per pair (align_coarse.hip): nothing to poll -- the host goes on to
    // enqueue the first steps of the level behind them
    Range range("coarse");
    rc = run_coarse(ctx, w, cfg, bp);
    if (rc != DVO_HIP_OK) return rc;
    if (!ctx->deferred.empty()) {
      const int rc_deferred = flush_deferred(ctx);
      if (rc_deferred != DVO_HIP_OK) {
        w.err = ctx->err;
        return rc_deferred;
      }
    }
    level_from = cfg->first_level - bp.coarse_levels;
  }
  // ---- the slow lane of a batch (round 6, option "overlap_tails") ------------------------------------------------------------------------
  // The pairs of a batch need different numbers of passes on a level, and the launch chain runs a level until its LAST pair has left it: on
  // the bench's 1024-pair batches 11 of a step's 32 iterations are such tail launches, a few dozen pairs each, while the chip waits (1.1 ms
  // of 11).  In the reference every match() runs on its own from start to end (dense_tracking.cpp:200-357).  Here, once at most 1/8 of
  // the pairs is still on a level, those pairs leave the batch's chain for good: they get a flag byte and a place in a list
  // (k_mark_stragglers), the chain goes on to the next level without them (LevelGeom::skip_flags), and a SLOW LANE -- a second stream, its
  // own partial rows and residual pairs, its own status words -- runs them to the end of the match with launches over the list
  // (LevelGeom::pair_list): the rest of that level, then level after level behind the main chain, taking up the stragglers the chain
  // sheds on the way.  The lane works on one level at a time and never gets ahead of the chain: it leaves a level when none of its members is
  // active on it AND the chain has left it (so nobody can still arrive there).  The batch ends when both have.  A pair's arithmetic does
  // not know in which launch it runs: the records are the synchronous chain's bit for bit (tests/test_gpu_overlap.py).
  struct LevelSchedule {
    bool fused_ll = false, two_waves = false;
    int ll_blocks = 0;
  };
  auto schedule_of = [&](int level) {
    const LevelGeom& g = bp.geom[level];
    LevelSchedule ls;
    const int fuse_opt = ctx->opt_fused_ll_pixels;
    ls.fused_ll = g.w * g.h <= (fuse_opt > 0 ? fuse_opt : (policy.fused_loglik_on_large_levels(n) && !ctx->opt_deterministic ? kFusedLoglikMaxPixelsBatch : kFusedLoglikMaxPixels));
    ls.ll_blocks = ctx->opt_ll_blocks > 0 ? std::min(ctx->opt_ll_blocks, kLlBlocksPerPair)
                   : (!g.compact || ctx->opt_deterministic ? kLlBlocksPerPair : policy.loglik_blocks(n));
    ls.two_waves = ctx->opt_solver_waves == 2 ||
                   (ctx->opt_solver_waves == 0 && ls.fused_ll && !g.compact && g.tiles_x * g.tiles_y <= 32 && policy.solver_two_waves(n));
    return ls;
  };
  struct SlowLane {
    bool live = false, done = false;
    int level = -1;                  // the level its launches work on
    int cap = 0;                     // entries of the current list: an upper bound of the flagged pairs
    int list_sel = 0;                // which of the two list buffers is the current one (the other may still be read by launches in flight)
    int seen = 0;                    // status words of the lane read so far (they complete in order: one stream)
    int valid_from = 0;              // steps before this one say nothing about the lane's level as it is now (another level, or members have arrived since)
    int last_active = -1;
  } sl;
  int tail_next = 0;                 // steps of the lane enqueued in this batch (index of its next status word, behind the chain's)
  int main_level = level_from;       // the level the batch's own chain works on (below last_level: through)
  unsigned char* const lane_flags = w.tail_flags.as<unsigned char>();
  auto lane_list = [&](int sel) { return w.tail_list.as<int>() + size_t(sel) * align_up(size_t(n), 64); };
  if (overlap_batch) launch_clear_flags(s, lane_flags, n);
  auto tail_enqueue = [&](int count) {
    const LevelSchedule ls = schedule_of(sl.level);
    LevelGeom g = bp.geom[sl.level];
    g.pair_list = lane_list(sl.list_sel);
    const PairPtrs* pp = bp.pair_ptrs + size_t(sl.level) * n;
    float* lp = w.tail_partials.as<float>();
    float2* lscr = w.tail_scratch.as<float2>();
    double* lll = w.tail_ll.as<double>();
    for (int c = 0; c < count && size_t(tail_next) < tail_steps_cap; ++c, ++tail_next) {
      const size_t idx = main_steps + size_t(tail_next);
      launch_residual_reduce(w.tail_stream, ctx->opt_variant, bp.rpw[sl.level], sl.level == 0, g, pp, states, sl.cap, lp, lscr, w.win_fallbacks.as<unsigned long long>(),
                             w.f16_range_flag);
      if (!ls.fused_ll) launch_loglik(w.tail_stream, g, states, sl.cap, lp, lscr, lll, ls.ll_blocks, ctx->opt_deterministic != 0);
      launch_solver_step(w.tail_stream, states, sl.cap, bp.prm, g, lp, lll, ls.ll_blocks, ls.fused_ll ? lscr : nullptr, d_levels, d_iters,
                         tallies + idx, w.host_status + idx, false, cfg->first_level - sl.level, nullptr);
      ctx->overlapped_steps += 1;
    }
    (void)hipEventRecord(w.tail_end, w.tail_stream);           // (the batch's end waits for the latest record: everything enqueued so far)
  };
  // what the lane's status words say by now, and what follows from it (never waits)
  auto tail_service = [&]() {
    if (!sl.live || sl.done) return;
    while (sl.seen < tail_next) {
      const int v = static_cast<volatile int*>(w.host_status)[main_steps + size_t(sl.seen)];
      if (!(v & kStepDoneFlag)) break;
      if (sl.seen >= sl.valid_from) sl.last_active = v & ~kStepDoneFlag;
      sl.seen += 1;
    }
    if (sl.seen > sl.valid_from && sl.last_active == 0) {      // nobody of the lane is active on its level
      if (sl.level <= main_level) return;                      // (the chain is still on it: its stragglers may yet arrive)
      if (sl.level == cfg->last_level) {
        sl.done = true;
        return;
      }
      // on to the next level: the members that have left this one begin it (those that arrived further down are on theirs already)
      const int lv = sl.level - 1;
      launch_level_begin(w.tail_stream, states, n, bp.prm, bp.geom[lv], lv, bp.pair_ptrs + size_t(lv) * n, d_levels, nullptr, lane_flags, 1, sl.level);
      sl.level = lv;
      sl.valid_from = tail_next;
      sl.last_active = -1;
      tail_enqueue(3);
      return;
    }
    if (tail_next - sl.seen < 2) tail_enqueue(2);
  };
  // the chain sheds the pairs still active on `level` (at most `active` of them) to the lane
  auto tail_shed = [&](int level, int active) -> int {
    Range range("shed");
    const int next_sel = sl.live ? sl.list_sel ^ 1 : 0;
    const int cap = std::min(n, sl.cap + active);
    launch_mark_stragglers(s, states, n, level, lane_flags, lane_list(next_sel), cap);
    DVO_WS_TRY(w, hipEventRecord(w.tail_split, s));
    DVO_WS_TRY(w, hipStreamWaitEvent(w.tail_stream, w.tail_split, 0));
    sl.cap = cap;
    sl.list_sel = next_sel;
    if (!sl.live) {
      sl.live = true;
      sl.level = level;
    }
    sl.valid_from = tail_next;                                 // (whatever level the lane is on: its launches read the new list from here on)
    sl.last_active = -1;
    tail_enqueue(3);
    ctx->overlapped_tails += 1;
    return DVO_HIP_OK;
  };
  // the chain's wait for a step, looking after the lane meanwhile
  auto wait_main = [&](int idx, int* active) -> int {
    if (!sl.live || sl.done) return wait_for_step(w, idx, active);
    volatile int* word = w.host_status + idx;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
      const int v = *word;
      if (v & kStepDoneFlag) {
        *active = v & ~kStepDoneFlag;
        tail_service();
        return DVO_HIP_OK;
      }
      if ((spins & 0x3f) == 0) tail_service();
      if ((spins & 0xfffff) == 0) {
        const int rc_check = step_wait_check(w, t0);
        if (rc_check != DVO_HIP_OK) return rc_check;
      }
    }
  };

  for (int level = level_from; level >= cfg->last_level; --level) {
    main_level = level;
    LevelGeom g = bp.geom[level];
    if (sl.live) g.skip_flags = lane_flags;                    // (the lane's pairs are not this chain's any more)
    const PairPtrs* pp = bp.pair_ptrs + size_t(level) * n;
    static const char* const kPrep[kMaxLevels] = {"prep L0", "prep L1", "prep L2", "prep L3", "prep L4", "prep L5", "prep L6", "prep L7"};
    static const char* const kErr[kMaxLevels] = {"err L0", "err L1", "err L2", "err L3", "err L4", "err L5", "err L6", "err L7"};
    static const char* const kLinsys[kMaxLevels] = {"linsys L0", "linsys L1", "linsys L2", "linsys L3", "linsys L4", "linsys L5", "linsys L6", "linsys L7"};
    // Batches of up to 256 pairs: every level but the launch path's first is begun pair by pair by the solver steps of the level before
    // it, and the results are written by the steps behind the last level (NextLevel) -- no launch between two levels, none at the end:
    // builds alternated on one box, 16 / 64 / 128 pairs 0.558 -> 0.553 / 1.196 -> 1.187 / 1.825 -> 1.807 ms per step; 256 and 1024
    // pairs level, 512 pairs 5.79 -> 5.88 (the launches k_level_begin / k_finish stay there)
    const bool hand_over = policy.level_hand_over(n);
    if (level == level_from || !hand_over) {
      Range range(kPrep[level]);
      // (the slow lane's pairs begin their levels there)
      launch_level_begin(s, states, n, bp.prm, g, level, pp, d_levels, level == cfg->first_level ? w.t_init.as<double>() : nullptr,
                         sl.live ? lane_flags : nullptr, 0);
    }
    NextLevel next;
    std::memset(&next, 0, sizeof(next));
    if (hand_over && level > cfg->last_level) {
      const LevelGeom& gn = bp.geom[level - 1];
      next.valid = 1;
      next.level = level - 1;
      next.fx = gn.fx; next.fy = gn.fy; next.ox = gn.ox; next.oy = gn.oy;
      next.pairs = bp.pair_ptrs + size_t(level - 1) * n;
    } else if (hand_over) {
      next.results = w.results.as<dvo_hip_result>();           // the last level: a pair that has left it gets its result written
    }
    // levels this small run the log-likelihood sweep inside the solver workgroup (one launch less per iteration)
    // (measured, scripts/ab_match.py fused_ll_pixels: 128 pairs 2.152 -> 2.117 ms with level 1 fused; 16 pairs 0.756 -> 0.799, one
    // pair 0.524 -> 0.547: a lone workgroup per pair is slower than 32 blocks when the chip is empty)
    // (round 5, packed residual pairs, the streaming step beside its ingest, scripts/r5_midsize.py: level 1 in a launch of its own
    // 64 pairs 1.456 -> 1.378 ms, 128 pairs 1.939 -> 1.908, 256 pairs 3.308 -> 3.286, 512 pairs 5.859 -> 5.880: fused from 512 pairs)
    const LevelSchedule schedule = schedule_of(level);         // (the rule: schedule_of, above -- the slow lane runs its pairs by the same one)
    const bool fused_ll = schedule.fused_ll;
    // Chunks of `per_sync` iterations are enqueued ONE AHEAD of the poll: while the host waits for the status word of
    // chunk k, chunk k+1 is already queued, so the GPU never idles for a host round trip.  Iterations enqueued past the
    // end of the level are no-ops (workgroups exit on !active).
    // workgroups per pair of the log-likelihood pass: each begins by reducing the pair's scale sums (a latency chain of ~10 us), which
    // a batch that fills the device anyway pays once per workgroup for nothing -- fewer, longer ones then (option ll_blocks to override)
    // (round 5, mid-size batches, scripts/r5_midsize.py: 16 instead of 32 workgroups per pair 64 / 128 / 200 pairs 1.179 -> 1.173 /
    // 1.775 -> 1.760 / 2.558 -> 2.538 ms per step; 8: 1.183 / 1.760 / 2.546)
    const int ll_blocks = schedule.ll_blocks;
    // the solver step of the smallest levels in two-wavefront workgroups (four per compute unit instead of two): a batch that otherwise
    // needs two goes of 512 resident workgroups (option solver_waves 2 / 4 to force; the records do not depend on it).  Measured at
    // 1024 pairs (scripts/r4_trace.sh): 80 x 60 46 -> 36 us per step; 160 x 120 with packed residuals 77 -> 90 (two wavefronts walk
    // its 96 slots in six rounds instead of three), 320 x 240 and the finest level level: those keep four.
    const bool solver_two_waves = schedule.two_waves;
    // The step in the sweep's launch (round 6): where the log-likelihood pass runs inside the solver step anyway and the level's sweep
    // has the instantiation, the workgroup that completes a pair's last tile runs the pair's step -- ONE launch per iteration
    const bool tail = ctx->opt_sweep_tail != 0 && fused_ll && sweep_has_tail(ctx->opt_variant, bp.rpw[level], g);
    // (option 2: the WIDE half of the step -- reduction and log-likelihood, its memory round trips -- in the sweep's tail, the serial half
    // in a one-wavefront launch behind it)
    double* pair_sums = tail && ctx->opt_sweep_tail == 2 ? w.pair_sums.as<double>() : nullptr;
    // (what the chain's launches of this level cover: every pair, or -- its last steps, see "hold" below -- the list of those still active)
    LevelGeom g_run = g;
    int n_run = n;
    auto enqueue_chunk = [&](int count) {
      for (int c = 0; c < count; ++c, ++step) {
        if (tail) {
          Range range(kErr[level]);
          const SolverStepArgs a = make_solver_step_args(states, n, bp.prm, partials, ll_partials, ll_blocks, scratch, d_levels, d_iters, tallies + step, w.host_status + step,
                                                         cfg->first_level - level, &next, arrivals, pair_sums);
          launch_residual_reduce(s, ctx->opt_variant, bp.rpw[level], level == 0, g, pp, states, n, partials, scratch, w.win_fallbacks.as<unsigned long long>(),
                                 w.f16_range_flag, &a);
          if (pair_sums) {
            Range range2(kLinsys[level]);
            launch_solver_serial(s, n, g, a);
          }
          ctx->tail_steps += 1;
          continue;
        }
        {
          Range range(kErr[level]);
          launch_residual_reduce(s, ctx->opt_variant, bp.rpw[level], level == 0, g_run, pp, states, n_run, partials, scratch, w.win_fallbacks.as<unsigned long long>(),
                                 w.f16_range_flag);
          if (!fused_ll) launch_loglik(s, g_run, states, n_run, partials, scratch, ll_partials, ll_blocks, ctx->opt_deterministic != 0);
        }
        if (g_run.pair_list) ctx->listed_steps += 1;
        Range range(kLinsys[level]);
        launch_solver_step(s, states, n_run, bp.prm, g_run, partials, ll_partials, ll_blocks, fused_ll ? scratch : nullptr, d_levels, d_iters,
                           tallies + step, w.host_status + step, solver_two_waves, cfg->first_level - level, &next);   // (the level record every pair on this level is at: fetched with the state)
      }
    };
    int enqueued = std::min(per_sync, per_level);
    enqueue_chunk(enqueued);
    // (option "defer_ingest_pixels": not before the chain has reached a level of that many pixels -- the frame build is bound by memory, and
    // beside it the latency-bound sweeps and steps of the SMALL levels run 30-50 % longer (80 x 60: 52 -> 77 us, 160 x 120: 159 -> 212 us per
    // 1024-pair launch), the issue-bound sweeps of the large ones 2 %: a large batch's ingest belongs beside levels 1 and 0)
    const bool flush_here = ctx->opt_defer_ingest_pixels <= 0 || g.w * g.h >= ctx->opt_defer_ingest_pixels || level == cfg->last_level;
    if (!ctx->deferred.empty() && flush_here) {
      // A recorded ingest of the caller's next batch (option "defer_ingest") is carried out now, behind the first launches of this
      // batch.  It is ~0.1-0.5 ms of host time during which nothing more would be enqueued here: where the level's steps are short,
      // a few more of them go out first (a pair needs more than four passes on its first level; a step too many exits at once).
      // (how many: the ingest of 2 n frames is ~0.3 us of host time per frame, a step of a small level ~20 us -- measured, builds
      // alternated on one box: seven instead of three more 256 pairs 3.288 -> 3.216 ms per step, 128 pairs 1.837 -> 1.856, 64 pairs
      // 1.199 -> 1.210)
      if (size_t(g.tiles_x) * g.tiles_y * size_t(n) < 65536) {
        const int lead = std::min(policy.deferred_ingest_lead(n) * per_sync, per_level - enqueued);
        if (lead > 0) {
          enqueue_chunk(lead);
          enqueued += lead;
        }
      }
      const int rc_deferred = flush_deferred(ctx);
      if (rc_deferred != DVO_HIP_OK) {
        w.err = ctx->err;
        return rc_deferred;
      }
    }
    int watched = step - 1;                                  // last step of the chunk whose outcome is awaited
    // (the slow lane: every level but the last may shed its stragglers to it -- on the last one nothing follows that they could run beside)
    const bool list_tails = ctx->opt_tail_lists != 0 && !tail && !hand_over && sweep_takes_pair_list(ctx->opt_variant, bp.rpw[level], g);
    const bool may_shed = overlap_batch && level > cfg->last_level && !tail && sweep_takes_pair_list(ctx->opt_variant, bp.rpw[level], g);
    // Where an EMPTY step is expensive -- the dispatcher needs 95 us for the 307 200 workgroups of a 1024-pair finest-level sweep
    // that all exit at once, 114 us with its log-likelihood and solver launches -- the step ahead of the poll is not enqueued once
    // only a few pairs are left on the level: the host then waits for the outcome first (a bubble of ~15 us if another step is
    // needed).  Elsewhere the speculative step is cheaper than the bubble.
    const bool empty_step_is_costly = size_t(g.tiles_x) * g.tiles_y * size_t(n) >= (ctx->opt_tail_speculation >= 2 ? size_t(ctx->opt_tail_speculation) : kCostlyEmptyStepWorkgroups) && ctx->opt_tail_speculation != 1;
    int last_active = n;
    for (;;) {
      const bool hold = empty_step_is_costly && last_active * 8 <= n;
      int more = hold ? 0 : std::min(per_sync, per_level - enqueued);
      if (more > 0) {
        enqueue_chunk(more);
        enqueued += more;
      }
      int active = 0;
      rc = wait_main(watched, &active);
      if (rc != DVO_HIP_OK) return rc;
      last_active = active;
      if (may_shed && active > 0 && size_t(active) * size_t(ctx->opt_overlap_fraction) <= size_t(n)) {
        // the pairs still on the level (at most `active`: the steps enqueued ahead of this poll only take some away) go to the slow lane
        rc = tail_shed(level, active);
        if (rc != DVO_HIP_OK) return rc;
        break;
      }
      if (hold && active > 0) {                              // (the held-back step is needed after all)
        // ... by `active` pairs, none of whose steps is in flight: from here on the launches cover the list of them
        if (list_tails && !g_run.pair_list) {
          int* const own_list = w.tail_list.as<int>() + 2 * align_up(size_t(n), 64);
          launch_mark_stragglers(s, states, n, level, sl.live ? lane_flags : nullptr, own_list, active, true);
          g_run.pair_list = own_list;
          n_run = active;
        }
        more = std::min(per_sync, per_level - enqueued);
        if (more > 0) {
          enqueue_chunk(more);
          enqueued += more;
        }
      }
      if (active == 0 || more <= 0) break;                   // every pair left this level (or the iteration cap is reached)
      watched = step - 1;
    }
    // The pairs that ended the level in the step just awaited are handed over (NextLevel) by the step that was enqueued ahead of the poll.
    // If there is none -- the step was held back, or the iteration cap is reached -- one solver launch does nothing else.
    if (hand_over && step - 1 <= watched) {
      Range range(kLinsys[level]);
      launch_solver_step(s, states, n, bp.prm, g, partials, ll_partials, ll_blocks, nullptr, d_levels, d_iters, tallies + step, w.host_status + step,
                         solver_two_waves, cfg->first_level - level, &next);
      ++step;
    }
  }

  if (sl.live) {
    // the chain is through; the slow lane runs its pairs to the end of the match (from here on it may leave any level)
    main_level = cfg->last_level - 1;
    const auto t_wait = std::chrono::steady_clock::now();
    tail_service();
    while (!sl.done) {
      if (tail_next == 0 || size_t(tail_next) >= tail_steps_cap) {
        w.err = "match: the slow lane of the batch ran out of steps";
        return DVO_HIP_ERR_HIP;
      }
      int unused = 0;
      rc = wait_for_step(w, int(main_steps) + tail_next - 1, &unused);
      if (rc != DVO_HIP_OK) return rc;
      tail_service();
    }
    ctx->tail_drains += 1;
    ctx->tail_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_wait).count();
    DVO_WS_TRY(w, hipStreamWaitEvent(s, w.tail_end, 0));       // (k_finish and the copies below follow the lane's last step)
  }
  const bool resident_used = rp.levels > 0;
  std::vector<dvo_hip_level_stats> hl;
  std::vector<dvo_hip_iteration_stats> hi;
  const dvo_hip_level_stats* hl_src = nullptr;
  const dvo_hip_iteration_stats* hi_src = nullptr;
  std::chrono::steady_clock::time_point t_enqueued, t_done;
  if (rp.direct) {
    // the kernel writes results (and statistics) into pinned host memory and counts the pairs done: the host thread watches that
    // word -- no copy command, no stream synchronisation on the way out
    t_enqueued = std::chrono::steady_clock::now();
    rc = wait_for_direct(w, n);
    w.device_may_lag = true;
    if (rc != DVO_HIP_OK) return rc;
    t_done = std::chrono::steady_clock::now();
    if (w.host_status[w.resident_error_word] == 0) {
      std::memcpy(results, w.direct_results.p, size_t(n) * sizeof(dvo_hip_result));
      if (levels && cap_levels > 0) hl_src = w.direct_levels.as<dvo_hip_level_stats>();
      if (iters && cap_iters > 0) hi_src = w.direct_iters.as<dvo_hip_iteration_stats>();
    } else {
      DVO_WS_TRY(w, sync_stream(s));                           // every group has to be gone before the batch is repeated
    }
  } else {
    if (level_from >= cfg->last_level && policy.finish_launch(n)) launch_finish(s, states, n, bp.prm, d_levels, d_iters, w.results.as<dvo_hip_result>());
    // (else the results are in place: written by the resident launch, or by the solver steps behind the pairs' last level)
    DVO_WS_TRY(w, hipMemcpyAsync(results, w.results.p, size_t(n) * sizeof(dvo_hip_result), hipMemcpyDeviceToHost, s));
    if (levels && cap_levels > 0) {
      hl.resize(size_t(n) * bp.cap_levels);
      DVO_WS_TRY(w, hipMemcpyAsync(hl.data(), d_levels, hl.size() * sizeof(dvo_hip_level_stats), hipMemcpyDeviceToHost, s));
      hl_src = hl.data();
    }
    if (iters && cap_iters > 0) {
      hi.resize(size_t(n) * bp.cap_iters);
      DVO_WS_TRY(w, hipMemcpyAsync(hi.data(), d_iters, hi.size() * sizeof(dvo_hip_iteration_stats), hipMemcpyDeviceToHost, s));
      hi_src = hi.data();
    }
    t_enqueued = std::chrono::steady_clock::now();
    DVO_WS_TRY(w, sync_stream(s));
    DVO_WS_TRY(w, hipGetLastError());
    t_done = std::chrono::steady_clock::now();
  }
  if (resident_used && w.host_status[w.resident_error_word] != 0) {
    // A group of the resident kernel gave up waiting for its peers: its workgroups were not all on the device at once (the
    // device is shared with another process that does the same, or a compute-unit mask shrank it).  Nothing is wrong with the
    // batch: it runs again, one launch per step, and the context stops using groups.
    w.resident_sequence = 0xffffffffu;                       // the exchange rows are in an unknown state: cleared before the next use
    ctx->resident_timeouts += 1;
    // (not after a single incident: a launch that started late once -- the device busy with another process for a moment -- is no
    // reason to give up the latency path for the rest of the context's life)
    if (ctx->resident_timeouts >= 3) ctx->opt_resident_group = 1;
    for (int i = 0; i < n; ++i) std::memcpy(results[i].transformation, &tinit[size_t(i) * 16], 16 * sizeof(double));
    const int keep = ctx->opt_resident;
    ctx->opt_resident = 0;
    w.needs_drain = true;
    rc = ensure_batch_roles(ctx, n, refs, curs, cfg);          // (the launch path may read another flavour of the current planes)
    if (rc == DVO_HIP_OK) rc = run_batch(ctx, n, refs, curs, cfg, results, levels, cap_levels, iters, cap_iters);
    ctx->opt_resident = keep;
    return rc;
  }
  // Pairs in which a Jacobian component of some pixel was beyond +-65504: the f16 high / low split of the Gram operands (variants 7-9)
  // does not represent it, the sweep's epilogue has raised the pair's word.  Those pairs run again with the f32 Gram of the same sweep
  // family (variant 6): the whole batch when most of it is concerned (one chain of launches either way; the batches that follow then
  // start on the f32 Gram, kF32GramHoldBatches), otherwise ONLY the flagged pairs, as a batch of their own, behind the copy-out below
  // (round 5: a single close depth step used to repeat all 1024 pairs of a batch).
  std::vector<int> out_of_range;
  if (ctx->opt_variant >= 7)
    for (int i = 0; i < n; ++i)
      if (static_cast<volatile int*>(w.f16_range_flag)[i] != 0) {
        static_cast<volatile int*>(w.f16_range_flag)[i] = 0;
        out_of_range.push_back(i);
      }
  // (the repeats below prepare role planes and run another batch: two groups of one batch -- dvo_hip_match_batch -- do not do that at once)
  std::unique_lock<std::mutex> rare_path;
  if (!out_of_range.empty()) rare_path = std::unique_lock<std::mutex>(g_rare_path_mutex);
  if (!out_of_range.empty() && out_of_range.size() * 2 >= size_t(n)) {
    ctx->f16_range_repeats += (long long)out_of_range.size();
    ctx->f32_gram_hold = kF32GramHoldBatches;
    for (int i = 0; i < n; ++i) std::memcpy(results[i].transformation, &tinit[size_t(i) * 16], 16 * sizeof(double));
    ctx->opt_variant = 6;                                      // (restored by variant_scope)
    rc = ensure_batch_roles(ctx, n, refs, curs, cfg);          // (another sweep, maybe another flavour of the current planes)
    if (rc != DVO_HIP_OK) return rc;
    return run_batch(ctx, n, refs, curs, cfg, results, levels, cap_levels, iters, cap_iters);
  }
  bool truncated = false;
  for (int i = 0; i < n; ++i) {
    if (hl_src) {
      const int nl = std::min(results[i].n_levels, std::min(cap_levels, bp.cap_levels));
      std::memcpy(levels + size_t(i) * cap_levels, hl_src + size_t(i) * bp.cap_levels, size_t(nl) * sizeof(dvo_hip_level_stats));
      truncated |= results[i].n_levels > cap_levels;
    }
    if (hi_src) {
      const int ni = std::min(results[i].n_iterations_total, std::min(cap_iters, bp.cap_iters));
      std::memcpy(iters + size_t(i) * cap_iters, hi_src + size_t(i) * bp.cap_iters, size_t(ni) * sizeof(dvo_hip_iteration_stats));
      truncated |= results[i].n_iterations_total > cap_iters;
    }
  }
  {
    using std::chrono::duration_cast;
    using std::chrono::nanoseconds;
    ctx->host_ns[0] += duration_cast<nanoseconds>(t_launch - ctx->batch_entry).count();
    ctx->host_ns[1] += duration_cast<nanoseconds>(t_enqueued - t_launch).count();
    ctx->host_ns[2] += duration_cast<nanoseconds>(t_done - t_enqueued).count();
    ctx->host_ns[3] += duration_cast<nanoseconds>(std::chrono::steady_clock::now() - t_done).count();
    ctx->host_batches += 1;
  }
  w.needs_drain = false;
  if (!out_of_range.empty()) {
    // the flagged pairs again, f32 Gram, as a batch of their own; their records replace the ones just copied out
    const int m = int(out_of_range.size());
    ctx->f16_range_repeats += m;
    std::vector<dvo_hip_frame*> r2(m), c2(m);
    std::vector<dvo_hip_result> res2(m);
    for (int k = 0; k < m; ++k) {
      const int i = out_of_range[k];
      r2[k] = refs[i];
      c2[k] = curs[i];
      res2[k] = results[i];
      std::memcpy(res2[k].transformation, &tinit[size_t(i) * 16], 16 * sizeof(double));
    }
    const bool want_l = levels && cap_levels > 0, want_i = iters && cap_iters > 0;
    std::vector<dvo_hip_level_stats> l2(want_l ? size_t(m) * cap_levels : 0);
    std::vector<dvo_hip_iteration_stats> i2(want_i ? size_t(m) * cap_iters : 0);
    ctx->opt_variant = 6;                                      // (restored by variant_scope)
    rc = ensure_batch_roles(ctx, m, r2.data(), c2.data(), cfg);
    if (rc == DVO_HIP_OK)
      rc = run_batch(ctx, m, r2.data(), c2.data(), cfg, res2.data(), want_l ? l2.data() : nullptr, cap_levels, want_i ? i2.data() : nullptr, cap_iters);
    if (rc != DVO_HIP_OK && rc != DVO_HIP_ERR_CAPACITY) return rc;
    truncated |= rc == DVO_HIP_ERR_CAPACITY;
    for (int k = 0; k < m; ++k) {
      const int i = out_of_range[k];
      results[i] = res2[k];
      if (want_l) std::memcpy(levels + size_t(i) * cap_levels, l2.data() + size_t(k) * cap_levels, size_t(cap_levels) * sizeof(dvo_hip_level_stats));
      if (want_i) std::memcpy(iters + size_t(i) * cap_iters, i2.data() + size_t(k) * cap_iters, size_t(cap_iters) * sizeof(dvo_hip_iteration_stats));
    }
  }
  if (truncated) {
    w.err = "match: statistics arrays too small (results are valid)";
    return DVO_HIP_ERR_CAPACITY;
  }
  return DVO_HIP_OK;
}

// The Gram schedule of a batch, decided ONCE, before the planes of the roles are prepared for it (round-4 advisor finding: deciding it
// inside run_batch prepared two flavours of the current role for every new frame).  The default accumulates on the f16 matrix pipe
// from exact high + low operand pairs, which cannot represent a Jacobian component beyond +-65504; pairs that meet one are repeated
// with the f32 Gram (run_batch).  Two cases take the f32 Gram from the start: option "deterministic" (a pair's bits must not depend
// on whether ANOTHER pair of its batch left the f16 range), and the batches right after a batch MOST of whose pairs left it (a
// tracking sequence with a close depth step would otherwise pay twice on every frame; this makes the arithmetic of those batches
// depend on the context's history -- documented in dvo_hip.h, reset by option "variant").
struct EffectiveVariantScope {
  dvo_hip_context* c; int keep;
  explicit EffectiveVariantScope(dvo_hip_context* ctx) : c(ctx), keep(ctx->opt_variant) {
    if (c->opt_variant >= 7 && (c->opt_deterministic || c->f32_gram_hold > 0)) {
      if (c->f32_gram_hold > 0) c->f32_gram_hold -= 1;
      c->opt_variant = 6;
    }
  }
  ~EffectiveVariantScope() { c->opt_variant = keep; }
};

#include "capi_groups.inc"   // concurrent pair groups (option "batch_groups"): run_batch_grouped, destroy_group_workers

// preparation for the parity / measurement entry points
int prepare_single(dvo_hip_context* ctx, int n, dvo_hip_frame* const* refs, dvo_hip_frame* const* curs, const dvo_hip_config* cfg, BatchPlan& bp) {
  int rc = validate_batch(ctx, n, refs, curs, cfg);
  if (rc != DVO_HIP_OK) return rc;
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  rc = ensure_batch_roles(ctx, n, refs, curs, cfg, /*launch_path_only=*/true);
  if (rc != DVO_HIP_OK) return rc;
  make_plan(ctx, refs[0]->cam, cfg, n, bp);
  rc = prepare_buffers(ctx->ws[0], cfg, refs, curs, bp);
  if (rc == DVO_HIP_OK && sync_stream(ctx->stream) != hipSuccess) rc = DVO_HIP_ERR_HIP;
  if (rc != DVO_HIP_OK) ctx->err = ctx->ws[0].err;
  return rc;
}

}  // namespace

extern "C" {

int dvo_hip_get_counter(dvo_hip_context* ctx, const char* key, long long* value);   // (capi_options.inc, below)

const char* dvo_hip_version(void) { return "dvo_hip 0.1 (gfx950)"; }

namespace {
// The host CPU's _mm_rcp_ps as a table (option "ref_compat").  The instruction's result is probed, not assumed: over all 2^23
// mantissas of [1, 2) the smallest k is found for which the result depends on the leading k mantissa bits only (11 on the Intel
// Xeon this was developed on, 12 on the EPYC 9575F of the MI355X box), and the exact scaling with the exponent is checked on a
// sample.  A CPU whose instruction does not have that form (k > 16) is refused.
int build_rcp_table(dvo_hip_context* ctx) {
  auto rcp = [](float x) { return _mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(x))); };
  auto at = [](unsigned m) { unsigned b = 0x3f800000u | m; float x; std::memcpy(&x, &b, 4); return x; };
  // change points of the result along the mantissa: their common alignment gives k
  unsigned all_changes = 0;
  float prev = rcp(at(0));
  for (unsigned m = 1; m < (1u << 23); ++m) {
    const float r = rcp(at(m));
    if (r != prev) { all_changes |= m; prev = r; }
  }
  int shift = 0;
  while (shift < 23 && !((all_changes >> shift) & 1u)) ++shift;     // every change point is a multiple of 2^shift
  const int k = 23 - shift;
  if (k > 16) return fail(ctx, DVO_HIP_ERR_INVALID, "ref_compat: this CPU's _mm_rcp_ps is not a table on at most 16 mantissa bits");
  for (unsigned m = 0; m < (1u << 23); m += 4099)
    for (int e = -24; e <= 24; e += 3) {
      const float s = std::ldexp(1.0f, e);
      if (rcp(at(m) * s) != rcp(at(m)) / s) return fail(ctx, DVO_HIP_ERR_INVALID, "ref_compat: this CPU's _mm_rcp_ps does not scale exactly with the exponent");
    }
  std::vector<float> table(size_t(1) << k);
  for (unsigned i = 0; i < table.size(); ++i) table[i] = rcp(at(i << shift));
  // a 16-bit copy for the sweep that keeps the table in LDS: every value is in (0.5, 1], i.e. bits in [0x3f000000, 0x3f800000]; an
  // instruction with 11-12 bits of precision leaves the low mantissa bits zero (0x3f7ff800 is the OR of all results on the Intel Xeon
  // this was developed on).  Checked, not assumed; a table that does not pack, or one of more than 2^12 entries, is used from memory.
  bool packs = k <= 12;
  std::vector<uint16_t> packed(table.size());
  for (size_t i = 0; i < table.size() && packs; ++i) {
    unsigned b;
    std::memcpy(&b, &table[i], 4);
    packs = b >= 0x3f000000u && b <= 0x3f800000u && ((b - 0x3f000000u) & 0xffu) == 0;
    packed[i] = uint16_t((b - 0x3f000000u) >> 8);
  }
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  DVO_HIP_TRY(ctx, ctx->rcp_table.reserve(table.size() * (sizeof(float) + sizeof(uint16_t))));
  DVO_HIP_TRY(ctx, hipMemcpy(ctx->rcp_table.p, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice));
  if (packs)
    DVO_HIP_TRY(ctx, hipMemcpy(ctx->rcp_table.as<float>() + table.size(), packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  ctx->rcp_shift = shift;
  ctx->rcp_packed = packs ? 1 : 0;
  return DVO_HIP_OK;
}
}  // namespace

int dvo_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dvo_hip_context_create(int device, dvo_hip_context** out) {
  if (!out) return DVO_HIP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "no HIP device available (libdvo_hip has no CPU fallback)";
    return DVO_HIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return DVO_HIP_ERR_INVALID;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
    return DVO_HIP_ERR_HIP;
  }
  dvo_hip_context* ctx = new dvo_hip_context();
  ctx->device = device;
  if (hipDeviceGetAttribute(&ctx->compute_units, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->compute_units = 0;
  int rc = workspace_create(ctx, 0);
  if (rc != DVO_HIP_OK) {
    g_create_error = "context setup: " + ctx->err;
    workspace_destroy(ctx->ws[0]);
    delete ctx;
    return DVO_HIP_ERR_HIP;
  }
  ctx->stream = ctx->ws[0].stream;
  ctx->ws[0].tables = &ctx->tables;
  int prio_least = 0, prio_greatest = 0;
  e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&ctx->build_stream, hipStreamNonBlocking, prio_least);
  for (int i = 0; i < dvo_hip_context::kBuildRing && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ctx->build_events[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->upload_done, hipEventDisableTiming);
  if (e != hipSuccess) {
    g_create_error = std::string("context setup (build stream): ") + hipGetErrorString(e);
    dvo_hip_context_destroy(ctx);
    return DVO_HIP_ERR_HIP;
  }
  // Warm-up of the wait path.  In the FIRST GPU process on a fresh box the first stream wait of a batch has been seen to return 14-24 ms
  // after the device had finished (19 us by its own time stamps; DESIGN.md section 8: not under the profiler, not in later
  // processes, with polling as with hipStreamSynchronize, and skipping one wait only moved it to the next) -- a one-time cost of the
  // runtime's host-side wait machinery, not of this engine.  Every stream of the context therefore waits three times on a
  // trivial command here, where no caller is timing; the longest of those waits is kept (counter "warmup_wait_us"), so that a stall
  // that still shows up in a first match can be told from one that was absorbed here.
  {
    long long longest = 0;
    unsigned char scratch_byte[8] = {0};
    void* dev = nullptr;
    if (hipMalloc(&dev, 64) == hipSuccess) {
      for (hipStream_t st : {ctx->stream, ctx->build_stream, ctx->upload_stream})
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipMemsetAsync(dev, 0, 64, st);
          (void)hipMemcpyAsync(scratch_byte, dev, 8, hipMemcpyDeviceToHost, st);
          const auto t0 = std::chrono::steady_clock::now();
          (void)hipStreamSynchronize(st);
          longest = std::max<long long>(longest, std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
        }
      (void)hipFree(dev);
    }
    ctx->warmup_wait_us = longest;
  }
  // DVO_HIP_REF_COMPAT=1: the reference-compatible arithmetic for callers that cannot set options -- the reference's own, unmodified
  // programs linked against the facade (tests/dropin)
  if (const char* env = std::getenv("DVO_HIP_REF_COMPAT"))
    if (env[0] == '1' && dvo_hip_set_option(ctx, "ref_compat", 1) != DVO_HIP_OK) {
      g_create_error = ctx->err;
      dvo_hip_context_destroy(ctx);
      return DVO_HIP_ERR_INVALID;
    }
  *out = ctx;
  return DVO_HIP_OK;
}

void dvo_hip_context_destroy(dvo_hip_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  ctx->deferred.clear();                                       // (nobody is left to read what a recorded ingest would build)
  destroy_group_workers(ctx);
  if (ctx->upload_stream) (void)hipStreamSynchronize(ctx->upload_stream);
  if (ctx->build_stream) (void)hipStreamSynchronize(ctx->build_stream);
  for (Workspace& w : ctx->ws) workspace_destroy(w);
  if (ctx->upload_done) (void)hipEventDestroy(ctx->upload_done);
  if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
  for (hipEvent_t ev : ctx->build_events)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->build_stream) (void)hipStreamDestroy(ctx->build_stream);
  for (DevBuf& b : ctx->build_tbl) b.release();
  for (DevBuf* b : {&ctx->misc, &ctx->role_tbl_cur, &ctx->role_tbl_ref, &ctx->prep_tbl_cur, &ctx->prep_tbl_ref, &ctx->rcp_table}) b->release();
  for (DevBuf& b : ctx->upload_buf) b.release();
  for (const dvo_hip_context::PooledBlock& b : ctx->frame_pool) (void)hipFree(b.p);
  ctx->frame_pool.clear();
  ctx->tables.release();
  for (CameraGeom* c : ctx->cameras) {
    c->tables.release();
    delete c;
  }
  delete ctx;
}

const char* dvo_hip_last_error(const dvo_hip_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

void* dvo_hip_context_stream(dvo_hip_context* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

int dvo_hip_context_device(const dvo_hip_context* ctx) { return ctx ? ctx->device : -1; }

#include "capi_options.inc"   // dvo_hip_get_counter, dvo_hip_set_option


int dvo_hip_frame_create_raw(dvo_hip_context* ctx, int width, int height, const float K[4], const uint8_t* grey,
                             const uint16_t* raw_depth, float depth_scale, int levels, dvo_hip_frame** out) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || !out || !grey || !raw_depth || !K) return fail(ctx, DVO_HIP_ERR_INVALID, "frame_create_raw: null argument");
  size_t raw_off;
  dvo_hip_frame* f = nullptr;
  int rc = frame_alloc(ctx, width, height, K, levels, &f, &raw_off);
  if (rc != DVO_HIP_OK) return rc;
  const size_t n = size_t(width) * height;
  char* stage = f->pool.as<char>() + raw_off;
  uint16_t* d_raw = reinterpret_cast<uint16_t*>(stage);
  uint8_t* d_grey = reinterpret_cast<uint8_t*>(stage + n * 2);
  hipError_t e = hipMemcpyAsync(d_raw, raw_depth, n * 2, hipMemcpyHostToDevice, ctx->build_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_grey, grey, n, hipMemcpyHostToDevice, ctx->build_stream);
  if (e == hipSuccess) {
    const void* g[1] = {d_grey};
    const void* r[1] = {d_raw};
    rc = frames_build(ctx, 1, &f, g, r, depth_scale);
    if (rc == DVO_HIP_OK) e = sync_stream(ctx->build_stream);
  }
  if (e != hipSuccess) ctx->err = std::string("frame_create_raw: ") + hipGetErrorString(e);
  if (e != hipSuccess || rc != DVO_HIP_OK) {
    dvo_hip_frame_destroy(ctx, f);
    return rc != DVO_HIP_OK ? rc : DVO_HIP_ERR_HIP;
  }
  *out = f;
  return DVO_HIP_OK;
}

int dvo_hip_frame_create_raw_device(dvo_hip_context* ctx, int width, int height, const float K[4], const void* grey_dev,
                                    const void* raw_depth_dev, float depth_scale, int levels, dvo_hip_frame** out) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || !out || !grey_dev || !raw_depth_dev || !K) return fail(ctx, DVO_HIP_ERR_INVALID, "frame_create_raw_device: null argument");
  size_t raw_off;
  dvo_hip_frame* f = nullptr;
  int rc = frame_alloc(ctx, width, height, K, levels, &f, &raw_off);
  if (rc != DVO_HIP_OK) return rc;
  const void* g[1] = {grey_dev};
  const void* r[1] = {raw_depth_dev};
  rc = frames_build(ctx, 1, &f, g, r, depth_scale);
  if (rc != DVO_HIP_OK) {
    dvo_hip_frame_destroy(ctx, f);
    return rc;
  }
  *out = f;   // asynchronous (build stream): every later use of the frame is ordered after the build
  return DVO_HIP_OK;
}

namespace {

int check_prepare_args(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, int role, const dvo_hip_config* cfg, const char* who) {
  if (!ctx || n_frames < 1 || !frames || !cfg || (role != DVO_HIP_ROLE_CURRENT && role != DVO_HIP_ROLE_REFERENCE))
    return fail(ctx, DVO_HIP_ERR_INVALID, who);
  if (cfg->first_level < cfg->last_level || cfg->last_level < 0 || cfg->first_level >= kMaxLevels)
    return fail(ctx, DVO_HIP_ERR_INVALID, "need 0 <= last_level <= first_level < DVO_HIP_MAX_LEVELS");
  for (int i = 0; i < n_frames; ++i) {
    if (!frames[i]) return fail(ctx, DVO_HIP_ERR_INVALID, who);
    if (frames[i]->cam != frames[0]->cam || frames[i]->levels <= cfg->first_level)
      return fail(ctx, DVO_HIP_ERR_INVALID, "frames must share the camera and have first_level + 1 levels");
  }
  return DVO_HIP_OK;
}

// role planes of levels cfg->last_level .. cfg->first_level on the build stream (what is already there is skipped)
int prepare_roles(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, int role, const dvo_hip_config* cfg) {
  const bool ref = role == DVO_HIP_ROLE_REFERENCE;
  int want[kMaxLevels];
  for (int l = 0; l < kMaxLevels; ++l) want[l] = l < frames[0]->cam->levels ? eager_current_flavor(ctx, frames[0]->cam, l, n_frames) : kCurAB;
  float ithr = ref ? cfg->intensity_derivative_threshold : 0.0f, dthr = ref ? cfg->depth_derivative_threshold : 0.0f;
  std::vector<dvo_hip_frame*> speculative;
  if (ref && (ithr < 0.0f || dthr < 0.0f)) {
    // A negative threshold asks for a SPECULATIVE preparation (a caller that does not know the tracker the frame will meet: the facade's
    // buildAccelerationStructure): with the thresholds of the context's last match, and only for frames that hold no selection at
    // all -- a frame selected for other thresholds keeps its planes instead of alternating between two selections.
    ithr = ctx->last_sel_ithr; dthr = ctx->last_sel_dthr;
    for (int i = 0; i < n_frames; ++i) {
      bool any = false;
      for (int l = cfg->last_level; l <= cfg->first_level; ++l) any |= frames[i]->lv[l].selected;
      if (!any) speculative.push_back(frames[i]);
    }
    if (speculative.empty()) return DVO_HIP_OK;
    n_frames = int(speculative.size());
    frames = speculative.data();
  }
  const int rc = ensure_roles(ctx, n_frames, frames, ref ? 1 : 0, cfg->last_level, cfg->first_level, ithr, dthr, /*eager=*/true, want);
  if (rc != DVO_HIP_OK) return rc;
  DVO_HIP_TRY(ctx, hipGetLastError());
  return DVO_HIP_OK;
}

// ingest of device-resident raw planes, optionally straight into a role (role < 0: none)
int update_raw_device(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const void* const* grey_dev,
                      const void* const* raw_depth_dev, float depth_scale, int role, const dvo_hip_config* cfg, bool keep_raw_copy = true) {
  const bool ref = role == DVO_HIP_ROLE_REFERENCE;
  const int fused = role >= 0 && cfg->last_level == 0 ? (ref ? 1 : 0) : -1;   // level 0 is built in the same pass if it is used at all
  int rc = frames_build(ctx, n_frames, frames, grey_dev, raw_depth_dev, depth_scale, fused, ref ? cfg->intensity_derivative_threshold : 0.0f,
                        ref ? cfg->depth_derivative_threshold : 0.0f, keep_raw_copy);
  if (rc == DVO_HIP_OK && role >= 0) rc = prepare_roles(ctx, n_frames, frames, role, cfg);
  return rc;
}

// carry out the recorded ingests (option "defer_ingest"), oldest first; the first failure is returned, the list is empty afterwards
int flush_deferred(dvo_hip_context* ctx) {
  if (ctx->deferred.empty()) return DVO_HIP_OK;
  std::vector<dvo_hip_context::DeferredIngest> list;
  list.swap(ctx->deferred);
  int rc = DVO_HIP_OK;
  for (dvo_hip_context::DeferredIngest& d : list) {
    for (dvo_hip_frame* f : d.frames) f->deferred = 0;
    if (rc != DVO_HIP_OK) continue;
    ctx->deferred_ingests += 1;
    rc = update_raw_device(ctx, int(d.frames.size()), d.frames.data(), d.grey.data(), d.raw.data(), d.depth_scale, d.role, &d.cfg, d.keep_raw_copy);
  }
  return rc;
}

}  // namespace

int dvo_hip_frames_update_raw_device(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const void* const* grey_dev,
                                     const void* const* raw_depth_dev, float depth_scale) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || n_frames < 1 || !frames || !grey_dev || !raw_depth_dev) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw_device: null argument");
  for (int i = 0; i < n_frames; ++i)
    if (!frames[i] || !grey_dev[i] || !raw_depth_dev[i]) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw_device: null entry");
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return update_raw_device(ctx, n_frames, frames, grey_dev, raw_depth_dev, depth_scale, -1, nullptr);
}

// defer / keep_raw_copy: -1 = what the context's options say ("defer_ingest", "keep_raw_copy"), 0 / 1 = for this call only
static int update_raw_device_as(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const void* const* grey_dev,
                                const void* const* raw_depth_dev, float depth_scale, int role, const dvo_hip_config* cfg, int defer, int keep_raw_copy) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  int rc = check_prepare_args(ctx, n_frames, frames, role, cfg, "frames_update_raw_device_as: bad argument");
  if (rc != DVO_HIP_OK) return rc;
  if (!grey_dev || !raw_depth_dev) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw_device_as: null argument");
  for (int i = 0; i < n_frames; ++i)
    if (!grey_dev[i] || !raw_depth_dev[i]) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw_device_as: null entry");
  const bool keep = keep_raw_copy < 0 ? ctx->opt_keep_raw_copy != 0 : keep_raw_copy != 0;
  if (defer < 0 ? ctx->opt_defer_ingest != 0 : defer != 0) {
    dvo_hip_context::DeferredIngest d;
    d.frames.assign(frames, frames + n_frames);
    d.grey.assign(grey_dev, grey_dev + n_frames);
    d.raw.assign(raw_depth_dev, raw_depth_dev + n_frames);
    d.depth_scale = depth_scale;
    d.role = role;
    d.cfg = *cfg;
    d.keep_raw_copy = keep;
    for (int i = 0; i < n_frames; ++i) frames[i]->deferred = 1;
    ctx->deferred.push_back(std::move(d));
    return DVO_HIP_OK;
  }
  DVO_FLUSH_DEFERRED(ctx);                                     // (nothing overtakes a recorded ingest)
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return update_raw_device(ctx, n_frames, frames, grey_dev, raw_depth_dev, depth_scale, role, cfg, keep);
}

int dvo_hip_frames_update_raw_device_as(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const void* const* grey_dev,
                                        const void* const* raw_depth_dev, float depth_scale, int role, const dvo_hip_config* cfg) {
  return update_raw_device_as(ctx, n_frames, frames, grey_dev, raw_depth_dev, depth_scale, role, cfg, -1, -1);
}

int dvo_hip_frames_update_raw_device_as_ex(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const void* const* grey_dev,
                                           const void* const* raw_depth_dev, float depth_scale, int role, const dvo_hip_config* cfg, unsigned flags) {
  return update_raw_device_as(ctx, n_frames, frames, grey_dev, raw_depth_dev, depth_scale, role, cfg, (flags & DVO_HIP_INGEST_DEFER) ? 1 : 0,
                              (flags & DVO_HIP_INGEST_NO_RAW_COPY) ? 0 : 1);
}

int dvo_hip_flush_deferred(dvo_hip_context* ctx) {
  if (!ctx) return DVO_HIP_ERR_INVALID;
  std::unique_lock<std::recursive_mutex> guard(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  return DVO_HIP_OK;
}

// Streaming ingest from HOST memory: DMA of the raw planes into a transfer buffer on the upload stream, then the batched
// build on the build stream.  Returns at once; from pinned memory (dvo_hip_host_alloc) the transfers are truly asynchronous,
// from pageable memory the runtime stages them (correct, but the call then blocks for most of the copy).
static int update_raw_host(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const uint8_t* const* grey,
                           const uint16_t* const* raw_depth, float depth_scale, int role, const dvo_hip_config* cfg, int keep_raw_copy = -1) {
  if (!ctx || n_frames < 1 || !frames || !grey || !raw_depth) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw: null argument");
  for (int i = 0; i < n_frames; ++i) {
    if (!frames[i] || !grey[i] || !raw_depth[i]) return fail(ctx, DVO_HIP_ERR_INVALID, "frames_update_raw: null entry");
    if (frames[i]->cam != frames[0]->cam) return fail(ctx, DVO_HIP_ERR_INVALID, "frames of one build batch must share camera and levels");
  }
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = size_t(frames[0]->lv[0].w) * frames[0]->lv[0].h;
  const size_t slot_bytes = (n * 3 + 1) & ~size_t(1);           // per frame: [u16 depth][u8 grey], padded to an even size
  const unsigned b = ctx->upload_next++ % dvo_hip_context::kUploadRing;
  DevBuf& buf = ctx->upload_buf[b];
  // the previous contents of this buffer may still be read by the build they were uploaded for
  int rc = wait_for_ticket(ctx, ctx->upload_buf_seq[b], /*upload=*/true);
  if (rc != DVO_HIP_OK) return rc;
  if (buf.bytes < slot_bytes * size_t(n_frames)) {
    DVO_HIP_TRY(ctx, hipStreamSynchronize(ctx->build_stream));   // growing = free + malloc
    DVO_HIP_TRY(ctx, buf.reserve(slot_bytes * size_t(n_frames)));
  }
  char* base = buf.as<char>();
  std::vector<const void*> g(static_cast<size_t>(n_frames)), r(static_cast<size_t>(n_frames));
  for (int i = 0; i < n_frames;) {
    const char* hd = reinterpret_cast<const char*>(raw_depth[i]);
    if (reinterpret_cast<const char*>(grey[i]) != hd + n * 2) {   // separate planes: two transfers for this frame
      DVO_HIP_TRY(ctx, hipMemcpyAsync(base + slot_bytes * i, hd, n * 2, hipMemcpyHostToDevice, ctx->upload_stream));
      DVO_HIP_TRY(ctx, hipMemcpyAsync(base + slot_bytes * i + n * 2, grey[i], n, hipMemcpyHostToDevice, ctx->upload_stream));
      ++i;
      continue;
    }
    int j = i + 1;                                               // frames in the slot layout that follow each other in host memory
    while (j < n_frames && reinterpret_cast<const char*>(raw_depth[j]) == hd + slot_bytes * size_t(j - i) &&
           reinterpret_cast<const char*>(grey[j]) == reinterpret_cast<const char*>(raw_depth[j]) + n * 2)
      ++j;
    DVO_HIP_TRY(ctx, hipMemcpyAsync(base + slot_bytes * i, hd, slot_bytes * size_t(j - i), hipMemcpyHostToDevice, ctx->upload_stream));
    i = j;
  }
  for (int i = 0; i < n_frames; ++i) {
    r[size_t(i)] = base + slot_bytes * i;
    g[size_t(i)] = base + slot_bytes * i + n * 2;
  }
  DVO_HIP_TRY(ctx, hipEventRecord(ctx->upload_done, ctx->upload_stream));
  DVO_HIP_TRY(ctx, hipStreamWaitEvent(ctx->build_stream, ctx->upload_done, 0));
  rc = update_raw_device(ctx, n_frames, frames, g.data(), r.data(), depth_scale, role, cfg, keep_raw_copy < 0 ? ctx->opt_keep_raw_copy != 0 : keep_raw_copy != 0);
  ctx->upload_buf_seq[b] = ctx->build_seq;          // the newest ticket is behind every reader of the buffer
  return rc;
}

int dvo_hip_frames_update_raw(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const uint8_t* const* grey,
                              const uint16_t* const* raw_depth, float depth_scale) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  return update_raw_host(ctx, n_frames, frames, grey, raw_depth, depth_scale, -1, nullptr);
}

int dvo_hip_frames_update_raw_as(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const uint8_t* const* grey,
                                 const uint16_t* const* raw_depth, float depth_scale, int role, const dvo_hip_config* cfg) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  const int rc = check_prepare_args(ctx, n_frames, frames, role, cfg, "frames_update_raw_as: bad argument");
  if (rc != DVO_HIP_OK) return rc;
  return update_raw_host(ctx, n_frames, frames, grey, raw_depth, depth_scale, role, cfg);
}

int dvo_hip_frames_update_raw_as_ex(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, const uint8_t* const* grey,
                                    const uint16_t* const* raw_depth, float depth_scale, int role, const dvo_hip_config* cfg, unsigned flags) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  const int rc = check_prepare_args(ctx, n_frames, frames, role, cfg, "frames_update_raw_as_ex: bad argument");
  if (rc != DVO_HIP_OK) return rc;
  return update_raw_host(ctx, n_frames, frames, grey, raw_depth, depth_scale, role, cfg, (flags & DVO_HIP_INGEST_NO_RAW_COPY) ? 0 : 1);
}

int dvo_hip_upload_wait(dvo_hip_context* ctx) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx) return DVO_HIP_ERR_INVALID;
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  DVO_HIP_TRY(ctx, hipStreamSynchronize(ctx->upload_stream));
  return DVO_HIP_OK;
}

int dvo_hip_host_alloc(dvo_hip_context* ctx, size_t bytes, void** out) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  if (!ctx || !out || bytes == 0) return fail(ctx, DVO_HIP_ERR_INVALID, "host_alloc: bad argument");
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  DVO_HIP_TRY(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return DVO_HIP_OK;
}

void dvo_hip_host_free(dvo_hip_context* ctx, void* p) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  if (!p) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

int dvo_hip_frames_prepare(dvo_hip_context* ctx, int n_frames, dvo_hip_frame* const* frames, int role, const dvo_hip_config* cfg) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  const int rc = check_prepare_args(ctx, n_frames, frames, role, cfg, "frames_prepare: bad argument");
  if (rc != DVO_HIP_OK) return rc;
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  return prepare_roles(ctx, n_frames, frames, role, cfg);
}

int dvo_hip_frame_update_raw_device(dvo_hip_context* ctx, dvo_hip_frame* frame, const void* grey_dev, const void* raw_depth_dev,
                                    float depth_scale) {
  dvo_hip_frame* f[1] = {frame};
  const void* g[1] = {grey_dev};
  const void* r[1] = {raw_depth_dev};
  return dvo_hip_frames_update_raw_device(ctx, 1, f, g, r, depth_scale);
}

void dvo_hip_frame_destroy(dvo_hip_context* ctx, dvo_hip_frame* frame) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  if (!frame) return;
  if (ctx && !ctx->deferred.empty()) (void)flush_deferred(ctx);   // (a recorded ingest may name this frame)
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    ctx->build_tbl_frames.clear();   // a later frame may be given the same address
    if (frame->pool.p && ctx->frame_pool.size() < dvo_hip_context::kFramePoolMaxBlocks &&
        ctx->frame_pool_bytes + frame->pool.bytes <= dvo_hip_context::kFramePoolMaxBytes) {
      ctx->frame_pool.push_back({frame->pool.p, frame->pool.bytes});   // kept for the next frame of this size (see frame_pool)
      ctx->frame_pool_bytes += frame->pool.bytes;
      frame->pool.p = nullptr;
      frame->pool.bytes = 0;
    } else {
      (void)hipStreamSynchronize(ctx->stream);
      if (ctx->upload_stream) (void)hipStreamSynchronize(ctx->upload_stream);
      if (ctx->build_stream) (void)hipStreamSynchronize(ctx->build_stream);
    }
  }
  frame->pool.release();
  delete frame;
}

int dvo_hip_frame_info(const dvo_hip_frame* frame, int level, int* width, int* height, float K[4]) {
  if (!frame || level < 0 || level >= frame->levels) return DVO_HIP_ERR_INVALID;
  if (width) *width = frame->lv[level].w;
  if (height) *height = frame->lv[level].h;
  if (K) std::memcpy(K, frame->cam->K[level], 16);
  return DVO_HIP_OK;
}

int dvo_hip_frame_download_plane(dvo_hip_context* ctx, dvo_hip_frame* frame, int level, int plane, float* out) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || !frame || !out || level < 0 || level >= frame->levels || plane < 0 || plane > 5)
    return fail(ctx, DVO_HIP_ERR_INVALID, "frame_download_plane: bad argument");
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  dvo_hip_frame* one[1] = {frame};
  int rc = wait_for_build(ctx, 1, one);
  if (rc == DVO_HIP_OK) rc = ensure_roles(ctx, 1, one, 0, level, level, 0.0f, 0.0f);
  if (rc != DVO_HIP_OK) return rc;
  const FrameLevel& L = frame->lv[level];
  const size_t n = size_t(L.w) * L.h;
  DVO_HIP_TRY(ctx, ctx->misc.reserve(n * 4));
  launch_unpack_plane(ctx->stream, L.A, L.B, int(n), plane, ctx->misc.as<float>());
  DVO_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->misc.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
  DVO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return DVO_HIP_OK;
}

int dvo_hip_frame_select(dvo_hip_context* ctx, dvo_hip_frame* frame, int level, float ithr, float dthr, int* n_selected,
                         uint8_t* mask_or_null) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || !frame || level < 0 || level >= frame->levels) return fail(ctx, DVO_HIP_ERR_INVALID, "frame_select: bad argument");
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = size_t(frame->lv[level].w) * frame->lv[level].h;
  uint8_t* mask_dev = nullptr;
  if (mask_or_null) {
    DVO_HIP_TRY(ctx, ctx->misc.reserve(n));
    mask_dev = ctx->misc.as<uint8_t>();
  }
  dvo_hip_frame* one[1] = {frame};
  int rc = wait_for_build(ctx, 1, one);
  if (rc == DVO_HIP_OK) rc = ensure_roles(ctx, 1, one, 1, level, level, ithr, dthr);
  if (rc == DVO_HIP_OK && mask_dev) rc = ensure_roles(ctx, 1, one, 0, level, level, 0.0f, 0.0f);
  if (rc != DVO_HIP_OK) return rc;
  if (mask_dev) {   // the mask is not kept on the device: recompute it from the sampling planes
    FrameLevel& L = frame->lv[level];
    DVO_HIP_TRY(ctx, hipMemsetAsync(frame->sel_count + level, 0, sizeof(int), ctx->stream));
    launch_select_pack(ctx->stream, L.A, L.B, L.w * L.h, ithr, dthr, L.R, frame->sel_count + level, mask_dev);
  }
  int count = 0;
  DVO_HIP_TRY(ctx, hipMemcpyAsync(&count, frame->sel_count + level, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (mask_or_null) DVO_HIP_TRY(ctx, hipMemcpyAsync(mask_or_null, mask_dev, n, hipMemcpyDeviceToHost, ctx->stream));
  DVO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (n_selected) *n_selected = count;
  return DVO_HIP_OK;
}

int dvo_hip_match_batch(dvo_hip_context* ctx, int n_pairs, dvo_hip_frame* const* references, dvo_hip_frame* const* currents,
                        const dvo_hip_config* cfg, dvo_hip_result* results, dvo_hip_level_stats* levels, int cap_levels,
                        dvo_hip_iteration_stats* iters, int cap_iters) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  if (!results) return fail(ctx, DVO_HIP_ERR_INVALID, "match: results is null");
  int rc = validate_batch(ctx, n_pairs, references, currents, cfg);
  if (rc != DVO_HIP_OK) return rc;
  if (cfg->use_initial_estimate)
    for (int i = 0; i < n_pairs; ++i)
      for (int k = 0; k < 16; ++k)
        if (!std::isfinite(results[i].transformation[k]))
          return fail(ctx, DVO_HIP_ERR_INVALID, "match: provided initialization is NaN (dense_tracking.cpp:139)");
  ctx->batch_entry = std::chrono::steady_clock::now();
  DVO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // a recorded ingest (option "defer_ingest") of frames this batch aligns comes first; of other frames: behind the first launches (run_batch)
  if (!ctx->deferred.empty()) {
    bool mine = false;
    for (int i = 0; i < n_pairs && !mine; ++i) mine = references[i]->deferred || currents[i]->deferred;
    if (mine) DVO_FLUSH_DEFERRED(ctx);
  }
  EffectiveVariantScope effective_variant_scope(ctx);
  rc = ensure_batch_roles(ctx, n_pairs, references, currents, cfg);
  if (rc != DVO_HIP_OK) return rc;

  const int groups = batch_groups_of(ctx, n_pairs);
  if (groups > 1) {
    rc = run_batch_grouped(ctx, groups, n_pairs, references, currents, cfg, results, levels, cap_levels, iters, cap_iters);
  } else {
    rc = run_batch(ctx, n_pairs, references, currents, cfg, results, levels, cap_levels, iters, cap_iters);
    if (rc != DVO_HIP_OK) ctx->err = ctx->ws[0].err;
  }
  if (!ctx->deferred.empty()) {                                // (a path without launches to hide it behind, or a batch that ended early)
    const int rc_deferred = flush_deferred(ctx);
    if (rc == DVO_HIP_OK) rc = rc_deferred;
  }
  return rc;
}

int dvo_hip_match(dvo_hip_context* ctx, dvo_hip_frame* reference, dvo_hip_frame* current, const dvo_hip_config* cfg,
                  dvo_hip_result* result, dvo_hip_level_stats* levels, int cap_levels, dvo_hip_iteration_stats* iters, int cap_iters) {
  auto alone = [&]() {
    dvo_hip_frame* r[1] = {reference};
    dvo_hip_frame* c[1] = {current};
    return dvo_hip_match_batch(ctx, 1, r, c, cfg, result, levels, cap_levels, iters, cap_iters);
  };
  if (!ctx || !reference || !current || !cfg || !result || !ctx->opt_rendezvous) return alone();
  typedef dvo_hip_context::MatchRequest Request;
  constexpr int kExpectCalls = 16;                             // lone calls that still wait after the last meeting / collision (each
                                                               // wait is up to 60 us against a 0.18-0.36 ms match: 64 was too many)
  auto same_config = [](const dvo_hip_config& a, const dvo_hip_config& b) {   // (field by field: the padding bytes of a caller's struct are not its business)
    return a.first_level == b.first_level && a.last_level == b.last_level && a.max_iterations_per_level == b.max_iterations_per_level &&
           a.use_initial_estimate == b.use_initial_estimate && a.precision == b.precision && a.mu == b.mu &&
           a.intensity_derivative_threshold == b.intensity_derivative_threshold && a.depth_derivative_threshold == b.depth_derivative_threshold;
  };
  constexpr auto kPartnerWait = std::chrono::microseconds(60);
  Request me;
  me.reference = reference; me.current = current; me.cfg = cfg; me.result = result;
  me.levels = levels; me.cap_levels = cap_levels; me.iters = iters; me.cap_iters = cap_iters;
  Request* partner = nullptr;
  bool waiting = false;
  {
    std::lock_guard<std::mutex> lock(ctx->rendezvous_mutex);
    Request* w = ctx->rendezvous_waiting;
    if (w && w->current == current && same_config(*w->cfg, *cfg)) {
      partner = w;
      ctx->rendezvous_waiting = nullptr;
      w->state.store(1, std::memory_order_release);
      ctx->rendezvous_expect = kExpectCalls;
    } else if (ctx->rendezvous_busy_current == current) {
      ctx->rendezvous_expect = kExpectCalls;                   // a collision: the other caller of this frame is already on the device
    } else if (!w && ctx->rendezvous_expect > 0) {
      ctx->rendezvous_expect -= 1;
      ctx->rendezvous_waiting = &me;
      waiting = true;
    }
  }
  if (partner) {
    // this thread runs both: the partner's pair first (it arrived first), one batch, statistics through a common stride
    dvo_hip_frame* r[2] = {partner->reference, reference};
    dvo_hip_frame* c[2] = {partner->current, current};
    dvo_hip_result res[2] = {*partner->result, *result};
    const bool want_levels = (partner->levels && partner->cap_levels > 0) || (levels && cap_levels > 0);
    const bool want_iters = (partner->iters && partner->cap_iters > 0) || (iters && cap_iters > 0);
    const int cl = want_levels ? cfg->first_level - cfg->last_level + 1 : 0;
    const int ci = want_iters ? cl * std::max(cfg->max_iterations_per_level, 1) : 0;
    std::vector<dvo_hip_level_stats> lv(size_t(2) * std::max(cl, 0));
    std::vector<dvo_hip_iteration_stats> it(size_t(2) * std::max(ci, 0));
    int rc = dvo_hip_match_batch(ctx, 2, r, c, cfg, res, want_levels ? lv.data() : nullptr, cl, want_iters ? it.data() : nullptr, ci);
    Request* both[2] = {partner, &me};
    int rcs[2] = {rc, rc};
    if (rc != DVO_HIP_OK && rc != DVO_HIP_ERR_CAPACITY && rc != DVO_HIP_ERR_INVALID) {
      // a HIP or device error: both callers get it at once (two more attempts would only double the time to report a dead device)
      partner->rc = rc;
      partner->state.store(2, std::memory_order_release);
      return rc;
    }
    if (rc == DVO_HIP_ERR_INVALID) {
      // the merged batch was refused (one request's initial estimate is not finite, a frame of another context, ...): each request
      // runs alone and gets ITS OWN status -- a valid match does not fail because of the partner it happened to meet
      for (int k = 0; k < 2; ++k) {
        Request* q = both[k];
        dvo_hip_frame* r1[1] = {q->reference};
        dvo_hip_frame* c1[1] = {q->current};
        rcs[k] = dvo_hip_match_batch(ctx, 1, r1, c1, q->cfg, q->result, q->levels, q->cap_levels, q->iters, q->cap_iters);
      }
      partner->rc = rcs[0];
      partner->state.store(2, std::memory_order_release);
      return rcs[1];
    }
    for (int k = 0; k < 2; ++k) {
      Request* q = both[k];
      *q->result = res[k];
      rcs[k] = DVO_HIP_OK;
      if (q->levels && q->cap_levels > 0) {
        const int nl = std::min(res[k].n_levels, std::min(q->cap_levels, cl));
        std::memcpy(q->levels, lv.data() + size_t(k) * cl, size_t(std::max(nl, 0)) * sizeof(dvo_hip_level_stats));
        if (res[k].n_levels > q->cap_levels) rcs[k] = DVO_HIP_ERR_CAPACITY;
      }
      if (q->iters && q->cap_iters > 0) {
        const int ni = std::min(res[k].n_iterations_total, std::min(q->cap_iters, ci));
        std::memcpy(q->iters, it.data() + size_t(k) * ci, size_t(std::max(ni, 0)) * sizeof(dvo_hip_iteration_stats));
        if (res[k].n_iterations_total > q->cap_iters) rcs[k] = DVO_HIP_ERR_CAPACITY;
      }
    }
    {
      std::lock_guard<std::mutex> lock(ctx->rendezvous_mutex);
      ctx->rendezvous_pairs += 1;
    }
    partner->rc = rcs[0];
    partner->state.store(2, std::memory_order_release);
    return rcs[1];
  }
  if (waiting) {
    const auto t0 = std::chrono::steady_clock::now();
    while (me.state.load(std::memory_order_acquire) == 0) {
      if (std::chrono::steady_clock::now() - t0 > kPartnerWait) {
        std::lock_guard<std::mutex> lock(ctx->rendezvous_mutex);
        if (ctx->rendezvous_waiting == &me) {
          ctx->rendezvous_waiting = nullptr;
          waiting = false;                                     // nobody came: alone after all
        }
        break;                                                 // (else: taken in this very moment)
      }
    }
    if (waiting) {
      while (me.state.load(std::memory_order_acquire) != 2) std::this_thread::yield();   // the partner's thread runs the batch
      return me.rc;
    }
  }
  {
    std::lock_guard<std::mutex> lock(ctx->rendezvous_mutex);
    ctx->rendezvous_busy_current = current;
  }
  const int rc = alone();
  {
    std::lock_guard<std::mutex> lock(ctx->rendezvous_mutex);
    if (ctx->rendezvous_busy_current == current) ctx->rendezvous_busy_current = nullptr;
  }
  return rc;
}

int dvo_hip_level_iteration(dvo_hip_context* ctx, dvo_hip_frame* reference, dvo_hip_frame* current, int level, float ithr,
                            float dthr, const float T34[12], const float P_prev[4], int first_iteration_on_level,
                            dvo_hip_iteration_out* out, float* residuals_or_null) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!ctx || !reference || !current || !T34 || !P_prev || !out) return fail(ctx, DVO_HIP_ERR_INVALID, "level_iteration: null argument");
  dvo_hip_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.first_level = level;
  cfg.last_level = level;
  cfg.max_iterations_per_level = 1;
  cfg.intensity_derivative_threshold = ithr;
  cfg.depth_derivative_threshold = dthr;
  if (level < 0 || level >= reference->levels || level >= current->levels) return fail(ctx, DVO_HIP_ERR_INVALID, "level_iteration: bad level");
  dvo_hip_frame* r[1] = {reference};
  dvo_hip_frame* c[1] = {current};
  BatchPlan bp;
  int rc = prepare_single(ctx, 1, r, c, &cfg, bp);
  if (rc != DVO_HIP_OK) return rc;
  hipStream_t s = ctx->stream;
  LevelGeom g = bp.geom[level];
  if (residuals_or_null) g.compact = 0;                        // (by pixel: one pair at every pixel's place)
  const size_t npx = size_t(g.w) * g.h;
  DVO_HIP_TRY(ctx, ctx->misc.reserve(256 + sizeof(dvo_hip_iteration_out)));
  float* d_T = ctx->misc.as<float>();
  float* d_P = d_T + 12;
  dvo_hip_iteration_out* d_out = reinterpret_cast<dvo_hip_iteration_out*>(ctx->misc.as<char>() + 256);
  DVO_HIP_TRY(ctx, hipMemcpyAsync(d_T, T34, 48, hipMemcpyHostToDevice, s));
  DVO_HIP_TRY(ctx, hipMemcpyAsync(d_P, P_prev, 16, hipMemcpyHostToDevice, s));
  PairState* states = ctx->ws[0].states.as<PairState>();
  DVO_HIP_TRY(ctx, hipMemsetAsync(states, 0, sizeof(PairState), s));
  launch_set_fixed_state(s, states, g, d_T, d_P, first_iteration_on_level ? 1 : 0);
  const PairPtrs* pp = bp.pair_ptrs + size_t(level);
  launch_residual_reduce(s, ctx->opt_variant, bp.rpw[level], level == 0, g, pp, states, 1, ctx->ws[0].partials.as<float>(), ctx->ws[0].scratch.as<float2>(),
                         ctx->ws[0].win_fallbacks.as<unsigned long long>(), ctx->ws[0].f16_range_flag);
  launch_loglik(s, g, states, 1, ctx->ws[0].partials.as<float>(), ctx->ws[0].scratch.as<float2>(), ctx->ws[0].ll_partials.as<double>(), kLlBlocksPerPair);
  int n_sel = 0;
  DVO_HIP_TRY(ctx, hipMemcpyAsync(&n_sel, reference->sel_count + level, sizeof(int), hipMemcpyDeviceToHost, s));
  DVO_HIP_TRY(ctx, hipStreamSynchronize(s));
  if (*static_cast<volatile int*>(ctx->ws[0].f16_range_flag) != 0) {   // (see run_batch: beyond the f16 range, again with the f32 Gram)
    *static_cast<volatile int*>(ctx->ws[0].f16_range_flag) = 0;
    ctx->f16_range_repeats += 1;
    // geometry and plane flavours of the f32 schedule (a level whose width is no multiple of 64 is the gathering sweep's there)
    const int keep_variant = ctx->opt_variant;
    ctx->opt_variant = 6;
    rc = prepare_single(ctx, 1, r, c, &cfg, bp);
    ctx->opt_variant = keep_variant;
    if (rc != DVO_HIP_OK) return rc;
    g = bp.geom[level];
    pp = bp.pair_ptrs + size_t(level);
    launch_residual_reduce(s, 6, bp.rpw[level], level == 0, g, pp, states, 1, ctx->ws[0].partials.as<float>(), ctx->ws[0].scratch.as<float2>(),
                           ctx->ws[0].win_fallbacks.as<unsigned long long>());
    launch_loglik(s, g, states, 1, ctx->ws[0].partials.as<float>(), ctx->ws[0].scratch.as<float2>(), ctx->ws[0].ll_partials.as<double>(), kLlBlocksPerPair);
    DVO_HIP_TRY(ctx, hipStreamSynchronize(s));
  }
  launch_single_shot_out(s, g, ctx->ws[0].partials.as<float>(), ctx->ws[0].ll_partials.as<double>(), kLlBlocksPerPair, n_sel, d_out);
  DVO_HIP_TRY(ctx, hipMemcpyAsync(out, d_out, sizeof(dvo_hip_iteration_out), hipMemcpyDeviceToHost, s));
  if (residuals_or_null) DVO_HIP_TRY(ctx, hipMemcpyAsync(residuals_or_null, ctx->ws[0].scratch.p, npx * sizeof(float2), hipMemcpyDeviceToHost, s));
  DVO_HIP_TRY(ctx, hipStreamSynchronize(s));
  DVO_HIP_TRY(ctx, hipGetLastError());
  if (residuals_or_null) {
    // a pixel without a constraint is a NaN PAIR on this boundary; the contracted sweep marks it in the first component only (which
    // is what the log-likelihood pass tests)
    float* r = reinterpret_cast<float*>(residuals_or_null);
    for (size_t i = 0; i < npx; ++i)
      if (r[2 * i] != r[2 * i]) r[2 * i + 1] = r[2 * i];
  }
  return DVO_HIP_OK;
}

int dvo_hip_time_residual_kernel(dvo_hip_context* ctx, int n_pairs, dvo_hip_frame* const* references, dvo_hip_frame* const* currents,
                                 int level, int warm_iterations, int reps, float* avg_ms) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!avg_ms || reps < 1 || warm_iterations < 0) return fail(ctx, DVO_HIP_ERR_INVALID, "time_residual_kernel: bad argument");
  dvo_hip_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.first_level = level;
  cfg.last_level = level;
  cfg.max_iterations_per_level = warm_iterations + 2;
  cfg.precision = 0.0;                                      // the warm-up iterations are never cut short by the stopping rule
  BatchPlan bp;
  int rc = prepare_single(ctx, n_pairs, references, currents, &cfg, bp);
  if (rc != DVO_HIP_OK) return rc;
  Workspace& w = ctx->ws[0];
  rc = ensure_host_status(w, 8);
  if (rc != DVO_HIP_OK) { ctx->err = w.err; return rc; }
  hipStream_t s = ctx->stream;
  const LevelGeom& g = bp.geom[level];
  const PairPtrs* pp = bp.pair_ptrs + size_t(level) * bp.n;
  std::vector<double> tinit(size_t(bp.n) * 16, 0.0);
  for (int i = 0; i < bp.n; ++i)
    for (int k = 0; k < 4; ++k) tinit[size_t(i) * 16 + k * 5] = 1.0;
  DVO_HIP_TRY(ctx, hipMemcpyAsync(w.t_init.p, tinit.data(), tinit.size() * sizeof(double), hipMemcpyHostToDevice, s));
  ctx->tables.forget(w.t_init.p);
  PairState* states = w.states.as<PairState>();
  launch_init_pairs(s, states, bp.n, bp.prm, w.t_init.as<double>());
  launch_level_begin(s, states, bp.n, bp.prm, g, level, pp, w.lvl_stats.as<dvo_hip_level_stats>());
  auto sweep = [&]() {
    launch_residual_reduce(s, ctx->opt_variant, bp.rpw[level], level == 0, g, pp, states, bp.n, w.partials.as<float>(), w.scratch.as<float2>(),
                           w.win_fallbacks.as<unsigned long long>());
  };
  // `warm_iterations` Gauss-Newton steps first: the timed sweeps then run where the sweeps of a match run -- at the transform
  // the solver moved to, with the t-distribution weights on (first = 0) -- instead of at the identity with unit weights
  for (int it = 0; it < warm_iterations; ++it) {
    sweep();
    launch_loglik(s, g, states, bp.n, w.partials.as<float>(), w.scratch.as<float2>(), w.ll_partials.as<double>(), kLlBlocksPerPair);
    DVO_HIP_TRY(ctx, hipMemsetAsync(w.counters.p, 0, sizeof(unsigned long long), s));
    launch_solver_step(s, states, bp.n, bp.prm, g, w.partials.as<float>(), w.ll_partials.as<double>(), kLlBlocksPerPair, nullptr,
                       w.lvl_stats.as<dvo_hip_level_stats>(), w.it_stats.as<dvo_hip_iteration_stats>(), w.counters.as<unsigned long long>(),
                       w.host_status);
  }
  // a pair whose last warm-up step was rejected (log-likelihood not improved) left the level: the timed sweeps cover every pair
  if (warm_iterations > 0) launch_force_active(s, states, bp.n);
  hipEvent_t e0, e1;
  DVO_HIP_TRY(ctx, hipEventCreate(&e0));
  DVO_HIP_TRY(ctx, hipEventCreate(&e1));
  sweep();   // warm
  DVO_HIP_TRY(ctx, hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) sweep();
  DVO_HIP_TRY(ctx, hipEventRecord(e1, s));
  DVO_HIP_TRY(ctx, hipEventSynchronize(e1));
  float ms = 0;
  DVO_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  DVO_HIP_TRY(ctx, hipGetLastError());
  *avg_ms = ms / float(reps);
  return DVO_HIP_OK;
}

int dvo_hip_time_stream_mix(dvo_hip_context* ctx, int n_pairs, dvo_hip_frame* const* references, dvo_hip_frame* const* currents,
                            int level, int with_write, int reps, float* avg_ms) {
  std::unique_lock<std::recursive_mutex> guard;
  if (ctx) guard = std::unique_lock<std::recursive_mutex>(ctx->mutex);
  DVO_FLUSH_DEFERRED(ctx);
  if (!avg_ms || reps < 1) return fail(ctx, DVO_HIP_ERR_INVALID, "time_stream_mix: bad argument");
  dvo_hip_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.first_level = level;
  cfg.last_level = level;
  cfg.max_iterations_per_level = 1;
  BatchPlan bp;
  int rc = prepare_single(ctx, n_pairs, references, currents, &cfg, bp);
  if (rc != DVO_HIP_OK) return rc;
  hipStream_t s = ctx->stream;
  const LevelGeom& g = bp.geom[level];
  const PairPtrs* pp = bp.pair_ptrs + size_t(level) * bp.n;
  // without the write the kernel folds what it read into one float per workgroup (the partials buffer is large enough)
  float2* scratch = with_write ? ctx->ws[0].scratch.as<float2>() : nullptr;
  float* sink = ctx->ws[0].partials.as<float>();
  hipEvent_t e0, e1;
  DVO_HIP_TRY(ctx, hipEventCreate(&e0));
  DVO_HIP_TRY(ctx, hipEventCreate(&e1));
  const bool window_planes = level_uses_window(ctx, g.w, g.h) || level_uses_small(ctx, g.w, g.h);   // the planes the level's sweep really reads
  launch_stream_mix(s, pp, bp.n, g.w * g.h, scratch, sink, window_planes);   // warm
  DVO_HIP_TRY(ctx, hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) launch_stream_mix(s, pp, bp.n, g.w * g.h, scratch, sink, window_planes);
  DVO_HIP_TRY(ctx, hipEventRecord(e1, s));
  DVO_HIP_TRY(ctx, hipEventSynchronize(e1));
  float ms = 0;
  DVO_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  DVO_HIP_TRY(ctx, hipGetLastError());
  *avg_ms = ms / float(reps);
  return DVO_HIP_OK;
}

}  // extern "C"
