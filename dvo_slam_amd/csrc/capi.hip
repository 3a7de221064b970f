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

#include "capi_frames.inc"     // cameras, frame allocation and build, role planes (ensure_roles)
#include "capi_schedule.inc"   // batch plan, buffers, waits, resident / coarse plans, run_batch

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
