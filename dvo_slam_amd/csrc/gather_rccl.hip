// gather_rccl.hip -- the one exchange of the multi-GPU path behind the C-ABI: an all-gather of fixed-size result records over RCCL
// (xGMI between the GPUs of a node), one process per GPU.
//
// The alignment shards by frame pair and needs no communication while it runs (the reference spreads independent
// match() calls over TBB workers and concatenates, dvo_slam/src/keyframe_graph.cpp:576-593).  What travels afterwards is 256 bytes per
// pair.  A C++ host -- what dvo_benchmark and keyframe_graph.cpp are -- calls this directly; bench.py does too (dvo_slam_amd/parallel.py,
// NativeRecordGatherer), so that the path a multi-GPU run exercises is the one a C++ caller gets.
//
// RCCL is bound at run time (dlopen of librccl.so.1: a process that already holds RCCL -- PyTorch ships its own copy under the same
// soname -- gets that very instance, a plain C++ host the one of the ROCm installation): libdvo_hip.so itself has no link-time
// dependency on it, and a single-GPU user never loads it.
//
// A gather is asynchronous and double-buffered: begin() stages the caller's records in pinned memory, copies them to the device,
// enqueues ncclAllGather and the copy back on the communicator's OWN stream, and returns; end() waits for that slot.  The records of
// step k travel while step k + 1 is aligned on the context's stream (the context stream is busy with the next batch by then).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/dvo_hip.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("DVO_HIP_RCCL_LIBRARY");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
      api.error = dlerror();
    }
    if (!api.handle) return;
    auto sym = [&](const char* s) {
      void* p = dlsym(api.handle, s);
      if (!p) api.error = std::string("librccl: missing symbol ") + s;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
      dlclose(api.handle);
      api.handle = nullptr;
    }
  });
  return api;
}

thread_local std::string g_comm_error;

}  // namespace

struct dvo_hip_comm {
  int device = 0, rank = 0, n_ranks = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  static const int kSlots = 2;
  struct Slot {
    void* host_in = nullptr;      // pinned: this rank's block
    void* host_out = nullptr;     // pinned: every rank's block, in rank order
    void* dev_in = nullptr;
    void* dev_out = nullptr;
    size_t block_bytes = 0;       // capacity per rank
    hipEvent_t done = nullptr;
    bool in_flight = false;
    size_t bytes_in_flight = 0;
  } slot[kSlots];
  unsigned next = 0;
  std::string err;
  std::mutex mutex;
};

namespace {

int comm_fail(dvo_hip_comm* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_comm_error = msg;
  return code;
}

void release_slot(dvo_hip_comm::Slot& s) {
  if (s.host_in) (void)hipHostFree(s.host_in);
  if (s.host_out) (void)hipHostFree(s.host_out);
  if (s.dev_in) (void)hipFree(s.dev_in);
  if (s.dev_out) (void)hipFree(s.dev_out);
  s.host_in = s.host_out = s.dev_in = s.dev_out = nullptr;
  s.block_bytes = 0;
}

}  // namespace

extern "C" {

const char* dvo_hip_comm_last_error(const dvo_hip_comm* comm) { return comm ? comm->err.c_str() : g_comm_error.c_str(); }

int dvo_hip_comm_get_unique_id(void* id) {
  if (!id) return DVO_HIP_ERR_INVALID;
  static_assert(sizeof(ncclUniqueId) == DVO_HIP_COMM_ID_BYTES, "the C-ABI carries RCCL's unique id as opaque bytes");
  RcclApi& api = rccl();
  if (!api.handle) return comm_fail(nullptr, DVO_HIP_ERR_NO_DEVICE, "RCCL is not available: " + api.error);
  ncclUniqueId uid;
  const ncclResult_t r = api.GetUniqueId(&uid);
  if (r != ncclSuccess) return comm_fail(nullptr, DVO_HIP_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
  std::memcpy(id, &uid, sizeof(uid));
  return DVO_HIP_OK;
}

int dvo_hip_comm_create(dvo_hip_context* ctx, const void* id, int rank, int n_ranks, dvo_hip_comm** out) {
  if (!out) return DVO_HIP_ERR_INVALID;
  *out = nullptr;
  if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return comm_fail(nullptr, DVO_HIP_ERR_INVALID, "comm_create: bad argument");
  RcclApi& api = rccl();
  if (!api.handle) return comm_fail(nullptr, DVO_HIP_ERR_NO_DEVICE, "RCCL is not available: " + api.error);
  const int device = dvo_hip_context_device(ctx);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return comm_fail(nullptr, DVO_HIP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  dvo_hip_comm* c = new dvo_hip_comm();
  c->device = device;
  c->rank = rank;
  c->n_ranks = n_ranks;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = api.CommInitRank(&c->comm, n_ranks, uid, rank);
  if (r != ncclSuccess) {
    const std::string msg = std::string("ncclCommInitRank: ") + api.GetErrorString(r);
    delete c;
    return comm_fail(nullptr, DVO_HIP_ERR_HIP, msg);
  }
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  for (dvo_hip_comm::Slot& s : c->slot)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
  if (e != hipSuccess) {
    const std::string msg = std::string("comm_create: ") + hipGetErrorString(e);
    dvo_hip_comm_destroy(c);
    return comm_fail(nullptr, DVO_HIP_ERR_HIP, msg);
  }
  *out = c;
  return DVO_HIP_OK;
}

void dvo_hip_comm_destroy(dvo_hip_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  for (dvo_hip_comm::Slot& s : c->slot) {
    release_slot(s);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int dvo_hip_comm_rank(const dvo_hip_comm* c) { return c ? c->rank : -1; }
int dvo_hip_comm_size(const dvo_hip_comm* c) { return c ? c->n_ranks : 0; }

int dvo_hip_gather_records_begin(dvo_hip_comm* c, const void* mine, size_t bytes_mine, size_t bytes_per_rank, int* ticket) {
  if (!c || !ticket || (!mine && bytes_mine) || bytes_mine > bytes_per_rank || bytes_per_rank == 0 || bytes_per_rank % 8 != 0)
    return comm_fail(c, DVO_HIP_ERR_INVALID, "gather_records_begin: bad argument (a block is a multiple of 8 bytes, bytes_mine <= bytes_per_rank)");
  std::lock_guard<std::mutex> lock(c->mutex);
  RcclApi& api = rccl();
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return comm_fail(c, DVO_HIP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  const int k = int(c->next++ % dvo_hip_comm::kSlots);
  dvo_hip_comm::Slot& s = c->slot[k];
  if (s.in_flight) return comm_fail(c, DVO_HIP_ERR_INVALID, "gather_records_begin: both slots are in flight (call gather_records_end first)");
  if (s.block_bytes < bytes_per_rank) {
    release_slot(s);
    e = hipHostMalloc(&s.host_in, bytes_per_rank, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&s.host_out, bytes_per_rank * size_t(c->n_ranks), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&s.dev_in, bytes_per_rank);
    if (e == hipSuccess) e = hipMalloc(&s.dev_out, bytes_per_rank * size_t(c->n_ranks));
    if (e != hipSuccess) {
      release_slot(s);
      return comm_fail(c, DVO_HIP_ERR_HIP, std::string("gather_records_begin: ") + hipGetErrorString(e));
    }
    s.block_bytes = bytes_per_rank;
  }
  std::memcpy(s.host_in, mine, bytes_mine);
  if (bytes_mine < bytes_per_rank) std::memset(static_cast<char*>(s.host_in) + bytes_mine, 0, bytes_per_rank - bytes_mine);   // (a rank with a smaller share: padded)
  e = hipMemcpyAsync(s.dev_in, s.host_in, bytes_per_rank, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return comm_fail(c, DVO_HIP_ERR_HIP, std::string("gather_records_begin: ") + hipGetErrorString(e));
  const ncclResult_t r = api.AllGather(s.dev_in, s.dev_out, bytes_per_rank / 8, ncclFloat64, c->comm, c->stream);
  if (r != ncclSuccess) return comm_fail(c, DVO_HIP_ERR_HIP, std::string("ncclAllGather: ") + api.GetErrorString(r));
  e = hipMemcpyAsync(s.host_out, s.dev_out, bytes_per_rank * size_t(c->n_ranks), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipEventRecord(s.done, c->stream);
  if (e != hipSuccess) return comm_fail(c, DVO_HIP_ERR_HIP, std::string("gather_records_begin: ") + hipGetErrorString(e));
  s.in_flight = true;
  s.bytes_in_flight = bytes_per_rank;
  *ticket = k;
  return DVO_HIP_OK;
}

int dvo_hip_gather_records_end(dvo_hip_comm* c, int ticket, void* all, size_t all_bytes) {
  if (!c || ticket < 0 || ticket >= dvo_hip_comm::kSlots || !all) return comm_fail(c, DVO_HIP_ERR_INVALID, "gather_records_end: bad argument");
  std::lock_guard<std::mutex> lock(c->mutex);
  dvo_hip_comm::Slot& s = c->slot[ticket];
  if (!s.in_flight) return comm_fail(c, DVO_HIP_ERR_INVALID, "gather_records_end: no gather in flight under this ticket");
  const size_t total = s.bytes_in_flight * size_t(c->n_ranks);
  if (all_bytes < total) return comm_fail(c, DVO_HIP_ERR_CAPACITY, "gather_records_end: the output holds fewer than n_ranks blocks");
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipEventSynchronize(s.done);
  s.in_flight = false;
  if (e != hipSuccess) return comm_fail(c, DVO_HIP_ERR_HIP, std::string("gather_records_end: ") + hipGetErrorString(e));
  std::memcpy(all, s.host_out, total);
  return DVO_HIP_OK;
}

int dvo_hip_gather_records(dvo_hip_comm* c, const void* mine, size_t bytes_mine, size_t bytes_per_rank, void* all, size_t all_bytes) {
  int ticket = -1;
  const int rc = dvo_hip_gather_records_begin(c, mine, bytes_mine, bytes_per_rank, &ticket);
  if (rc != DVO_HIP_OK) return rc;
  return dvo_hip_gather_records_end(c, ticket, all, all_bytes);
}

}  // extern "C"
