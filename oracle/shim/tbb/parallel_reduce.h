// oracle/shim/tbb/parallel_reduce.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  tbb::parallel_reduce as ONE of the executions
// TBB may choose, chosen to be reproducible: the range is halved down to its grain size (simple_partitioner), every split gets a
// split-constructed body, the halves are joined left to right.  With DVO_SHIM_TBB_THREADS the first two levels of the recursion run
// on threads of their own (four concurrent leaves: the reference's validators then call DenseTracker::match concurrently, like under
// TBB); without it the same tree is walked serially, so both builds of a caller see the same partition.
#pragma once
#include "blocked_range.h"
#include <cstdio>
#include <cstdlib>
#ifdef DVO_SHIM_TBB_THREADS
#include <thread>
#endif
namespace tbb {
namespace shim {
// which concurrent branch of a parallel_reduce the calling thread is (0 = the caller's own); enumerable_thread_specific keys on it
inline int& worker_slot() { static thread_local int slot = 0; return slot; }
template <typename R, typename B> void reduce_tree(R& r, B& body, int depth) {
  if (!r.is_divisible()) { body(r); return; }
  R right(r, split());
  B right_body(body, split());
#ifdef DVO_SHIM_TBB_THREADS
  if (depth < 2) {
    const int slot = worker_slot() + (1 << depth);
    std::thread t([&right, &right_body, depth, slot] { worker_slot() = slot; reduce_tree(right, right_body, depth + 1); });
    reduce_tree(r, body, depth + 1);
    t.join();
    body.join(right_body);
    return;
  }
#endif
  reduce_tree(r, body, depth + 1);
  reduce_tree(right, right_body, depth + 1);
  body.join(right_body);
}
}  // namespace shim
template <typename R, typename B> void parallel_reduce(const R& range, B& body) {
  R r(range);
  if (std::getenv("DVO_SHIM_TBB_TRACE")) std::fprintf(stderr, "tbb::parallel_reduce (stand-in): %zu items, grain %zu\n", r.size(), r.grainsize());
  shim::reduce_tree(r, body, 0);
}
}  // namespace tbb
