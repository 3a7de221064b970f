// tests/cpp/shim_check.cpp -- the stand-ins the drop-in builds lean on (oracle/shim/tbb, oracle/shim/g2o), checked by themselves:
//   * tbb::parallel_reduce walks the SAME partition with and without threads (compile with / without -DDVO_SHIM_TBB_THREADS and compare
//     the printed leaves): leaves in range order, joins left to right, a split body per split;
//   * enumerable_thread_specific hands every concurrent branch its own element, the caller's branch reusing one;
//   * the g2o container: incidence sets, removeEdge, changeId, the Cauchy kernel and an EdgeSE3's chi2 on a hand-made graph.
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include <tbb/concurrent_queue.h>
#include <tbb/enumerable_thread_specific.h>
#include <tbb/mutex.h>
#include <tbb/parallel_reduce.h>
#include <tbb/tbb_thread.h>

#include <g2o/core/robust_kernel_impl.h>
#include <g2o/core/sparse_optimizer.h>
#include <g2o/types/slam3d/edge_se3.h>

struct Leaves {
  std::string log;                                        // "[b,e)" of every leaf this body (and the ones joined into it) saw, in order
  tbb::enumerable_thread_specific<int>* slots;
  explicit Leaves(tbb::enumerable_thread_specific<int>* s) : slots(s) {}
  Leaves(const Leaves& o, tbb::split) : slots(o.slots) {}
  void operator()(const tbb::blocked_range<int>& r) {
    log += "[" + std::to_string(r.begin()) + "," + std::to_string(r.end()) + ")";
    slots->local() += 1;
  }
  void join(Leaves& right) { log += right.log; }
};

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

int main() {
  // ---- partition ----
  for (int n : {1, 2, 5, 13, 28}) {
    for (int grain : {1, 3, 100}) {
      tbb::enumerable_thread_specific<int> slots;
      Leaves body(&slots);
      tbb::parallel_reduce(tbb::blocked_range<int>(0, n, grain), body);
      std::printf("n %d grain %d: %s slots %zu\n", n, grain, body.log.c_str(), slots.size());
      // the leaves cover [0, n) in order without gaps
      int expect = 0;
      size_t pos = 0;
      while (pos < body.log.size()) {
        int b = 0, e = 0, used = 0;
        CHECK(std::sscanf(body.log.c_str() + pos, "[%d,%d)%n", &b, &e, &used) == 2);
        CHECK(b == expect && e > b && e - b <= (grain > n ? n : grain));
        expect = e;
        pos += size_t(used);
      }
      CHECK(expect == n);
      CHECK(slots.size() >= 1 && slots.size() <= 4);
    }
  }
  // ---- queue / thread / mutex ----
  {
    tbb::concurrent_bounded_queue<int> q;
    int sum = 0;
    tbb::mutex m;
    {
      tbb::tbb_thread consumer([&] { for (;;) { int v; q.pop(v); if (v < 0) break; tbb::mutex::scoped_lock l(m); sum += v; } });
      for (int i = 1; i <= 100; ++i) q.push(i);
      q.push(-1);
      consumer.join();
    }
    CHECK(sum == 5050 && q.empty());
    tbb::mutex::scoped_lock a;
    CHECK(a.try_acquire(m));
    tbb::mutex::scoped_lock b;
    CHECK(!b.try_acquire(m));
  }
  // ---- g2o container ----
  {
    g2o::SparseOptimizer g;
    g2o::VertexSE3* v[3];
    for (int i = 0; i < 3; ++i) {
      v[i] = new g2o::VertexSE3();
      v[i]->setId(i + 1);
      Eigen::Isometry3d p = Eigen::Isometry3d::Identity();
      p(0, 3) = 0.1 * i;
      v[i]->setEstimate(p);
      CHECK(g.addVertex(v[i]));
    }
    CHECK(!g.addVertex(v[0]) && g.vertex(2) == v[1] && g.vertex(9) == 0);
    g2o::EdgeSE3* e = new g2o::EdgeSE3();
    e->setId(7);
    e->setVertex(0, v[0]);
    e->setVertex(1, v[2]);
    Eigen::Isometry3d m = Eigen::Isometry3d::Identity();
    m(0, 3) = 0.25;                                         // the estimates say 0.2: error 0.05 along x
    e->setMeasurement(m);
    Eigen::Matrix<double, 6, 6> info;
    info.setIdentity();
    info(0, 0) = 400.0;
    e->setInformation(info);
    g2o::RobustKernelCauchy* k = new g2o::RobustKernelCauchy();
    k->setDelta(5);
    e->setRobustKernel(k);
    CHECK(g.addEdge(e) && v[0]->edges().count(e) == 1 && v[2]->edges().count(e) == 1 && v[1]->edges().empty());
    CHECK(std::fabs(e->chi2() - 400.0 * 0.05 * 0.05) < 1e-12);
    Eigen::Vector3d rho;
    e->robustKernel()->robustify(e->chi2(), rho);
    CHECK(std::fabs(rho[1] - 1.0 / (1.0 + 1.0 / 25.0)) < 1e-15 && std::fabs(rho[0] - 25.0 * std::log(1.0 + 1.0 / 25.0)) < 1e-13);
    CHECK(g.changeId(v[1], -4) && g.vertex(-4) == v[1] && g.vertex(2) == 0 && v[1]->id() == -4);
    CHECK(g.optimize(10) == 0 && g.optimizeCalls() == 1);
    CHECK(g.removeEdge(e) && g.edges().empty() && v[0]->edges().empty() && v[2]->edges().empty());
  }
  std::printf(fails ? "shim_check: %d FAILED\n" : "shim_check: ok\n", fails);
  return fails ? 1 : 0;
}
