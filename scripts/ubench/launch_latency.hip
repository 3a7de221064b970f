// launch_latency.hip -- what a chain of dependent tiny kernels costs per kernel on this system: plain stream launches vs one
// hipGraph of the same chain (the Gauss-Newton loop of a single pair is ~64 such kernels).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_tiny(int* p, int n_blocks_work) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}

int main() {
  int* d;
  hipMalloc(&d, 4);
  hipMemset(d, 0, 4);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int chain = 64, reps = 50;
  for (int grid : {1, 128, 2400}) {
    // plain launches
    for (int i = 0; i < chain; ++i) k_tiny<<<grid, 256, 0, s>>>(d, grid);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < chain; ++i) k_tiny<<<grid, 256, 0, s>>>(d, grid);
      hipStreamSynchronize(s);
    }
    double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    // the same chain as a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; ++i) k_tiny<<<grid, 256, 0, s>>>(d, grid);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
    }
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * chain);
    std::printf("grid %5d: %.2f us per kernel in a stream chain of %d, %.2f us per kernel as one graph\n", grid, us_stream, chain, us_graph);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  // the Gauss-Newton loop's shape: short graphs (one or two iterations of 3 kernels) launched back to back, 63 kernels in all
  for (int per_graph : {3, 6, 9}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < per_graph; ++i) k_tiny<<<128, 256, 0, s>>>(d, 128);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    const int launches = 63 / per_graph;
    for (int i = 0; i < launches; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < launches; ++i) hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * launches * per_graph);
    std::printf("graphs of %d kernels, %d launches back to back: %.2f us per kernel\n", per_graph, launches, us);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
