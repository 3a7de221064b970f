// tests/dropin/graph_visualizer_stub.cpp -- TEST INFRASTRUCTURE: the two members of dvo_slam::visualization::GraphVisualizer the
// reference's KeyframeTracker calls (RViz display, out of scope), for the builds that link the reference's REAL keyframe graph
// (dvo_slam/src/keyframe_graph.cpp) instead of the counting sink oracle/ref_graph_stub.cpp, which carries the same two no-ops.
#include <dvo_slam/keyframe_graph.h>
#include <dvo_slam/visualization/graph_visualizer.h>

namespace dvo_slam {
namespace visualization {
// (they do not touch `this`: benchmark_slam.cpp hands KeyframeTracker an UNINITIALISED GraphVisualizer* when no visualisation is
// configured, dvo_benchmark/src/benchmark_slam.cpp:358, 407 -- reference behaviour, left alone)
void GraphVisualizer::setGraph(KeyframeGraph*) {}
void GraphVisualizer::update() {}
}  // namespace visualization
}  // namespace dvo_slam
