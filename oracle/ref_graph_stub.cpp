// oracle/ref_graph_stub.cpp -- TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libdvo_ref.so, see ref_bridge.cpp).
//
// The reference's tracking front end (dvo_slam/src/keyframe_tracker.cpp, compiled unmodified) owns a KeyframeGraph, the pose-graph
// back end (dvo_slam/src/keyframe_graph.cpp: g2o optimisation, TBB worker thread, RViz markers).  The back end is outside this
// engine's scope; for the builds that do not link it (tests/dropin: benchmark_slam_graph does) this file defines the
// members of the class DECLARED in
// dvo_slam/include/dvo_slam/keyframe_graph.h:45-78 that the front end links against, as a sink that counts the completed local maps.
#include <cstdlib>

#include <dvo_slam/keyframe_graph.h>
#include <dvo_slam/visualization/graph_visualizer.h>

int g_ref_completed_local_maps = 0;

namespace dvo_slam {
namespace internal {
class KeyframeGraphImpl {
 public:
  KeyframeGraphConfig config;
  KeyframeVector keyframes;
  g2o::SparseOptimizer graph;
};
}  // namespace internal

KeyframeGraph::KeyframeGraph() : impl_(new internal::KeyframeGraphImpl()) {}
KeyframeGraph::~KeyframeGraph() {}
const KeyframeGraphConfig& KeyframeGraph::configuration() const { return impl_->config; }
void KeyframeGraph::configure(const KeyframeGraphConfig& config) { impl_->config = config; }
void KeyframeGraph::configureValidationTracking(const dvo::DenseTracker::Config&) {}
void KeyframeGraph::add(const LocalMap::Ptr&) { ++g_ref_completed_local_maps; }
void KeyframeGraph::finalOptimization() {}
void KeyframeGraph::addMapChangedCallback(const KeyframeGraph::MapChangedCallback&) {}
const KeyframeVector& KeyframeGraph::keyframes() const { return impl_->keyframes; }
const g2o::SparseOptimizer& KeyframeGraph::graph() const { return impl_->graph; }
cv::Mat KeyframeGraph::computeIntensityErrorImage(int, bool) const { std::abort(); }
void KeyframeGraph::debugLoopClosureConstraint(int, int) const { std::abort(); }

namespace visualization {
// (no-ops that do not touch `this`: benchmark_slam.cpp hands KeyframeTracker an UNINITIALISED GraphVisualizer* when no
// visualisation is configured, dvo_benchmark/src/benchmark_slam.cpp:358, 407 -- reference behaviour, left alone)
void GraphVisualizer::setGraph(KeyframeGraph*) {}
void GraphVisualizer::update() {}
}  // namespace visualization
}  // namespace dvo_slam
