// local_tracker_check.cpp -- runs include/dvo_slam/local_tracker.h over a TUM-layout folder the way KeyframeTracker drives the
// reference's LocalTracker (dvo_slam/src/keyframe_tracker.cpp:52-72, 216-246): first two frames open a local map, every further
// frame is update()d; a distance criterion on the keyframe alignment (keyframe_tracker.cpp:153-156) decides when a new keyframe
// is due.  Prints one line per frame: "<switched> <16 pose numbers>" for tests/test_local_tracker.py.
//   local_tracker_check <assoc.txt> <max_translational_distance>            distance criterion + an always-true second slot
//   local_tracker_check <assoc.txt> <max_translational_distance> selection  the KeyframeTracker criteria (dvo_slam/keyframe_selection.h)
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include <dvo_benchmark/file_reader.h>
#include <dvo_benchmark/image_io.h>
#include <dvo_benchmark/rgbd_pair.h>
#include <dvo_slam/keyframe_selection.h>
#include <dvo_slam/local_tracker.h>

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string assoc = argv[1];
  const double max_distance = std::atof(argv[2]);
  dvo_benchmark::FileReader<dvo_benchmark::RgbdPair> reader(assoc);
  reader.skipComments();
  std::vector<dvo_benchmark::RgbdPair> pairs;
  reader.readAllEntries(pairs);
  const std::string folder = assoc.substr(0, assoc.find_last_of("/") + 1);

  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();   // launch/benchmark.yaml
  cfg.FirstLevel = 3; cfg.LastLevel = 1; cfg.MaxIterationsPerLevel = 50; cfg.Precision = 1e-4; cfg.Mu = 0.05; cfg.UseInitialEstimate = true;

  const dvo_benchmark::PngImage first = dvo_benchmark::readPng(folder + pairs[0].RgbFile());
  const float s = float(first.width) / 640.0f;
  dvo::core::RgbdCameraPyramid camera(first.width, first.height, dvo::core::IntrinsicMatrix::create(517.3f * s, 516.5f * s, 318.6f * s, 255.3f * s));
  camera.build(cfg.getNumLevels());
  std::vector<dvo::core::RgbdImagePyramidPtr> frames;
  for (size_t k = 0; k < pairs.size(); ++k)
    frames.push_back(camera.create(dvo_benchmark::greyFloatFromPng(dvo_benchmark::readPng(folder + pairs[k].RgbFile())),
                                   dvo_benchmark::depthFloatFromPng(dvo_benchmark::readPng(folder + pairs[k].DepthFile()), 1.0f / 5000.0f)));

  dvo_slam::LocalTracker tracker;
  tracker.configure(cfg);
  int completed = 0, votes_cast = 0;
  const bool use_selection = argc >= 4 && std::string(argv[3]) == "selection";
  dvo_slam::KeyframeTrackerConfig selection_cfg;
  selection_cfg.MaxTranslationalDistance = max_distance;
  dvo_slam::KeyframeSelection selection(selection_cfg);
  if (use_selection) selection.install(tracker);
  if (!use_selection) tracker.addAcceptCallback([&](const dvo_slam::LocalTracker&, const dvo_slam::LocalTracker::TrackingResult&, const dvo_slam::LocalTracker::TrackingResult& r_keyframe) {
    double m[16];
    dvo::compat::affine_to_rowmajor(r_keyframe.Transformation, m);
    ++votes_cast;
    return std::sqrt(m[3] * m[3] + m[7] * m[7] + m[11] * m[11]) < max_distance;
  });
  if (!use_selection) tracker.addAcceptCallback([&](const dvo_slam::LocalTracker&, const dvo_slam::LocalTracker::TrackingResult&, const dvo_slam::LocalTracker::TrackingResult&) {
    ++votes_cast;     // a second slot: must be asked even when the first one already vetoed
    return true;
  });
  tracker.addMapCompleteCallback([&](const dvo_slam::LocalTracker&, const dvo_slam::LocalMap::Ptr&) { ++completed; });

  auto print_pose = [](int switched, const dvo::core::AffineTransformd& pose) {
    double m[16];
    dvo::compat::affine_to_rowmajor(pose, m);
    std::printf("%d", switched);
    for (int i = 0; i < 16; ++i) std::printf(" %.17g", m[i]);
    std::printf("\n");
  };
  dvo::core::AffineTransformd pose;
  tracker.initNewLocalMap(frames[0], frames[1]);
  tracker.getCurrentPose(pose);
  print_pose(0, pose);
  for (size_t k = 2; k < frames.size(); ++k) {
    const int before = completed;
    tracker.update(frames[k], pose);
    print_pose(completed - before, pose);
  }
  std::fprintf(stderr, "local maps completed %d, accept votes cast %d, measurements in the open map %zu\n", completed, votes_cast,
               tracker.getLocalMap()->measurements().size());
  return (use_selection || votes_cast == 2 * int(frames.size() - 2)) ? 0 : 4;
}
