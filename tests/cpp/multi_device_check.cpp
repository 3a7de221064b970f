// multi_device_check.cpp -- dvo::DenseTracker::matchBatch over frames that live on SEVERAL device contexts inside one process
// (include/dvo/dense_tracking.h: one sub-batch per context, one host thread each, records concatenated in the caller's order --
// the reference's own model for independent pairs, dvo_slam/src/keyframe_graph.cpp:576-593, with GPUs in place of TBB workers).
//   multi_device_check <assoc.txt> <contexts>
// Frame pairs (k, k+1) of a TUM-layout folder are created round-robin on `contexts` contexts: device i mod
// dvo_hip_device_count() each -- on a one-GPU box they are independent contexts of device 0 (own streams and workspaces, run
// concurrently), on an 8-GPU node one context per GPU.  Prints one line per pair: 16 numbers of the transform, for
// tests/test_capi.py to compare with the single-context run.
#include <cstdio>
#include <string>
#include <vector>

#include <dvo/dense_tracking.h>
#include <dvo_benchmark/file_reader.h>
#include <dvo_benchmark/image_io.h>
#include <dvo_benchmark/rgbd_pair.h>

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string assoc = argv[1];
  const int n_contexts = std::atoi(argv[2]);
  dvo_benchmark::FileReader<dvo_benchmark::RgbdPair> reader(assoc);
  reader.skipComments();
  std::vector<dvo_benchmark::RgbdPair> entries;
  reader.readAllEntries(entries);
  const std::string folder = assoc.substr(0, assoc.find_last_of("/") + 1);

  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.FirstLevel = 3; cfg.LastLevel = 1; cfg.MaxIterationsPerLevel = 50; cfg.Precision = 1e-4;
  const dvo_benchmark::PngImage first = dvo_benchmark::readPng(folder + entries[0].RgbFile());
  const float s = float(first.width) / 640.0f;
  dvo::core::RgbdCameraPyramid camera(first.width, first.height, dvo::core::IntrinsicMatrix::create(517.3f * s, 516.5f * s, 318.6f * s, 255.3f * s));
  camera.build(cfg.getNumLevels());

  const int devices = dvo::core::DeviceContext::deviceCount();
  std::vector<dvo_hip_context*> contexts;
  for (int c = 0; c < n_contexts; ++c)
    contexts.push_back(c < devices ? dvo::core::DeviceContext::forDevice(c) : dvo::core::DeviceContext::createAdditional(c % devices));
  // sub-batches of different sizes would otherwise get different numbers of workgroups per pair in the resident kernel, i.e.
  // another summation order: pinned, like rows_per_wave for the launch path, so that the records can be compared to the bit
  for (size_t c = 0; c < contexts.size(); ++c)
    if (dvo_hip_set_option(contexts[c], "resident_group", 16) != DVO_HIP_OK) return 2;

  // pair p = (frame p, frame p + 1), both frames created on context p mod n (a frame shared by two pairs on different contexts
  // is uploaded to each, as SURVEY.md 8e prescribes)
  const size_t n_pairs = entries.size() - 1;
  std::vector<dvo::core::RgbdImagePyramidPtr> keep;
  std::vector<dvo::core::RgbdImagePyramid*> refs, curs;
  std::vector<dvo::DenseTracker::Result> results(n_pairs);
  std::vector<dvo::DenseTracker::Result*> out;
  for (size_t p = 0; p < n_pairs; ++p) {
    dvo::core::DeviceContext::Scope scope(contexts[p % contexts.size()]);
    for (size_t k = p; k <= p + 1; ++k)
      keep.push_back(camera.create(dvo_benchmark::greyFloatFromPng(dvo_benchmark::readPng(folder + entries[k].RgbFile())),
                                   dvo_benchmark::depthFloatFromPng(dvo_benchmark::readPng(folder + entries[k].DepthFile()), 1.0f / 5000.0f)));
    refs.push_back(keep[keep.size() - 2].get());
    curs.push_back(keep[keep.size() - 1].get());
    out.push_back(&results[p]);
  }
  dvo::DenseTracker tracker(cfg);
  tracker.matchBatch(refs, curs, out);
  for (size_t p = 0; p < n_pairs; ++p) {
    double m[16];
    dvo::compat::affine_to_rowmajor(results[p].Transformation, m);
    for (int i = 0; i < 16; ++i) std::printf("%s%.17g", i ? " " : "", m[i]);
    std::printf("\n");
    if (results[p].isNaN() || results[p].Statistics.Levels.size() != 3) return 4;
  }
  keep.clear();
  for (size_t c = 0; c < contexts.size(); ++c) {               // (stderr: did a workgroup group ever wait in vain?)
    long long timeouts = 0, launches = 0;
    dvo_hip_get_counter(contexts[c], "resident_timeouts", &timeouts);
    dvo_hip_get_counter(contexts[c], "resident_launches", &launches);
    std::fprintf(stderr, "context %zu: %lld resident launches, %lld time-outs\n", c, launches, timeouts);
  }
  for (int c = devices; c < n_contexts; ++c) dvo_hip_context_destroy(contexts[size_t(c)]);
  return 0;
}
