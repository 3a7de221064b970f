// benchmark.cpp -- frame-to-frame visual odometry over a TUM RGB-D association file on the MI355X engine.
//
// The replay loop of the reference's benchmark driver (dvo_benchmark/src/benchmark.cpp:402-478; image loading as in
// dvo_benchmark/src/benchmark_slam.cpp:46-93) written against the facade in include/dvo/ and the readers in
// include/dvo_benchmark/: for every associated RGB-D frame, align it to its predecessor with dvo::DenseTracker,
// chain `trajectory = trajectory * relative` and print `stamp tx ty tz qx qy qz qw`.  The ROS parameter server is
// replaced by command-line flags carrying the names of launch/benchmark.yaml; visualisation is out of scope.
//
//   dvo_benchmark --rgbdpair_file <assoc.txt> [--groundtruth_file <groundtruth.txt>] [--trajectory_file <out.txt>]
//                 [--coarsest_level 3] [--finest_level 1] [--max_iterations 50] [--precision 1e-4] [--mu 0.05]
//                 [--use_initial_estimate 1] [--min_intensity_deriv 0] [--min_depth_deriv 0]
//                 [--fx 517.3 --fy 516.5 --ox 318.6 --oy 255.3]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include <dvo/dense_tracking.h>
#include <dvo_benchmark/file_reader.h>
#include <dvo_benchmark/groundtruth.h>
#include <dvo_benchmark/image_io.h>
#include <dvo_benchmark/rgbd_pair.h>
#include <dvo_benchmark/tools.h>

namespace {

// benchmark_slam.cpp:46-93: decode, BGR->grey, raw depth -> metres, hand both planes to the camera pyramid
dvo::core::RgbdImagePyramidPtr load(dvo::core::RgbdCameraPyramid& camera, const std::string& rgb_file, const std::string& depth_file) {
  const dvo_benchmark::PngImage rgb = dvo_benchmark::readPng(rgb_file), depth = dvo_benchmark::readPng(depth_file);
  if (rgb.empty() || depth.empty()) return dvo::core::RgbdImagePyramidPtr();
  return camera.create(dvo_benchmark::greyFloatFromPng(rgb), dvo_benchmark::depthFloatFromPng(depth, 1.0f / 5000.0f));
}

struct Options {
  std::map<std::string, std::string> kv;
  std::string str(const char* key, const char* dflt) const {
    std::map<std::string, std::string>::const_iterator it = kv.find(key);
    return it == kv.end() ? std::string(dflt) : it->second;
  }
  double num(const char* key, double dflt) const {
    std::map<std::string, std::string>::const_iterator it = kv.find(key);
    return it == kv.end() ? dflt : std::atof(it->second.c_str());
  }
};

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::strncmp(argv[i], "--", 2) != 0) { std::fprintf(stderr, "unexpected argument %s\n", argv[i]); return 2; }
    opt.kv[argv[i] + 2] = argv[i + 1];
  }
  const std::string pair_file = opt.str("rgbdpair_file", "");
  if (pair_file.empty()) { std::fprintf(stderr, "usage: %s --rgbdpair_file assoc.txt [--trajectory_file out.txt] ...\n", argv[0]); return 2; }

  // launch/benchmark.yaml
  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.FirstLevel = int(opt.num("coarsest_level", 3));
  cfg.LastLevel = int(opt.num("finest_level", 1));
  cfg.MaxIterationsPerLevel = int(opt.num("max_iterations", 50));
  cfg.Precision = opt.num("precision", 1e-4);
  cfg.Mu = opt.num("mu", 0.05);
  cfg.UseInitialEstimate = opt.num("use_initial_estimate", 1) != 0;
  cfg.IntensityDerivativeThreshold = float(opt.num("min_intensity_deriv", 0));
  cfg.DepthDerivativeThreshold = float(opt.num("min_depth_deriv", 0));
  if (!cfg.IsSane()) { std::fprintf(stderr, "coarsest_level must be >= finest_level\n"); return 2; }

  dvo_benchmark::FileReader<dvo_benchmark::RgbdPair> pair_reader(pair_file);
  pair_reader.skipComments();
  if (!pair_reader.next()) { std::fprintf(stderr, "%s: no association entries\n", pair_file.c_str()); return 1; }
  std::vector<dvo_benchmark::RgbdPair> pairs;
  pair_reader.readAllEntries(pairs);
  const std::string folder = pair_file.substr(0, pair_file.find_last_of("/") + 1);

  // initial pose: the ground truth at the first frame when a ground-truth file is given (benchmark.cpp:391-400)
  dvo::core::AffineTransformd trajectory, relative;
  trajectory.setIdentity();
  relative.setIdentity();
  const std::string gt_file = opt.str("groundtruth_file", "");
  if (!gt_file.empty()) {
    dvo_benchmark::FileReader<dvo_benchmark::Groundtruth> gt_reader(gt_file);
    gt_reader.skipComments();
    if (!gt_reader.next()) { std::fprintf(stderr, "%s: no ground-truth entries\n", gt_file.c_str()); return 1; }
    dvo_benchmark::findClosestEntry(gt_reader, pairs.front().RgbTimestamp());
    dvo_benchmark::toPoseEigen(gt_reader.entry(), trajectory);
  }

  std::ofstream trajectory_file;
  const std::string out_file = opt.str("trajectory_file", "");
  if (!out_file.empty()) trajectory_file.open(out_file.c_str());
  std::ostream& out = out_file.empty() ? std::cout : trajectory_file;
  out << std::setprecision(17);

  dvo::DenseTracker dense_tracker(cfg);
  dvo::core::RgbdCameraPyramid* camera = 0;
  dvo::core::RgbdImagePyramidPtr reference, current;
  double match_seconds = 0;
  size_t matches = 0, failures = 0;
  for (std::vector<dvo_benchmark::RgbdPair>::const_iterator it = pairs.begin(); it != pairs.end(); ++it) {
    if (!camera) {   // the reference hard-codes 640x480 fr1 intrinsics (benchmark_slam.cpp:384-389); scale them to the file
      const dvo_benchmark::PngImage first = dvo_benchmark::readPng(folder + it->RgbFile());
      if (first.empty()) { std::fprintf(stderr, "cannot read %s\n", (folder + it->RgbFile()).c_str()); return 1; }
      const float s = float(first.width) / 640.0f;
      dvo::core::IntrinsicMatrix intrinsics = dvo::core::IntrinsicMatrix::create(
          float(opt.num("fx", 517.3 * s)), float(opt.num("fy", 516.5 * s)), float(opt.num("ox", 318.6 * s)), float(opt.num("oy", 255.3 * s)));
      camera = new dvo::core::RgbdCameraPyramid(first.width, first.height, intrinsics);
      camera->build(cfg.getNumLevels());
    }
    reference = current;
    current = load(*camera, folder + it->RgbFile(), folder + it->DepthFile());
    if (!current) { std::fprintf(stderr, "skipping unreadable frame %s\n", it->RgbFile().c_str()); current = reference; continue; }
    if (reference) {
      const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      dvo::DenseTracker::Result result;
      result.Transformation = relative;            // last motion = initial guess when use_initial_estimate
      dense_tracker.match(*reference, *current, result);
      match_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      ++matches;
      if (result.isNaN()) {                        // Q16: failure is reported through the result, not the return value
        ++failures;
        relative.setIdentity();
      } else {
        relative = result.Transformation;
      }
      trajectory = trajectory * relative;          // benchmark.cpp:463
    }
    double q[4], m[16];
    dvo_benchmark::toQuaternion(trajectory, q);
    dvo::compat::affine_to_rowmajor(trajectory, m);
    out << it->RgbTimestamp() << " " << m[3] << " " << m[7] << " " << m[11] << " " << q[0] << " " << q[1] << " " << q[2] << " " << q[3] << " "
        << std::endl;
  }
  std::fprintf(stderr, "frames %zu matches %zu failures %zu  %.3f ms/match (incl. pyramid build + upload)\n", pairs.size(), matches, failures,
               matches ? 1e3 * match_seconds / double(matches) : 0.0);
  delete camera;
  return 0;
}
