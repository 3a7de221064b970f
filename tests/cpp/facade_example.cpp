// facade_example.cpp -- the dvo_benchmark call pattern (dvo_benchmark/src/benchmark_slam.cpp:384-415, 486) against the
// header-only facade in include/dvo/.  Reads two float frames (intensity 0..255, depth metres) from a raw file written
// by tests/test_gpu_parity.py, aligns them with the reference's default API and prints the 4x4 result row-major.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <dvo/dense_tracking.h>

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s frames.raw width height first_level last_level\n", argv[0]); return 2; }
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<float> buf(size_t(w) * h * 4);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(buf.data(), sizeof(float), buf.size(), f) != buf.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  std::fclose(f);
  const size_t n = size_t(w) * h;
  // intrinsics scaled from the fr1 values the reference hard-codes
  const float s = float(w) / 640.0f;
  dvo::core::IntrinsicMatrix intrinsics = dvo::core::IntrinsicMatrix::create(517.3f * s, 516.5f * s, 318.6f * s, 255.3f * s);
  dvo::core::RgbdCameraPyramid camera(w, h, intrinsics);
  dvo::DenseTracker::Config cfg = dvo::DenseTracker::getDefaultConfig();
  cfg.FirstLevel = std::atoi(argv[4]);
  cfg.LastLevel = std::atoi(argv[5]);
  camera.build(cfg.getNumLevels());
  dvo::core::RgbdImagePyramidPtr reference = camera.create(dvo::compat::ImageMat(h, w, buf.data()), dvo::compat::ImageMat(h, w, buf.data() + n));
  dvo::core::RgbdImagePyramidPtr current = camera.create(dvo::compat::ImageMat(h, w, buf.data() + 2 * n), dvo::compat::ImageMat(h, w, buf.data() + 3 * n));
  dvo::DenseTracker tracker(cfg);
  dvo::DenseTracker::Result result;
  bool ok = tracker.match(*reference, *current, result);
  double m[16];
  dvo::compat::affine_to_rowmajor(result.Transformation, m);
  std::printf("ok %d nan %d levels %zu\n", int(ok), int(result.isNaN()), result.Statistics.Levels.size());
  for (int i = 0; i < 16; ++i) std::printf("%.17g%c", m[i], i % 4 == 3 ? '\n' : ' ');
  for (size_t l = 0; l < result.Statistics.Levels.size(); ++l)
    std::printf("level %zu valid %zu iterations %zu termination %d\n", result.Statistics.Levels[l].Id, result.Statistics.Levels[l].ValidPixels,
                result.Statistics.Levels[l].Iterations.size(), int(result.Statistics.Levels[l].TerminationCriterion));
  // Affine3d overload (dvo_ros/src/camera_dense_tracking.cpp:269)
  dvo::core::AffineTransformd T;
  tracker.match(*reference, *current, T);
  return 0;
}
