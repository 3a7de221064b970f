// io_check.cpp -- exercises include/dvo_benchmark/ (readers, PNG decoder, pose conversion) without a GPU: prints what
// it parsed so tests/test_tum.py can compare against the Python twin (dvo_slam_amd/tum.py).
//   io_check assoc <assoc.txt>            -> one line per entry
//   io_check gt <groundtruth.txt> <stamp> -> entry found by findClosestEntry + its pose, re-converted to a quaternion
//   io_check png <file>                   -> width height channels bits + sum and weighted checksum of the samples
//   io_check frame <rgb.png> <depth.png>  -> checksums of the float planes `load` produces
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>

#include <dvo_benchmark/file_reader.h>
#include <dvo_benchmark/groundtruth.h>
#include <dvo_benchmark/image_io.h>
#include <dvo_benchmark/rgbd_pair.h>
#include <dvo_benchmark/tools.h>

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  if (mode == "assoc") {
    dvo_benchmark::FileReader<dvo_benchmark::RgbdPair> r(argv[2]);
    r.skipComments();
    std::vector<dvo_benchmark::RgbdPair> all;
    r.readAllEntries(all);
    for (size_t i = 0; i < all.size(); ++i) std::cout << all[i];
  } else if (mode == "gt" && argc >= 4) {
    dvo_benchmark::FileReader<dvo_benchmark::Groundtruth> r(argv[2]);
    r.skipComments();
    if (!r.next()) return 1;
    const bool found = dvo_benchmark::findClosestEntry(r, dvo_benchmark::Time(std::atof(argv[3])));
    dvo::core::AffineTransformd T;
    dvo_benchmark::toPoseEigen(r.entry(), T);
    double m[16], q[4];
    dvo::compat::affine_to_rowmajor(T, m);
    dvo_benchmark::toQuaternion(T, q);
    std::printf("%d %.6f\n", int(found), r.entry().Timestamp().toSec());
    for (int i = 0; i < 16; ++i) std::printf("%.17g ", m[i]);
    std::printf("\n%.17g %.17g %.17g %.17g\n", q[0], q[1], q[2], q[3]);
  } else if (mode == "png") {
    const dvo_benchmark::PngImage p = dvo_benchmark::readPng(argv[2]);
    unsigned long long sum = 0, wsum = 0;
    for (int y = 0; y < p.height; ++y)
      for (int x = 0; x < p.width; ++x)
        for (int c = 0; c < p.channels; ++c) {
          const unsigned long long s = p.sample(x, y, c);
          sum += s;
          wsum = (wsum * 31 + s) % 1000000007ULL;
        }
    std::printf("%d %d %d %d %llu %llu\n", p.width, p.height, p.channels, p.bit_depth, sum, wsum);
  } else if (mode == "frame" && argc >= 4) {
    const dvo::compat::ImageMat g = dvo_benchmark::greyFloatFromPng(dvo_benchmark::readPng(argv[2]));
    const dvo::compat::ImageMat d = dvo_benchmark::depthFloatFromPng(dvo_benchmark::readPng(argv[3]), 1.0f / 5000.0f);
    double gs = 0, ds = 0;
    long nan = 0;
    const float* gp = dvo::compat::image_ptr(g);
    const float* dp = dvo::compat::image_ptr(d);
    const size_t n = size_t(dvo::compat::image_rows(g)) * dvo::compat::image_cols(g);
    for (size_t i = 0; i < n; ++i) {
      gs += gp[i] * double(i % 97 + 1);
      if (std::isnan(dp[i])) ++nan; else ds += dp[i] * double(i % 89 + 1);
    }
    std::printf("%.17g %.17g %ld\n", gs, ds, nan);
  } else {
    return 2;
  }
  return 0;
}
