// dvo_benchmark/tools.h -- pose conversion and timestamp lookup used by the benchmark drivers
// (reference: dvo_benchmark/include/dvo_benchmark/tools.h:51-106).
#pragma once

#include <cmath>
#include <stdexcept>

#include "dvo/core/datatypes.h"
#include "dvo_benchmark/file_reader.h"
#include "dvo_benchmark/groundtruth.h"

namespace dvo_benchmark {

// quaternion (x, y, z, w) + position -> rigid transform
inline void toPoseEigen(const Groundtruth& gt, dvo::core::AffineTransformd& pose) {
  double x = gt.OrientationX(), y = gt.OrientationY(), z = gt.OrientationZ(), w = gt.OrientationW();
  const double n = std::sqrt(x * x + y * y + z * z + w * w);
  x /= n; y /= n; z /= n; w /= n;
  const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
                       2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                       2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  const double t[3] = {gt.PositionX(), gt.PositionY(), gt.PositionZ()};
  double m[16] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2], 0, 0, 0, 1};
  dvo::compat::affine_from_rowmajor(m, pose);
}

// rigid transform -> quaternion (x, y, z, w), w >= 0: what the trajectory writer prints (benchmark_slam.cpp:490-504)
inline void toQuaternion(const dvo::core::AffineTransformd& pose, double q[4]) {
  double m[16];
  dvo::compat::affine_to_rowmajor(pose, m);
  const double r00 = m[0], r11 = m[5], r22 = m[10], tr = r00 + r11 + r22;
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[0] = (m[9] - m[6]) / s; q[1] = (m[2] - m[8]) / s; q[2] = (m[4] - m[1]) / s; q[3] = 0.25 * s;
  } else {
    const int i = (r00 >= r11 && r00 >= r22) ? 0 : (r11 >= r22 ? 1 : 2), j = (i + 1) % 3, k = (i + 2) % 3;
    const double s = std::sqrt(1.0 + m[i * 4 + i] - m[j * 4 + j] - m[k * 4 + k]) * 2;
    q[i] = 0.25 * s;
    q[j] = (m[j * 4 + i] + m[i * 4 + j]) / s;
    q[k] = (m[k * 4 + i] + m[i * 4 + k]) / s;
    q[3] = (m[k * 4 + j] - m[j * 4 + k]) / s;
  }
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
}

// Forward-only scan: afterwards reader.entry() is the first record whose stamp is >= reference (or the last record
// of the file).  false once the file ran out.
template <class EntryT>
bool findClosestEntry(FileReader<EntryT>& reader, const Time& reference) {
  if (reader.entry().Timestamp() >= reference) return true;
  bool more = reader.next();
  while (more && reader.entry().Timestamp() < reference) more = reader.next();
  return more;
}

// the records bracketing `reference`; the reader must currently be before it
template <class EntryT>
bool findClosestEntries(FileReader<EntryT>& reader, const Time& reference, EntryT& before, EntryT& after) {
  if (reader.entry().Timestamp() >= reference) throw std::runtime_error("findClosestEntries: reader is already past the reference stamp");
  bool more;
  do {
    before = reader.entry();
    more = reader.next();
  } while (more && reader.entry().Timestamp() < reference);
  after = reader.entry();
  return more;
}

}  // namespace dvo_benchmark
