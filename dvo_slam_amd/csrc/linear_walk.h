// linear_walk.h -- a level walked as ONE row of w*h pixels in 64-pixel segments (the resident kernel always, the sweep kernel for
// levels narrower than a tile): row and column of flat pixel index idx without an integer division.
#pragma once

#include "hd_compat.h"

namespace dvo_hip {

// idx < 2^24 (the callers check): float(idx) is exact, and one float multiply by the rounded 1/w lands within one row of the true
// quotient, which the two corrections settle.  tests/test_emul_device.py checks every index of a set of level sizes on the host.
DVO_HD void locate_pixel(int idx, int w, float inv_w, int& row, int& col) {
  row = int(float(idx) * inv_w);
  col = idx - row * w;
  if (col < 0) { col += w; row -= 1; }
  if (col >= w) { col -= w; row += 1; }
}

}  // namespace dvo_hip
