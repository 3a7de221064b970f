// dvo/core/point_selection_predicates.h -- the predicates live with PointSelection in this facade (point_selection.h); the
// reference's callers include this header by name (dvo_slam/src/local_tracker.cpp:24).
#pragma once

#include "point_selection.h"
