// oracle/shim/g2o/core/estimate_propagator.h -- TEST INFRASTRUCTURE, see hyper_graph.h (included by keyframe_graph.cpp, not used).
#pragma once
