// tests/dropin/shim -- TEST INFRASTRUCTURE: RViz marker server header that keyframe_graph.cpp includes and does not use.
#pragma once
