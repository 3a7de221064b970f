// tests/dropin/shim -- TEST INFRASTRUCTURE: RViz marker message header that keyframe_graph.cpp includes and does not use.
#pragma once
