// tests/dropin/shim (see boost/thread/mutex.hpp)
#pragma once
#include <boost/thread/mutex.hpp>
