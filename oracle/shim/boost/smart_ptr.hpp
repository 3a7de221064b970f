// oracle/shim/boost/smart_ptr.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  boost's smart pointers are the standard ones.
#pragma once
#include <memory>
namespace boost {
using std::shared_ptr;
using std::make_shared;
using std::enable_shared_from_this;
template <typename T> using scoped_ptr = std::unique_ptr<T>;
}  // namespace boost
