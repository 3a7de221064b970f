// oracle/shim/boost/bind.hpp -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.
#pragma once
#include <functional>
namespace boost { using std::bind; }
using std::placeholders::_1;
using std::placeholders::_2;
using std::placeholders::_3;
using std::placeholders::_4;
