#pragma once
#include <functional>
namespace boost { using std::function; }
