#pragma once
#include "smart_ptr.hpp"
