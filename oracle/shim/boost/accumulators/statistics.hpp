#pragma once
#include "accumulators.hpp"
