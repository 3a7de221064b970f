#pragma once
#include "block_solver.h"
