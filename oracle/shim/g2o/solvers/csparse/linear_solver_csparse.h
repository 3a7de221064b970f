#pragma once
#include "../../core/block_solver.h"
namespace g2o { template <typename M> class LinearSolverCSparse : public LinearSolver<M> {}; }
