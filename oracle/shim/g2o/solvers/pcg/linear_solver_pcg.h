// oracle/shim/g2o/solvers/pcg/linear_solver_pcg.h -- TEST INFRASTRUCTURE, see ../../core/sparse_optimizer.h: constructed, never run.
#pragma once
#include "../../core/block_solver.h"
namespace g2o { template <typename M> class LinearSolverPCG : public LinearSolver<M> {}; }
