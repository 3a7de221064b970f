#!/bin/bash
# Experiment build of libdvo_hip with wall-clock stamps inside k_solver_step (-DDVO_SOLVER_CLOCKS) and k_match_resident
# (-DDVO_RESIDENT_CLOCKS): scripts/ubench/_build/libdvo_hip_clk.so, used through DVO_HIP_LIBRARY by solver_clocks.py /
# resident_clocks.py.  Not part of the product.
set -e
cd "$(dirname "$0")/../../dvo_slam_amd/csrc"
make -s
OUT=../../scripts/ubench/_build
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS -DDVO_RESIDENT_CLOCKS -c align_resident.hip -o $OUT/align_resident_clk.o
/opt/rocm/bin/hipcc $FLAGS -DDVO_SOLVER_CLOCKS -c solver_kernels.hip -o $OUT/solver_kernels_clk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libdvo_hip_clk.so capi.o pyramid_kernels.o ingest_strips.o align_kernels.o align_mfma.o align_window.o align_fast.o \
    $OUT/solver_kernels_clk.o $OUT/align_resident_clk.o -L/opt/rocm/lib -lrocprofiler-sdk-roctx
