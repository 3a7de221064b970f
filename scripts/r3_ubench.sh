#!/bin/bash
# Round 3 micro-benchmarks on the GPU box: vector-ALU issue rate, the short division against the IEEE division.
mkdir -p scripts/ubench/_build gpurun_out/r3u
cd scripts/ubench
hipcc --offload-arch=gfx950 -O3 -o _build/valu_rate valu_rate.hip && _build/valu_rate | tee ../../gpurun_out/r3u/valu_rate.txt
hipcc --offload-arch=gfx950 -O3 -I../../dvo_slam_amd/csrc -o _build/div_check div_check.hip && _build/div_check | tee ../../gpurun_out/r3u/div_check.txt
