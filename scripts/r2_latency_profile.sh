#!/bin/bash
# Round-2 evidence for the latency path (the resident match kernel): A/B against the launch-per-step path over batch sizes with the
# host thread's time split, the cooperative-launch cost, the kernel's own stage clocks (experiment build), and the kernel timeline
# of one single-pair match on either path.  Everything under gpurun_out/r02_latency/.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_latency
mkdir -p $O
cd $R
timeout 300 python scripts/resident_ab.py 1 2 4 8 16 32 64 128 256 --host > $O/ab.txt 2>&1; cat $O/ab.txt | cut -c1-220
timeout 100 python scripts/resident_ab.py 1 2 --cooperative > $O/ab_cooperative.txt 2>&1; cat $O/ab_cooperative.txt | cut -c1-220
DVO_HIP_LIBRARY=$R/scripts/ubench/_build/libdvo_hip_clk.so timeout 100 python scripts/ubench/resident_clocks.py 1 > $O/clocks_1.txt 2>&1
DVO_HIP_LIBRARY=$R/scripts/ubench/_build/libdvo_hip_clk.so timeout 100 python scripts/ubench/resident_clocks.py 2 > $O/clocks_2.txt 2>&1
DVO_HIP_LIBRARY=$R/scripts/ubench/_build/libdvo_hip_clk.so timeout 100 python scripts/ubench/solver_clocks.py 1 > $O/solver_clocks_launch_path.txt 2>&1
cat $O/clocks_1.txt
for mode in 0 -1; do
  ( cd /tmp && DVO_RESIDENT=$mode timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof$mode -o one -- python $R/scripts/single_pair_trace.py 1 > $O/trace$mode.log 2>&1 )
  f=$(find $O/prof$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/timeline_last.py "$f" > $O/timeline_resident_$mode.txt 2>&1 && tail -12 $O/timeline_resident_$mode.txt
  rm -rf $O/prof$mode
done
timeout 100 python scripts/resident_ab_yaml.py > $O/ab_frontend_config.txt 2>&1; cat $O/ab_frontend_config.txt | cut -c1-200
