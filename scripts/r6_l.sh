#!/bin/bash
# Round 6, visit l: is the gain of several contexts on one GPU the groups' running free (out of phase), or their running side by side?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06l
mkdir -p $O
cd $R
timeout 900 python scripts/r6_groups.py 1024 1,2,3,1,2,3 20 256,192,96,256,192,96 > $O/groups_free.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_free.txt
timeout 900 python scripts/r6_groups.py 1024 2,3,2,3 20 192,96,192,96 sync > $O/groups_sync.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_sync.txt
