#!/bin/bash
# Round 6, visit j2: the tuned streaming loop split over 1 / 2 / 3 / 4 contexts (host threads) on one GPU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
timeout 900 python scripts/r6_groups.py 1024 1,2,2,4,1,2,3 20 256,128,256,64,256,192,96 > $O/groups_1024.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_1024.txt
timeout 600 python scripts/r6_groups.py 512 1,2,2,1,2 20 256,128,256,256,128 > $O/groups_512.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_512.txt
timeout 600 python scripts/r6_groups.py 256 1,2,1,2 20 256,128,256,128 > $O/groups_256.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_256.txt
timeout 600 python scripts/r6_groups.py 128 1,2,1,2 20 256,128,256,128 > $O/groups_128.txt 2>&1 < /dev/null; grep "pairs per step" $O/groups_128.txt
