#!/bin/bash
# A/B on one box: build libdvo_hip.so of another commit into scripts/ubench/_build/base/libdvo_hip.so (used through DVO_HIP_LIBRARY).
#   scripts/build_base.sh <commit>
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
C=${1:-HEAD}
T=$(mktemp -d)
git -C "$R" archive "$C" dvo_slam_amd/csrc include | tar -x -C "$T"
mkdir -p "$T/dvo_slam_amd/lib"
make -C "$T/dvo_slam_amd/csrc" -s -j4
mkdir -p "$R/scripts/ubench/_build"
mkdir -p "$R/scripts/ubench/_build/base"
cp "$T/dvo_slam_amd/lib/libdvo_hip.so" "$R/scripts/ubench/_build/base/libdvo_hip.so"
cp "$R/dvo_slam_amd/lib/libdvo_stream.so" "$R/scripts/ubench/_build/base/"
rm -rf "$T"
echo "base = $(git -C "$R" log --oneline -1 "$C")"
