#!/bin/bash
# Registers, scratch, occupancy and LDS of every kernel of one translation unit (compiler's resource-usage remarks):
#   scripts/kernel_resources.sh align_fast.hip [extra hipcc flags]
cd "$(dirname "$0")/../dvo_slam_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unknown-pragmas -DDVO_WITH_ROCTX "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kr_$$.o 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); printf "%-8s vgpr %-4s agpr %-3s scratch %-4s waves/simd %-2s lds %-6s %s\n", "", v, a, s, o, l, name}' |
  while read -r line; do n=$(echo "$line" | awk '{print $NF}'); echo "$(echo "$line" | sed 's/ [^ ]*$//') $(echo "$n" | c++filt | sed 's/(.*//')"; done
rm -f /tmp/kr_$$.o
