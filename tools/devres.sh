#!/bin/bash
# Developer loop: device-only compile of the library's kernels (seconds) and the register / spill / occupancy lines of the
# cost kernels. usage: tools/devres.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -c \
  -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/derp_dev.o facebook360_dep_amd/csrc/derp_capi.hip 2>&1 |
  awk '/error/ {print} /Function Name/ {n=$NF; sub(/.*Function Name: /,"",$0); name=$1} /VGPRs:|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill/ {if (name ~ /k_ping_pong[A-Z]|k_random_proposals|k_cost_map|k_brute_costs/) {v=$0; sub(/.*remark: +/,"",v); sub(/\[-Rpass.*/,"",v); printf "%-40s %s\n", substr(name,1,40), v}}'
