#!/bin/bash
# Developer A/B: build facebook360_dep_amd/libderp_var_<name>.so for every "<name>:<extra hipcc flags>" argument
# (tools/variants.sh then runs parity + bench on each through DERP_LIB). Runs here: hipcc cross-compiles gfx950.
#   tools/build_variants.sh "base:-DDERP_ATAN_LUT=0" "sc:-DDERP_SSD_SCALAR=1 -fno-slp-vectorize"
cd "$(dirname "$0")/.."
rm -f facebook360_dep_amd/libderp_var_*.so
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $flags \
      -o facebook360_dep_amd/libderp_var_$name.so facebook360_dep_amd/csrc/derp_capi.hip facebook360_dep_amd/csrc/derp_images.cpp -lz -ldl 2>&1 | grep -E "error|warning: v" ; echo "built $name ($flags)" ) &
done
wait
ls -la facebook360_dep_amd/libderp_var_*.so
