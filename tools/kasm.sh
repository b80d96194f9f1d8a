#!/bin/bash
# Developer loop: gfx950 assembly of one kernel of the library (device-only compile), into /tmp/<kernel>.s, plus where its
# scratch (spill) accesses sit. usage: tools/kasm.sh k_ping_pong [extra hipcc flags]
k=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S "$@" \
  -o /tmp/derp_dev.s facebook360_dep_amd/csrc/derp_capi.hip 2>/dev/null
awk -v k="$k" '$0 ~ "^_ZN4derp[0-9]+" k "E.*:" {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/derp_dev.s > /tmp/$k.s
echo "$(wc -l < /tmp/$k.s) lines in /tmp/$k.s; v_ $(grep -c '^\s*v_' /tmp/$k.s)  s_ $(grep -c '^\s*s_' /tmp/$k.s)  global/flat $(grep -cE '^\s*(global|flat)_' /tmp/$k.s)  ds_ $(grep -c '^\s*ds_' /tmp/$k.s)  scratch $(grep -c '^\s*scratch_' /tmp/$k.s)"
awk '/^\.LBB/ {lab=$1} /scratch_/ {print NR": "lab" "$1" "$2" "$3" "$4}' /tmp/$k.s | head -${KASM_N:-40}
