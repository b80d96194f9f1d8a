#!/bin/bash
# Developer A/B: for every facebook360_dep_amd/libderp_var_*.so run a parity subset and the bench (2 frames, no CPU
# legs); prints one line per library. Usage (on the GPU box): tools/variants.sh [extra bench.py flags]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in facebook360_dep_amd/libderp_var_*.so; do
  name=$(basename $lib .so)
  if [ -z "$VARIANTS_NO_PARITY" ]; then
    DERP_LIB=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu \
      -k "cost_map or brute_force or random_proposals or full_pyramid or config1_full or option_matrix or lean_atan2" > gpurun_out/var_$name.pytest 2>&1
    echo "$name parity: $(tail -1 gpurun_out/var_$name.pytest)"
  fi
  DERP_LIB=$PWD/$lib timeout 600 python bench.py --frames 2 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame "$@" > /tmp/v.json 2>/tmp/v.err || { echo "$lib FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
s = d["stage_ms_per_step"]
print("%-40s %8.1f Mpix/s  %7.2f ms/frame  pp0 %.2f ms  random %.1f  pingpong %.1f  brute %.2f bilateral %.1f" % (
    sys.argv[1], d["value"], d["ms_per_frame"], d["roofline"]["kernel_ms"], s["random_proposals"] / 2,
    s["ping_pong"] / 2, s["brute_force"] / 2, s["bilateral"] / 2))
PY
done
