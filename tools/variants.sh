#!/bin/bash
# Developer A/B: run the bench (2 frames, no CPU legs) against every facebook360_dep_amd/libderp_var_*.so
cd $GRAFT_REPO_ROOT
for lib in facebook360_dep_amd/libderp_hip.so facebook360_dep_amd/libderp_var_*.so; do
  DERP_LIB=$PWD/$lib python bench.py --frames 2 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame "$@" > /tmp/v.json 2>/tmp/v.err || { echo "$lib FAILED"; tail -3 /tmp/v.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
s = d["stage_ms_per_step"]
print("%-48s %8.1f Mpix/s  %7.2f ms/frame  pp0 %.2f ms  random %.1f  pingpong %.1f  bilateral %.1f" % (
    sys.argv[1].split("/")[-1], d["value"], d["ms_per_frame"], d["roofline"]["kernel_ms"], s["random_proposals"] / 2,
    s["ping_pong"] / 2, s["bilateral"] / 2))
PY
done
