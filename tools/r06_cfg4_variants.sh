# cfg4 A/B of the developer variants in facebook360_dep_amd/libderp_var_*.so (one frame, 24 x 4096^2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for lib in facebook360_dep_amd/libderp_var_*.so; do
  name=$(basename $lib .so)
  for w in "" $EXTRA_WAVES; do
    DERP_RANDOM_WAVES=$w DERP_LIB=$PWD/$lib timeout 900 python bench.py --config cfg4 --frames 1 --temporal 0 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/v.json 2>/tmp/v.err || { echo "$name FAILED"; tail -3 /tmp/v.err; continue; }
    python - "$name waves=${w:-default}" <<'PY'
import json, sys
d = json.load(open("/tmp/v.json"))
s = d["stage_ms_per_step"]
print("cfg4 %-36s %7.1f Mpix/s  random %.1f pingpong %.1f proj_warp %.1f reproject %.1f bilateral %.1f" % (
    sys.argv[1], d["value"], s["random_proposals"], s["ping_pong"], s["proj_warp"], s["reproject"], s["bilateral"]))
PY
  done
done
