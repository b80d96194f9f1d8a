cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-single-frame > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/${TAG}_$name.err; return; }
  python - $name gpurun_out/${TAG}_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-12s %7.1f Mpix/s  %8.2f ms/step  crc_ok=%s  lanes_wall %.1f ms/step  levels %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("result_crc_matches_n1"), d["stage_ms_per_step"].get("coarse_levels_on_lanes", 0), d["level_ms_per_frame"][3:]))
PY
}
for rep in 1 2; do
run spans DERP_SEQ_LANE_SPANS=1
run muted X=1
run muted512 DERP_SEQ_LANE_MAX_WIDTH=512
done
