cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sequence.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-single-frame > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/${TAG}_$name.err; return; }
  python - $name gpurun_out/${TAG}_$name.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
lv = d.get("level_ms_per_frame")
print("%-12s %7.1f Mpix/s  %7.2f ms/step  crc_ok=%s  levels %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("result_crc_matches_n1"), lv))
PY
}
run nolanes DERP_SEQ_LANES=0
run lanes256 DERP_SEQ_LANES=8
run lanes512 DERP_SEQ_LANES=8 DERP_SEQ_LANE_MAX_WIDTH=512
run lanes1024 DERP_SEQ_LANES=8 DERP_SEQ_LANE_MAX_WIDTH=1024
run lanes4 DERP_SEQ_LANES=4
run nolanes DERP_SEQ_LANES=0
