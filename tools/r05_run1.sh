#!/bin/bash
# round 5, GPU run 1: parity of the new random-proposal path + warp identity + scratch-free ping-pong, then A/B timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_tables or golden_fixture or cost_map or brute_force or random_proposals or full_pyramid or destination or config1_full or config2_rig or option_matrix or edge_cases or camera_types or non_square" > gpurun_out/r05_run1_parity.txt 2>&1
echo "parity: $(tail -1 gpurun_out/r05_run1_parity.txt)"
timeout 600 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu > gpurun_out/r05_run1_fullsize.txt 2>&1
echo "fullsize: $(tail -1 gpurun_out/r05_run1_fullsize.txt)"
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/r05_run1_variants.txt
for w in 2; do
  DERP_RANDOM_WAVES=$w DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_new.so python bench.py --frames 2 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/w.json 2>/tmp/w.err
  python - <<PY | tee -a gpurun_out/r05_run1_variants.txt
import json
d=json.load(open("/tmp/w.json")); s=d["stage_ms_per_step"]
print("new RANDOM_WAVES=$w  %.1f Mpix/s pp0 %.2f random %.1f pingpong %.1f" % (d["value"], d["roofline"]["kernel_ms"], s["random_proposals"]/2, s["ping_pong"]/2))
PY
done
for v in r4like new; do
  for w in 0 2; do
    DERP_RANDOM_WAVES=$w DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python bench.py --config cfg4 --frames 1 --temporal 0 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/c4.json 2>/tmp/c4.err || { echo cfg4 $v FAILED; tail -3 /tmp/c4.err; continue; }
    python - <<PY | tee -a gpurun_out/r05_run1_variants.txt
import json
d=json.load(open("/tmp/c4.json")); s=d["stage_ms_per_step"]
print("cfg4 $v waves=$w %.1f Mpix/s random %.1f pingpong %.1f proj_warp %.1f reproject %.1f" % (d["value"], s["random_proposals"], s["ping_pong"], s["proj_warp"], s["reproject"]))
PY
  done
done
