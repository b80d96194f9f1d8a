#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "destination or config1_full or config2_rig or option_matrix or edge_cases or camera_types or non_square or mismatch" > gpurun_out/r05_run2_parity.txt 2>&1
echo "parity: $(tail -1 gpurun_out/r05_run2_parity.txt)"
DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_union.so python tools/union_probe.py cfg2 2>&1 | tee gpurun_out/r05_run2_union.txt
for v in bias tiled; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python bench.py --config cfg4 --frames 1 --temporal 0 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/c4.json 2>/tmp/c4.err || { echo cfg4 $v FAILED; tail -3 /tmp/c4.err; continue; }
  python - <<PY | tee -a gpurun_out/r05_run2_cfg4.txt
import json
d=json.load(open("/tmp/c4.json")); s=d["stage_ms_per_step"]
print("cfg4 $v %.1f Mpix/s random %.1f pingpong %.1f proj_warp %.1f reproject %.1f" % (d["value"], s["random_proposals"], s["ping_pong"], s["proj_warp"], s["reproject"]))
PY
done
for v in new bias r4like; do
  for grp in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmc_v
    DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_v -o p -- python bench.py --frames 2 --steps 1 --warmup 0 --no-cpu-baseline --no-single-frame > /dev/null 2> gpurun_out/r05_run2_pmc_${v}_$tag.err
    python tools/pmc_summarize.py /tmp/pmc_v gpurun_out/r05_run2_pmc_${v}_$tag.json > /dev/null
  done
done
python - <<'PY' | tee gpurun_out/r05_run2_pmc.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_run2_pmc_*_*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if "k_random_proposals" in k or ("k_ping_pong" in k and "commit" not in k) or "k_reproject" in k:
            print(f.split("pmc_")[1], k[:40], {c:(x["max"], x["sum"]) for c,x in v.items()})
PY
