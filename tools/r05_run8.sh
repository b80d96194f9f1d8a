#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in pipe2 pipe2pk; do
DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_proposals or full_pyramid or config1_full" > gpurun_out/r05_run8_${v}_parity.txt 2>&1
echo "$v parity: $(tail -1 gpurun_out/r05_run8_${v}_parity.txt)"
done
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/r05_run8_variants.txt
for v in current pipe2; do
  DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_$v.so timeout 900 python bench.py --config cfg4 --frames 1 --temporal 0 --steps 2 --warmup 1 --no-cpu-baseline --no-single-frame > /tmp/c4.json 2>/tmp/c4.err || { echo cfg4 $v FAILED; tail -3 /tmp/c4.err; continue; }
  python - <<PY | tee -a gpurun_out/r05_run8_variants.txt
import json
d=json.load(open("/tmp/c4.json")); s=d["stage_ms_per_step"]
print("cfg4 $v %.1f Mpix/s random %.1f pingpong %.1f" % (d["value"], s["random_proposals"], s["ping_pong"]))
PY
done
