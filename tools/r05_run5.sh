#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/r05_run5_variants.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -x -q -m gpu > gpurun_out/r05_run5_tests.txt 2>&1
echo "tests: $(grep -E 'passed|failed' gpurun_out/r05_run5_tests.txt | tail -1)"
bash tools/profile_round.sh r05
python -c "
import json
d=json.load(open('gpurun_out/r05_bench.json'))
print('bench', d['value'], d['ms_per_step'], d.get('config2_single_frame'), d['stage_ms_per_step'])
"
bash tools/profile_cfg4.sh r05cfg4
python -c "
import json
d=json.load(open('gpurun_out/r05cfg4_bench.json'))
print('cfg4', d['value'], d['stage_ms_per_step'])
"
