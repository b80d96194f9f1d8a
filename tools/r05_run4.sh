#!/bin/bash
# round 5, GPU run 4: the whole GPU suite, then the judged evidence (profile round on the final kernel sources)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --config small --frames 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r05_run4_small.json 2> gpurun_out/r05_run4_small.err || { echo "small bench FAILED"; tail -5 gpurun_out/r05_run4_small.err; }
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_run4_gpu_tests.txt 2>&1
echo "gpu tests: $(tail -1 gpurun_out/r05_run4_gpu_tests.txt)"
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
