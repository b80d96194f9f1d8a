#!/bin/bash
# round 5, last GPU call: what the driver runs at round end (smoke, the GPU suite, the default bench) on the committed tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final_smoke.txt 2>&1; tail -2 gpurun_out/r05_final_smoke.txt
python bench.py --no-cpu-baseline > gpurun_out/r05_bench_final_check.json 2> gpurun_out/r05_bench_final_check.err
python -c "
import json
d=json.load(open('gpurun_out/r05_bench_final_check.json'))
r=d['roofline']
print('final check', d['value'], d['config2_single_frame']['value'], 'frac', r['frac'], r.get('issue_frac'), 'hbm', r.get('hbm_frac'), 'stale', r.get('stale'))
print('random', r['random_proposals']['frac'], r['random_proposals'].get('issue_frac'), r['random_proposals'].get('hbm_frac'), r['random_proposals'].get('l2'))
print(d['result_crc_matches_n1'])
"
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05_final_gpu_tests.txt 2>&1
echo "gpu tests: $(grep -E 'passed|failed' gpurun_out/r05_final_gpu_tests.txt | tail -1)"
