#!/bin/bash
# Round 5: the GPU-box scripts behind profiles/r05_kernel_variants.txt, r05_pipeline_timing.txt and the r05_* counter files,
# one function per gpurun call ("bash tools/r05_experiments.sh run2"). Variant libraries come from tools/build_variants.sh
# (the flags of each run are listed in profiles/r05_kernel_variants.txt); they are developer A/Bs, not part of the product.

run1() {
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
}

run2() {
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
}

run3() {
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r05_run3_cli.txt 2>&1
echo "cli: $(tail -1 gpurun_out/r05_run3_cli.txt)"
python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_pipeline_after.txt 2>&1; tail -36 gpurun_out/r05_pipeline_after.txt
DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_packed3.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_proposals or full_pyramid or config1_full or option_matrix" > gpurun_out/r05_run3_packed3_parity.txt 2>&1
echo "packed3 parity: $(tail -1 gpurun_out/r05_run3_packed3_parity.txt)"
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/r05_run3_variants.txt
}

run4() {
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
}

run5() {
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
}

run6() {
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PIPELINE_KEEP=1 python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_threads_auto.txt 2>&1
root=$(grep DATASET_ROOT= gpurun_out/r05_threads_auto.txt | cut -d= -f2)
for t in 16 12; do
  PIPELINE_DATASET=$root PIPELINE_THREADS=$t python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_threads_$t.txt 2>&1
done
for f in auto 16 12; do echo "== threads $f"; grep -E "^DerpCLI +[01] |^TemporalBilateralFilter +[01] |^schedule:|^DerpSequence:|^outputs" gpurun_out/r05_threads_$f.txt | cut -c1-230; done
}

run8() {
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
}

final() {
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
}

"$@"
