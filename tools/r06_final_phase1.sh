# Round-6 evidence, phase 1 (GPU box): the full GPU suite, the counters and traces of the default bench and of config 4,
# the start-up split, the reference's three-binary schedule, the plain --gpus N command with all ranks on the one GPU.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/r06_gpu_tests.txt | tail -2
bash tools/profile_round.sh r06 > gpurun_out/r06_profile.log 2>&1
bash tools/profile_cfg4.sh r06cfg4 > gpurun_out/r06cfg4_profile.log 2>&1
python tools/startup_split.py 6 > gpurun_out/r06_startup_split.txt 2>&1; tail -6 gpurun_out/r06_startup_split.txt
timeout 900 python tools/pipeline_timing.py cfg2 8 > gpurun_out/r06_pipeline_timing.txt 2>&1; tail -5 gpurun_out/r06_pipeline_timing.txt
DERP_BENCH_SINGLE_DEVICE=1 timeout 1200 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 > gpurun_out/r06_plain_gpus8.json 2> gpurun_out/r06_plain_gpus8.err; echo "gpus8 rc=$? stdout lines: $(wc -l < gpurun_out/r06_plain_gpus8.json)"
ls gpurun_out | grep -c r06
