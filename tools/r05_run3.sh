#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_cli.py -x -q -m gpu > gpurun_out/r05_run3_cli.txt 2>&1
echo "cli: $(tail -1 gpurun_out/r05_run3_cli.txt)"
python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_pipeline_after.txt 2>&1; tail -36 gpurun_out/r05_pipeline_after.txt
DERP_LIB=$PWD/facebook360_dep_amd/libderp_var_packed3.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_proposals or full_pyramid or config1_full or option_matrix" > gpurun_out/r05_run3_packed3_parity.txt 2>&1
echo "packed3 parity: $(tail -1 gpurun_out/r05_run3_packed3_parity.txt)"
VARIANTS_NO_PARITY=1 tools/variants.sh 2>&1 | tee gpurun_out/r05_run3_variants.txt
