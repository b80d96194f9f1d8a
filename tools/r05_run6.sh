#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
PIPELINE_KEEP=1 python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_threads_auto.txt 2>&1
root=$(grep DATASET_ROOT= gpurun_out/r05_threads_auto.txt | cut -d= -f2)
for t in 16 12; do
  PIPELINE_DATASET=$root PIPELINE_THREADS=$t python tools/pipeline_timing.py cfg2 8 > gpurun_out/r05_threads_$t.txt 2>&1
done
for f in auto 16 12; do echo "== threads $f"; grep -E "^DerpCLI +[01] |^TemporalBilateralFilter +[01] |^schedule:|^DerpSequence:|^outputs" gpurun_out/r05_threads_$f.txt | cut -c1-230; done
