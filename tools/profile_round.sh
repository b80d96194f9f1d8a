#!/bin/bash
# Collect the round's judged evidence on the GPU box into gpurun_out/ (copy into profiles/ afterwards):
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
#   <tag>_pmc_FETCH_SIZE.json / _WRITE_SIZE.json   separate --pmc passes (per-kernel sum / max dispatch)
#   <tag>_bench.json         the un-profiled bench line
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2> /dev/null
cp /tmp/prof_$tag/${tag}_kernel_stats.csv gpurun_out/${tag}_kernel_stats_full.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> /dev/null
  python tools/pmc_summarize.py /tmp/pmc_$c gpurun_out/${tag}_pmc_$c.json
done
