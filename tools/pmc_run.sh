#!/bin/bash
# Developer tool: collect PMC counter groups for the bench (one rocprofv3 pass per group) and
# summarise per kernel into gpurun_out/pmc_<tag>.json.   usage: tools/pmc_run.sh <tag> "<C1 C2 ...>" [bench args]
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $counters --output-format csv -d /tmp/pmc_$tag -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/pmc_$tag.err
python tools/pmc_summarize.py /tmp/pmc_$tag gpurun_out/pmc_$tag.json
