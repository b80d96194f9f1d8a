#!/bin/bash
# Developer tool: collect PMC counter groups for the bench (one rocprofv3 pass per group) and
# summarise per kernel into gpurun_out/pmc_<tag>.json.   usage: tools/pmc_run.sh <tag> "<C1 C2 ...>" [bench args]
# (under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then sit on the box for its whole limit)
tag=$1; counters=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc_$tag
timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $counters --output-format csv -d /tmp/pmc_$tag -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/pmc_$tag.err
python tools/pmc_summarize.py /tmp/pmc_$tag gpurun_out/pmc_$tag.json
