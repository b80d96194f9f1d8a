#!/bin/bash
# rocprofv3 --kernel-trace --stats of bin/DerpSequence itself (the drop-in, not the Python bench) on the 8-frame
# config-2 dataset, from PNGs on disk to PFMs on disk. usage (on the GPU box): tools/seq_cli_profile.sh <tag>
set -e
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
root=/tmp/derp_seq_prof
rm -rf $root && mkdir -p $root
python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from facebook360_dep_amd import synth
n, res, widths = synth.config("cfg2")
rig = synth.make_rig(n, res)
synth.write_dataset("$root/in", rig, list(range(8)), synth.level_sizes(res, res, widths))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $root/prof -o seq -- $GRAFT_REPO_ROOT/facebook360_dep_amd/bin/DerpSequence \
  --input_root=$root/in --output_root=$root/out --first=000000 --last=000007 > $root/log.txt 2>&1 || true
grep -E "TOTAL|level [01]:" $root/log.txt | cut -c1-200
f=$(find $root/prof -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_seq_cli_kernel_stats.csv
head -8 "$f" | cut -c1-160
