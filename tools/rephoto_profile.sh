#!/bin/bash
# Developer tool: kernel-time split of one bin/ComputeRephotographyErrors run (rocprofv3 --kernel-trace --stats) on a
# synthetic config.  usage: tools/rephoto_profile.sh [config]   -> gpurun_out/rephoto_kernel_stats.csv
cfg=${1:-cfg2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - "$cfg" <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from facebook360_dep_amd import synth
name = sys.argv[1]
n, res, widths = synth.config(name)
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
root = "/tmp/rephoto_prof"
os.makedirs(root, exist_ok=True)
synth.write_dataset(root, rig, [0], sizes)
BIN = os.path.join(os.getcwd(), "facebook360_dep_amd", "bin")
subprocess.run([os.path.join(BIN, "DerpCLI"), "--input_root=" + root, "--output_root=" + root + "/out", "--first=000000",
                "--last=000000", "--resolution=%d" % res] + (["--partial_coverage"] if n <= 4 else []), check=True, capture_output=True)
PY
rm -rf /tmp/prof_rephoto
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rephoto -o r -- facebook360_dep_amd/bin/ComputeRephotographyErrors --first=000000 --last=000000 --output=/tmp/rephoto_prof/rephoto --rig=/tmp/rephoto_prof/rigs/rig_calibrated.json --color=/tmp/rephoto_prof/video/color_levels/level_0 --disparity=/tmp/rephoto_prof/out/disparity_levels/level_0 2>&1 | tail -3
cp /tmp/prof_rephoto/*kernel_stats.csv gpurun_out/rephoto_kernel_stats.csv 2>/dev/null || find /tmp/prof_rephoto -name "*kernel_stats.csv" -exec cp {} gpurun_out/rephoto_kernel_stats.csv \;
head -12 gpurun_out/rephoto_kernel_stats.csv | cut -c1-200
