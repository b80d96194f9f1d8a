#!/usr/bin/env python
"""Measure the drop-in, not only the kernels: write a BASELINE-config-2 dataset in the
reference's on-disk layout, run bin/DerpCLI (or bin/DerpSequence: the same frames with the per-level temporal
filter, resident in HBM) on it from disk and print its own timing lines (per-frame wall, decode / upload /
compute / download / write split, TOTAL). usage: tools/cli_timing.py [config] [frames] [threads] [binary]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from facebook360_dep_amd import synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
threads = sys.argv[3] if len(sys.argv) > 3 else "-1"
binary = sys.argv[4] if len(sys.argv) > 4 else "DerpCLI"
n, res, widths = synth.config(cfg)
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
root = tempfile.mkdtemp(prefix="derp_cli_", dir="/tmp")
t0 = time.time()
synth.write_dataset(root, rig, list(range(frames)), sizes)
print("dataset: %d frame(s) of %s written in %.1f s under %s" % (frames, cfg, time.time() - t0, root))
for binary, threads in [(b, t) for b in binary.split(",") for t in threads.split(",")]:
    out = os.path.join(root, "out_%s_%s" % (binary, threads))
    t0 = time.time()
    p = subprocess.run([os.path.join(ROOT, "facebook360_dep_amd", "bin", binary), "--input_root=" + root,
                        "--output_root=" + out, "--first=000000", "--last=%06d" % (frames - 1), "--resolution=%d" % res,
                        "--threads=" + threads] + (["--partial_coverage"] if n <= 4 else []), capture_output=True, text=True)
    wall = time.time() - t0
    print("%s --threads=%s rc=%d, wall %.2f s for %d frame(s) = %.1f Mpix/s from disk to disk" % (
        binary, threads, p.returncode, wall, frames, frames * n * res * res / wall / 1e6))
    for line in p.stderr.splitlines():
        if "-- I/O" in line or "-- TOTAL" in line or "-- rank" in line or "-- inputs" in line or "-- start-up" in line or \
                re.search(r"\(level \d+\)$", line):
            print(line)
    if p.returncode:
        print(p.stderr[-2000:])
