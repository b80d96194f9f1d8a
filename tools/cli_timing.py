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
# CLI_EXTRA="--resident_frames=5;" runs every binary once per ';'-separated flag set (empty = the plain run) and
# compares the level-0 PFMs of the runs byte for byte
extras = os.environ.get("CLI_EXTRA", "").split(";") if os.environ.get("CLI_EXTRA") else [""]
outs = []
for binary, threads, extra in [(b, t, e) for b in binary.split(",") for t in threads.split(",") for e in extras]:
    out = os.path.join(root, "out_%s_%s_%d" % (binary.replace("/", "_").replace("@", "_").replace("=", "_"), threads, len(outs)))
    outs.append(out)
    t0 = time.time()
    # "<dir>/<binary>" = another build of the binary (A/B on one box: the round-3 build lives in bin_r3 with its library)
    env = dict(os.environ)
    label = binary
    while "@" in binary:  # "KEY=VALUE@binary": that environment variable for this run
        kv, binary = binary.split("@", 1)
        env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    exe = os.path.join(ROOT, "facebook360_dep_amd", binary) if "/" in binary else os.path.join(ROOT, "facebook360_dep_amd", "bin", binary)
    if "/" in binary:
        env["LD_LIBRARY_PATH"] = os.path.dirname(exe) + ":" + env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe, "--input_root=" + root,
                        "--output_root=" + out, "--first=000000", "--last=%06d" % (frames - 1), "--resolution=%d" % res,
                        "--threads=" + threads] + (["--partial_coverage"] if n <= 4 else []) + extra.split(),
                       capture_output=True, text=True, env=env)
    wall = time.time() - t0
    print("%s --threads=%s %s rc=%d, wall %.2f s for %d frame(s) = %.1f Mpix/s from disk to disk" % (
        label, threads, extra, p.returncode, wall, frames, frames * n * res * res / wall / 1e6))
    for line in p.stderr.splitlines():
        if "-- I/O" in line or "-- TOTAL" in line or "-- rank" in line or "-- inputs" in line or "-- start-up" in line or \
                "-- level" in line or "-- waited" in line or "-- released" in line or \
                re.search(r"\(level \d+\)$", line):
            print(line)
        if "frame slot(s) in HBM" in line:
            print(line)
    if p.returncode:
        print(p.stderr[-2000:])
if len(outs) > 1:
    import filecmp

    base = os.path.join(outs[0], "disparity_levels", "level_0")
    for other in outs[1:]:
        same = total = 0
        for cam in sorted(os.listdir(base)):
            for f in sorted(os.listdir(os.path.join(base, cam))):
                total += 1
                same += filecmp.cmp(os.path.join(base, cam, f), os.path.join(other, "disparity_levels", "level_0", cam, f),
                                    shallow=False)
        print("%s vs %s: %d of %d level-0 files byte-identical" % (os.path.basename(other), os.path.basename(outs[0]), same, total))
