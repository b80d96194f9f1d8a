#!/usr/bin/env python
"""Developer tool: rephotography score (bin/ComputeRephotographyErrors) of the estimated level-0
disparity against the score the analytic ground-truth disparity gets on the same synthetic rig."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facebook360_dep_amd import synth, imageio as dio

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2s"
n, res, widths = synth.config(name)
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
root = tempfile.mkdtemp(prefix="rephoto_")
synth.write_dataset(root, rig, [0], sizes)
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "facebook360_dep_amd", "bin")
out = os.path.join(root, "out")
subprocess.run([os.path.join(BIN, "DerpCLI"), "--input_root=" + root, "--output_root=" + out, "--first=000000",
                "--last=000000", "--resolution=%d" % res] + (["--partial_coverage"] if n <= 4 else []),
               check=True, capture_output=True)
frame = synth.make_frame(rig, sizes)
truth = os.path.join(root, "truth")
for cam, t in zip(rig["cameras"], frame["truth"]):
    os.makedirs(os.path.join(truth, cam["id"]), exist_ok=True)
    fov = np.isfinite(dio.read_pfm(os.path.join(out, "disparity_levels", "level_0", cam["id"], "000000.pfm")))
    dio.write_pfm(os.path.join(truth, cam["id"], "000000.pfm"), np.where(fov, t, np.nan).astype(np.float32))
for label, disp in (("estimated", os.path.join(out, "disparity_levels", "level_0")), ("ground truth", truth)):
    t0 = time.time()
    p = subprocess.run([os.path.join(BIN, "ComputeRephotographyErrors"), "--first=000000", "--last=000000",
                        "--output=" + os.path.join(root, "rephoto_" + label.split()[0]),
                        "--rig=" + os.path.join(root, "rigs", "rig_calibrated.json"),
                        "--color=" + os.path.join(root, "video", "color_levels", "level_0"), "--disparity=" + disp],
                       capture_output=True, text=True, check=True)
    print(name, label, "disparity:", p.stderr.strip().splitlines()[-1].split("] ")[-1],
          "(ComputeRephotographyErrors: %.1f s wall)" % (time.time() - t0))
