#!/usr/bin/env python
"""Developer measurement: what a freshly started process pays before its first result — HIP runtime +
context (derp_create), the first kernel launch (which loads the library's one 370 KB code object), a second launch.
Run in a fresh interpreter per sample: python tools/startup_split.py [n]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    t0 = time.perf_counter()
    from facebook360_dep_amd import derp, synth  # (imports torch first: one HIP runtime per process)
    t1 = time.perf_counter()
    rig = synth.make_rig(16, 2048)
    g = derp.Derp(rig["cameras"])
    t2 = time.perf_counter()
    g.fov_mask(0, 256, 256)
    t3 = time.perf_counter()
    g.fov_mask(1, 256, 256)
    t4 = time.perf_counter()
    print("imports %.3f s | derp_create (HIP runtime + context) %.3f s | first kernel (code object load + launch + copy) %.4f s | "
          "second kernel %.4f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
else:
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one"])
