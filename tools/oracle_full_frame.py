#!/usr/bin/env python
"""The CPU oracle on ONE full BASELINE config-2 frame (16 x 2048^2, all 10 levels), timed — the measurement behind
bench.py's extrapolated `cpu_baseline`. Writes gpurun_out/oracle_full_frame.json (copied to
profiles/). Takes minutes."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from facebook360_dep_amd import synth  # noqa: E402
from tests import common  # noqa: E402

n, res, widths = synth.config("cfg2")
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
frame = synth.make_frame(rig, sizes, frame=0, seed=360, device="cuda")
levels = {}
prev = None
t_all = time.time()
for level in range(len(sizes) - 1, -1, -1):
    t0 = time.time()
    L = common.oracle_level(rig, sizes, frame, level, res, res, prev, threads=-1)
    L.process()
    prev = [L.get_dst(d)[0] for d in range(L.D)]
    levels[level] = round(time.time() - t0, 2)
    print("level %d: %.1f s" % (level, levels[level]), flush=True)
total = time.time() - t_all
out = {"workload": "BASELINE config 2 in full: frame 0 of the 16-camera 2048^2 rig, all %d levels, CPU oracle" % len(sizes),
       "seconds": round(total, 1), "value": round(n * res * res / total / 1e6, 4), "unit": "Mpix/s",
       "cores": bench.usable_cpus(), "threads": os.cpu_count(), "seconds_per_level": levels}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "oracle_full_frame.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
