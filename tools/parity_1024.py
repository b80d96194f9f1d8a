#!/usr/bin/env python
"""Developer check (GPU box, ~1 min): config 2's rig at 1024^2, full 9-level pyramid, HIP path vs the oracle
on all host cores. Too slow for the suite; result recorded in DESIGN.md section 5."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facebook360_dep_amd import derp, synth
from tests import common
n, res = 16, 1024
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, [w for w in synth.WIDTHS if w <= res])
frame = synth.make_frame(rig, sizes, device="cuda")
t = time.time(); cnt = {}
ref = common.oracle_pyramid(rig, sizes, frame, res, res, counters=cnt)
print("oracle s", time.time() - t)
g = derp.Derp(rig["cameras"]); g.set_pyramid(sizes, res, res); g.upload_frame(frame); g.process_pyramid(); g.synchronize()
nbad = npx = 0
for level in ref:
    for d in range(n):
        bad, rel = common.compare_disparity(g.download_disparity(level, d), ref[level][d], 1e-4)
        nbad += bad; npx += ref[level][d].size
print("16 x 1024^2 full pyramid: %d of %d pixels outside 1e-4; counters equal: %s" % (nbad, npx, g.counters()["n_cost"] == sum(v["n_cost"] for v in cnt.values())))
