#!/usr/bin/env python
"""Developer check: destination batching (DERP_TABLE_BUDGET_GB) must not change results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facebook360_dep_amd import derp, synth

n, res, widths = synth.config("cfg2s")
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
frame = synth.make_frame(rig, sizes)
outs = []
for budget in (None, "0.5"):
    if budget:
        os.environ["DERP_TABLE_BUDGET_GB"] = budget
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid(); g.synchronize()
    outs.append([g.download_disparity(0, d) for d in range(n)])
    print("budget", budget, "counters", g.counters())
    g.close()
same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*outs))
print("batched == unbatched:", same)
sys.exit(0 if same else 1)
