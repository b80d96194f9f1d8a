#!/usr/bin/env python
"""Developer measurement (library built with -DDERP_COUNT_UNION): how many SSD iterations the random-proposal waves walk
(the union of their lanes' visible sources, counted per active lane) against the pairs their lanes actually evaluate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facebook360_dep_amd import derp, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n, res, widths = synth.config(cfg)
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
g = derp.Derp(rig["cameras"], partial_coverage=int(n <= 4))
g.set_pyramid(sizes, res, res)
g.upload_frame(synth.make_frame(rig, sizes, frame=0, seed=360, device="cuda"))
g.profile_reset()
g.profile_enable(True)
g.process_pyramid()
g.synchronize()
for lv in range(3):
    q = g.profile_query("random_proposals", lv)
    slots_rand = g.profile_memoised("random_proposals", lv)
    print("level %d: n_cost %d n_pair %d (%.2f per cost); random-candidate lane-slots walked %d; ms %.2f" % (
        lv, q["n_cost"], q["n_pair"], q["n_pair"] / max(q["n_cost"], 1), slots_rand, q["ms"]))
print("first-evaluation lane-slots (all levels, 'insufficient' slot):", g.counters()["insufficient"])
g.close()
