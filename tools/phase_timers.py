#!/usr/bin/env python
"""Developer tool: where a cost-kernel wave spends its cycles. Needs a library built with -DDERP_PHASE_TIMERS=1 or 2 (its
counter slots then carry s_memtime cycles instead of cost / pair counts — see derp_kernels.h) as DERP_LIB.
usage: DERP_LIB=.../libderp_var_t1.so python tools/phase_timers.py 1|2 [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facebook360_dep_amd import derp, synth  # noqa: E402

mode = int(sys.argv[1])
n, res, widths = synth.config(sys.argv[2] if len(sys.argv) > 2 else "cfg2")
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
frame = synth.make_frame(rig, sizes, device="cuda")
g = derp.Derp(rig["cameras"])
g.set_pyramid(sizes, res, res)
g.upload_frame(frame)
g.process_pyramid()
g.synchronize()
for stage in ("random_proposals", "ping_pong"):
    for lvl in (0, 1):
        q = g.profile_query(stage, lvl)
        w, h = sizes[lvl]
        waves = n * ((w + 15) // 16) * ((h + 15) // 16) * 4
        a, b, m = q["n_cost"], q["n_pair"], g.profile_memoised(stage, lvl)
        if mode == 1:
            print("%-17s level %d: %.2f ms; per wave (of %d launched): projection + taps %.0f, SSD walk %.0f, selection %.0f cycles"
                  % (stage, lvl, q["ms"], waves, a / waves, b / waves, m / waves))
        else:
            print("%-17s level %d: %.2f ms; per wave (of %d launched): kernel body %.0f, outside computeCost %.0f cycles"
                  % (stage, lvl, q["ms"], waves, a / waves, b / waves))
g.close()
