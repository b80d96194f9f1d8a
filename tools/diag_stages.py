#!/usr/bin/env python
"""Developer tool: per-stage GPU-vs-oracle mismatch counts on a rig, each level started from the
ORACLE's previous-level result so that divergences do not compound."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facebook360_dep_amd import derp, synth
from tests import common
from oracle import oracle_lib as O

ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rig = synth.make_rig(ncam, res)
sizes = synth.level_sizes(res, res, [w for w in [128, 100, 80, 60, 50] if w <= res])
frame = synth.make_frame(rig, sizes)
g = derp.Derp(rig["cameras"])
g.set_pyramid(sizes, res, res)
g.upload_frame(frame)

def neq(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())

prev = None
for level in range(len(sizes) - 1, -1, -1):
    L = common.oracle_level(rig, sizes, frame, level, res, res, prev)
    L.reproject_colors()
    g.level_begin(level)
    g.stage("reproject_colors")
    tab = 0
    for d in range(ncam):
        for s in range(ncam):
            if s != d:
                tab += neq(g.debug(d, s, "warp"), L.proj(d, s, "warp")) + int((g.debug(d, s, "color") != L.proj(d, s, "color")).sum())
    start = [L.get_dst(d)[0] for d in range(ncam)]
    for d in range(ncam):
        g.set_level_disparity(d, start[d])
    rep = {"tables": tab}
    for st, of in (("brute_force", L.brute_force), ("random_proposals", L.random_proposals), ("ping_pong", L.ping_pong),
                   ("bilateral", L.bilateral), ("median", L.median), ("mask_fov", L.mask_fov)):
        of()
        g.stage(st)
        bad = 0; worst = 0.0
        for d in range(ncam):
            od = L.get_dst(d)[0]; gd = g.get_level_disparity(d)
            bad += neq(gd, od)
            b, r = common.compare_disparity(gd, od, 1e-4); worst = max(worst, r)
        rep[st] = (bad, "%.2g" % worst)
        # re-sync the GPU to the oracle after each stage so the next one is judged in isolation
        for d in range(ncam):
            g.set_level_disparity(d, L.get_dst(d)[0])
    print("level", level, sizes[level], rep)
    prev = [L.get_dst(d)[0] for d in range(ncam)]
