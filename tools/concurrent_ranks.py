#!/usr/bin/env python
"""Developer experiment: the 8-frame bench sequence on ONE GPU, split over `world` emulated ranks (own context,
stream, tables and frame slots each; loopback halo exchange). Kernels of different ranks run concurrently on
different HIP streams: does the chip fill better?   usage: tools/concurrent_ranks.py [config] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facebook360_dep_amd import derp, sequence, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n, res, widths = synth.config(cfg)
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
data = {t: synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cuda") for t in range(frames)}
for world in (1, 2, 4):
    made = []
    for rank in range(world):
        g = derp.Derp(rig["cameras"], partial_coverage=int(n <= 4))
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, 0, frames - 1, rank, world)
        for t in r.owned:
            r.upload_frame(t, data[t])
        made.append((g, r))
    runners = [r for (_, r) in made]
    for rep in range(3):
        for g, _ in made:
            g.synchronize()
        t0 = time.perf_counter()
        if world == 1:
            runners[0].run()
            made[0][0].synchronize()
        else:
            sequence.run_loopback(runners, len(sizes) - 1)
        dt = time.perf_counter() - t0
        print("world %d rep %d: %.1f ms per frame, %.1f Mpix/s" % (world, rep, dt / frames * 1e3, frames * n * res * res / dt / 1e6), flush=True)
    for g, _ in made:
        g.close()
