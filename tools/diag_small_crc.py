#!/usr/bin/env python
"""Developer check: bench.py's `small` workload (6 cameras, 160^2, 4 frames, temporal filter) on the GPU against
the CPU oracle, value by value — where do the CRCs part?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from facebook360_dep_amd import derp, sequence, synth  # noqa: E402
from tests import common  # noqa: E402

n, res, widths = synth.config("small")
rig = synth.make_rig(n, res)
sizes = synth.level_sizes(res, res, widths)
seq = common.OracleSequence(rig, sizes, res, 0, 3, threads=-1, partial_coverage=False)
sequence.run_schedule(seq, list(range(len(sizes) - 1, -1, -1)), 0, 3, 0, 1)
g = derp.Derp(rig["cameras"])
g.set_pyramid(sizes, res, res)
r = sequence.SequenceRunner(g, 0, 3)
for t in r.owned:
    r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu"))
r.run()
g.synchronize()
print("oracle crc", {t: "%08x" % v for t, v in seq.result_crc().items()})
print("gpu    crc", {t: "%08x" % v for t, v in r.result_crc().items()})
for level in range(len(sizes) - 1, -1, -1):
    for t in range(4):
        for d in range(n):
            got = r.download_disparity(t, level, d)
            want = seq.disp[t][level][d].numpy()
            neq = ~((got == want) | (np.isnan(got) & np.isnan(want)))
            bits = (got.view(np.uint32) != want.view(np.uint32)) & ~neq
            if neq.any() or bits.any():
                ys, xs = np.nonzero(neq | bits)
                print("level %d frame %d dst %d: %d values differ, %d equal values with different bits; first at (%d, %d): gpu %r (%08x) oracle %r (%08x)"
                      % (level, t, d, int(neq.sum()), int(bits.sum()), xs[0], ys[0], got[ys[0], xs[0]],
                         got.view(np.uint32)[ys[0], xs[0]], want[ys[0], xs[0]], want.view(np.uint32)[ys[0], xs[0]]))
g.close()
