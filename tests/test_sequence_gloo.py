"""The N>1 path on CPU: two processes (gloo, world_size 2), one frame each, the per-level
barrier schedule of scripts/render/pipeline.py:364-408 with the temporal window exchanged by
all_gather. Compute is the CPU oracle standing in for the HIP library; the result must equal a
single-process emulation of the same schedule over both frames, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from facebook360_dep_amd import sequence, synth
from oracle import oracle_lib as O
from tests import common


def test_temporal_window():
    # TemporalBilateralFilter.cpp:96-119 with --time_radius=2 on an 8-frame sequence
    assert [sequence.temporal_window(t, 0, 7, 2) for t in range(8)] == [
        (0, 2), (0, 3), (0, 4), (1, 5), (2, 6), (3, 7), (4, 7), (5, 7)]
    assert sequence.temporal_window(0, 0, 0, 2) == (0, 0)


def _setup():
    n, res, widths = synth.config("tiny")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    return n, res, rig, sizes


class _OracleFrame:
    """One frame's state: what a rank holds (the oracle stands in for the GPU library)."""

    def __init__(self, rig, sizes, res, frame_index):
        self.rig, self.sizes, self.res = rig, sizes, res
        self.frame = synth.make_frame(rig, sizes, frame=frame_index, seed=360 + frame_index, device="cpu")
        self.disp = {}
        self.n = len(rig["cameras"])
        self.rd = O.Rig(rig["cameras"]).normalize()

    def process_level(self, level):
        prev = self.disp.get(level + 1)
        L = common.oracle_level(self.rig, self.sizes, self.frame, level, self.res, self.res, prev,
                                partial_coverage=True, threads=2)
        L.process()
        self.disp[level] = [L.get_dst(d)[0] for d in range(self.n)]
        self.mask = [L.fov_mask(d) for d in range(self.n)]

    def views(self, level):
        return (torch.from_numpy(np.stack(self.disp[level])),
                torch.from_numpy(np.stack(self.frame["color"][level])),
                torch.from_numpy(np.stack(self.mask)))

    def temporal(self, level, guides, disps, masks, offset):
        out = []
        for d in range(self.n):
            out.append(O.temporal_filter([g[d].numpy() for g in guides], [x[d].numpy() for x in disps],
                                         [m[d].numpy() for m in masks], offset, 0.01,
                                         O.temporal_space_radius(level), 0.5, 1.0, 0.5, threads=2))
        return torch.from_numpy(np.stack(out))

    def write_back(self, level, filtered):
        self.disp[level] = [filtered[d].numpy().copy() for d in range(self.n)]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, res, rig, sizes = _setup()
    fr = _OracleFrame(rig, sizes, res, rank)
    levels = list(range(len(sizes) - 1, -1, -1))
    sequence.run_level_schedule(rank, world, levels, fr.process_level, fr.views, fr.temporal, fr.write_back, dist=dist)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.stack(fr.disp[0]))
    dist.destroy_process_group()


def test_two_ranks_match_single_process(tmp_path):
    n, res, rig, sizes = _setup()
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # single-process emulation: both frames, level by level, window = frames {0, 1} for both
    frames = [_OracleFrame(rig, sizes, res, t) for t in range(world)]
    for level in range(len(sizes) - 1, -1, -1):
        for f in frames:
            f.process_level(level)
        views = [f.views(level) for f in frames]
        filtered = []
        for t, f in enumerate(frames):
            lo, hi = sequence.temporal_window(t, 0, world - 1, 2)
            filtered.append(f.temporal(level, [views[i][1] for i in range(lo, hi + 1)],
                                       [views[i][0] for i in range(lo, hi + 1)],
                                       [views[i][2] for i in range(lo, hi + 1)], t - lo))
        for f, x in zip(frames, filtered):
            f.write_back(level, x)
    for t in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % t))
        ref = np.stack(frames[t].disp[0])
        same = (got == ref) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), (t, int((~same).sum()))
    # the temporal stage really mixed the two frames: rank 0's result differs from an unfiltered run
    solo = _OracleFrame(rig, sizes, res, 0)
    for level in range(len(sizes) - 1, -1, -1):
        solo.process_level(level)
    assert not np.array_equal(np.nan_to_num(np.stack(solo.disp[0])), np.nan_to_num(np.stack(frames[0].disp[0])))
