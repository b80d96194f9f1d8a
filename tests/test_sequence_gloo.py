"""The N>1 path on CPU: two processes (gloo, world_size 2), one frame each, the per-level
barrier schedule of scripts/render/pipeline.py:364-408 with the temporal window exchanged by
point-to-point neighbour exchange. Compute is the CPU oracle standing in for the HIP library; the result must equal a
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

    def static_masks(self, level):
        # fov masks depend on rig + level size only (no foreground masks in this test)
        rs, rd, d2s = common.oracle_rigs(self.rig)
        w, h = self.sizes[level]
        L = O.Level(rs, rd, d2s, O.make_params(level, len(self.sizes), w, h, self.res, self.res))
        return torch.from_numpy(np.stack([L.fov_mask(d) for d in range(self.n)]))

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
    # inputs of the neighbour frames, fetched once before the level loop
    static = {}
    for level in levels:
        guides = sequence.neighbour_exchange(torch.from_numpy(np.stack(fr.frame["color"][level])), rank, world, dist)
        masks = sequence.neighbour_exchange(fr.static_masks(level), rank, world, dist)
        static[level] = (guides, masks)
    sequence.run_level_schedule(rank, world, levels, fr.process_level, lambda lv: fr.views(lv)[0],
                                lambda lv: static[lv], fr.temporal, fr.write_back, dist=dist)
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


def _worker_modes(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = (torch.arange(24, dtype=torch.float32).reshape(2, 3, 4) + 100 * rank)
    u = (torch.arange(24, dtype=torch.int32).reshape(2, 3, 4) + 1000 * rank).to(torch.uint16)
    res = {}
    for mode in ("p2p", "allgather"):
        sequence.MODE = mode
        a = sequence.neighbour_exchange(x, rank, world, dist, radius=1)
        b = sequence.neighbour_exchange(u, rank, world, dist, radius=1)
        res[mode] = (torch.stack(a).numpy(), torch.stack(b).to(torch.int32).numpy())
    sequence.MODE = "p2p"
    np.savez(os.path.join(out_dir, "modes%d.npz" % rank), p0=res["p2p"][0], p1=res["p2p"][1], a0=res["allgather"][0],
             a1=res["allgather"][1])
    dist.destroy_process_group()


def test_exchange_modes_agree_three_ranks(tmp_path):
    """Window clamping with radius 1 on three ranks: rank 0 sees {0,1}, rank 1 {0,1,2}, rank 2 {1,2};
    the point-to-point and the all_gather transports return the same tensors (incl. a uint16 payload)."""
    world = 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_modes, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank, frames in ((0, [0, 1]), (1, [0, 1, 2]), (2, [1, 2])):
        z = np.load(os.path.join(str(tmp_path), "modes%d.npz" % rank))
        exp = np.stack([np.arange(24, dtype=np.float32).reshape(2, 3, 4) + 100 * f for f in frames])
        assert np.array_equal(z["p0"], exp) and np.array_equal(z["a0"], exp)
        expu = np.stack([np.arange(24).reshape(2, 3, 4) + 1000 * f for f in frames])
        assert np.array_equal(z["p1"], expu) and np.array_equal(z["a1"], expu)
