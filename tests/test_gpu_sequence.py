"""BASELINE config 3 on the GPU: the per-level barrier schedule (DerpCLI(L) -> temporal filter(L) ->
Transfer -> L-1; scripts/render/pipeline.py:364-408, TemporalBilateralFilter.cpp:96-184) driven by the
library's sequence driver (`derp_seq_*`) with HIP compute, compared with the CPU oracle running the same
schedule — every level of every frame — on one rank, on several ranks emulated on one GPU (loopback
transport), and on two processes sharing cuda:0 over gloo (external transport). Also: the RCCL transport's
self-test on a 1-rank communicator, and `derp_dev_*` / `derp_temporal_filter_dev` through device pointers."""
import os
import socket

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

FIRST, LAST = 0, 4  # 5 frames, radius 2: windows {0,1,2} {0..3} {0..4} {1..4} {2,3,4} — clamped at both ends


def _bad(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    return int((~((got == want) | (np.isnan(got) & np.isnan(want)))).sum())


def _setup(name):
    from facebook360_dep_amd import synth

    n, res, widths = synth.config(name)
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    return n, res, rig, sizes


def _gpu_runner(rig, sizes, res, first, last, rank=0, world=1, **opts):
    from facebook360_dep_amd import derp, sequence, synth

    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, res, res)
    r = sequence.SequenceRunner(g, first, last, rank, world, **opts)
    for t in r.owned:
        r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu"))
    return g, r


_oracle = {}


def _oracle_sequence(name, first, last):
    """Single-process oracle run of the schedule (cached per rig): .disp[t][level], .raw[(t, level)]."""
    from facebook360_dep_amd import sequence

    key = (name, first, last)
    if key not in _oracle:
        n, res, rig, sizes = _setup(name)
        seq = common.OracleSequence(rig, sizes, res, first, last, threads=-1)
        sequence.run_schedule(seq, list(range(len(sizes) - 1, -1, -1)), first, last, 0, 1)
        _oracle[key] = seq
    return _oracle[key]


def _compare_with_oracle(runners, ref, n, sizes):
    """Every level of every frame; the libm caveat of test_gpu_parity applies (glibc vs OCML last ulp)."""
    worst = 0
    for r in runners:
        for t in r.owned:
            for level in range(len(sizes)):
                for d in range(n):
                    got = r.download_disparity(t, level, d)
                    want = ref.disp[t][level][d].numpy()
                    bad, _ = common.compare_disparity(got, want, 1e-4)
                    assert bad == 0, (r.rank, t, level, d, bad)
                    worst = max(worst, _bad(got, want))
    return worst


def test_sequence_one_rank_against_oracle(built):
    """5 frames resident on one GPU (frame slots), full pyramid, temporal filter seeded level to level."""
    n, res, rig, sizes = _setup("tiny")
    g, r = _gpu_runner(rig, sizes, res, FIRST, LAST)
    assert r.owned == [0, 1, 2, 3, 4] and r.halo == [] and g.frame_slots()[0] == 5
    r.run()
    g.synchronize()
    ref = _oracle_sequence("tiny", FIRST, LAST)
    assert _compare_with_oracle([r], ref, n, sizes) == 0  # bit-equal on this rig
    # the stage did something and the next level really starts from the FILTERED level: an unfiltered
    # pyramid of frame 2 differs at level 0
    from facebook360_dep_amd import derp, synth

    solo = derp.Derp(rig["cameras"], partial_coverage=1)
    solo.set_pyramid(sizes, res, res)
    solo.upload_frame(synth.make_frame(rig, sizes, frame=2, seed=362, device="cpu"))
    solo.process_pyramid()
    assert _bad(solo.download_disparity(0, 0), r.download_disparity(2, 0, 0)) > 0
    assert _bad(solo.download_disparity(len(sizes) - 1, 0), ref.raw[(2, len(sizes) - 1)][0]) == 0
    solo.close()
    st = r.stats()
    assert st["bytes_sent"] == 0 and st["bytes_received"] == 0
    g.close()


@pytest.mark.parametrize("world,partition", [(2, 0), (3, 0), (3, 1), (5, 0)])
def test_sequence_ranks_on_one_gpu_loopback(built, world, partition):
    """`world` ranks emulated on one GPU, each with its own context and frame slots; the halo frames'
    raw level disparity moves device to device along the plan. Result = the oracle's, for the block and
    the cyclic partition, including one frame per rank (world 5 = the one-frame-per-GPU shape)."""
    from facebook360_dep_amd import sequence

    n, res, rig, sizes = _setup("tiny")
    made = [_gpu_runner(rig, sizes, res, FIRST, LAST, rank, world, partition=partition) for rank in range(world)]
    runners = [r for (_, r) in made]
    assert sorted(t for r in runners for t in r.owned) == list(range(FIRST, LAST + 1))
    sequence.run_loopback(runners, len(sizes) - 1)
    ref = _oracle_sequence("tiny", FIRST, LAST)
    assert _compare_with_oracle(runners, ref, n, sizes) == 0
    # exactly the planned traffic: colour pyramid once + one disparity level per (transfer, level)
    plan = sequence.plan(FIRST, LAST, world, 2, partition)
    px = sum(w * h for (w, h) in sizes)
    assert sum(r.stats()["bytes_received"] for r in runners) == len(plan) * px * n * (8 + 4)
    assert sum(r.stats()["bytes_sent"] for r in runners) == len(plan) * px * n * (8 + 4)
    for (g, r) in made:
        g.close()


def test_sequence_edge_cases(built):
    """More ranks than frames (a rank that owns nothing takes part in every phase and moves no data), a
    one-frame sequence (window = the frame itself: the filter still runs, on one frame), time_radius 0."""
    from facebook360_dep_amd import sequence

    n, res, rig, sizes = _setup("tiny")
    made = [_gpu_runner(rig, sizes, res, 0, 1, rank, 3) for rank in range(3)]
    runners = [r for (_, r) in made]
    assert [r.owned for r in runners] == [[0], [1], []] and runners[2].halo == []
    sequence.run_loopback(runners, len(sizes) - 1)
    ref = _oracle_sequence("tiny", 0, 1)
    assert _compare_with_oracle(runners, ref, n, sizes) == 0
    assert runners[2].stats()["bytes_received"] == 0
    for (g, r) in made:
        g.close()
    # one frame: temporalJointBilateralFilter over a single frame is still a 3x3 spatial filter
    g, r = _gpu_runner(rig, sizes, res, 3, 3)
    r.run()
    one = common.OracleSequence(rig, sizes, res, 3, 3, threads=-1)
    sequence.run_schedule(one, list(range(len(sizes) - 1, -1, -1)), 3, 3, 0, 1)
    assert _compare_with_oracle([r], one, n, sizes) == 0
    assert _bad(one.raw[(3, 0)][0], one.disp[3][0][0].numpy()) > 0
    g.close()
    # time_radius 0 on two frames: no halo, each frame filtered on its own
    g, r = _gpu_runner(rig, sizes, res, 0, 1, time_radius=0)
    r.run()
    zero = common.OracleSequence(rig, sizes, res, 0, 1, radius=0, threads=-1)
    sequence.run_schedule(zero, list(range(len(sizes) - 1, -1, -1)), 0, 1, 0, 1, radius=0)
    assert _compare_with_oracle([r], zero, n, sizes) == 0
    g.close()


def test_sequence_eight_frames_one_per_rank(built):
    """The bench's N = 8 shape in miniature: 8 frames, 8 ranks (one frame each, <= 4 halo frames per rank),
    emulated on one GPU — against the oracle and against the 1-rank run of the same sequence."""
    from facebook360_dep_amd import sequence

    n, res, rig, sizes = _setup("tiny")
    first, last, world = 0, 7, 8
    made = [_gpu_runner(rig, sizes, res, first, last, rank, world) for rank in range(world)]
    runners = [r for (_, r) in made]
    assert [r.owned for r in runners] == [[t] for t in range(8)]
    assert runners[0].halo == [1, 2] and runners[3].halo == [1, 2, 4, 5] and runners[7].halo == [5, 6]
    sequence.run_loopback(runners, len(sizes) - 1)
    ref = _oracle_sequence("tiny", first, last)
    assert _compare_with_oracle(runners, ref, n, sizes) == 0
    g1, r1 = _gpu_runner(rig, sizes, res, first, last)
    r1.run()
    for t in range(8):
        for d in range(n):
            assert _bad(runners[t].download_disparity(t, 0, d), r1.download_disparity(t, 0, d)) == 0
    g1.close()
    for (g, r) in made:
        g.close()


def test_sequence_sixteen_cameras_against_oracle(built):
    """Config 3's rig (16 cameras) at 128^2, 3 frames split over 2 emulated ranks."""
    from facebook360_dep_amd import sequence, synth

    n, res, widths = 16, 128, [128, 100, 80, 60, 50]
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    first, last = 0, 2
    ref = common.OracleSequence(rig, sizes, res, first, last, threads=-1)
    sequence.run_schedule(ref, list(range(len(sizes) - 1, -1, -1)), first, last, 0, 1)
    made = []
    for rank in range(2):
        from facebook360_dep_amd import derp

        g = derp.Derp(rig["cameras"])
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, first, last, rank, 2)
        for t in r.owned:
            r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu"))
        made.append((g, r))
    sequence.run_loopback([r for (_, r) in made], len(sizes) - 1)
    flips = _compare_with_oracle([r for (_, r) in made], ref, n, sizes)
    common.observed("sequence_sixteen_cameras_float_differences", flips)
    for (g, r) in made:
        g.close()


def test_config3_full_size_two_ranks_equal_one(built):
    """BASELINE config 3 at its own size — 16 cameras x 2048^2, 8 frames, full 10-level pyramid, temporal filter —
    on one rank and on two ranks emulated on the GPU (loopback transport, own context / tables / frame slots
    each): level-0 disparities bit-equal frame by frame, exchanged bytes = the plan's, NaN <=> outside the FOV
    mask, median error against the analytic scene, and the CRCs bench.py prints for this workload."""
    import json
    import os

    from facebook360_dep_amd import derp, sequence, synth

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    first, last = 0, 7
    w0, h0 = sizes[0]

    truth = {}

    def make(rank, world):
        g = derp.Derp(rig["cameras"])
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, first, last, rank, world)
        for t in r.owned:
            frame = synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cuda")
            r.upload_frame(t, frame)
            for (tt, d) in ((0, 0), (5, 9)):
                if tt == t:
                    truth[(t, d)] = np.asarray(frame["truth"][d])
        return g, r

    g1, r1 = make(0, 1)
    r1.run()
    g1.synchronize()
    crc1 = r1.result_crc()
    # properties of the 1-rank result (the oracle cannot run this size): two frames, two destinations
    for t, d in ((0, 0), (5, 9)):
        got = r1.download_disparity(t, 0, d)
        fov = g1.fov_mask(d, w0, h0).astype(bool)
        assert np.array_equal(np.isnan(got), ~fov)
        err = np.abs(got[fov] - truth[(t, d)][fov]) / truth[(t, d)][fov]
        assert np.median(err) < 0.01, (t, d, float(np.median(err)))
    keep = {(t, d): r1.download_disparity(t, 0, d) for t in (0, 3, 4, 7) for d in (0, 7, 15)}
    g1.close()

    made = [make(rank, 2) for rank in range(2)]
    runners = [r for (_, r) in made]
    assert [r.owned for r in runners] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert runners[0].halo == [4, 5] and runners[1].halo == [2, 3]
    sequence.run_loopback(runners, len(sizes) - 1)
    crc2 = {}
    for r in runners:
        crc2.update(r.result_crc())
    assert crc2 == crc1
    for (t, d), want in keep.items():
        assert _bad(runners[t // 4].download_disparity(t, 0, d), want) == 0, (t, d)
    plan = sequence.plan(first, last, 2, 2, 0)
    px = sum(w * h for (w, h) in sizes)
    assert len(plan) == 4
    assert sum(r.stats()["bytes_received"] for r in runners) == len(plan) * px * n * (8 + 4)
    for (g, r) in made:
        g.close()
    # the values bench.py's default workload prints at every --gpus N
    got = {str(t): "%08x" % v for t, v in sorted(crc1.items())}
    if os.environ.get("DERP_RECORD_BASELINE"):
        with open(os.path.join(os.path.dirname(common._RECORD_PATH), "bench_result_crc_cfg2_8.json"), "w") as f:
            json.dump(got, f, indent=1, sort_keys=True)
    else:
        with open(os.path.join(os.path.dirname(common._BASELINE_PATH), "bench_result_crc.json")) as f:
            assert got == json.load(f)["cfg2_8"]


def test_sequence_foreground_masks(built):
    """Temporal masking (pipeline.py:386): mask = fg & fov per window frame; fg masks of halo frames are
    exchanged with the inputs."""
    from facebook360_dep_amd import derp, sequence, synth

    n, res, rig, sizes = _setup("tiny")
    first, last = 0, 2
    ref = common.OracleSequence(rig, sizes, res, first, last, threads=-1, use_foreground_masks=True)
    sequence.run_schedule(ref, list(range(len(sizes) - 1, -1, -1)), first, last, 0, 1)
    made = []
    for rank in range(2):
        g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=1)
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, first, last, rank, 2, use_foreground_masks=1)
        for t in r.owned:
            r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu", with_masks=True))
        made.append((g, r))
    sequence.run_loopback([r for (_, r) in made], len(sizes) - 1)
    assert _compare_with_oracle([r for (_, r) in made], ref, n, sizes) == 0
    for (g, r) in made:
        g.close()


def _levels_equal(a, b, frames, n, sizes):
    for t in frames:
        for level in range(len(sizes)):
            for d in range(n):
                assert _bad(a.download_disparity(t, level, d), b.download_disparity(t, level, d)) == 0, (t, level, d)


@pytest.mark.parametrize("masks", [False, True])
def test_sequence_out_of_core_equals_resident(built, masks):
    """resident_frames = 2R + 1 = 5 device slots for 8 owned frames: inputs stream per level from host memory, raw
    levels and results live in page-locked host buffers — every level of every frame equals the fully resident run
    bit for bit (with and without foreground masks / temporal masking), and both equal the oracle."""
    from facebook360_dep_amd import derp, sequence, synth

    n, res, rig, sizes = _setup("tiny")
    first, last = 0, 7

    def run(resident):
        g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=int(masks))
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, first, last, 0, 1, use_foreground_masks=int(masks), resident_frames=resident)
        for t in r.owned:
            r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu", with_masks=masks))
        r.run()
        g.synchronize()
        return g, r

    g0, r0 = run(0)
    g1, r1 = run(5)
    assert not r0.streaming and r1.streaming and g1.frame_slots()[0] == 5 and g0.frame_slots()[0] == 8
    _levels_equal(r0, r1, range(first, last + 1), n, sizes)
    if not masks:
        ref = _oracle_sequence("tiny", first, last)
        assert _compare_with_oracle([r1], ref, n, sizes) == 0
    g0.close()
    g1.close()
    with pytest.raises(derp.DerpError):  # fewer slots than the temporal window
        g = derp.Derp(rig["cameras"], partial_coverage=1)
        g.set_pyramid(sizes, res, res)
        try:
            sequence.SequenceRunner(g, first, last, 0, 1, resident_frames=3)
        finally:
            g.close()


def test_sequence_out_of_core_two_ranks_and_no_filter(built):
    """Out of core on two emulated ranks (6 owned frames each, 5 slots: the frames the neighbour's window reaches
    into keep fixed device buffers for the exchange) equals the resident single-rank run; without the temporal
    filter ONE slot serves any number of frames."""
    from facebook360_dep_amd import derp, sequence, synth

    n, res, rig, sizes = _setup("tiny")
    first, last = 0, 11
    g0, r0 = _gpu_runner(rig, sizes, res, first, last)
    r0.run()
    made = [_gpu_runner(rig, sizes, res, first, last, rank, 2, resident_frames=5) for rank in range(2)]
    runners = [r for (_, r) in made]
    assert all(r.streaming for r in runners)
    sequence.run_loopback(runners, len(sizes) - 1)
    for r in runners:
        _levels_equal(r0, r, r.owned, n, sizes)
    plan = sequence.plan(first, last, 2, 2, 0)
    px = sum(w * h for (w, h) in sizes)
    assert sum(r.stats()["bytes_received"] for r in runners) == len(plan) * px * n * (8 + 4)
    for (g, r) in made:
        g.close()
    g0.close()
    g2, r2 = _gpu_runner(rig, sizes, res, 0, 3, do_temporal_filter=0, resident_frames=1)
    g3, r3 = _gpu_runner(rig, sizes, res, 0, 3, do_temporal_filter=0)
    r2.run()
    r3.run()
    assert r2.streaming and g2.frame_slots()[0] == 1
    _levels_equal(r2, r3, range(4), n, sizes)
    g2.close()
    g3.close()


def test_sequence_frames_filtered_ahead_of_the_level(built):
    """derp_seq_level_filter_frame: a frame is filtered as soon as its window is computed — before the level's last
    frame — into its scratch, from where derp_seq_download_filtered reads it over the copy stream; the Transfer still
    waits for derp_seq_level_filter. Every level of every frame equals the oracle, and the early download equals what
    the slot holds after the Transfer."""
    n, res, rig, sizes = _setup("tiny")
    ref = _oracle_sequence("tiny", FIRST, LAST)
    g, r = _gpu_runner(rig, sizes, res, FIRST, LAST)
    early = {}
    for level in range(len(sizes) - 1, -1, -1):
        done = 0
        for t in r.owned:
            r.compute_frame(level, t)
            while done < len(r.owned) and r.filter_frame(level, r.owned[done]):
                early[(r.owned[done], level)] = [r.download_filtered(r.owned[done], level, d) for d in range(n)]
                done += 1
            # windows reach 2 frames ahead: frame t - 2 is the newest that can be filtered after frame t's compute
            assert done == max(0, t - FIRST - 1) or t == LAST
        assert done == len(r.owned)  # the last frame's compute completes every window (one rank: no halo)
        r.filter(level)  # nothing left to filter: the Transfer
    g.synchronize()
    assert _compare_with_oracle([r], ref, n, sizes) == 0
    for (t, level), planes in early.items():
        for d in range(n):
            assert _bad(planes[d], r.download_disparity(t, level, d)) == 0, (t, level, d)
    # a frame that is not owned is an error; out of turn (window incomplete) is "not yet", not an error
    r.compute_frame(0, FIRST)
    assert r.filter_frame(0, FIRST) is False
    with pytest.raises(Exception):
        r.filter_frame(0, LAST + 5)
    g.close()


def test_sequence_exchange_stream_statistics(built):
    """The level's exchange runs on the library's exchange stream; what the compute stream waited of it cannot exceed what the
    exchange took, a rank with halo frames reports both, and `derp_device_memory` answers with the device's free / total HBM."""
    from facebook360_dep_amd import sequence

    n, res, rig, sizes = _setup("tiny")
    made = [_gpu_runner(rig, sizes, res, FIRST, LAST, rank, 2) for rank in range(2)]
    sequence.run_loopback([r for (_, r) in made], len(sizes) - 1)
    for g, r in made:
        g.synchronize()
        st = r.stats()
        assert st["bytes_received"] > 0 and st["exchange_ms"] > 0
        assert 0.0 <= st["exchange_exposed_ms"] <= st["exchange_ms"] + 0.5
        free, total = g.device_memory()
        assert 0 < free <= total and total > (64 << 30)
    ref = _oracle_sequence("tiny", FIRST, LAST)
    assert _compare_with_oracle([r for (_, r) in made], ref, n, sizes) == 0
    for g, r in made:
        r.close()
        g.close()


def test_sequence_work_lanes_equal_frame_after_frame(built, monkeypatch):
    """derp_seq_level_compute runs the frames of a coarse level on work lanes (own working set + stream per frame, the
    level's warps shared): every level of every frame must equal the frame-after-frame order bit for bit, with and
    without masks, and with fewer lanes than frames (a lane then takes several frames of the level in turn)."""
    from facebook360_dep_amd import synth

    n, res, rig, sizes = _setup("small")

    def run(masks, **env):
        for k in ("DERP_SEQ_LANES", "DERP_SEQ_LANE_MAX_WIDTH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        from facebook360_dep_amd import derp, sequence

        g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=int(masks))
        g.set_pyramid(sizes, res, res)
        r = sequence.SequenceRunner(g, 0, 5, use_foreground_masks=int(masks))
        for t in r.owned:
            r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu", with_masks=masks))
        r.run()
        g.synchronize()
        out = {(t, lvl): [r.download_disparity(t, lvl, d) for d in range(n)] for t in r.owned for lvl in range(len(sizes))}
        r.close()
        g.close()
        return out

    for masks in (False, True):
        plain = run(masks, DERP_SEQ_LANES="0")
        for env in ({}, {"DERP_SEQ_LANES": "3"}, {"DERP_SEQ_LANE_MAX_WIDTH": "100000"}):
            laned = run(masks, **env)
            assert plain.keys() == laned.keys()
            bad = sum(_bad(a, b) for k in plain for a, b in zip(plain[k], laned[k]))
            assert bad == 0, (masks, env, bad)


def test_sequence_phases_out_of_order_are_refused(built):
    """derp_seq_level_filter without the level's compute, or (with halo frames) without its exchange, fails
    instead of filtering stale data."""
    from facebook360_dep_amd import derp, sequence

    n, res, rig, sizes = _setup("tiny")
    top = len(sizes) - 1
    g, r = _gpu_runner(rig, sizes, res, 0, 2)
    with pytest.raises(derp.DerpError):
        r.filter(top)
    r.compute(top)
    r.filter(top)  # one rank: nothing to exchange
    g.close()
    made = [_gpu_runner(rig, sizes, res, 0, 3, rank, 2) for rank in range(2)]
    runners = [r for (_, r) in made]
    for r in runners:
        r.attach_loopback(runners)
    for r in runners:
        r.exchange_inputs()
    for r in runners:
        r.compute(top)
    with pytest.raises(derp.DerpError):
        runners[0].filter(top)  # halo level not exchanged yet
    for r in runners:
        r.exchange_level(top)
    for r in runners:
        r.filter(top)
    for (g, r) in made:
        g.close()


def test_sequence_without_temporal_filter_is_replicas(built):
    """do_temporal_filter = 0: no halo, no exchange, every frame equals a stand-alone pyramid."""
    from facebook360_dep_amd import derp, synth

    n, res, rig, sizes = _setup("tiny")
    g, r = _gpu_runner(rig, sizes, res, 0, 1, do_temporal_filter=0)
    r.run()
    solo = derp.Derp(rig["cameras"], partial_coverage=1)
    solo.set_pyramid(sizes, res, res)
    solo.upload_frame(synth.make_frame(rig, sizes, frame=1, seed=361, device="cpu"))
    solo.process_pyramid()
    for d in range(n):
        assert _bad(r.download_disparity(1, 0, d), solo.download_disparity(0, d)) == 0
    solo.close()
    g.close()


def test_rccl_transport_selftest_single_rank(built):
    """librccl is bound at run time; a 1-rank communicator sends to itself on the library's stream."""
    from facebook360_dep_amd import sequence

    n, res, rig, sizes = _setup("tiny")
    g, r = _gpu_runner(rig, sizes, res, 0, 1)
    r.attach_rccl(sequence.rccl_unique_id())
    r.selftest(1 << 16)
    r.run()  # world 1 over the RCCL transport object: no transfers, same result path
    g.synchronize()
    ref = _oracle_sequence("tiny", 0, 1)
    assert _compare_with_oracle([r], ref, n, sizes) == 0
    g.close()


def test_frame_slots_and_dev_pointers(built):
    """derp_select_frame switches which frame uploads / process / derp_dev_* refer to; derp_dev_mask is
    complete on return and lives in its own buffer; derp_temporal_filter_dev on those device pointers
    equals the host-pointer entry point."""
    import torch

    from facebook360_dep_amd import derp, synth

    n, res, rig, sizes = _setup("tiny")
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, res, res)
    g.set_frame_slots(3)
    frames = [synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu") for t in range(3)]
    for t in range(3):
        g.select_frame(t)
        g.upload_frame(frames[t])
        g.process_pyramid()
    g.synchronize()
    with pytest.raises(derp.DerpError):
        g.select_frame(3)
    level = 0
    w, h = sizes[level]
    ptrs = {"g": [], "d": [], "m": []}
    host = {"d": [], "m": None}
    for t in range(3):
        g.select_frame(t)
        ptrs["g"].append(g.dev_color(level, 1)[0])
        ptrs["d"].append(g.dev_disparity(level, 1)[0])
        host["d"].append(g.download_disparity(level, 1))
    assert len(set(ptrs["d"])) == 3 and len(set(ptrs["g"])) == 3  # three distinct resident pyramids
    mp_, nb = g.dev_mask(level, 1)
    assert nb == w * h

    class _A:  # read the mask back through torch's view of the pointer
        __cuda_array_interface__ = {"shape": (h, w), "typestr": "|u1", "data": (mp_, False), "version": 3}

    mask = torch.as_tensor(_A(), device="cuda").cpu().numpy()
    assert np.array_equal(mask, g.fov_mask(1, w, h))
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    g.temporal_filter_dev(ptrs["g"], ptrs["d"], [mp_] * 3, w, h, 1, 0.01, 1, 0.5, 1.0, 0.5, out.data_ptr())
    g.synchronize()
    want = g.temporal_filter([frames[t]["color"][level][1] for t in range(3)], host["d"], [mask] * 3, 1, 0.01, 1, 0.5,
                             1.0, 0.5)
    assert _bad(out.cpu().numpy(), want) == 0
    # slot 0's result is still frame 0's: equal to a stand-alone run
    solo = derp.Derp(rig["cameras"], partial_coverage=1)
    solo.set_pyramid(sizes, res, res)
    solo.upload_frame(frames[0])
    solo.process_pyramid()
    g.select_frame(0)
    assert _bad(g.download_disparity(0, 2), solo.download_disparity(0, 2)) == 0
    solo.close()
    g.close()


def _gloo_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from facebook360_dep_amd import sequence

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, res, rig, sizes = _setup("tiny")
    g, r = _gpu_runner(rig, sizes, res, FIRST, LAST, rank, world)
    r.attach_torch(dist, "p2p", stage_on_host=True)  # gloo moves host tensors; the GPU path stages through pinned memory
    r.exchange_inputs()
    r.run()
    g.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             **{"f%d_l%d" % (t, lv): np.stack([r.download_disparity(t, lv, d) for d in range(n)])
                for t in r.owned for lv in range(len(sizes))})
    g.close()
    dist.destroy_process_group()


def test_sequence_two_processes_one_gpu_gloo(built, tmp_path):
    """world_size 2, both ranks on cuda:0, HIP compute, the external transport driven by torch.distributed."""
    import torch.multiprocessing as mp

    from facebook360_dep_amd import sequence

    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gloo_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ref = _oracle_sequence("tiny", FIRST, LAST)
    n, res, rig, sizes = _setup("tiny")
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for t in sequence.owned_frames(FIRST, LAST, world, rank):
            for lv in range(len(sizes)):
                assert _bad(z["f%d_l%d" % (t, lv)], ref.disp[t][lv].numpy()) == 0, (rank, t, lv)
