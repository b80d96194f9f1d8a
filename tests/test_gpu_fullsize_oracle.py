"""Oracle parity AT THE SIZES BASELINE.json IS QUOTED ON. A whole frame at 16 x 2048^2 costs the CPU oracle
minutes, one destination camera at one level costs seconds: for one destination the GPU's level-1 result of a
full-pyramid run seeds the oracle's level 0 (the between-level upsample, precomputeProjections, processLevel:
DerpCLI.cpp:220-323, Derp.cpp:1005-1034, with the size-dependent varNoiseFloor of PyramidLevel.h:232-236), and
the oracle's level-0 disparity and its computeCost / computeSSD counters are compared with the GPU's. The
destinations of a level do not interact (mismatches_start_level = -1), so a context with that single destination
computes the same map as the 16-destination run — asserted bit for bit — and its device counters are the
destination's own. Config 3 adds the filter stage (TemporalBilateralFilter.h:126-215) on the GPU's raw level-0
maps of the five window frames; config 5 the masks, the background and UpsampleDisparity's guided filter
(UpsampleDisparity.cpp:109-128) at 2048^2."""
import time

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


def _differ(a, b):
    """elements that differ bitwise, NaN == NaN"""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())


def _one_destination(rig, sizes, res, frame, cam, **opts):
    """Full pyramid for destination `cam` alone -> (level-1 map, level-0 map, level-0 counters per stage)."""
    from facebook360_dep_amd import derp

    g = derp.Derp(rig["cameras"], [rig["cameras"][cam]], **opts)
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    lvl1, lvl0 = g.download_disparity(1, 0), g.download_disparity(0, 0)
    cnt = {k: sum(g.profile_query(st, 0)[k] for st in ("random_proposals", "ping_pong")) for k in ("n_cost", "n_pair")}
    g.close()
    return lvl1, lvl0, cnt


def _oracle_level0(rig, sizes, res, frame, cam, prev, **opts):
    t = time.time()
    L = common.oracle_level(rig, sizes, frame, 0, res, res, prev=[prev], dst_ids=[rig["cameras"][cam]["id"]],
                            threads=-1, **opts)
    L.process()
    c = L.counters()
    print("oracle level 0 of one destination at %d^2, %d sources: %.1f s" % (res, len(rig["cameras"]) - 1, time.time() - t))
    assert c["check_failed"] == 0
    return L.get_dst(0)[0], c


def test_config2_level0_one_destination_against_oracle(built):
    """BASELINE config 2 (16 x 2048^2, 10 levels): destination 3's level 0 against the oracle."""
    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, device="cuda")
    cam = 3
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    full1, full0 = g.download_disparity(1, cam), g.download_disparity(0, cam)
    g.close()
    one1, one0, cnt = _one_destination(rig, sizes, res, frame, cam)
    assert _differ(one1, full1) == 0 and _differ(one0, full0) == 0, "destinations are not independent"
    ref, c = _oracle_level0(rig, sizes, res, frame, cam, full1)
    bad = _differ(full0, ref)
    print("config 2, camera %d, level 0: %d of %d values differ from the oracle" % (cam, bad, ref.size))
    assert common.compare_disparity(full0, ref, 1e-4)[0] <= 1e-5 * ref.size
    common.observed("fullsize.cfg2.level0.cam3.float_differences", bad)
    assert cnt["n_cost"] == c["n_cost"]
    assert cnt["n_pair"] == c["n_pair"]


def test_config5_level0_and_guided_upsample_against_oracle(built):
    """BASELINE config 5 at 2048^2: foreground masks + background disparity through level 0, then
    UpsampleDisparity level 1 -> level 0 with both masks, the background and the colour guide
    (generalizedJointBilateralFilter<float, Vec3f>, radius from the scale) — all against the oracle."""
    from facebook360_dep_amd import derp, synth
    from oracle import oracle_lib as O

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, with_masks=True, device="cuda")
    # a destination that sees the foreground planes
    cam = max(range(n), key=lambda d: int(frame["masks"][0][d].sum()))
    assert frame["masks"][0][cam].sum() > 1000
    lvl1, lvl0, cnt = _one_destination(rig, sizes, res, frame, cam, use_foreground_masks=1)
    ref, c = _oracle_level0(rig, sizes, res, frame, cam, lvl1, use_foreground_masks=True)
    bad = _differ(lvl0, ref)
    print("config 5, camera %d, level 0: %d of %d values differ from the oracle" % (cam, bad, ref.size))
    assert common.compare_disparity(lvl0, ref, 1e-4)[0] <= 1e-5 * ref.size
    common.observed("fullsize.cfg5.level0.float_differences", bad)
    assert cnt["n_cost"] == c["n_cost"] and cnt["n_pair"] == c["n_pair"]
    # UpsampleDisparity.cpp:65-144 as pipeline.py:409-443 calls it
    g = derp.Derp(rig["cameras"], use_foreground_masks=1)
    _, rd, _ = common.oracle_rigs(rig)
    w1, h1 = sizes[1]
    bg, fg1, fg0 = frame["bg_disp"][0][cam], frame["masks"][1][cam], frame["masks"][0][cam]
    guide = frame["color"][0][cam].astype(np.float32) * np.float32(1.0 / 65535.0)
    radius = O.upsample_radius(w1, res)
    assert radius > 0
    for masked in (True, False):
        if masked:
            up = g.upsample_disparity(cam, lvl1, res, res, bg_up=bg, fg=fg1, fg_up=fg0)
            want = O.upsample_disparity(rd, cam, lvl1, res, res, bg, fg1, fg0)
            mask = fg0
        else:
            up = g.upsample_disparity(cam, lvl1, res, res)
            want = O.upsample_disparity(rd, cam, lvl1, res, res)
            mask = np.ones((res, res), np.uint8)
        assert _differ(up, want) == 0
        t = time.time()
        got = g.joint_bilateral_f32(up, guide, mask, radius, 0.05, 0.5, 0.5, 1.0)
        want = O.joint_bilateral_f32(want, guide, mask, radius, 0.05, 0.5, 0.5, 1.0)
        print("guided filter radius %d at %d^2 (masked=%s): oracle + GPU %.1f s" % (radius, res, masked, time.time() - t))
        bad = _differ(got, want)
        assert common.compare_disparity(got, want, 1e-5)[0] == 0
        common.observed("fullsize.cfg5.guided_upsample.masked=%s" % masked, bad)
    g.close()


def test_config3_level0_and_temporal_filter_against_oracle(built):
    """BASELINE config 3 (8 frames of the 16 x 2048^2 rig, per-level temporal filter): levels 9..1 on the GPU
    with the filter in the loop; at level 0 frame 3's processLevel — seeded by the FILTERED level 1 — and then
    the filter over the GPU's raw level-0 maps of frames 1..5 are compared with the oracle for one camera."""
    from facebook360_dep_amd import derp, sequence, synth
    from oracle import oracle_lib as O

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    first, last, t0, cam = 0, 7, 3, 9
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    r = sequence.SequenceRunner(g, first, last)
    guides, frame3 = {}, None
    for t in r.owned:
        fr = synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cuda")
        r.upload_frame(t, fr)
        guides[t] = fr["color"][0][cam]
        if t == t0:
            frame3 = fr
    r.run(level_end=1)
    seed = r.download_disparity(t0, 1, cam)  # filtered level 1
    r.compute(0)
    lo, hi = sequence.temporal_window(t0, first, last, 2)
    assert (lo, hi) == (1, 5)
    raw = {t: r.download_disparity(t, 0, cam) for t in range(lo, hi + 1)}
    r.filter(0)
    g.synchronize()
    filtered = r.download_disparity(t0, 0, cam)
    fov = g.fov_mask(cam, res, res)
    r.close()
    g.close()
    # processLevel(level 0) of frame 3
    ref, _ = _oracle_level0(rig, sizes, res, frame3, cam, seed)
    bad = _differ(raw[t0], ref)
    print("config 3, frame %d camera %d, raw level 0: %d of %d values differ from the oracle" % (t0, cam, bad, ref.size))
    assert common.compare_disparity(raw[t0], ref, 1e-4)[0] <= 1e-5 * ref.size
    common.observed("fullsize.cfg3.level0.raw.float_differences", bad)
    # the filter: sigma 0.01, weights (b, g, b) — TemporalBilateralFilter.cpp:176-178
    t = time.time()
    want = O.temporal_filter([guides[u] for u in range(lo, hi + 1)], [raw[u] for u in range(lo, hi + 1)],
                             [fov] * (hi - lo + 1), t0 - lo, 0.01, O.temporal_space_radius(0), 0.5, 1.0, 0.5, threads=-1)
    print("oracle temporal filter at %d^2, 5 frames: %.1f s" % (res, time.time() - t))
    bad = _differ(filtered, want)
    assert common.compare_disparity(filtered, want, 1e-5)[0] == 0
    common.observed("fullsize.cfg3.level0.filtered.float_differences", bad)


def test_config4_level0_one_destination_against_oracle(built):
    """BASELINE config 4 (24 x 4096^2, 11 levels): destination 11's level 0 against the oracle
    (varNoiseFloor scales with (W / heightFull)^2, parallax up to ~800 px, 23 sources)."""
    import torch

    from facebook360_dep_amd import synth

    n, res, widths = synth.config("cfg4")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, device="cuda")
    torch.cuda.empty_cache()
    cam = 11
    lvl1, lvl0, cnt = _one_destination(rig, sizes, res, frame, cam)
    ref, c = _oracle_level0(rig, sizes, res, frame, cam, lvl1)
    bad = _differ(lvl0, ref)
    print("config 4, camera %d, level 0: %d of %d values differ from the oracle" % (cam, bad, ref.size))
    assert common.compare_disparity(lvl0, ref, 1e-4)[0] <= 1e-5 * ref.size
    common.observed("fullsize.cfg4.level0.cam11.float_differences", bad)
    assert cnt["n_cost"] == c["n_cost"] and cnt["n_pair"] == c["n_pair"]


def test_config2_levels_2_and_1_all_destinations_and_a_second_level0_against_oracle(built):
    """BASELINE config 2, wider than one destination's level 0: levels 2 (512^2) and 1 (1024^2) of ALL 16
    destinations — each seeded by the GPU's level above, like DerpCLI's loop (DerpCLI.cpp:220-323) — and a second
    level-0 destination, against the oracle: every map bit for bit, and the levels' computeCost / computeSSD counters."""
    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, device="cuda")
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    maps = {lvl: [g.download_disparity(lvl, d) for d in range(n)] for lvl in (3, 2, 1)}
    cam = 12
    full0 = g.download_disparity(0, cam)
    cnt = {lvl: {k: sum(g.profile_query(st, lvl)[k] for st in ("random_proposals", "ping_pong")) for k in ("n_cost", "n_pair")}
           for lvl in (2, 1)}
    g.close()
    for lvl in (2, 1):
        t = time.time()
        L = common.oracle_level(rig, sizes, frame, lvl, res, res, prev=maps[lvl + 1], threads=-1)
        L.process()
        c = L.counters()
        print("oracle level %d of all %d destinations at %d^2: %.1f s" % (lvl, n, sizes[lvl][0], time.time() - t))
        assert c["check_failed"] == 0
        bad = sum(_differ(maps[lvl][d], L.get_dst(d)[0]) for d in range(n))
        print("config 2, level %d, all destinations: %d of %d values differ from the oracle" % (lvl, bad, n * maps[lvl][0].size))
        for d in range(n):
            assert common.compare_disparity(maps[lvl][d], L.get_dst(d)[0], 1e-4)[0] <= 1e-5 * maps[lvl][d].size
        common.observed("fullsize.cfg2.level%d.all_destinations.float_differences" % lvl, bad)
        assert cnt[lvl]["n_cost"] == c["n_cost"]
        assert cnt[lvl]["n_pair"] == c["n_pair"]
    one1, one0, cnt0 = _one_destination(rig, sizes, res, frame, cam)
    assert _differ(one1, maps[1][cam]) == 0 and _differ(one0, full0) == 0, "destinations are not independent"
    ref, c = _oracle_level0(rig, sizes, res, frame, cam, maps[1][cam])
    bad = _differ(full0, ref)
    print("config 2, camera %d, level 0: %d of %d values differ from the oracle" % (cam, bad, ref.size))
    assert common.compare_disparity(full0, ref, 1e-4)[0] <= 1e-5 * ref.size
    common.observed("fullsize.cfg2.level0.cam12.float_differences", bad)
    assert cnt0["n_cost"] == c["n_cost"] and cnt0["n_pair"] == c["n_pair"]


def test_config3_temporal_filter_clamped_windows_against_oracle(built):
    """BASELINE config 3 at 2048^2, the windows populateMinMaxFrame clamps at the ends of the sequence
    (TemporalBilateralFilter.cpp:96-119): frame 0 filters over frames 0..2, frame 7 over 5..7. The GPU's raw level-0 maps
    of those frames go through the oracle's filter; the filtered maps of both end frames must equal it bit for bit."""
    from facebook360_dep_amd import derp, sequence, synth
    from oracle import oracle_lib as O

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    first, last, cam = 0, 7, 5
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    r = sequence.SequenceRunner(g, first, last)
    guides = {}
    for t in r.owned:
        fr = synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cuda")
        r.upload_frame(t, fr)
        guides[t] = fr["color"][0][cam]
    r.run(level_end=1)
    r.compute(0)
    windows = {t0: sequence.temporal_window(t0, first, last, 2) for t0 in (first, last)}
    assert windows == {0: (0, 2), 7: (5, 7)}
    raw = {t: r.download_disparity(t, 0, cam) for t in (0, 1, 2, 5, 6, 7)}
    r.filter(0)
    g.synchronize()
    filtered = {t0: r.download_disparity(t0, 0, cam) for t0 in windows}
    fov = g.fov_mask(cam, res, res)
    r.close()
    g.close()
    for t0, (lo, hi) in windows.items():
        t = time.time()
        want = O.temporal_filter([guides[u] for u in range(lo, hi + 1)], [raw[u] for u in range(lo, hi + 1)],
                                 [fov] * (hi - lo + 1), t0 - lo, 0.01, O.temporal_space_radius(0), 0.5, 1.0, 0.5, threads=-1)
        print("oracle temporal filter at %d^2, frame %d over %d..%d: %.1f s" % (res, t0, lo, hi, time.time() - t))
        bad = _differ(filtered[t0], want)
        assert common.compare_disparity(filtered[t0], want, 1e-5)[0] == 0
        common.observed("fullsize.cfg3.level0.filtered.clamped_window.frame%d" % t0, bad)
