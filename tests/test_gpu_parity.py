"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs. Integer / index / mask work must be bit-exact; float disparity within 1e-4 relative
(BASELINE.json north_star) — in practice the stages below are bit-exact except where fp64
libm (host) and OCML (device) transcendentals differ in the last ulp."""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def small(built):
    from facebook360_dep_amd import synth

    n, res, widths = synth.config("small")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, with_masks=True)
    return dict(rig=rig, sizes=sizes, frame=frame, res=res, n=n)


@pytest.fixture(scope="module")
def gpu(small):
    from facebook360_dep_amd import derp

    g = derp.Derp(small["rig"]["cameras"], partial_coverage=1)
    g.set_pyramid(small["sizes"], small["res"], small["res"])
    g.upload_frame({"color": small["frame"]["color"]})
    yield g
    g.close()


def _float_equal(a, b):
    """count of elements that differ bitwise, NaN == NaN"""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    return int((~same).sum())


def test_level_tables(small, gpu):
    """fov masks, variances, projection warps, reprojected colours and colour biases."""
    level = 1
    L = common.oracle_level(small["rig"], small["sizes"], small["frame"], level, small["res"], small["res"],
                            partial_coverage=True)
    L.reproject_colors()
    gpu.level_begin(level)
    gpu.stage("reproject_colors")
    n = small["n"]
    for d in range(n):
        assert np.array_equal(gpu.debug(d, 0, "fov"), L.fov_mask(d)), "fov mask (bit-exact) dst %d" % d
    for s in range(n):
        assert _float_equal(gpu.debug(0, s, "variance"), L.variance(s)) == 0, "variance src %d" % s
    total = {"warp": 0, "color": 0, "bias": 0}
    for d in range(n):
        for s in range(n):
            if s == d:
                continue
            w_bad = _float_equal(gpu.debug(d, s, "warp"), L.proj(d, s, "warp"))
            c_bad = int((gpu.debug(d, s, "color") != L.proj(d, s, "color")).sum())
            b_bad = int((gpu.debug(d, s, "bias") != L.proj(d, s, "bias")).sum())
            total["warp"] += w_bad
            total["color"] += c_bad
            total["bias"] += b_bad
            # fp64 atan2/sin/cos differ in the last ulp between glibc and OCML: allow a handful of
            # float-rounding flips per table, nothing systematic
            assert w_bad <= 4 and c_bad <= 12 and b_bad <= 40, (d, s, w_bad, c_bad, b_bad)
    print("table elements differing from the oracle:", total)
    common.observed("level_tables.small.level1", total)  # the observed counts themselves are pinned


def test_against_committed_golden_fixture(built):
    """The HIP path against tests/golden/oracle_tiny.npz directly (no oracle call at test time): the
    'tiny' rig's pyramids with and without foreground masks, level-1 tables, and the sibling stages."""
    import os

    from facebook360_dep_amd import derp, synth

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_tiny.npz"))
    n, res, widths = synth.config("tiny")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, with_masks=True, device="cpu")
    assert np.array_equal(frame["color"][1][0], gold["input_color_l1_cam0"])
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, res, res)
    g.upload_frame({"color": frame["color"]})
    g.process_pyramid()
    g.synchronize()
    disps = [g.download_disparity(0, d) for d in range(n)]
    for d in range(n):
        assert common.compare_disparity(disps[d], gold["plain_l0"][d], TOL)[0] == 0
        assert common.compare_disparity(g.download_disparity(2, d), gold["plain_l2"][d], TOL)[0] == 0
    c = g.counters()
    assert c["n_cost"] == int(gold["plain_counters"][:, 0].sum()) and c["n_pair"] == int(gold["plain_counters"][:, 1].sum())
    g.level_begin(1)
    g.stage("reproject_colors")
    flips = [_float_equal(g.debug(1, 0, "warp"), gold["warp_1_0"]), int((g.debug(1, 0, "color") != gold["color_1_0"]).sum()),
             int((g.debug(1, 0, "bias") != gold["bias_1_0"]).sum())]
    assert flips[0] <= 4 and flips[1] <= 12 and flips[2] <= 40
    common.observed("golden_fixture.tiny.tables_1_0", flips)
    assert _float_equal(g.debug(0, 2, "variance"), gold["variance_2"]) == 0
    assert np.array_equal(g.debug(3, 0, "fov"), gold["fov_3"])
    # sibling stages on the golden disparities
    gd = [np.asarray(x, dtype=np.float32) for x in gold["plain_l0"]]
    rendered = g.rephotograph(1, frame["color"][0], gd)
    assert _float_equal(rendered, gold["rephoto_cam1"]) == 0
    mask = (np.isfinite(gd[1]) & (gd[1] > 0)).astype(np.uint8)
    x = frame["color"][0][1].astype(np.float32) * (np.float32(1.0) / np.float32(65535.0)) * mask[..., None]
    score = g.ssim(x, gold["rephoto_cam1"][..., :3], 1)
    assert _float_equal(score, gold["ssim_cam1"]) == 0
    assert _float_equal(g.ssim(x, gold["rephoto_cam1"][..., :3], 2, 0, 0, 1), gold["ncc_cam1_r2"]) == 0
    assert derp.average_score(score, mask) == list(gold["mssim_cam1"])
    g.rephotograph_upload(frame["color"][0], gd)
    centre = rig["cameras"][1]["origin"]
    assert _float_equal(g.canopy_cubemap([0, 1, 0, 0], centre, 32), gold["canopy_only_cam1"]) == 0
    assert _float_equal(g.canopy_cubemap([1, 0, 1, 1], centre, 32), gold["canopy_others_cam1"]) == 0
    frame1 = synth.make_frame(rig, sizes, frame=3, device="cpu")
    assert np.array_equal(g.generate_foreground_mask(frame["color"][0][0], frame1["color"][0][0], 1, 0.04, 4),
                          gold["fgmask_cam0"])
    fgd = np.where(frame["masks"][0][0] == 1, gd[0], 0).astype(np.float32)
    assert np.array_equal(g.layer_disparities(fgd, frame["bg_disp"][0][0]), gold["layers_cam0"])
    g.close()
    g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=1)
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    for d in range(n):
        assert common.compare_disparity(g.download_disparity(0, d), gold["fg_l0"][d], TOL)[0] == 0
    g.close()
    # BASELINE config 3's schedule (3 frames, temporal filter at every level) against the committed result
    from facebook360_dep_amd import sequence

    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, res, res)
    r = sequence.SequenceRunner(g, 0, 2)
    for t in r.owned:
        fr = synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu")
        if t == 2:  # the rendering does not depend on the host that renders it
            assert np.array_equal(fr["color"][1][0], gold["seq3_input_color_l1_cam0_frame2"])
        r.upload_frame(t, fr)
    r.compute(len(sizes) - 1)
    g.synchronize()
    assert _float_equal(np.stack([r.download_disparity(1, len(sizes) - 1, d) for d in range(n)]), gold["seq3_raw_l2_frame1"]) == 0
    r.run()
    g.synchronize()
    for t in range(3):
        for d in range(n):
            assert _float_equal(r.download_disparity(t, 0, d), gold["seq3_l0"][t][d]) == 0, (t, d)
    g.close()


def test_cost_map(small, gpu):
    """computeCost on a random disparity field: cost and confidence."""
    level = 1
    L = common.oracle_level(small["rig"], small["sizes"], small["frame"], level, small["res"], small["res"],
                            partial_coverage=True)
    L.reproject_colors()
    gpu.level_begin(level)
    gpu.stage("reproject_colors")
    w, h = small["sizes"][level]
    rng = np.random.default_rng(5)
    disp = (1.0 / rng.uniform(0.6, 30.0, size=(h, w))).astype(np.float32)
    for d in (0, 3):
        rc, rf = L.cost_map(d, disp)
        gc, gf = gpu.cost_map(d, disp)
        inner = (slice(1, h - 1), slice(1, w - 1))
        bad_c, rel = common.compare_disparity(gc[inner], rc[inner], 1e-6)
        assert _float_equal(gf[inner], rf[inner]) == 0
        exact = _float_equal(gc[inner], rc[inner])
        print("dst", d, "cost not bit-exact:", exact, "of", gc[inner].size, "max rel", rel)
        assert bad_c == 0, (d, bad_c, rel)
        assert exact <= 0.001 * gc[inner].size


def test_brute_force(small, gpu):
    level = len(small["sizes"]) - 1
    L = common.oracle_level(small["rig"], small["sizes"], small["frame"], level, small["res"], small["res"],
                            partial_coverage=True)
    L.reproject_colors()
    L.brute_force()
    gpu.level_begin(level)
    gpu.stage("reproject_colors")
    gpu.stage("brute_force")
    gpu.synchronize()
    for d in range(small["n"]):
        rd, rc, rf = L.get_dst(d)
        gd = gpu.get_level_disparity(d)
        gc, gf = gpu.download_cost(level, d)
        bad, rel = common.compare_disparity(gd, rd, TOL)
        assert bad == 0, ("disparity", d, bad, rel)
        assert common.compare_disparity(gc, rc, 1e-5)[0] == 0
        assert _float_equal(gf, rf) == 0
    oc = L.counters()
    gcnt = gpu.profile_query("brute_force", level)
    assert gcnt["n_cost"] == oc["n_cost"], (gcnt, oc)
    assert abs(gcnt["n_pair"] - oc["n_pair"]) <= 4, (gcnt, oc)


def test_random_proposals_and_ping_pong(small, gpu):
    level = 0
    rng = np.random.default_rng(11)
    w, h = small["sizes"][level]
    # smooth-ish start: truth perturbed, as an upsampled coarse estimate would be
    start = [(small["frame"]["truth"][d] * rng.uniform(0.8, 1.25, size=(h, w))).astype(np.float32)
             for d in range(small["n"])]
    L = common.oracle_level(small["rig"], small["sizes"], small["frame"], level, small["res"], small["res"],
                            partial_coverage=True)
    L.reproject_colors()
    gpu.level_begin(level)
    gpu.stage("reproject_colors")
    for d in range(small["n"]):
        L.set_dst(d, disparity=start[d])
        gpu.set_level_disparity(d, start[d])
    L.random_proposals()
    gpu.stage("random_proposals")
    changed = 0
    for d in range(small["n"]):
        rd, rc, rf = L.get_dst(d)
        gd = gpu.get_level_disparity(d)
        gc, gf = gpu.download_cost(level, d)
        assert _float_equal(gd, rd) == 0, ("random proposals disparity", d, _float_equal(gd, rd))
        assert common.compare_disparity(gc, rc, 1e-5)[0] == 0
        changed += int((rd != start[d]).sum())
    assert changed > 0, "random proposals changed nothing: the test would be vacuous"
    L.ping_pong()
    gpu.stage("ping_pong")
    for d in range(small["n"]):
        rd, rc, rf = L.get_dst(d)
        gd = gpu.get_level_disparity(d)
        gc, _ = gpu.download_cost(level, d)
        assert _float_equal(gd, rd) == 0, ("ping pong disparity", d, _float_equal(gd, rd))
        assert common.compare_disparity(gc, rc, 1e-5)[0] == 0
    oc = L.counters()
    rp = gpu.profile_query("random_proposals", level)
    pp = gpu.profile_query("ping_pong", level)
    assert rp["n_cost"] + pp["n_cost"] == oc["n_cost"]


def test_filters(small, gpu):
    from oracle import oracle_lib as O

    rng = np.random.default_rng(3)
    w, h = small["sizes"][0]
    guide = small["frame"]["color"][0][2]
    disp = (1.0 / rng.uniform(0.6, 30.0, size=(h, w))).astype(np.float32)
    disp[rng.random((h, w)) < 0.02] = np.nan
    mask = (rng.random((h, w)) > 0.1).astype(np.uint8)
    clean = np.nan_to_num(disp, nan=0.3)
    for radius in (1, 3, 5):
        ref = O.joint_bilateral_u16(clean, guide, mask, radius, 0.005, 0.5, 1.0, 1.0)
        got = gpu.joint_bilateral_u16(clean, guide, mask, radius, 0.005, 0.5, 1.0, 1.0)
        bad, rel = common.compare_disparity(got, ref, 1e-5)
        print("bilateral r=%d max rel %.3g bit-diff %d" % (radius, rel, _float_equal(got, ref)))
        assert bad == 0, (radius, bad, rel)
    gf = (guide.astype(np.float32) / 65535.0)
    ref = O.joint_bilateral_f32(clean, gf, mask, 5, 0.05, 0.5, 0.5, 1.0)
    got = gpu.joint_bilateral_f32(clean, gf, mask, 5, 0.05, 0.5, 0.5, 1.0)
    assert common.compare_disparity(got, ref, 1e-5)[0] == 0
    bg = rng.random((h, w)).astype(np.float32)
    for background in (None, bg):
        ref = O.masked_median(disp, background, mask, 1)
        got = gpu.masked_median(disp, background, mask, 1)
        assert _float_equal(got, ref) == 0


def test_upsample(small, gpu):
    from oracle import oracle_lib as O

    rs, rd, _ = common.oracle_rigs(small["rig"])
    rng = np.random.default_rng(9)
    for (sw, sh), (dw, dh) in (((50, 50), (60, 60)), ((60, 60), (80, 80)), ((64, 64), (128, 128)), ((40, 30), (100, 76))):
        disp = (1.0 / rng.uniform(0.6, 30.0, size=(sh, sw))).astype(np.float32)
        disp[rng.random((sh, sw)) < 0.05] = np.nan
        ref = O.upsample_disparity(rd, 1, disp, dw, dh)
        got = gpu.upsample_disparity(1, disp, dw, dh)
        assert _float_equal(got, ref) == 0, ((sw, sh), (dw, dh), _float_equal(got, ref))
        fg = (rng.random((sh, sw)) > 0.4).astype(np.uint8)
        fg_up = (rng.random((dh, dw)) > 0.3).astype(np.uint8)
        bg_up = rng.random((dh, dw)).astype(np.float32)
        ref = O.upsample_disparity(rd, 1, disp, dw, dh, bg_up, fg, fg_up)
        got = gpu.upsample_disparity(1, disp, dw, dh, bg_up, fg, fg_up)
        assert _float_equal(got, ref) == 0, ("masked", (sw, sh), (dw, dh))


def test_temporal_filter(small, gpu):
    from oracle import oracle_lib as O

    rng = np.random.default_rng(21)
    w, h = small["sizes"][0]
    n = 5
    base = small["frame"]["color"][0][1].astype(np.int64)
    guides = [np.clip(base + rng.integers(-300, 300, size=base.shape), 0, 65535).astype(np.uint16) for _ in range(n)]
    disps = [(1.0 / rng.uniform(0.6, 30.0, size=(h, w))).astype(np.float32) for _ in range(n)]
    masks = [(rng.random((h, w)) > 0.05).astype(np.uint8) for _ in range(n)]
    for off, cnt in ((2, 5), (0, 3), (1, 2)):
        ref = O.temporal_filter(guides[:cnt], disps[:cnt], masks[:cnt], off, 0.01, 1, 0.5, 1.0, 0.5)
        got = gpu.temporal_filter(guides[:cnt], disps[:cnt], masks[:cnt], off, 0.01, 1, 0.5, 1.0, 0.5)
        bad, rel = common.compare_disparity(got, ref, 1e-5)
        assert bad == 0, (off, cnt, bad, rel)


def test_temporal_filter_window_longer_than_one_launch(small, gpu):
    """--time_radius has no upper limit in the reference (TemporalBilateralFilter.cpp:55,108-109). One launch of the
    filter kernel walks 31 frames; longer windows run as consecutive launches that carry the two float sums, i.e. the
    same additions in the same order: 70 frames (radius 34 + 1 clamped side: three launches) and 32 frames (the first
    window one launch cannot hold), the filtered frame in the first, a middle and the last chunk — bit for bit."""
    from oracle import oracle_lib as O

    rng = np.random.default_rng(22)
    w, h = small["sizes"][2]
    base = small["frame"]["color"][2][0].astype(np.int64)
    n = 70
    guides = [np.clip(base + rng.integers(-300, 300, size=base.shape), 0, 65535).astype(np.uint16) for _ in range(n)]
    disps = [(1.0 / rng.uniform(0.6, 30.0, size=(h, w))).astype(np.float32) for _ in range(n)]
    masks = [(rng.random((h, w)) > 0.05).astype(np.uint8) for _ in range(n)]
    for off, cnt in ((35, 70), (3, 70), (69, 70), (31, 32), (0, 32)):
        ref = O.temporal_filter(guides[:cnt], disps[:cnt], masks[:cnt], off, 0.01, 1, 0.5, 1.0, 0.5)
        got = gpu.temporal_filter(guides[:cnt], disps[:cnt], masks[:cnt], off, 0.01, 1, 0.5, 1.0, 0.5)
        assert _float_equal(got, ref) == 0, (off, cnt)


def test_config3_temporal_filter_full_size_properties(built):
    """BASELINE config 3's filter stage at 2048^2 (one camera, 5 frames of the moving scene) through
    properties of temporalJointBilateralFilter (TemporalBilateralFilter.h:126-172): the weights depend
    on the colour guides only, so scaling every disparity by 2 scales the result by exactly 2; identical
    frames give back the input (to rounding); pixels outside the mask pass through untouched; the
    window clamped at the start of the sequence (populateMinMaxFrame) is a different, valid filter."""
    from facebook360_dep_amd import derp, synth

    res = 2048
    rig = synth.make_rig(16, res)
    cam = rig["cameras"][3]
    frames = [synth.render_camera(cam, res, res, frame=f, device="cuda") for f in range(5)]
    guides = [f[0] for f in frames]
    disps = [np.asarray(f[1], dtype=np.float32) for f in frames]
    rng = np.random.default_rng(3)
    masks = [(rng.random((res, res)) > 0.02).astype(np.uint8) for _ in range(5)]
    g = derp.Derp(rig["cameras"])
    args = (0.01, 1, 0.5, 1.0, 0.5)  # sigma, space radius, weights (b, g, b) — TemporalBilateralFilter.cpp:55,165-178
    out = g.temporal_filter(guides, disps, masks, 2, *args)
    assert np.isfinite(out[masks[2] == 1]).all()
    assert np.array_equal(out[masks[2] == 0], disps[2][masks[2] == 0])
    lo, hi = np.minimum.reduce(disps), np.maximum.reduce(disps)
    m = masks[2] == 1
    assert (out[m] >= lo[m] * (1 - 1e-5)).all() and (out[m] <= hi[m] * (1 + 1e-5)).all()  # a convex combination
    out2 = g.temporal_filter(guides, [2.0 * d for d in disps], masks, 2, *args)
    assert np.array_equal(out2, 2.0 * out)
    same = g.temporal_filter([guides[2]] * 5, [disps[2]] * 5, [masks[2]] * 5, 2, *args)
    assert np.abs(same[m] - disps[2][m]).max() <= 4e-6 * disps[2][m].max()
    head = g.temporal_filter(guides[:3], disps[:3], masks[:3], 0, *args)  # frame 0 of the sequence: window [0, 2]
    assert np.isfinite(head[masks[0] == 1]).all() and not np.array_equal(head, out)
    g.close()


def _run_pyramid(small, **opts):
    from facebook360_dep_amd import derp

    ref = common.oracle_pyramid(small["rig"], small["sizes"], small["frame"], small["res"], small["res"], **opts)
    gopts = {k: int(v) if isinstance(v, bool) else v for k, v in opts.items() if k != "threads"}
    g = derp.Derp(small["rig"]["cameras"], **gopts)
    g.set_pyramid(small["sizes"], small["res"], small["res"])
    g.upload_frame(small["frame"] if opts.get("use_foreground_masks") else {"color": small["frame"]["color"]})
    g.process_pyramid()
    g.synchronize()
    stats = {}
    for level in sorted(ref):
        nbad = npx = 0
        worst = 0.0
        for d in range(small["n"]):
            got = g.download_disparity(level, d)
            bad, rel = common.compare_disparity(got, ref[level][d], TOL)
            nbad += bad
            npx += got.size
            worst = max(worst, rel)
        stats[level] = (nbad, npx, worst)
    g.close()
    return stats


def test_full_pyramid(small):
    """processLevel over the whole pyramid (brute force, proposals, ping-pong, bilateral, median,
    maskFov, Lanczos hand-off): disparity within 1e-4 relative of the oracle."""
    stats = _run_pyramid(small, partial_coverage=True)
    print("full pyramid (level: bad, pixels, max rel):", stats)
    for level, (bad, npx, worst) in stats.items():
        assert bad <= 1e-4 * npx, (level, bad, npx, worst)
    common.observed("full_pyramid.small", {str(level): bad for level, (bad, _, _) in stats.items()})


def test_full_pyramid_foreground_masks(small):
    """BASELINE config 5's mask path: foreground masks + background disparity through every stage."""
    stats = _run_pyramid(small, partial_coverage=True, use_foreground_masks=True)
    print("fg-mask pyramid (level: bad, pixels, max rel):", stats)
    for level, (bad, npx, worst) in stats.items():
        assert bad <= 1e-4 * npx, (level, bad, npx, worst)
    common.observed("full_pyramid_fg.small", {str(level): bad for level, (bad, _, _) in stats.items()})


def test_destination_subset(small):
    """--cameras=cam4,cam1: destinations filtered and reordered (ImageUtil.cpp:110-125)."""
    from facebook360_dep_amd import derp

    ids = ["cam4", "cam1"]
    ref = common.oracle_pyramid(small["rig"], small["sizes"], small["frame"], small["res"], small["res"],
                                dst_ids=ids, partial_coverage=True)
    cams = small["rig"]["cameras"]
    g = derp.Derp(cams, derp.filter_destinations(cams, ",".join(ids)), partial_coverage=1)
    g.set_pyramid(small["sizes"], small["res"], small["res"])
    g.upload_frame({"color": small["frame"]["color"]})
    g.process_pyramid()
    g.synchronize()
    for d in range(2):
        bad, rel = common.compare_disparity(g.download_disparity(0, d), ref[0][d], TOL)
        assert bad <= 2, (d, bad, rel)
    g.close()


def test_destination_batching(small, monkeypatch):
    """Config 4's memory path: when the projection tables of all destinations do not fit the table
    budget, destinations are processed in batches (DERP_TABLE_BUDGET_GB caps the budget). Results and counters
    must not depend on the batch size; the number of batches really taken is read back (one ping-pong span per
    batch at level 0) so that the test cannot silently stop exercising the batched path."""
    from facebook360_dep_amd import derp

    def run():
        g = derp.Derp(small["rig"]["cameras"], partial_coverage=1)
        g.set_pyramid(small["sizes"], small["res"], small["res"])
        g.upload_frame({"color": small["frame"]["color"]})
        g.profile_enable(True)
        g.process_pyramid()
        g.synchronize()
        out = [[g.download_disparity(level, d) for d in range(small["n"])] for level in range(len(small["sizes"]))]
        c = g.counters()
        batches = g.profile_query("ping_pong", 0)["launches"]
        g.close()
        return out, c, batches

    whole, c_whole, b_whole = run()
    assert b_whole == 1
    seen = []
    for budget in ("0.012", "0.006"):
        monkeypatch.setenv("DERP_TABLE_BUDGET_GB", budget)
        batched, c_batched, b = run()
        seen.append(b)
        assert c_batched == c_whole
        for a, b_ in zip(whole, batched):
            for x, y in zip(a, b_):
                assert _float_equal(x, y) == 0
    print("destination batches at level 0 under 12 MB / 6 MB table budgets:", seen)
    assert 1 < seen[0] <= seen[1] <= small["n"] and seen[1] > seen[0]
    monkeypatch.setenv("DERP_TABLE_BUDGET_GB", "0.0001")
    g = derp.Derp(small["rig"]["cameras"], partial_coverage=1)
    g.set_pyramid(small["sizes"], small["res"], small["res"])
    g.upload_frame({"color": small["frame"]["color"]})
    with pytest.raises(derp.DerpError, match="table budget"):
        g.process_pyramid()
    g.close()


def test_cost_kernel_register_budgets_agree(small, monkeypatch):
    """k_ping_pong / k_random_proposals exist under two register budgets — four waves per SIMD (128 VGPRs, up to 16 cameras)
    and three (their _w3 twins, what rigs beyond 16 cameras launch: the pair slots bound the occupancy through LDS there).
    Same body: forced either way (DERP_COST_WAVES) a pyramid must come out bit for bit the same, counters included."""
    from facebook360_dep_amd import derp

    def run(waves):
        monkeypatch.setenv("DERP_COST_WAVES", waves)
        g = derp.Derp(small["rig"]["cameras"], partial_coverage=1)
        g.set_pyramid(small["sizes"], small["res"], small["res"])
        g.upload_frame({"color": small["frame"]["color"]})
        g.process_pyramid()
        g.synchronize()
        out = [[g.download_disparity(level, d) for d in range(small["n"])] for level in range(len(small["sizes"]))]
        c = g.counters()
        g.close()
        return out, c

    four, c4 = run("4")
    three, c3 = run("3")
    assert c3 == c4
    for a, b in zip(four, three):
        for x, y in zip(a, b):
            assert _float_equal(x, y) == 0


def test_sixteen_camera_rig_full_pyramid(built):
    """BASELINE config 2's camera count (16 on a Fibonacci sphere, up to 15 sources per cost, which
    exercises every branch of the nth_element restatement) at a size the oracle finishes in seconds."""
    from facebook360_dep_amd import derp, synth

    res = 128
    rig = synth.make_rig(16, res)
    sizes = synth.level_sizes(res, res, [128, 100, 80, 60, 50])
    frame = synth.make_frame(rig, sizes)
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, counters=cnt)
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()  # also the coverage CHECK: a full sphere of cameras must not trip it
    nbad = npx = 0
    for level in ref:
        for d in range(16):
            bad, rel = common.compare_disparity(g.download_disparity(level, d), ref[level][d], TOL)
            nbad += bad
            npx += ref[level][d].size
    print("16-camera rig: %d of %d pixels outside 1e-4" % (nbad, npx))
    assert nbad <= 1e-4 * npx
    common.observed("sixteen_camera_rig.128", nbad)
    got = g.counters()
    assert got["n_cost"] == sum(c["n_cost"] for c in cnt.values())
    assert abs(got["n_pair"] - sum(c["n_pair"] for c in cnt.values())) <= 1e-6 * got["n_pair"] + 4
    g.close()


def test_config2_full_size_properties(built):
    """BASELINE config 2 at full size (16 cameras, 2048^2, 10 levels) — too big for the oracle, so
    checked through size-independent properties: bit-identical reruns, NaN exactly outside the FOV
    mask, disparities inside the search range (up to Lanczos ringing), agreement with the analytic
    scene, and cost-evaluation counters equal to the closed form of the schedule."""
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
    first = [g.download_disparity(0, d) for d in (0, 7, 15)]
    c1 = g.counters()
    g.reset_counters()
    g.process_pyramid()
    g.synchronize()
    c2 = g.counters()
    assert c1 == c2
    for i, d in enumerate((0, 7, 15)):
        again = g.download_disparity(0, d)
        assert _float_equal(first[i], again) == 0, "rerun differs"
        fov = g.fov_mask(d, res, res)
        assert np.array_equal(np.isnan(again), fov == 0)
        v = again[fov == 1]
        assert v.min() > 0 and v.max() < 2.0 * 1.05
        truth = frame["truth"][d][fov == 1]
        rel = np.abs(v - truth) / truth
        print("cam %d: median rel err vs analytic scene %.4f, 90th pct %.4f" % (d, np.median(rel), np.percentile(rel, 90)))
        assert np.median(rel) < 0.01 and np.percentile(rel, 90) < 0.1
    # schedule: 150 costs/px at the coarsest level over fov interior; (1 + P) + 9 per gated pixel above
    lvl = len(sizes) - 1
    w, h = sizes[lvl]
    interior = sum(int(g.fov_mask(d, w, h)[1:-1, 1:-1].sum()) for d in range(n))
    assert g.profile_query("brute_force", lvl)["n_cost"] == 150 * interior
    for level in (0, 3):
        w, h = sizes[level]
        interior = sum(int(g.fov_mask(d, w, h)[1:-1, 1:-1].sum()) for d in range(n))
        pp = g.profile_query("ping_pong", level)["n_cost"]
        rp = g.profile_query("random_proposals", level)["n_cost"]
        assert pp <= 9 * interior and pp >= 0.9 * 9 * interior  # textured scene: nearly every pixel passes the variance gate
        assert rp <= 3 * interior and rp % 3 == 0
    g.close()


def test_config4_full_size_properties(built, monkeypatch):
    """BASELINE config 4 at full size (24 cameras, 4096^2, 11 levels — the projection tables of all 24
    destinations do not fit the table budget, so destinations are processed in batches): bit-identical
    reruns, NaN exactly outside the FOV mask, range, agreement with the analytic scene, closed-form
    evaluation counts; and, at 24 x 1024^2 where an unbatched run fits, batched == unbatched bit for bit."""
    import torch

    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg4")
    assert (n, res) == (24, 4096) and len(widths) == 11
    rig = synth.make_rig(n, res)
    # --- batched == unbatched at a size where both fit
    small_res = 1024
    rig_s = synth.make_rig(n, small_res)
    sizes_s = synth.level_sizes(small_res, small_res, synth.WIDTHS)
    frame_s = synth.make_frame(rig_s, sizes_s, device="cuda")

    def run_small():
        g = derp.Derp(rig_s["cameras"])
        g.set_pyramid(sizes_s, small_res, small_res)
        g.upload_frame(frame_s)
        g.process_pyramid()
        g.synchronize()
        out = [g.download_disparity(0, d) for d in range(n)]
        c = g.counters()
        g.close()
        return out, c

    whole, c_whole = run_small()
    monkeypatch.setenv("DERP_TABLE_BUDGET_GB", "6")  # 0.6 GB of tables per destination at 1024^2 -> batches of <= 10
    batched, c_batched = run_small()
    monkeypatch.delenv("DERP_TABLE_BUDGET_GB")
    assert c_batched == c_whole
    assert sum(_float_equal(a, b) for a, b in zip(whole, batched)) == 0
    del whole, batched, frame_s
    # --- full size under the real budget
    sizes = synth.level_sizes(res, res, widths)
    assert sizes[0] == (4096, 4096) and len(sizes) == 11
    frame = synth.make_frame(rig, sizes, device="cuda")
    torch.cuda.empty_cache()
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    cams = (0, 11, 23)
    first = [g.download_disparity(0, d) for d in cams]
    c1 = g.counters()
    g.reset_counters()
    g.process_pyramid()
    g.synchronize()
    assert g.counters() == c1
    for i, d in enumerate(cams):
        again = g.download_disparity(0, d)
        assert _float_equal(first[i], again) == 0, "rerun differs"
        fov = g.fov_mask(d, res, res)
        assert np.array_equal(np.isnan(again), fov == 0)
        v = again[fov == 1]
        assert v.min() > 0 and v.max() < 2.0 * 1.05
        truth = frame["truth"][d][fov == 1]
        rel = np.abs(v - truth) / truth
        print("cfg4 cam %d: median rel err vs analytic scene %.4f, 90th pct %.4f" % (d, np.median(rel), np.percentile(rel, 90)))
        assert np.median(rel) < 0.01 and np.percentile(rel, 90) < 0.1
    lvl = len(sizes) - 1
    w, h = sizes[lvl]
    interior = sum(int(g.fov_mask(d, w, h)[1:-1, 1:-1].sum()) for d in range(n))
    assert g.profile_query("brute_force", lvl)["n_cost"] == 150 * interior
    for level in (0, 4):
        w, h = sizes[level]
        interior = sum(int(g.fov_mask(d, w, h)[1:-1, 1:-1].sum()) for d in range(n))
        pp = g.profile_query("ping_pong", level)["n_cost"]
        rp = g.profile_query("random_proposals", level)["n_cost"]
        # at 4096^2 the same scene is sampled twice as finely: more pixels fall under the variance gate
        assert pp <= 9 * interior and pp >= 0.8 * 9 * interior
        assert rp <= 3 * interior and rp % 3 == 0
    g.close()


def test_config1_full_size_against_oracle(built):
    """BASELINE config 1 (the reference's own CPU-runnable case: 4 cameras, 512^2, 8 levels) in full:
    every level of every camera against the oracle, plus the cost-evaluation counters."""
    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg1")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    assert [w for w, _ in sizes] == [512, 256, 200, 128, 100, 80, 60, 50]
    frame = synth.make_frame(rig, sizes, device="cuda")
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, counters=cnt, partial_coverage=True)
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    nbad = npx = 0
    for level in ref:
        for d in range(n):
            bad, rel = common.compare_disparity(g.download_disparity(level, d), ref[level][d], TOL)
            nbad += bad
            npx += ref[level][d].size
    print("config 1: %d of %d pixels outside 1e-4" % (nbad, npx))
    assert nbad == 0
    c = g.counters()
    assert c["n_cost"] == sum(v["n_cost"] for v in cnt.values())
    assert c["n_pair"] == sum(v["n_pair"] for v in cnt.values())
    g.close()


def test_config2_rig_at_512_against_oracle(built):
    """BASELINE config 2's rig (16 cameras, Fibonacci sphere) at 512^2 with the full 8-level pyramid,
    every level of every camera against the oracle (the largest size the oracle finishes in seconds
    on the GPU box's host cores)."""
    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg2s")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, device="cuda")
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, counters=cnt)
    g = derp.Derp(rig["cameras"])
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    nbad = npx = 0
    for level in ref:
        for d in range(n):
            bad, rel = common.compare_disparity(g.download_disparity(level, d), ref[level][d], TOL)
            nbad += bad
            npx += ref[level][d].size
    print("config-2 rig at 512^2: %d of %d pixels outside 1e-4" % (nbad, npx))
    assert nbad <= 1e-5 * npx
    common.observed("config2_rig.512", nbad)
    c = g.counters()
    assert c["n_cost"] == sum(v["n_cost"] for v in cnt.values())
    g.close()


def test_config5_full_size_properties(built):
    """BASELINE config 5 at full size (16 cameras, 2048^2, foreground masks + background disparity,
    then UpsampleDisparity level 1 -> level 0 with the colour guide) through properties the domain
    offers: outside the foreground mask the result IS the background disparity, inside it is never
    behind the background, NaN exactly outside the FOV, reruns are bit-identical, and the masked
    upsample returns background exactly where UpsampleDisparityLib.cpp:118-139 says."""
    from facebook360_dep_amd import derp, synth

    n, res, widths = synth.config("cfg2")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, with_masks=True, device="cuda")
    g = derp.Derp(rig["cameras"], use_foreground_masks=1)
    g.set_pyramid(sizes, res, res)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    cams = (0, 5, 15)
    first = {d: g.download_disparity(0, d) for d in cams}
    g.process_pyramid()
    g.synchronize()
    any_fg = 0
    for d in cams:
        disp = g.download_disparity(0, d)
        assert _float_equal(disp, first[d]) == 0, "rerun differs"
        fov = g.fov_mask(d, res, res) == 1
        fg = frame["masks"][0][d] == 1
        bg = frame["bg_disp"][0][d]
        assert np.array_equal(np.isnan(disp), ~fov)
        outside = fov & ~fg
        assert np.array_equal(disp[outside], bg[outside])
        inside = fov & fg
        any_fg += int(inside.sum())
        if inside.any():
            # foreground is never placed behind the background, up to the median / bilateral blend
            assert (disp[inside] >= bg[inside] * (1 - 1e-3)).mean() > 0.99
            truth = frame["truth"][d][inside]
            rel = np.abs(disp[inside] - truth) / truth
            assert np.median(rel) < 0.02
        # UpsampleDisparity level 1 -> level 0 (masked nearest + spiral fill + background)
        w1, h1 = sizes[1]
        up = g.upsample_disparity(d, g.download_disparity(1, d), res, res, bg_up=bg, fg=frame["masks"][1][d],
                                  fg_up=frame["masks"][0][d])
        # outside fov & fg every pixel ends as background (the last step replaces what is still NaN / 0)
        assert np.isfinite(up).all()
        assert np.array_equal(up[~inside], bg[~inside])
        assert (up[inside] > 0).all()
    assert any_fg > 0
    g.close()


def test_mismatch_handling(small):
    """handleDisparityMismatches (Derp.cpp:553-748, --mismatches_start_level): the only stage in which
    one destination reads the other cameras' disparities."""
    from facebook360_dep_amd import derp

    opts = dict(partial_coverage=True, mismatches_start_level=1)
    # stage level, from identical inputs
    level = 0
    rng = np.random.default_rng(4)
    w, h = small["sizes"][level]
    start = [(small["frame"]["truth"][d] * rng.uniform(0.7, 1.4, size=(h, w))).astype(np.float32)
             for d in range(small["n"])]
    L = common.oracle_level(small["rig"], small["sizes"], small["frame"], level, small["res"], small["res"], **opts)
    g = derp.Derp(small["rig"]["cameras"], partial_coverage=1, mismatches_start_level=1)
    g.set_pyramid(small["sizes"], small["res"], small["res"])
    g.upload_frame({"color": small["frame"]["color"]})
    g.level_begin(level)
    for d in range(small["n"]):
        L.set_dst(d, disparity=start[d])
        g.set_level_disparity(d, start[d])
    L.mismatches()
    g.stage("mismatches")
    flagged = 0
    for d in range(small["n"]):
        assert _float_equal(g.get_level_disparity(d), L.get_dst(d)[0]) == 0, d
        assert np.array_equal(g.mismatch_mask(d), L.mismatch_mask(d)), d
        flagged += int(L.mismatch_mask(d).sum())
    assert flagged > 0, "no pixel was flagged: the test would be vacuous"
    # whole pyramid with the stage switched on
    ref = common.oracle_pyramid(small["rig"], small["sizes"], small["frame"], small["res"], small["res"], **opts)
    g.process_pyramid()
    g.synchronize()
    for d in range(small["n"]):
        bad, rel = common.compare_disparity(g.download_disparity(0, d), ref[0][d], TOL)
        assert bad == 0, (d, bad, rel)
    assert g.profile_query("mismatches", 0)["launches"] >= 0
    g.close()


def test_layer_disparities(gpu):
    from oracle import oracle_lib as O

    rng = np.random.default_rng(8)
    fg = rng.uniform(-0.2, 1.3, size=(37, 53)).astype(np.float32)
    fg[rng.random(fg.shape) < 0.2] = np.nan
    fg[rng.random(fg.shape) < 0.1] = 0.0
    bg = rng.uniform(0.0, 1.1, size=fg.shape).astype(np.float32)
    assert np.array_equal(gpu.layer_disparities(fg, bg), O.layer_disparities(fg, bg))


def test_pyramid_builder(built):
    """scripts/render/resize.py on the GPU: one full-size frame in, every level in HBM; then the depth
    path on the built pyramid equals the oracle fed with the oracle-built pyramid."""
    from facebook360_dep_amd import derp, synth
    from oracle import oracle_lib as O

    res = 256
    rig = synth.make_rig(4, res)
    sizes = synth.level_sizes(res, res, [256, 200, 128, 100, 80, 60, 50])
    full = synth.make_frame(rig, [(res, res)], with_masks=True)
    g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=1)
    g.set_pyramid(sizes, res, res)
    color = [[None] * 4 for _ in sizes]
    masks = [[None] * 4 for _ in sizes]
    bgd = [[None] * 4 for _ in sizes]
    for s in range(4):
        m255 = full["masks"][0][s] * 255
        g.build_pyramid_color(s, full["color"][0][s])
        g.build_pyramid_foreground_mask(s, m255, 127)
        g.build_pyramid_background_disparity(s, full["bg_disp"][0][s])
        for li, (w, h) in enumerate(sizes):
            color[li][s] = O.cv_resize_area(full["color"][0][s], w, h)
            masks[li][s] = (O.cv_resize_area(m255, w, h) > 127).astype(np.uint8)
            bgd[li][s] = O.cv_resize_area(full["bg_disp"][0][s], w, h)
            assert np.array_equal(g.download_level_color(li, s), color[li][s]), ("colour", li, s)
            assert np.array_equal(g.download_level_mask(li, s), masks[li][s]), ("mask", li, s)
            assert _float_equal(g.download_level_background(li, s), bgd[li][s]) == 0, ("background", li, s)
    # stand-alone entry point, incl. a non-square fractional case
    rng = np.random.default_rng(2)
    img = rng.integers(0, 65536, size=(216, 336, 3)).astype(np.uint16)
    for dw, dh in ((168, 108), (100, 64), (50, 32)):
        assert np.array_equal(g.resize_area(img, dw, dh), O.cv_resize_area(img, dw, dh)), (dw, dh)
    frame = {"color": color, "masks": masks, "bg_disp": bgd}
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, partial_coverage=True, use_foreground_masks=True)
    g.process_pyramid()
    g.synchronize()
    for d in range(4):
        bad, rel = common.compare_disparity(g.download_disparity(0, d), ref[0][d], TOL)
        assert bad == 0, (d, bad, rel)
    g.close()


def test_resize_area_vec3f(gpu):
    """cv_util::resizeImage<cv::Vec3f> (CvUtil.h:139-147), the colour guide of UpsampleDisparity.cpp:117:
    integer scale 2 and 3 (float block sums, no 2x2 integer shortcut), fractional scales, identity."""
    from oracle import oracle_lib as O

    rng = np.random.default_rng(7)
    src = rng.random((90, 120, 3), dtype=np.float32)
    for (dw, dh) in [(60, 45), (40, 30), (80, 60), (97, 71), (120, 90), (33, 90)]:
        got = gpu.resize_area(src, dw, dh)
        want = O.cv_resize_area(src, dw, dh)
        assert got.shape == want.shape == (dh, dw, 3)
        assert _float_equal(got, want) == 0, (dw, dh)
    # ENLARGED along at least one axis: cv::resize(INTER_AREA) emulates area interpolation with its bilinear machinery
    # and area-mode taps (a colour guide smaller than UpsampleDisparity's output): integer and fractional factors, one
    # axis only, one axis up and the other down, scalar float planes
    for (dw, dh) in [(240, 180), (360, 270), (151, 113), (120, 200), (300, 90), (200, 45), (60, 180)]:
        got = gpu.resize_area(src, dw, dh)
        want = O.cv_resize_area(src, dw, dh)
        assert got.shape == want.shape == (dh, dw, 3)
        assert _float_equal(got, want) == 0, (dw, dh)
        # sanity of the restatement itself: a convex combination of source texels, exact on a constant image
        assert want.min() >= src.min() and want.max() <= src.max()
    flat = np.full((40, 50, 3), np.float32(0.3))
    assert np.allclose(O.cv_resize_area(flat, 125, 73), 0.3, rtol=0, atol=1e-7)
    plane = rng.random((31, 47), dtype=np.float32)
    assert _float_equal(gpu.resize_area(plane, 100, 64), O.cv_resize_area(plane, 100, 64)) == 0


def test_generate_foreground_mask(gpu):
    """BackgroundSubtractionUtil.h:20-60 — foreground masks must be BIT-EXACT (north star)."""
    from facebook360_dep_amd import synth
    from oracle import oracle_lib as O

    rig = synth.make_rig(4, 200)
    cam = rig["cameras"][1]
    bg = synth.render_camera(cam, 200, 200, frame=0)[0]
    rng = np.random.default_rng(6)
    fr = bg.astype(np.int64) + rng.integers(-600, 600, size=bg.shape)  # sensor noise
    yy, xx = np.mgrid[0:200, 0:200]
    blob = ((yy - 90) ** 2 + (xx - 120) ** 2 < 40 ** 2) | ((abs(yy - 30) < 3) & (xx > 20) & (xx < 150))
    fr[blob] = fr[blob] * 0.5 + 9000  # a foreground object
    fr = np.clip(fr, 0, 65535).astype(np.uint16)
    total = 0
    for blur, thr, morph in ((1, 0.04, 4), (0, 0.04, 0), (2, 0.02, 5), (3, 0.1, 3), (1, 0.0, 1)):
        ref = O.generate_foreground_mask(bg, fr, blur, thr, morph)
        got = gpu.generate_foreground_mask(bg, fr, blur, thr, morph)
        assert np.array_equal(got, ref), (blur, thr, morph, int((got != ref).sum()))
        total += int(ref.sum())
    assert 0 < total < 5 * 200 * 200


def test_reference_test_rig_non_square(built):
    """The reference's own 16-camera test rig (res/test/rigs/rig.json: real calibration, 3360x2160,
    off-centre principal points, 3-term distortion, rotation matrices that need re-unitarising) on
    non-square pyramid levels (200x130, 128x82, 100x64 — resize.py's even-height rule)."""
    import json
    import os

    from facebook360_dep_amd import derp, synth

    rig = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_test_rig.json")))
    sizes = synth.level_sizes(3360, 2160, [200, 128, 100])
    assert sizes == [(200, 130), (128, 82), (100, 64)]
    frame = synth.make_frame(rig, sizes)
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, 3360, 2160, counters=cnt, partial_coverage=True)
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, 3360, 2160)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    nbad = npx = 0
    for level in ref:
        for d in range(16):
            got = g.download_disparity(level, d)
            assert got.shape == (sizes[level][1], sizes[level][0])
            bad, rel = common.compare_disparity(got, ref[level][d], TOL)
            nbad += bad
            npx += got.size
    print("reference test rig: %d of %d pixels outside 1e-4" % (nbad, npx))
    assert nbad <= 1e-4 * npx
    common.observed("reference_test_rig", nbad)
    assert g.counters()["n_cost"] == sum(c["n_cost"] for c in cnt.values())
    # the subset / reorder path on the same rig (DerpTest.cpp's "cam4,cam15,cam0")
    sub = derp.filter_destinations(rig["cameras"], "cam4,cam15,cam0")
    g2 = derp.Derp(rig["cameras"], sub, partial_coverage=1)
    g2.set_pyramid(sizes, 3360, 2160)
    g2.upload_frame(frame)
    g2.process_pyramid()
    g2.synchronize()
    for k, d in enumerate((4, 15, 0)):
        assert _float_equal(g2.download_disparity(0, k), g.download_disparity(0, d)) == 0
    g.close()
    g2.close()


def test_rephotography_score(small, gpu):
    """computeSSIM / averageScore (RephotographyUtil.h:38-108) bit-exact against the
    oracle, MSSIM and NCC, radii 1..3; camera-space rephotography (z-buffer + bilinear fetch) bit-exact."""
    from facebook360_dep_amd import derp
    from oracle import oracle_lib as O

    rig, frame = small["rig"], small["frame"]
    rng = np.random.default_rng(11)
    w, h = 93, 61
    x = rng.random((h, w, 3), dtype=np.float32)
    y = np.clip(x + 0.1 * rng.standard_normal(x.shape).astype(np.float32), 0, 1).astype(np.float32)
    y[5:9, 7:30] = 0  # flat patches: sigma = 0 on one side
    x[40:48, 50:70] = 0.5
    mask = (rng.random((h, w)) > 0.2).astype(np.uint8)
    for radius in (1, 2, 3):
        for abg in ((1, 1, 1), (0, 0, 1)):
            ref = O.compute_ssim(x, y, radius, *abg)
            got = gpu.ssim(x, y, radius, *abg)
            assert _float_equal(got, ref) == 0, (radius, abg)
            a_ref, a_got = O.average_score(ref, mask), derp.average_score(got, mask)
            assert a_ref == a_got
            assert derp.format_results(a_got) == O.format_results(a_ref)
    assert 0.5 < derp.average_score(gpu.ssim(x, y), mask)[1] < 1.0
    with pytest.raises(derp.DerpError):
        gpu.ssim(x, y, 1, 0.5, 1, 1)
    # rephotography with the level-0 truth disparity and with a NaN / zero / inf riddled one
    R = O.Rig(rig["cameras"]).normalize()
    cols = frame["color"][0]
    truth = [np.asarray(t, dtype=np.float32) for t in frame["truth"]]
    broken = [t.copy() for t in truth]
    broken[1][10:20, :] = np.nan
    broken[2][:, 30:40] = 0
    broken[3][50:60, 50:60] = np.inf
    for disps in (truth, broken):
        for target in (0, len(cols) - 1, 2):
            ref = O.rephotograph(R, target, cols, disps)
            got = gpu.rephotograph(target, cols, disps)
            assert _float_equal(got, ref) == 0
            assert 0.2 < (ref[..., 3] > 0).mean() <= 1.0


def _mixed_type_rig(res):
    """Six cameras on a 40-degree arc, two each RECTILINEAR / EQUISOLID / ORTHOGRAPHIC plus FTHETA
    neighbours, some with and some without distortion and explicit fov (Camera.h:301-378)."""
    import math

    from facebook360_dep_amd import synth

    base = synth.make_rig(6, res, layout="arc")["cameras"]
    spec = [
        ("RECTILINEAR", 0.50, 0.75, (-0.02, 0.001, 0.0)),
        ("FTHETA", 0.36, math.pi / 2, synth.DISTORTION),
        ("EQUISOLID", 0.36, 1.4, (0.0, 0.0, 0.0)),
        ("ORTHOGRAPHIC", 0.52, 1.2, (0.01, 0.0, 0.0)),
        ("RECTILINEAR", 0.55, None, (0.0, 0.0, 0.0)),
        ("EQUISOLID", 0.40, None, (-0.01, 0.002, -0.0005)),
    ]
    cams = []
    for i, (cam, (kind, f, fov, dist)) in enumerate(zip(base, spec)):
        az = math.radians(-20.0 + 8.0 * i)
        fwd = np.array([math.cos(az), math.sin(az), 0.05 * ((i % 2) * 2 - 1)])
        fw, up, right = synth._frame(fwd)
        cam = dict(cam, type=kind, focal=[f * res, -f * res], forward=fw.tolist(), up=up.tolist(),
                   right=right.tolist(), origin=(0.25 * fw).tolist())
        cam.pop("fov", None)
        cam.pop("distortion", None)
        if fov is not None:
            cam["fov"] = fov
        if any(dist):
            cam["distortion"] = list(dist)
        cams.append(cam)
    return {"cameras": cams}


def test_canopy_cubemap(built):
    """The rephotography renderer (CanopyScene::cubemap, CanopyScene.cpp:198-374, as
    ComputeRephotographyErrors.cpp:77-95 drives it) against the oracle's restatement: the camera alone and
    all the other cameras, seen from the camera, on analytic and on perturbed disparities (depth
    discontinuities, a NaN hole, stretched triangles). Bit for bit, including the {0,1} alpha; then the MSSIM
    of the pair through derp_ssim / derp_average_score."""
    from facebook360_dep_amd import derp, synth
    from oracle import oracle_lib as O

    n, res = 6, 96
    rig = synth.make_rig(n, res)
    frame = synth.make_frame(rig, [(res, res)], device="cpu")
    colors = frame["color"][0]
    disps = [d.copy() for d in frame["truth"]]
    rng = np.random.default_rng(3)
    disps[1][20:40, 30:50] *= 2.5            # a foreground slab: long stretched triangles at its border
    disps[3][60:64, 10:14] = np.nan          # a hole
    disps[4] = disps[4] * (1 + 0.02 * rng.standard_normal(disps[4].shape)).astype(np.float32)
    R = O.Rig(rig["cameras"]).normalize()
    g = derp.Derp(rig["cameras"])
    g.rephotograph_upload(colors, disps)
    E = 64
    for i in (0, 3):
        centre = rig["cameras"][i]["origin"]
        only = [int(s == i) for s in range(n)]
        others = [int(s != i) for s in range(n)]
        got_ref, got_ren = g.canopy_cubemap(only, centre, E), g.canopy_cubemap(others, centre, E)
        want_ref = O.canopy_cubemap(R, colors, disps, only, centre, E)
        want_ren = O.canopy_cubemap(R, colors, disps, others, centre, E)
        assert got_ref.shape == (6 * E, E, 4)
        assert _float_equal(got_ref, want_ref) == 0 and _float_equal(got_ren, want_ren) == 0, i
        alpha = got_ref[..., 3]
        assert set(np.unique(alpha)) <= {0.0, 1.0} and 0.3 < alpha.mean() < 0.7  # a 180-degree camera: half the sphere
        assert (got_ren[..., 3] > 0).mean() > 0.95
        mask = (alpha > 0).astype(np.uint8)
        score = g.ssim(got_ref[..., :3], got_ren[..., :3], 1)
        want = O.compute_ssim(want_ref[..., :3], want_ren[..., :3], 1)
        assert _float_equal(score, want) == 0
        avg = derp.average_score(score, mask)
        print("camera %d MSSIM from its neighbours: %s" % (i, derp.format_results(avg)))
        assert all(0.2 < v <= 1.0 for v in avg)
    g.close()


def test_lean_atan2_against_libm(gpu):
    """The FTHETA projection of the cost kernels uses its own fp64 atan2(y >= 0, x) (fdlibm's atan with the argument
    reduction on the pair, one division): at most 2 ulp from this host's libm (99.9 % within 1) over the angles a
    camera meets, all breakpoints of the reduction, both signs of x, tiny and huge ratios; exact at the axes."""
    rng = np.random.default_rng(7)
    n = 1 << 20
    ys, xs = [], []
    mag = 10.0 ** rng.uniform(-6, 6, n)
    ys.append(np.abs(rng.normal(size=n)) * mag)          # any ratio, any scale
    xs.append(rng.normal(size=n) * 10.0 ** rng.uniform(-6, 6, n))
    th = rng.uniform(0, np.pi, n)                           # uniform in angle: what a fisheye rig produces
    r = 10.0 ** rng.uniform(-2, 3, n)
    ys.append(r * np.sin(th))
    xs.append(r * np.cos(th))
    for c in (0.4375, 0.6875, 1.1875, 2.4375):              # the reduction's breakpoints, a few ulps either side
        x = rng.uniform(0.5, 2.0, n // 4) * rng.choice([-1.0, 1.0], n // 4)
        y = np.abs(x) * c * (1.0 + rng.integers(-8, 9, n // 4) * 2.0 ** -52)
        ys.append(y)
        xs.append(x)
    ys.append(np.array([0.0, 0.0, 1.0, 3.0, 1e-300, 1e300, 1.0]))
    xs.append(np.array([1.0, -1.0, 0.0, -0.0, 1.0, 1.0, 1e300]))
    y, x = np.abs(np.concatenate(ys)), np.concatenate(xs)
    got = gpu.debug_atan2_ypos(y, x)
    want = np.arctan2(y, x)
    ulp = np.abs(got - want) / np.spacing(np.maximum(want, 2.0 ** -1000))
    assert np.all(np.isfinite(got)) and float(ulp.max()) <= 2.0 and float((ulp > 1).mean()) < 1e-3, float(ulp.max())
    assert got[-7] == 0.0 and got[-6] == np.pi and got[-5] == np.pi / 2 and got[-4] == np.pi / 2
    print("lean atan2: %d arguments, %.2f %% equal to libm, %.4f %% more than 1 ulp away, max %.1f ulp"
          % (y.size, 100.0 * float((ulp == 0).mean()), 100.0 * float((ulp > 1).mean()), float(ulp.max())))


def test_camera_types(built):
    """RECTILINEAR, EQUISOLID and ORTHOGRAPHIC cameras (Camera.h:301-378) through the whole pyramid:
    warp tables (unproject + project), FOV masks and the cost loop's fp64 `sees` for every type."""
    from facebook360_dep_amd import derp, synth

    rig = _mixed_type_rig(120)
    sizes = [(120, 120), (80, 80), (50, 50)]
    frame = synth.make_frame(rig, sizes)
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, 120, 120, counters=cnt, partial_coverage=True)
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid(sizes, 120, 120)
    g.upload_frame(frame)
    g.process_pyramid()
    g.synchronize()
    nbad = npx = nvalid = 0
    for level in ref:
        for d in range(6):
            got = g.download_disparity(level, d)
            bad, rel = common.compare_disparity(got, ref[level][d], TOL)
            nbad += bad
            npx += got.size
            nvalid += int(np.isfinite(got).sum())
    print("mixed camera types: %d of %d pixels outside 1e-4 (%d finite)" % (nbad, npx, nvalid))
    assert nbad == 0
    assert nvalid > 0.3 * npx
    c = g.counters()
    assert c["n_cost"] == sum(v["n_cost"] for v in cnt.values())
    assert c["n_pair"] == sum(v["n_pair"] for v in cnt.values())
    g.close()


@pytest.mark.parametrize("opts", [
    dict(ping_pong_iterations=2),
    dict(random_proposals=0),
    dict(random_proposals=5, ping_pong_iterations=3),
    dict(do_bilateral_filter=False, do_median_filter=False),
    dict(min_depth_m=1.0, max_depth_m=50.0, var_noise_floor=1e-3, var_high_thresh=5e-2),
    dict(mismatches_start_level=2, ping_pong_iterations=2),
])
def test_option_matrix(small, opts):
    """DerpCLI flags that change the schedule (DerpCLI.cpp:44-67), each against the oracle."""
    stats = _run_pyramid(small, partial_coverage=True, **opts)
    for level, (bad, npx, worst) in stats.items():
        assert bad <= 1e-4 * npx, (opts, level, bad, npx, worst)
    common.observed("option_matrix.small." + ",".join("%s=%s" % kv for kv in sorted(opts.items())),
                    {str(level): bad for level, (bad, _, _) in stats.items()})


def test_edge_cases(built):
    """Degenerate inputs: tiny ragged levels, an all-background camera, an all-foreground-but-empty
    mask, constant (zero-variance) imagery, and a single-level pyramid."""
    from facebook360_dep_amd import derp, synth

    rig = synth.make_rig(4, 64)
    sizes = [(64, 64), (24, 18), (7, 5)]  # ragged, odd, and a level whose interior is 5 x 3 pixels
    frame = synth.make_frame(rig, [(64, 64)], with_masks=True)
    color = [[synth.resize_area(frame["color"][0][s].astype(np.float64), w, h).round().astype(np.uint16)
              for s in range(4)] for (w, h) in sizes]
    masks = [[(synth.resize_area(frame["masks"][0][s] * 255.0, w, h) > 127).astype(np.uint8) for s in range(4)]
             for (w, h) in sizes]
    bgd = [[synth.resize_area(frame["bg_disp"][0][s], w, h).astype(np.float32) for s in range(4)] for (w, h) in sizes]
    for lvl in range(3):
        masks[lvl][1][:] = 0                       # camera 1: everything is background
        color[lvl][2][:] = 31000                   # camera 2: constant image, zero variance everywhere
    fr = {"color": color, "masks": masks, "bg_disp": bgd}
    for use_fg in (False, True):
        ref = common.oracle_pyramid(rig, sizes, fr, 64, 64, partial_coverage=True, use_foreground_masks=use_fg)
        g = derp.Derp(rig["cameras"], partial_coverage=1, use_foreground_masks=int(use_fg))
        g.set_pyramid(sizes, 64, 64)
        g.upload_frame(fr if use_fg else {"color": color})
        g.process_pyramid()
        g.synchronize()
        for level in ref:
            for d in range(4):
                bad, rel = common.compare_disparity(g.download_disparity(level, d), ref[level][d], TOL)
                assert bad == 0, (use_fg, level, d, bad, rel)
        if use_fg:  # the all-background camera reproduces its background disparity inside the FOV
            got = g.download_disparity(0, 1)
            fov = g.fov_mask(1, 64, 64) == 1
            assert np.array_equal(got[fov], bgd[0][1][fov])
        g.close()
    # single-level pyramid: brute force + filters only
    ref = common.oracle_pyramid(rig, [(64, 64)], {"color": [color[0]]}, 64, 64, partial_coverage=True)
    g = derp.Derp(rig["cameras"], partial_coverage=1)
    g.set_pyramid([(64, 64)], 64, 64)
    g.upload_frame({"color": [color[0]]})
    g.process_pyramid()
    g.synchronize()
    for d in range(4):
        assert common.compare_disparity(g.download_disparity(0, d), ref[0][d], TOL)[0] == 0
    # API misuse is an error code + message, never a crash
    with pytest.raises(derp.DerpError):
        g.process_level(3)
    with pytest.raises(derp.DerpError):
        g.set_options(random_proposals=-1)
    g.close()
