"""The reference-named executables (DerpCLI, TemporalBilateralFilter, UpsampleDisparity) run as
processes on the reference's on-disk layout, the way scripts/render/worker.py:66-107 and
scripts/test/test_master_class.py:210-238 drive the originals: flags in, files out, exit status
as the error channel. Outputs are compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "facebook360_dep_amd", "bin")


@pytest.fixture(scope="module")
def dataset(built, tmp_path_factory):
    from facebook360_dep_amd import synth

    root = str(tmp_path_factory.mktemp("derp_in"))
    n, res, widths = synth.config("tiny")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    synth.write_dataset(root, rig, [0, 1, 2], sizes, with_masks=True)
    frames = [synth.make_frame(rig, sizes, f, 360 + f, with_masks=True) for f in (0, 1, 2)]
    return dict(root=root, rig=rig, sizes=sizes, res=res, n=n, frames=frames)


def run(binary, *flags, expect_ok=True, env=None, timeout=600):
    p = subprocess.run([os.path.join(BIN, binary)] + list(flags), capture_output=True, text=True, timeout=timeout, env=env)
    if expect_ok:
        assert p.returncode == 0, p.stderr[-3000:]
    return p


def test_derp_cli_layout_and_values(dataset, tmp_path):
    from facebook360_dep_amd import imageio as dio

    out = str(tmp_path / "out")
    p = run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--first=000000", "--last=000001",
            "--partial_coverage", "--output_formats=png,pfm", "--resolution=96", "--threads=4",
            "--log_dir=" + str(tmp_path))
    assert "-- TOTAL:" in p.stderr
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    for level in range(len(dataset["sizes"])):
        for cam in ids:
            d = os.path.join(out, "disparity_levels", "level_%d" % level, cam)
            assert sorted(os.listdir(d)) == ["000000.pfm", "000000.png", "000001.pfm", "000001.png"]
    assert sorted(os.listdir(os.path.join(out, "disparity"))) == sorted(ids)  # createLevelOutputDirs
    total_bad = 0
    for f in (0, 1):
        ref = common.oracle_pyramid(dataset["rig"], dataset["sizes"], dataset["frames"][f], dataset["res"],
                                    dataset["res"], partial_coverage=True)
        for level in ref:
            for d, cam in enumerate(ids):
                got = dio.read_pfm(os.path.join(out, "disparity_levels", "level_%d" % level, cam, "%06d.pfm" % f))
                bad, rel = common.compare_disparity(got, ref[level][d], 1e-4)
                assert bad <= 1, (f, level, cam, bad, rel)
                total_bad += bad
    common.observed("derp_cli.tiny.pixels_outside_1e-4", total_bad)
    png = dio.read_png(os.path.join(out, "disparity_levels", "level_0", ids[0], "000000.png"))
    pfm = dio.read_pfm(os.path.join(out, "disparity_levels", "level_0", ids[0], "000000.pfm"))
    exp = np.clip(np.rint(np.nan_to_num(pfm.astype(np.float32) * np.float32(65535.0), nan=0.0)), 0, 65535)
    assert png.dtype == np.uint16 and np.abs(png.astype(np.int64) - exp.astype(np.int64)).max() <= 1
    # the PFM container exactly as writeCvMat32FC1ToPFM lays it out (CvUtil.cpp:39-49; pinned by
    # tests/golden/ref_pfm.json): header literals, then the rows top first
    raw = open(os.path.join(out, "disparity_levels", "level_0", ids[0], "000000.pfm"), "rb").read()
    w0, h0 = dataset["sizes"][0]
    header = ("Pf\n%d %d\n-1.0\n" % (w0, h0)).encode()
    assert raw.startswith(header) and len(raw) == len(header) + 4 * w0 * h0
    assert np.array_equal(np.frombuffer(raw[len(header):], dtype="<f4").reshape(h0, w0), pfm, equal_nan=True)


def _converted_copy(src_root, dst_root, frames, colour, mask=None):
    """The dataset of `src_root` with every colour (and mask) PNG of `frames` re-encoded by colour(array, path_without_ext)
    / mask(...), which write the file under whatever extension they like."""
    import shutil

    from facebook360_dep_amd import imageio as dio

    shutil.copytree(src_root, dst_root)
    for kind, fn in (("color_levels", colour), ("foreground_masks_levels", mask)):
        base = os.path.join(dst_root, "video", kind)
        if fn is None or not os.path.isdir(base):
            continue
        for level in os.listdir(base):
            for cam in os.listdir(os.path.join(base, level)):
                for name in os.listdir(os.path.join(base, level, cam)):
                    path = os.path.join(base, level, cam, name)
                    a = dio.read_png(path) if name[:-4] in frames else None
                    os.remove(path)  # first: fn may write a .png of its own under the same name
                    if a is not None:
                        fn(a, path[:-4])


def test_derp_cli_reads_tiff_pnm_and_jpeg_inputs(dataset, tmp_path):
    """cv::imread takes whatever the colour / mask directories hold (CvUtil.cpp:23-29; the extension only names the
    file, ImageUtil.h:48-56). Lossless re-encodings of the inputs (16-bit tiled big-endian LZW TIFF with a predictor,
    PGM masks) must give byte-identical disparities; a JPEG data set must give what a PNG data set holding libjpeg's
    decode of the same files gives."""
    from tests.test_image_codecs import tiff_bytes

    flags = ["--first=000000", "--last=000000", "--partial_coverage", "--resolution=96", "--use_foreground_masks"]

    def level0(out):
        return {cam: open(os.path.join(out, "disparity_levels", "level_0", cam, "000000.pfm"), "rb").read()
                for cam in sorted(os.listdir(os.path.join(out, "disparity_levels", "level_0")))}

    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + str(tmp_path / "out_png"), *flags)
    want = level0(str(tmp_path / "out_png"))
    assert len(want) == dataset["n"]

    def tif(a, stem):  # imageio.read_png returns the reference's channel order (B, G, R): the file holds R, G, B
        open(stem + ".tif", "wb").write(tiff_bytes(np.ascontiguousarray(a[..., ::-1]), ">", 5, 2, (16, 16)))

    def pgm(a, stem):
        open(stem + ".pgm", "wb").write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]) + a.astype(np.uint8).tobytes())

    _converted_copy(dataset["root"], str(tmp_path / "in_tif"), ["000000"], tif, pgm)
    run("DerpCLI", "--input_root=" + str(tmp_path / "in_tif"), "--output_root=" + str(tmp_path / "out_tif"), *flags)
    assert level0(str(tmp_path / "out_tif")) == want

    Image = pytest.importorskip("PIL.Image")
    from facebook360_dep_amd import imageio as dio

    def jpg(a, stem):
        Image.fromarray((a[..., ::-1] >> 8).astype(np.uint8)).save(stem + ".jpg", quality=92, subsampling=2)

    def png_of_jpg(a, stem):
        jpg(a, stem)
        decoded = np.asarray(Image.open(stem + ".jpg"))
        os.remove(stem + ".jpg")
        dio.write_png8(stem + ".png", decoded[..., ::-1])

    _converted_copy(dataset["root"], str(tmp_path / "in_jpg"), ["000000"], jpg)
    _converted_copy(dataset["root"], str(tmp_path / "in_jpg_as_png"), ["000000"], png_of_jpg)
    run("DerpCLI", "--input_root=" + str(tmp_path / "in_jpg"), "--output_root=" + str(tmp_path / "out_jpg"), *flags)
    run("DerpCLI", "--input_root=" + str(tmp_path / "in_jpg_as_png"), "--output_root=" + str(tmp_path / "out_jpg_as_png"), *flags)
    got = level0(str(tmp_path / "out_jpg"))
    assert got == level0(str(tmp_path / "out_jpg_as_png")) and got != want


def test_derp_cli_resume_and_camera_subset(dataset, tmp_path):
    """Checkpoint / resume through the per-level PFMs (DerpCLI.cpp:153-155,276-303) and --cameras."""
    from facebook360_dep_amd import imageio as dio

    out = str(tmp_path / "out")
    common_flags = ["--input_root=" + dataset["root"], "--output_root=" + out, "--partial_coverage", "--resolution=96",
                    "--cameras=cam2,cam0", "--threads=0"]  # 0 = no I/O workers: decode and writes run inline
    run("DerpCLI", *common_flags, "--level_start=2", "--level_end=1")
    assert not os.path.exists(os.path.join(out, "disparity_levels", "level_0"))
    run("DerpCLI", *common_flags, "--level_start=0", "--level_end=0")
    ref = common.oracle_pyramid(dataset["rig"], dataset["sizes"], dataset["frames"][0], dataset["res"], dataset["res"],
                                dst_ids=["cam2", "cam0"], partial_coverage=True)
    assert sorted(os.listdir(os.path.join(out, "disparity_levels", "level_0"))) == ["cam0", "cam2"]
    for d, cam in enumerate(["cam2", "cam0"]):
        got = dio.read_pfm(os.path.join(out, "disparity_levels", "level_0", cam, "000000.pfm"))
        bad, rel = common.compare_disparity(got, ref[0][d], 1e-4)
        assert bad <= 1, (cam, bad, rel)


def test_derp_cli_foreground_masks_and_flagfile(dataset, tmp_path):
    from facebook360_dep_amd import imageio as dio

    out = str(tmp_path / "out")
    ff = tmp_path / "derp.flags"
    ff.write_text("# test flagfile\n--input_root\n--output_root\n--use_foreground_masks=1\n--partial_coverage=true\n"
                  "--do_median_filter=1\n--resolution=96\n")
    run("DerpCLI", "--flagfile=" + str(ff), "--input_root=" + dataset["root"], "--output_root", out)
    ref = common.oracle_pyramid(dataset["rig"], dataset["sizes"], dataset["frames"][0], dataset["res"], dataset["res"],
                                partial_coverage=True, use_foreground_masks=True)
    for d, cam in enumerate(c["id"] for c in dataset["rig"]["cameras"]):
        got = dio.read_pfm(os.path.join(out, "disparity_levels", "level_0", cam, "000000.pfm"))
        bad, rel = common.compare_disparity(got, ref[0][d], 1e-4)
        assert bad <= 1, (cam, bad, rel)


def test_derp_cli_errors_are_nonzero_exit(dataset, tmp_path):
    out = str(tmp_path / "out")
    p = run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--first=000000", "--last=000007",
            "--resolution=96", expect_ok=False)
    assert p.returncode != 0 and "Missing file" in p.stderr
    p = run("DerpCLI", "--output_root=" + out, expect_ok=False)
    assert p.returncode != 0 and "Check failed" in p.stderr
    p = run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--bogus_flag=1", expect_ok=False)
    assert p.returncode != 0
    # without --partial_coverage the coverage CHECK of computeBruteForceDisparity fires (Derp.cpp:339)
    p = run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--resolution=96", expect_ok=False)
    assert p.returncode != 0 and "Insufficient coverage" in p.stderr


def test_temporal_cli(dataset, tmp_path):
    from facebook360_dep_amd import imageio as dio
    from oracle import oracle_lib as O

    out = str(tmp_path / "out")
    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--first=000000", "--last=000002",
        "--partial_coverage", "--resolution=96", "--level_end=1")
    run("TemporalBilateralFilter", "--input_root=" + dataset["root"], "--output_root=" + out,
        "--rig=" + os.path.join(dataset["root"], "rigs", "rig_calibrated.json"), "--level=1", "--first=000000",
        "--last=000002")
    rs, rd, _ = common.oracle_rigs(dataset["rig"])
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    w, h = dataset["sizes"][1]
    for cur in (0, 1, 2):
        lo, hi = max(0, cur - 2), min(2, cur + 2)
        for d, cam in enumerate(ids):
            disps = [dio.read_pfm(os.path.join(out, "disparity_levels", "level_1", cam, "%06d.pfm" % f))
                     for f in range(lo, hi + 1)]
            guides = [dataset["frames"][f]["color"][1][d] for f in range(lo, hi + 1)]
            p = O.make_params(1, 3, w, h, dataset["res"], dataset["res"])
            fov = O.Level(rs, rd, list(range(dataset["n"])), p).fov_mask(d)
            ref = O.temporal_filter(guides, disps, [fov] * len(disps), cur - lo, 0.01, 1, 0.5, 1.0, 0.5)
            got = dio.read_pfm(os.path.join(out, "disparity_time_filtered_levels", "level_1", cam, "%06d.pfm" % cur))
            bad, rel = common.compare_disparity(got, ref, 1e-5)
            assert bad == 0, (cur, cam, bad, rel)


def test_temporal_cli_engine_equals_frame_by_frame(dataset, tmp_path):
    """TemporalBilateralFilter runs a level on the sequence driver (every frame of the chunk decoded and uploaded once, a
    frame filtered as soon as its window is in HBM) whenever the windows are those of a contiguous run of frames, and
    falls back to the reference's frame-by-frame loop (filterFrame, TemporalBilateralFilter.cpp:121-184) otherwise:
    the two write the same bytes — plain, a sub-range of the frames on disk (windows reach past --first / --last), with
    foreground masks, a --cameras subset, PNG output — and a gap in the frame numbering takes the fallback."""
    import shutil

    root = dataset["root"]
    rigf = os.path.join(root, "rigs", "rig_calibrated.json")
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    raw = str(tmp_path / "raw")
    run("DerpCLI", "--input_root=" + root, "--output_root=" + raw, "--first=000000", "--last=000002", "--partial_coverage",
        "--resolution=96", "--level_end=1")
    cases = [("all", ["--first=000000", "--last=000002"]),
             ("middle", ["--first=000001", "--last=000001", "--time_radius=1"]),
             ("masks", ["--first=000000", "--last=000002", "--use_foreground_masks", "--output_formats=pfm,png"]),
             ("subset", ["--first=000000", "--last=000001", "--cameras=%s,%s" % (ids[2], ids[0]), "--space_radius=2", "--sigma=0.02"])]
    for name, extra in cases:
        outs = {}
        for mode in ("engine", "legacy"):
            out = str(tmp_path / (name + "_" + mode))
            shutil.copytree(os.path.join(raw, "disparity_levels"), os.path.join(out, "disparity_levels"))
            env = dict(os.environ, DERP_TBF_LEGACY="1") if mode == "legacy" else None
            p = run("TemporalBilateralFilter", "--input_root=" + root, "--output_root=" + out, "--rig=" + rigf, "--level=1",
                    *extra, env=env)
            assert ("read once each" in p.stderr) == (mode == "engine"), p.stderr[-2000:]
            outs[mode] = os.path.join(out, "disparity_time_filtered_levels", "level_1")
        cams = sorted(os.listdir(outs["legacy"]))
        assert cams == sorted(os.listdir(outs["engine"])) and cams
        for cam in cams:
            files = sorted(os.listdir(os.path.join(outs["legacy"], cam)))
            assert files == sorted(os.listdir(os.path.join(outs["engine"], cam))) and files, (name, cam)
            for f in files:
                assert open(os.path.join(outs["legacy"], cam, f), "rb").read() == \
                    open(os.path.join(outs["engine"], cam, f), "rb").read(), (name, cam, f)
    # a chunk too large to keep resident (budget forced to nothing here) takes the frame-by-frame path as well
    big = str(tmp_path / "big")
    shutil.copytree(os.path.join(raw, "disparity_levels"), os.path.join(big, "disparity_levels"))
    p = run("TemporalBilateralFilter", "--input_root=" + root, "--output_root=" + big, "--rig=" + rigf, "--level=1",
            "--first=000000", "--last=000002", env=dict(os.environ, DERP_TBF_HBM_BUDGET_GB="0.000001"))
    assert "read once each" not in p.stderr and "frame by frame instead" in p.stderr
    ref_dir = os.path.join(str(tmp_path / "all_legacy"), "disparity_time_filtered_levels", "level_1")
    for cam in sorted(os.listdir(ref_dir)):
        for f in sorted(os.listdir(os.path.join(ref_dir, cam))):
            assert open(os.path.join(ref_dir, cam, f), "rb").read() == \
                open(os.path.join(big, "disparity_time_filtered_levels", "level_1", cam, f), "rb").read(), (cam, f)
    # a hole in the numbering (frame 1 of one input missing): windows are no longer those of one contiguous run
    gap = str(tmp_path / "gap")
    shutil.copytree(os.path.join(raw, "disparity_levels"), os.path.join(gap, "disparity_levels"))
    os.remove(os.path.join(gap, "disparity_levels", "level_1", ids[0], "000001.pfm"))
    p = run("TemporalBilateralFilter", "--input_root=" + root, "--output_root=" + gap, "--rig=" + rigf, "--level=1",
            "--first=000000", "--last=000000", "--time_radius=2", expect_ok=False)
    assert "read once each" not in p.stderr


def test_derp_sequence_cli_equals_the_three_binary_pipeline(dataset, tmp_path):
    """The depth_estimation stage of scripts/render/pipeline.py:364-408 two ways: (a) the way the reference
    orchestrates it — per level, DerpCLI on every frame, TemporalBilateralFilter on every frame, then "Transfer"
    (copy the filtered level over disparity_levels) — with this build's drop-in binaries and the file system in
    between; (b) bin/DerpSequence, which keeps the frames resident in HBM and fuses the three. Same files,
    bit for bit; and both equal the oracle's run of the schedule."""
    import shutil

    from facebook360_dep_amd import imageio as dio
    from facebook360_dep_amd import sequence

    root = dataset["root"]
    rigf = os.path.join(root, "rigs", "rig_calibrated.json")
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    n_levels = len(dataset["sizes"])
    common_flags = ["--input_root=" + root, "--first=000000", "--last=000002", "--partial_coverage", "--resolution=96"]
    # (a) three binaries + Transfer, level by level
    out_a = str(tmp_path / "a")
    for level in range(n_levels - 1, -1, -1):
        run("DerpCLI", *common_flags, "--output_root=" + out_a, "--level_start=%d" % level, "--level_end=%d" % level)
        run("TemporalBilateralFilter", "--input_root=" + root, "--output_root=" + out_a, "--rig=" + rigf,
            "--first=000000", "--last=000002", "--level=%d" % level)
        src = os.path.join(out_a, "disparity_time_filtered_levels", "level_%d" % level)
        dst = os.path.join(out_a, "disparity_levels", "level_%d" % level)
        shutil.rmtree(dst)
        shutil.copytree(src, dst)
    # (b) fused
    out_b = str(tmp_path / "b")
    p = run("DerpSequence", *common_flags, "--output_root=" + out_b)
    assert "-- TOTAL:" in p.stderr and "3 frame(s) owned, 0 halo" in p.stderr
    ref = common.OracleSequence(dataset["rig"], dataset["sizes"], dataset["res"], 0, 2, threads=-1,
                                frames={f: dataset["frames"][f] for f in range(3)})
    sequence.run_schedule(ref, list(range(n_levels - 1, -1, -1)), 0, 2, 0, 1)
    for level in range(n_levels):
        for d, cam in enumerate(ids):
            for f in range(3):
                name = os.path.join("level_%d" % level, cam, "%06d.pfm" % f)
                a = open(os.path.join(out_a, "disparity_levels", name), "rb").read()
                b = open(os.path.join(out_b, "disparity_levels", name), "rb").read()
                assert a == b, name
                assert b == open(os.path.join(out_b, "disparity_time_filtered_levels", name), "rb").read()
                got = dio.read_pfm(os.path.join(out_b, "disparity_levels", name))
                want = ref.disp[f][level][d].numpy()
                assert common.compare_disparity(got, want, 1e-4)[0] == 0, name
    # without the temporal filter DerpSequence degenerates to DerpCLI
    out_c = str(tmp_path / "c")
    run("DerpSequence", *common_flags, "--output_root=" + out_c, "--do_temporal_filter=false")
    out_d = str(tmp_path / "d")
    run("DerpCLI", *common_flags, "--output_root=" + out_d)
    name = os.path.join("disparity_levels", "level_0", ids[1], "000001.pfm")
    assert open(os.path.join(out_c, name), "rb").read() == open(os.path.join(out_d, name), "rb").read()
    assert not os.path.exists(os.path.join(out_c, "disparity_time_filtered_levels", "level_0", ids[1], "000001.pfm"))
    # out of core (--resident_frames): one frame slot in HBM, the frames stream level by level from host memory —
    # same files, byte for byte, with the filter (window of one frame: time_radius 0) and without it
    out_e, out_f, out_g = str(tmp_path / "e"), str(tmp_path / "f"), str(tmp_path / "g")
    p = run("DerpSequence", *common_flags, "--output_root=" + out_e, "--time_radius=0", "--resident_frames=1")
    assert "1 frame slot(s) in HBM (out of core)" in p.stderr
    run("DerpSequence", *common_flags, "--output_root=" + out_f, "--time_radius=0")
    run("DerpSequence", *common_flags, "--output_root=" + out_g, "--do_temporal_filter=false", "--resident_frames=1")
    for level in range(n_levels):
        for cam in ids:
            for f in range(3):
                name = os.path.join("disparity_levels", "level_%d" % level, cam, "%06d.pfm" % f)
                assert open(os.path.join(out_e, name), "rb").read() == open(os.path.join(out_f, name), "rb").read(), name
                assert open(os.path.join(out_g, name), "rb").read() == open(os.path.join(out_c, name), "rb").read(), name
    # PNG only at the finest level (pipeline.py:366-369 forces PFM above it)
    out_h = str(tmp_path / "h")
    run("DerpSequence", *common_flags, "--output_root=" + out_h, "--output_formats=pfm,png")
    assert os.path.exists(os.path.join(out_h, "disparity_levels", "level_0", ids[0], "000000.png"))
    assert not os.path.exists(os.path.join(out_h, "disparity_levels", "level_1", ids[0], "000000.png"))


def test_derp_sequence_cli_masks_subset_and_resume(dataset, tmp_path):
    """DerpSequence with foreground masks + temporal masking (pipeline.py:386), a --cameras subset, and a
    resume from level 1 on disk (--level_start): against the oracle schedule."""
    from facebook360_dep_amd import imageio as dio
    from facebook360_dep_amd import sequence

    root = dataset["root"]
    n_levels = len(dataset["sizes"])
    flags = ["--input_root=" + root, "--first=000000", "--last=000002", "--partial_coverage", "--resolution=96",
             "--use_foreground_masks", "--do_temporal_masking", "--cameras=cam2,cam0"]
    out = str(tmp_path / "o")
    run("DerpSequence", *flags, "--output_root=" + out, "--level_end=1")           # levels 2, 1
    run("DerpSequence", *flags, "--output_root=" + out, "--level_start=0")         # resumes from level 1 on disk

    class Seq(common.OracleSequence):  # destinations = the subset, sources = the whole rig
        def compute(self, level):
            for t in self.owned:
                prev = None
                if level + 1 < len(self.sizes):
                    prev = [self.disp[t][level + 1][d].numpy() for d in range(self.nd)]
                L = common.oracle_level(self.rig, self.sizes, self.frames[t], level, self.res, self.res, prev,
                                        dst_ids=["cam2", "cam0"], partial_coverage=True, threads=-1,
                                        use_foreground_masks=True)
                L.process()
                for d in range(self.nd):
                    self.disp[t][level][d] = __import__("torch").from_numpy(L.get_dst(d)[0])
                self.fov[level] = np.stack([L.fov_mask(d) for d in range(self.nd)])

    # one static background (--background_frame=000000) serves every frame, as synth.write_dataset lays it out
    frames = {f: dict(dataset["frames"][f], bg_disp=dataset["frames"][0]["bg_disp"]) for f in range(3)}
    ref = Seq(dataset["rig"], dataset["sizes"], dataset["res"], 0, 2, threads=-1, use_foreground_masks=True, frames=frames)
    ref.nd = 2
    src_of = [2, 0]

    def filt(level):  # the temporal stage on the destination subset: guides / fg masks of the matching source cameras
        import torch
        from oracle import oracle_lib as O
        outp = {}
        for t in ref.owned:
            lo, hi = sequence.temporal_window(t, 0, 2, 2)
            res = []
            for d in range(2):
                s = src_of[d]
                masks = [ref.fov[level][d] & ref.fg[u][level][s].numpy() for u in range(lo, hi + 1)]
                res.append(O.temporal_filter([ref.color[u][level][s].numpy() for u in range(lo, hi + 1)],
                                             [ref.disp[u][level][d].numpy() for u in range(lo, hi + 1)], masks, t - lo,
                                             0.01, O.temporal_space_radius(level), 0.5, 1.0, 0.5, threads=-1))
            outp[t] = torch.from_numpy(np.stack(res))
        for t in ref.owned:
            ref.disp[t][level][:2].copy_(outp[t])

    ref.filter = filt
    sequence.run_schedule(ref, list(range(n_levels - 1, -1, -1)), 0, 2, 0, 1)
    for level in range(n_levels):
        for d, cam in enumerate(["cam2", "cam0"]):
            for f in range(3):
                got = dio.read_pfm(os.path.join(out, "disparity_levels", "level_%d" % level, cam, "%06d.pfm" % f))
                want = ref.disp[f][level][d].numpy()
                assert common.compare_disparity(got, want, 1e-4)[0] == 0, (level, cam, f)
    assert sorted(os.listdir(os.path.join(out, "disparity_levels", "level_0"))) == ["cam0", "cam2"]


def test_derp_sequence_two_processes_exchange_through_files(dataset, tmp_path):
    """bin/DerpSequence --gpus 2 as two OS processes sharing the one GPU (DERP_SINGLE_DEVICE): the frames are
    partitioned, the halo frames' colour guides and raw level disparities move between the processes — through
    files (--exchange=files), and through the same files when the RCCL communicator cannot be built (RCCL refuses
    two ranks on one device: the default --exchange=rccl must notice, say so and fall back) — and every output
    file equals the single-process run's byte for byte."""
    root = dataset["root"]
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    n_levels = len(dataset["sizes"])
    flags = ["--input_root=" + root, "--first=000000", "--last=000002", "--partial_coverage", "--resolution=96"]
    one = str(tmp_path / "one")
    run("DerpSequence", *flags, "--output_root=" + one)
    env = dict(os.environ, DERP_SINGLE_DEVICE="1")
    for name, extra, expect in (("files", ["--exchange=files"], "halo exchange through files"),
                                ("fallback", [], "RCCL transport unavailable")):
        out = str(tmp_path / name)
        # what a job that crashed on this --output_root would have left behind: a rendezvous token with its go file
        # and halo files under the names this run will wait for — the start-up rendezvous must wipe them, not read them
        stale = os.path.join(out, ".derp_seq")
        os.makedirs(os.path.join(stale, "halo"))
        with open(os.path.join(stale, "token"), "w") as f:
            f.write("dead-job")
        with open(os.path.join(stale, "go.dead-job"), "w") as f:
            f.write("whatever\n")
        for level in range(n_levels):
            for kind in (0, 2):
                for frame, to in ((1, 1), (2, 0)):
                    with open(os.path.join(stale, "halo", "L%d_k%d_f%06d_to%d.bin" % (level, kind, frame, to)), "wb") as f:
                        f.write(b"\0" * (1 << 16))
        p = run("DerpSequence", *flags, "--output_root=" + out, "--gpus=2", *extra, env=env, timeout=150)
        assert expect in p.stderr and "2 frame(s) owned, 1 halo frame(s)" in p.stderr and \
            "1 frame(s) owned, 2 halo frame(s)" in p.stderr, p.stderr[-3000:]
        for kind in ("disparity_levels", "disparity_time_filtered_levels"):
            for level in range(n_levels):
                for cam in ids:
                    for f in range(3):
                        rel = os.path.join(kind, "level_%d" % level, cam, "%06d.pfm" % f)
                        assert open(os.path.join(out, rel), "rb").read() == open(os.path.join(one, rel), "rb").read(), (name, rel)
        assert not [d for d in os.listdir(out) if d.startswith(".halo") or d.startswith(".derp_seq")]


def test_derp_sequence_failing_rank_takes_the_job_down(dataset, tmp_path):
    """--gpus 2 on a one-GPU box: rank 1 has no device. The parent must report the failure and stop rank 0
    (which would otherwise wait in the RCCL rendezvous for ever) — non-zero exit, promptly."""
    import time

    t0 = time.time()
    p = run("DerpSequence", "--input_root=" + dataset["root"], "--output_root=" + str(tmp_path / "o"), "--first=000000",
            "--last=000002", "--partial_coverage", "--resolution=96", "--gpus=2", expect_ok=False)
    assert p.returncode != 0 and "ranks failed" in p.stderr
    assert time.time() - t0 < 120


def test_output_format_exr(dataset, tmp_path):
    """--output_formats=exr (PyramidLevel.h:515-516): a scan-line OpenEXR with one FLOAT channel next to the PFM that
    is always written — same floats, bit for bit (NaN outside the FOV included); an unknown format is refused."""
    from facebook360_dep_amd import imageio as dio

    out = str(tmp_path / "o")
    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--first=000000", "--last=000000",
        "--partial_coverage", "--resolution=96", "--output_formats=exr")
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    for level in range(len(dataset["sizes"])):
        base = os.path.join(out, "disparity_levels", "level_%d" % level, ids[1], "000000")
        a, b = dio.read_pfm(base + ".pfm"), dio.read_exr(base + ".exr")
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert open(base + ".exr", "rb").read(8) == b"\x76\x2f\x31\x01\x02\x00\x00\x00"
        assert not os.path.exists(base + ".png")
    p = run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + str(tmp_path / "x"), "--first=000000",
            "--last=000000", "--partial_coverage", "--resolution=96", "--output_formats=tiff", expect_ok=False)
    assert p.returncode != 0 and "Invalid output format" in p.stderr
    # downstream binaries find the .exr first (it sorts before the .pfm of the same frame, ImageUtil.h:48-56's
    # first-extension lookup) and must read it, as cv::imread does: same filtered level as from a pfm-only directory
    plain = str(tmp_path / "p")
    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + plain, "--first=000000", "--last=000000",
        "--partial_coverage", "--resolution=96")
    rigf = os.path.join(dataset["root"], "rigs", "rig_calibrated.json")
    for root_out in (out, plain):
        run("TemporalBilateralFilter", "--input_root=" + dataset["root"], "--output_root=" + root_out, "--rig=" + rigf,
            "--first=000000", "--last=000000", "--level=0")
    name = os.path.join("disparity_time_filtered_levels", "level_0", ids[1], "000000.pfm")
    assert open(os.path.join(out, name), "rb").read() == open(os.path.join(plain, name), "rb").read()


def test_upsample_cli(dataset, tmp_path):
    """BASELINE config 5's last step: UpsampleDisparity with colour guide, with and without masks."""
    from facebook360_dep_amd import imageio as dio
    from oracle import oracle_lib as O

    root = dataset["root"]
    out = str(tmp_path / "out")
    run("DerpCLI", "--input_root=" + root, "--output_root=" + out, "--partial_coverage", "--resolution=96",
        "--level_end=1")
    rigf = os.path.join(root, "rigs", "rig_calibrated.json")
    lvl1 = os.path.join(out, "disparity_levels", "level_1")
    color0 = os.path.join(root, "video", "color_levels", "level_0")
    w_up, h_up = dataset["sizes"][0]
    rs, rd, _ = common.oracle_rigs(dataset["rig"])
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    # (a) no masks: Lanczos + joint bilateral with the colour guide
    up_a = str(tmp_path / "up_a")
    run("UpsampleDisparity", "--rig=" + rigf, "--disparity=" + lvl1, "--output=" + up_a, "--resolution=%d" % w_up,
        "--color=" + color0)
    # (b) both masks + background
    up_b = str(tmp_path / "up_b")
    run("UpsampleDisparity", "--rig=" + rigf, "--disparity=" + lvl1, "--output=" + up_b, "--resolution=%d" % w_up,
        "--color=" + color0, "--output_formats=pfm,png",
        "--foreground_masks_in=" + os.path.join(root, "video", "foreground_masks_levels", "level_1"),
        "--foreground_masks_out=" + os.path.join(root, "video", "foreground_masks_levels", "level_0"),
        "--background_disp=" + os.path.join(root, "background", "disparity_levels", "level_0"))
    fr = dataset["frames"][0]
    for d, cam in enumerate(ids):
        disp = dio.read_pfm(os.path.join(lvl1, cam, "000000.pfm"))
        guide = fr["color"][0][d].astype(np.float32) * np.float32(1.0 / 65535.0)
        radius = O.upsample_radius(disp.shape[1], w_up)
        ref = O.upsample_disparity(rd, d, disp, w_up, h_up)
        ref = O.joint_bilateral_f32(ref, guide, np.ones((h_up, w_up), np.uint8), radius, 0.05, 0.5, 0.5, 1.0)
        got = dio.read_pfm(os.path.join(up_a, cam, "000000.pfm"))
        assert common.compare_disparity(got, ref, 1e-5)[0] == 0, cam
        ref = O.upsample_disparity(rd, d, disp, w_up, h_up, fr["bg_disp"][0][d], fr["masks"][1][d], fr["masks"][0][d])
        ref = O.joint_bilateral_f32(ref, guide, fr["masks"][0][d], radius, 0.05, 0.5, 0.5, 1.0)
        got = dio.read_pfm(os.path.join(up_b, cam, "000000.pfm"))
        assert common.compare_disparity(got, ref, 1e-5)[0] == 0, cam
        assert os.path.exists(os.path.join(up_b, cam, "000000.png"))
    # (c), (d) the colour guide arrives LARGER than the output, as the pipeline passes it (pipeline.py:409-443:
    # "the smallest colour level larger than our last level"), and is resized with INTER_AREA on Vec3f
    # (UpsampleDisparity.cpp:117, CvUtil.h:139-147): fractional scale 96 -> 64 and integer scale 96 -> 48
    lvl2 = os.path.join(out, "disparity_levels", "level_2")
    for (w_out, h_out) in (dataset["sizes"][1], dataset["sizes"][2]):
        up_c = str(tmp_path / ("up_c%d" % w_out))
        run("UpsampleDisparity", "--rig=" + rigf, "--disparity=" + lvl2, "--output=" + up_c,
            "--resolution=%d" % w_out, "--color=" + color0)
        for d, cam in enumerate(ids):
            disp = dio.read_pfm(os.path.join(lvl2, cam, "000000.pfm"))
            guide = O.cv_resize_area(fr["color"][0][d].astype(np.float32) * np.float32(1.0 / 65535.0), w_out, h_out)
            assert guide.shape == (h_out, w_out, 3)
            radius = O.upsample_radius(disp.shape[1], w_out)
            if disp.shape[1] == w_out:
                ref = np.where(np.isnan(disp), np.float32(1e-4), disp)  # cv::resize to the same size
            else:
                ref = O.upsample_disparity(rd, d, disp, w_out, h_out)
            ref = O.joint_bilateral_f32(ref, guide, np.ones((h_out, w_out), np.uint8), radius, 0.05, 0.5, 0.5, 1.0)
            got = dio.read_pfm(os.path.join(up_c, cam, "000000.pfm"))
            assert common.compare_disparity(got, ref, 1e-5)[0] == 0, (w_out, cam)
    # (e) a guide SMALLER than the output is enlarged the way cv::resize(INTER_AREA) enlarges (its bilinear emulation,
    # CvUtil.h:139-147): level-1 colour (64 px) for the 96-px output
    up_e = str(tmp_path / "up_e")
    run("UpsampleDisparity", "--rig=" + rigf, "--disparity=" + lvl2, "--output=" + up_e, "--resolution=%d" % w_up,
        "--color=" + os.path.join(root, "video", "color_levels", "level_1"))
    for d, cam in enumerate(ids):
        disp = dio.read_pfm(os.path.join(lvl2, cam, "000000.pfm"))
        guide = O.cv_resize_area(fr["color"][1][d].astype(np.float32) * np.float32(1.0 / 65535.0), w_up, h_up)
        assert guide.shape == (h_up, w_up, 3)
        radius = O.upsample_radius(disp.shape[1], w_up)
        ref = O.upsample_disparity(rd, d, disp, w_up, h_up)
        ref = O.joint_bilateral_f32(ref, guide, np.ones((h_up, w_up), np.uint8), radius, 0.05, 0.5, 0.5, 1.0)
        got = dio.read_pfm(os.path.join(up_e, cam, "000000.pfm"))
        assert common.compare_disparity(got, ref, 1e-5)[0] == 0, cam


def test_layer_disparities_cli(dataset, tmp_path):
    from facebook360_dep_amd import imageio as dio
    from oracle import oracle_lib as O

    out = str(tmp_path / "out")
    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--partial_coverage", "--resolution=96",
        "--use_foreground_masks", "--level_end=0", "--save_debug_images")
    fgdir = os.path.join(out, "disparity_levels", "level_0")
    bgdir = os.path.join(dataset["root"], "background", "disparity_levels", "level_0")
    layered = str(tmp_path / "layered")
    run("LayerDisparities", "--rig=" + os.path.join(dataset["root"], "rigs", "rig_calibrated.json"),
        "--background_disp=" + bgdir, "--foreground_disp=" + fgdir, "--output=" + layered)
    for cam in (c["id"] for c in dataset["rig"]["cameras"]):
        fg = dio.read_pfm(os.path.join(fgdir, cam, "000000.pfm"))
        bg = dio.read_pfm(os.path.join(bgdir, cam, "000000.pfm"))
        got = dio.read_png(os.path.join(layered, "disparity", cam, "000000.png"))
        assert got.dtype == np.uint8 and np.array_equal(got, O.layer_disparities(fg, bg))
        for t in ("cost", "confidence", "mismatches"):
            assert os.path.exists(os.path.join(out, t, "level_0", cam, "000000.png"))


def test_resize_module_builds_level_directories(dataset, tmp_path):
    """python -m facebook360_dep_amd.resize — the GPU twin of scripts/render/resize.py: full-size frames
    in <src>/<cam>/, level_<L>/<cam>/<frame>.png out, identical to the oracle's cv2.INTER_AREA restatement."""
    import sys

    from facebook360_dep_amd import imageio as dio
    from oracle import oracle_lib as O

    src = tmp_path / "color"
    rig = {"cameras": [dict(c, resolution=[512, 512]) for c in dataset["rig"]["cameras"][:2]]}
    rng = np.random.default_rng(12)
    imgs = {}
    for cam in rig["cameras"]:
        os.makedirs(src / cam["id"])
        imgs[cam["id"]] = rng.integers(0, 65536, size=(512, 512, 3)).astype(np.uint16)
        dio.write_png16(str(src / cam["id"] / "000000.png"), imgs[cam["id"]])
    rigf = tmp_path / "rig.json"
    rigf.write_text(__import__("json").dumps(rig))
    dst = tmp_path / "levels"
    p = subprocess.run([sys.executable, "-m", "facebook360_dep_amd.resize", "--src_dir", str(src), "--dst_dir", str(dst),
                        "--rig", str(rigf)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    widths = [2048, 1024, 512, 256, 200, 128, 100, 80, 60, 50]
    assert sorted(os.listdir(dst)) == sorted("level_%d" % i for i in range(10))
    for level in (2, 3, 4, 9):  # levels 0-1 would upsample a 512-px frame; resize.py is only run on larger inputs
        for cam in imgs:
            got = dio.read_png(str(dst / ("level_%d" % level) / cam / "000000.png"))
            assert np.array_equal(got, O.cv_resize_area(imgs[cam], widths[level], widths[level])), (level, cam)
    # sources in other containers (resize.py reads them with cv2.imread): 16-bit TIFF stays TIFF, level for level the
    # same samples as from the PNG; 8-bit JPEG comes out as JPEG: libjpeg's encoding of the resized libjpeg decode
    from tests.test_image_codecs import tiff_bytes

    cam = rig["cameras"][0]["id"]
    src2, dst2 = tmp_path / "color_tif", tmp_path / "levels_tif"
    for c in rig["cameras"]:
        os.makedirs(src2 / c["id"])
        open(str(src2 / c["id"] / "000000.tif"), "wb").write(tiff_bytes(np.ascontiguousarray(imgs[c["id"]][..., ::-1]), ">", 5, 2))
    p = subprocess.run([sys.executable, "-m", "facebook360_dep_amd.resize", "--src_dir", str(src2), "--dst_dir", str(dst2),
                        "--rig", str(rigf)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    for level in (2, 4, 9):
        got = dio.read_image(str(dst2 / ("level_%d" % level) / cam / "000000.tif"))
        assert np.array_equal(got, dio.read_png(str(dst / ("level_%d" % level) / cam / "000000.png"))), level
    Image = pytest.importorskip("PIL.Image")
    src3, dst3 = tmp_path / "color_jpg", tmp_path / "levels_jpg"
    smooth = {}
    for c in rig["cameras"]:
        os.makedirs(src3 / c["id"])
        y, x = np.mgrid[0:512, 0:512]
        rgb = np.stack([(np.sin(x / 40.0 + k) * 0.5 + 0.5) * 200 + np.cos(y / 30.0) * 40 for k in range(3)], -1).clip(0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(str(src3 / c["id"] / "000000.jpg"), quality=90)
        smooth[c["id"]] = np.asarray(Image.open(str(src3 / c["id"] / "000000.jpg")))[..., ::-1]
    p = subprocess.run([sys.executable, "-m", "facebook360_dep_amd.resize", "--src_dir", str(src3), "--dst_dir", str(dst3),
                        "--rig", str(rigf)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    for level in (2, 4, 9):  # JPEG in, JPEG out (cv2.imwrite's defaults: quality 95, 4:2:0), byte for byte libjpeg's file
        want = O.cv_resize_area(smooth[cam].astype(np.uint16), widths[level], widths[level]).astype(np.uint8)
        Image.fromarray(np.ascontiguousarray(want[..., ::-1])).save(str(tmp_path / "want.jpg"), quality=95)
        got = open(str(dst3 / ("level_%d" % level) / cam / "000000.jpg"), "rb").read()
        assert got == open(str(tmp_path / "want.jpg"), "rb").read(), level


def test_generate_foreground_masks_cli(dataset, tmp_path):
    """GenerateForegroundMasks on full-size frames wider than --width: INTER_AREA downscale, then masks;
    output PNG 0/255 bit-identical to the oracle chain."""
    from facebook360_dep_amd import imageio as dio, synth
    from oracle import oracle_lib as O

    rig = {"cameras": dataset["rig"]["cameras"][:2]}
    color, bgc = tmp_path / "color", tmp_path / "bgcolor"
    data = {}
    rng = np.random.default_rng(17)
    for cam in rig["cameras"]:
        bg = synth.render_camera(cam, 192, 192, frame=0)[0]
        fr = bg.copy()
        fr[60:120, 40:150] = rng.integers(0, 65536, size=(60, 110, 3))
        os.makedirs(color / cam["id"])
        os.makedirs(bgc / cam["id"])
        dio.write_png16(str(bgc / cam["id"] / "000000.png"), bg)
        dio.write_png16(str(color / cam["id"] / "000003.png"), fr)
        data[cam["id"]] = (bg, fr)
    rigf = tmp_path / "rig.json"
    rigf.write_text(__import__("json").dumps(rig))
    out = tmp_path / "masks"
    run("GenerateForegroundMasks", "--first=000003", "--last=000003", "--rig=" + str(rigf), "--color=" + str(color),
        "--background_color=" + str(bgc), "--foreground_masks=" + str(out), "--width=96")
    for cam, (bg, fr) in data.items():
        ref = O.generate_foreground_mask(O.cv_resize_area(bg, 96, 96), O.cv_resize_area(fr, 96, 96), 1, 0.04, 4)
        got = dio.read_png(str(out / cam / "000003.png"))
        assert got.dtype == np.uint8 and np.array_equal(got, ref * 255)
        assert ref.sum() > 500


def test_compute_rephotography_errors_cli(dataset, tmp_path):
    """The reference's quality gate, driven the way scripts/test/test_derp_cli.py:64-100 drives it:
    DerpCLI, then ComputeRephotographyErrors on one level, then the R / G / B percentages parsed from
    the last line of <log_dir>/ComputeRephotographyErrors.INFO. The numbers must equal the oracle's
    computeSSIM / averageScore on the oracle's CanopyScene cubemaps (camera alone vs all the others, centred
    on the camera; ComputeRephotographyErrors.cpp:137-160) of the same files."""
    from facebook360_dep_amd import imageio as dio
    from oracle import oracle_lib as O

    out = str(tmp_path / "out")
    logs = str(tmp_path / "logs")
    run("DerpCLI", "--input_root=" + dataset["root"], "--output_root=" + out, "--first=000000", "--last=000000",
        "--partial_coverage", "--resolution=96")
    color = os.path.join(dataset["root"], "video", "color_levels", "level_0")
    disp = os.path.join(out, "disparity_levels", "level_0")
    run("ComputeRephotographyErrors", "--first=000000", "--last=000000", "--output=" + out,
        "--rig=" + os.path.join(dataset["root"], "rigs", "rig_calibrated.json"), "--color=" + color,
        "--disparity=" + disp, "--log_dir=" + logs)
    last = open(os.path.join(logs, "ComputeRephotographyErrors.INFO")).readlines()[-1]
    assert "TOTAL average MSSIM: R " in last
    parts = last.split("%")
    got = [float(parts[i].split(" ")[-1]) for i in range(3)]  # parse_rephoto_errors: R, G, B
    ids = [c["id"] for c in dataset["rig"]["cameras"]]
    R = O.Rig(dataset["rig"]["cameras"]).normalize()
    cols = dataset["frames"][0]["color"][0]
    disps = [dio.read_pfm(os.path.join(disp, cam, "000000.pfm")) for cam in ids]
    total = np.zeros(3)
    edge = disps[0].shape[0]  # cubeHeight = colors[0].rows
    for t, cam in enumerate(ids):
        assert os.path.exists(os.path.join(out, "rephoto", cam, "000000.png"))
        centre = dataset["rig"]["cameras"][t]["origin"]
        ref = O.canopy_cubemap(R, cols, disps, [int(s == t) for s in range(len(ids))], centre, edge)
        ren = O.canopy_cubemap(R, cols, disps, [int(s != t) for s in range(len(ids))], centre, edge)
        mask = (ref[..., 3] > 0).astype(np.uint8)
        total += np.array(O.average_score(O.compute_ssim(ref[..., :3], ren[..., :3], 1), mask))
    total /= len(ids)
    exp = [float("%.2f" % (100 * total[2])), float("%.2f" % (100 * total[1])), float("%.2f" % (100 * total[0]))]
    assert got == exp, (got, exp)
    assert all(5.0 < v <= 100.0 for v in got)
    # NCC and a camera subset; bad method -> non-zero exit
    p = run("ComputeRephotographyErrors", "--first=000000", "--last=000000", "--output=" + out,
            "--rig=" + os.path.join(dataset["root"], "rigs", "rig_calibrated.json"), "--color=" + color,
            "--disparity=" + disp, "--method=NCC", "--cameras=cam1", "--stat_radius=2")
    assert "TOTAL average NCC: R " in p.stderr
    p = run("ComputeRephotographyErrors", "--first=000000", "--last=000000", "--output=" + out,
            "--rig=" + os.path.join(dataset["root"], "rigs", "rig_calibrated.json"), "--color=" + color,
            "--disparity=" + disp, "--method=PSNR", expect_ok=False)
    assert p.returncode != 0 and "Invalid method" in p.stderr
