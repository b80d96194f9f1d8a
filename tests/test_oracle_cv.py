"""The oracle's restated OpenCV primitives (oracle/oracle_cv.h) against independent implementations:
torch.grid_sample bicubic (a = -0.75, same kernel as cv::remap INTER_CUBIC), scipy uniform_filter
with mirror (= BORDER_REFLECT_101), and straightforward float64 numpy evaluations. OpenCV itself is
not available in this image; these cross-checks catch restatement slips, they do not pin OpenCV."""
import numpy as np
import pytest

from oracle import oracle_lib as O

rng = np.random.default_rng(42)


def test_remap_cubic_vs_grid_sample():
    import torch
    import torch.nn.functional as F

    h, w = 40, 52
    src = rng.integers(0, 65536, size=(h, w, 3)).astype(np.uint16)
    # coordinates on the 1/32-px lattice so remap's fixed-point quantisation is exact
    mx = rng.integers(-3 * 32, (w + 2) * 32, size=(30, 35)) / 32.0
    my = rng.integers(-3 * 32, (h + 2) * 32, size=(30, 35)) / 32.0
    mp = np.stack([mx, my], -1).astype(np.float32)
    got = O.cv_remap_cubic(src, mp).astype(np.float64)
    t = torch.from_numpy(src.astype(np.float64)).permute(2, 0, 1)[None]
    gx = 2 * torch.from_numpy(mx) / (w - 1) - 1
    gy = 2 * torch.from_numpy(my) / (h - 1) - 1
    grid = torch.stack([gx, gy], -1)[None]
    ref = F.grid_sample(t, grid, mode="bicubic", padding_mode="zeros", align_corners=True)[0].permute(1, 2, 0).numpy()
    ref = np.clip(np.rint(ref), 0, 65535)
    assert np.abs(got - ref).max() <= 1.0  # float32 vs float64 accumulation: at most one rounding step
    assert (got != ref).mean() < 0.01


def test_remap_nan_and_far_outside_is_zero():
    src = rng.integers(1, 65536, size=(16, 16, 3)).astype(np.uint16)
    mp = np.array([[[np.nan, 3.0], [3.0, np.nan], [1e9, 2.0], [-50.0, 4.0], [4.0, 40.0]]], dtype=np.float32)
    assert not O.cv_remap_cubic(src, mp).any()


def test_remap_identity_on_integer_coordinates():
    src = rng.integers(0, 65536, size=(20, 24, 3)).astype(np.uint16)
    ys, xs = np.mgrid[0:20, 0:24]
    mp = np.stack([xs, ys], -1).astype(np.float32)
    assert np.array_equal(O.cv_remap_cubic(src, mp), src)


def test_blur_u16_vs_scipy():
    from scipy.ndimage import uniform_filter

    src = rng.integers(0, 65536, size=(33, 47, 3)).astype(np.uint16)
    ref = np.stack([uniform_filter(src[..., c].astype(np.float64), 3, mode="mirror") for c in range(3)], -1)
    assert np.array_equal(O.cv_blur3_u16(src), np.rint(ref).astype(np.uint16))


def test_blur_f32_and_variance_vs_float64():
    from scipy.ndimage import uniform_filter

    src = rng.integers(0, 65536, size=(31, 29, 3)).astype(np.uint16)
    f = (src.astype(np.float32) * np.float32(1.0 / 65535.0))
    blur = O.cv_blur3_f32(f)
    ref = np.stack([uniform_filter(f[..., c].astype(np.float64), 3, mode="mirror") for c in range(3)], -1)
    assert np.abs(blur - ref).max() < 1e-7
    var = O.cv_variance(src).astype(np.float64)
    m = np.stack([uniform_filter(f[..., c].astype(np.float64), 3, mode="mirror") for c in range(3)], -1)
    m2 = np.stack([uniform_filter((f[..., c].astype(np.float64)) ** 2, 3, mode="mirror") for c in range(3)], -1)
    v = m2 - m * m
    ref = v[..., 0] * 0.3333 + v[..., 1] * 0.3334 + v[..., 2] * 0.3333
    assert np.abs(var - ref).max() < 2e-7


def _lanczos_ref(src, dw, dh):
    def axis(ssize, dsize):
        scale = ssize / dsize
        W = np.zeros((dsize, ssize))
        for d in range(dsize):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            x = float(np.float32(f - s))
            if x < np.finfo(np.float32).eps:
                c = np.zeros(8)
                c[3] = 1
            else:
                y = -(x + 3 - np.arange(8)) * np.pi * 0.25
                c = np.sin(y) / (y * y) * 0  # placeholder, replaced below
                # sin(pi x) sin(pi x / 4) / x^2 kernel in the form OpenCV evaluates it
                y0 = -(x + 3) * np.pi * 0.25
                s45 = np.sqrt(0.5)
                cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
                c = (cs[:, 0] * np.sin(y0) + cs[:, 1] * np.cos(y0)) / (y * y)
                c = c / c.sum()
            for k in range(8):
                W[d, min(max(s - 3 + k, 0), ssize - 1)] += c[k]
        return W

    return axis(src.shape[0], dh) @ src.astype(np.float64) @ axis(src.shape[1], dw).T


@pytest.mark.parametrize("shape,up", [((50, 50), (60, 60)), ((60, 60), (80, 80)), ((32, 40), (64, 80)), ((30, 44), (76, 100))])
def test_lanczos_vs_float64(shape, up):
    src = rng.random(shape).astype(np.float32)
    got = O.cv_resize_lanczos4(src, up[1], up[0])
    ref = _lanczos_ref(src, up[1], up[0])
    assert np.abs(got - ref).max() < 2e-6
    const = np.full(shape, 0.37, dtype=np.float32)
    assert np.abs(O.cv_resize_lanczos4(const, up[1], up[0]) - 0.37).max() < 1e-6


def test_nearest():
    src = rng.random((30, 41)).astype(np.float32)
    got = O.cv_resize_nearest(src, 100, 64)
    ys = np.minimum(np.floor(np.arange(64) * (30 / 64)).astype(int), 29)
    xs = np.minimum(np.floor(np.arange(100) * (41 / 100)).astype(int), 40)
    assert np.array_equal(got, src[ys][:, xs])


def test_radii():
    # Derp.cpp:876-878 -> [5,5,5,4,4,3,3,3,3,2] for levels 0..9
    assert [O.bilateral_radius(l) for l in range(10)] == [5, 5, 5, 4, 4, 3, 3, 3, 3, 2]
    assert [O.temporal_space_radius(l) for l in range(10)] == [1] * 10
    assert O.upsample_radius(1024, 2048) == 5 and O.upsample_radius(50, 60) == 2


def test_minstd_against_definition():
    # minstd_rand0: x <- 16807 x mod 2^31-1; generate_canonical<float,24>: (x-1) / 2^31 (float)
    x = 21
    exp = []
    for _ in range(16):
        x = (16807 * x) % 2147483647
        u = np.float32(np.float32(x - 1) / np.float32(2147483648.0))
        if u >= 1:
            u = np.nextafter(np.float32(1), np.float32(0))
        exp.append(np.float32(u * np.float32(2.0 - 0.5) + np.float32(0.5)))
    assert np.array_equal(O.minstd_uniform(21, 16, 0.5, 2.0), np.array(exp, dtype=np.float32))
    assert np.array_equal(O.minstd_uniform(0, 4, 0, 1), O.minstd_uniform(2147483647, 4, 0, 1))  # seed 0 -> state 1


def test_pfm_and_png_io(tmp_path):
    from facebook360_dep_amd import imageio as dio

    m = rng.random((7, 5)).astype(np.float32)
    m[2, 3] = np.nan
    p = str(tmp_path / "a.pfm")
    dio.write_pfm(p, m)
    raw = open(p, "rb").read()
    assert raw.startswith(b"Pf\n5 7\n-1.0\n")  # CvUtil.cpp:39-49
    assert raw[len(b"Pf\n5 7\n-1.0\n"):] == m.tobytes()  # rows top-to-bottom, little endian, no flip
    back = dio.read_pfm(p)
    assert np.array_equal(np.isnan(back), np.isnan(m)) and np.array_equal(np.nan_to_num(back), np.nan_to_num(m))
    img = rng.integers(0, 65536, size=(9, 6, 3)).astype(np.uint16)
    q = str(tmp_path / "a.png")
    dio.write_png16(q, img)
    assert np.array_equal(dio.load_color_u16(q), img)
    mask = (rng.random((9, 6)) > 0.5).astype(np.uint8)
    dio.write_png8(q, mask * 255)
    assert np.array_equal(dio.load_mask(q), mask)
    try:
        from PIL import Image

        assert np.array_equal(np.asarray(Image.open(q)), mask * 255)  # a third-party decoder agrees
    except ImportError:
        pass


def test_resize_area_paths_vs_exact_area_average():
    """cv2.resize INTER_AREA restatement (integer 2x2, integer NxN, fractional tables) against an exact
    float64 area average: never more than one rounding step apart."""
    from facebook360_dep_amd import synth

    img = rng.integers(0, 65536, size=(240, 320, 3)).astype(np.uint16)
    for dw, dh in ((160, 120), (80, 60), (40, 30), (100, 76), (50, 38), (320, 240)):
        got = O.cv_resize_area(img, dw, dh).astype(np.float64)
        ref = synth.resize_area(img, dw, dh)
        assert np.abs(got - ref).max() <= 0.5 + 1e-2, (dw, dh)
    # 2x2 integer path rounds half up: (a+b+c+d+2)>>2
    blk = np.array([[1, 2], [2, 1]], dtype=np.uint16)  # sum 6 -> (6+2)>>2 = 2
    tile = np.tile(blk[:, :, None], (4, 4, 3))
    assert (O.cv_resize_area(tile, 4, 4) == 2).all()
    msk = (rng.random((240, 320)) > 0.5).astype(np.uint8) * 255
    assert np.abs(O.cv_resize_area(msk, 100, 76).astype(np.float64) - synth.resize_area(msk, 100, 76)).max() <= 0.51
    f = rng.random((240, 320)).astype(np.float32)
    for dw, dh in ((160, 120), (100, 76), (40, 30)):
        assert np.abs(O.cv_resize_area(f, dw, dh) - synth.resize_area(f, dw, dh)).max() < 1e-6



def test_ssim_vs_float64():
    """computeSSIM restatement (RephotographyUtil.h:38-86) against a float64 numpy evaluation of the
    same formula; averageScore's NaN / mask exclusion; formatResults' R, G, B order."""
    rng = np.random.default_rng(5)
    x = rng.random((37, 52, 3), dtype=np.float32)
    y = np.clip(x + 0.05 * rng.standard_normal(x.shape).astype(np.float32), 0, 1).astype(np.float32)

    def blur(a, r):
        k = np.exp(-np.arange(-r, r + 1, dtype=np.float64) ** 2 / (2 * 1.5 ** 2))
        k /= k.sum()
        p = np.pad(a, ((r, r), (r, r), (0, 0)), mode="reflect")
        t = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(2 * r + 1))
        return sum(k[i] * t[i:i + a.shape[0]] for i in range(2 * r + 1))

    for r in (1, 2, 4):
        assert np.abs(O.gaussian_blur_f32c3(x, r) - blur(x.astype(np.float64), r)).max() < 1e-6
        X, Y = x.astype(np.float64), y.astype(np.float64)
        mx, my = blur(X, r), blur(Y, r)
        s2x, s2y, sxy = blur((X - mx) ** 2, r), blur((Y - my) ** 2, r), blur((X - mx) * (Y - my), r)
        sx, sy = np.sqrt(s2x), np.sqrt(s2y)
        c1, c2 = 1e-4, 9e-4
        lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
        con = (2 * sx * sy + c2) / (s2x + s2y + c2)
        st = (sxy + c2 / 2) / (sx * sy + c2 / 2)
        assert np.abs(O.compute_ssim(x, y, r) - con * lum * st).max() < 5e-5
        assert np.abs(O.compute_ssim(x, y, r, 0, 0, 1) - st).max() < 5e-5
    s = O.compute_ssim(x, x, 1)
    assert s.min() > 0.99999 and s.max() <= 1.0
    score = O.compute_ssim(x, y, 1)
    score[3, 4, 1] = np.nan
    mask = np.ones(x.shape[:2], np.uint8)
    mask[:10] = 0
    avg = O.average_score(score, mask)
    sel = score[10:].reshape(-1, 3).astype(np.float64)
    assert abs(avg[0] - sel[:, 0].mean()) < 1e-12
    assert abs(avg[1] - np.nanmean(sel[:, 1])) < 1e-12
    assert O.format_results([0.5, 0.25, 0.125]) == "R 12.50%, G 25.00%, B 50.00%"


def test_resize_area_enlargement_is_the_bilinear_emulation():
    """cv::resize(INTER_AREA) on an image that is enlarged along an axis (oracle_cv.h resizeLinearAreaF32): with an
    integer zoom factor every destination pixel lies inside one source pixel and the rule degenerates to pixel
    replication (OpenCV's documentation: "when the image is zoomed, it is similar to the INTER_NEAREST method"); with
    2 -> 3 pixels the middle one is the mean of both; a constant image stays constant; the result is a convex
    combination of source texels."""
    from oracle import oracle_lib as O

    rng = np.random.default_rng(3)
    src = rng.random((7, 9, 3), dtype=np.float32)
    for k in (2, 3, 5):
        got = O.cv_resize_area(src, 9 * k, 7 * k)
        assert np.array_equal(got, np.repeat(np.repeat(src, k, axis=0), k, axis=1)), k
    # one axis enlarged by an integer factor, the other untouched
    assert np.array_equal(O.cv_resize_area(src, 18, 7), np.repeat(src, 2, axis=1))
    row = np.array([[0.25, 0.75]], dtype=np.float32)
    assert np.array_equal(O.cv_resize_area(row, 3, 1), np.array([[0.25, 0.5, 0.75]], dtype=np.float32))
    plane = rng.random((11, 13), dtype=np.float32)
    up = O.cv_resize_area(plane, 31, 17)
    assert up.shape == (17, 31) and up.min() >= plane.min() and up.max() <= plane.max()
    # corners are the source corners (first tap weight 1 at the start, clamped single tap at the end)
    assert up[0, 0] == plane[0, 0] and up[-1, -1] == plane[-1, -1]
    flat = np.full((5, 6), np.float32(0.7))
    assert np.allclose(O.cv_resize_area(flat, 17, 9), 0.7, rtol=0, atol=2e-7)
