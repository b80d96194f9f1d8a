"""Properties of the rephotography renderer's restatement (oracle/oracle_canopy.h, CanopyScene::cubemap) that hold
whatever OpenGL leaves to the driver — on the CPU, no GPU: a scene of one flat colour renders that colour wherever
anything is drawn (mip chain, trilinear / anisotropic taps, the exp-weighted accumulation and the un-premultiply all
preserve a constant), alpha is 1 where drawn and 0 elsewhere, the six faces of a closed rig are mostly covered, the
render is deterministic, and a camera included twice weighs like one camera drawn twice (same colour)."""
import numpy as np

from facebook360_dep_amd import synth
from oracle import oracle_lib as O


def _scene(value):
    n, res, widths = synth.config("tiny")
    rig = synth.make_rig(n, res)
    R = O.Rig(rig["cameras"]).normalize()
    cols = [np.full((res, res, 3), value, dtype=np.uint16) for _ in range(n)]
    # smooth disparities with a step (a depth discontinuity stretches triangles) and a NaN hole
    yy, xx = np.mgrid[0:res, 0:res].astype(np.float32)
    disp = (0.25 + 0.1 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.float32)
    disp[:, res // 2:] += 0.3
    disp[10:14, 20:24] = np.nan
    return rig, R, cols, [disp.copy() for _ in range(n)], n


def test_flat_colour_scene_renders_that_colour():
    rig, R, cols, disps, n = _scene(32768)
    centre = rig["cameras"][1]["origin"]
    out = O.canopy_cubemap(R, cols, disps, [1] * n, centre, 32)
    assert out.shape == (6 * 32, 32, 4)
    drawn = out[..., 3] > 0
    assert drawn.mean() > 0.5  # a closed 4-camera ring covers most directions
    want = np.float32(32768) / np.float32(65535)
    assert np.all(np.abs(out[drawn][:, :3] - want) < 2e-6)
    assert np.all(out[drawn][:, 3] == 1.0) and np.all(out[~drawn] == 0.0)
    again = O.canopy_cubemap(R, cols, disps, [1] * n, centre, 32)
    assert np.array_equal(out, again)


def test_subset_of_cameras_covers_less_and_agrees_where_both_draw():
    rig, R, cols, disps, n = _scene(20000)
    centre = rig["cameras"][0]["origin"]
    every = O.canopy_cubemap(R, cols, disps, [1] * n, centre, 24)
    one = O.canopy_cubemap(R, cols, disps, [1] + [0] * (n - 1), centre, 24)
    a, b = every[..., 3] > 0, one[..., 3] > 0
    assert b.sum() > 0 and np.all(a[b]) and a.sum() > b.sum()  # what one camera draws, all of them draw too
    assert np.all(np.abs(every[b][:, :3] - one[b][:, :3]) < 2e-6)
