"""Deterministic synthetic rig + imagery generator.

Writes / returns data in the reference's layout: rig JSON as `Camera::loadRig` reads it
(source/util/Camera.cpp:30-75,251-258), colour as 16-bit BGR per pyramid level
(`video/color_levels/level_N/<cam>/<frame>.png`, source/util/ImageTypes.h:23), level sizes
from scripts/render/config.py:46 + resize.py:71-74.

numpy only; independent of both the HIP path and the oracle.
"""
import json
import math
import os

import numpy as np

WIDTHS = [2048, 1024, 512, 256, 200, 128, 100, 80, 60, 50]  # scripts/render/config.py:46
DISTORTION = (-0.034, 4.4e-4, -1.9e-3)  # magnitude of res/test/rigs/rig.json


def level_sizes(full_w, full_h, widths=None):
    """resize.py:71-74: height = round(ratio*w) made even. Levels whose width exceeds the
    full resolution are dropped (the 512-px rig of config 1 has 8 levels)."""
    widths = WIDTHS if widths is None else widths
    ratio = full_h / full_w
    out = []
    for w in widths:
        if w > full_w:
            continue
        h = int(round(ratio * w))
        h += h % 2
        out.append((w, h))
    return out


def _frame(forward):
    f = forward / np.linalg.norm(forward)
    helper = np.array([0.0, 0.0, 1.0]) if abs(f[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
    right = np.cross(f, helper)
    right /= np.linalg.norm(right)
    up = np.cross(right, f)  # right x up = -forward... check handedness below
    up /= np.linalg.norm(up)
    # reference requires right.cross(up).dot(forward) < 0 (Camera.cpp:78)
    if np.dot(np.cross(right, up), f) >= 0:
        right = -right
    return f, up, right


def make_rig(n_cams, res, radius=0.25, layout=None, fov=math.pi / 2):
    """n_cams FTHETA cameras of res x res on a sphere of `radius` metres, looking outward."""
    layout = layout or ("arc" if n_cams <= 4 else "fibonacci")
    cams = []
    for i in range(n_cams):
        if layout == "arc":
            az = math.radians(-45.0 + 90.0 * i / max(n_cams - 1, 1))
            d = np.array([math.cos(az), math.sin(az), 0.12 * ((i % 2) * 2 - 1)])
        else:
            z = 1.0 - 2.0 * (i + 0.5) / n_cams
            r = math.sqrt(max(0.0, 1.0 - z * z))
            phi = i * math.pi * (3.0 - math.sqrt(5.0))
            d = np.array([r * math.cos(phi), r * math.sin(phi), z])
        f, up, right = _frame(d)
        cams.append(
            {
                "version": 1,
                "type": "FTHETA",
                "id": "cam%d" % i,
                "origin": (radius * f).tolist(),
                "forward": f.tolist(),
                "up": up.tolist(),
                "right": right.tolist(),
                "resolution": [res, res],
                "focal": [0.36 * res, -0.36 * res],
                "principal": [res / 2.0, res / 2.0],
                "distortion": list(DISTORTION),
                "fov": fov,
            }
        )
    return {"cameras": cams}


# ---------------------------------------------------------------- camera (generator's own)
def _distort(r, d):
    r2 = r * r
    return r * (1 + r2 * (d[0] + r2 * (d[1] + r2 * d[2])))


def _undistort(y, d, iters=12):
    x = y.copy()
    for _ in range(iters):
        x2 = x * x
        f = x * (1 + x2 * (d[0] + x2 * (d[1] + x2 * d[2]))) - y
        df = 1 + x2 * (3 * d[0] + x2 * (5 * d[1] + x2 * 7 * d[2]))
        x = x - f / df
    return x


def pixel_rays(cam, w, h):
    """Unit ray directions (rig space) for the pixel centres of a w x h image of `cam`."""
    res = cam["resolution"]
    fx, fy = cam["focal"][0] * w / res[0], cam["focal"][1] * h / res[1]
    px, py = cam["principal"][0] * w / res[0], cam["principal"][1] * h / res[1]
    xs = (np.arange(w, dtype=np.float64) + 0.5 - px) / fx
    ys = (np.arange(h, dtype=np.float64) + 0.5 - py) / fy
    sx, sy = np.meshgrid(xs, ys)
    norm = np.sqrt(sx * sx + sy * sy)
    theta = _undistort(norm, cam.get("distortion", (0, 0, 0)))
    s = np.where(norm > 0, np.sin(theta) / np.maximum(norm, 1e-300), 0.0)
    cx, cy, cz = s * sx, s * sy, -np.cos(theta)
    R = np.array([cam["right"], cam["up"], (-np.asarray(cam["forward"])).tolist()])
    d = np.stack([cx, cy, cz], -1) @ R  # R^T * unit
    return d, theta


# ---------------------------------------------------------------- scene
PLANES = [  # (unit normal n, offset c: n.x = c, centre, half-size) — three inset planes 1.5–3 m
    (np.array([1.0, 0.0, 0.0]), 1.5, np.array([1.5, 0.2, 0.1]), 0.9),
    (np.array([-0.6, 0.8, 0.0]), 2.2, np.array([-1.32, 1.76, -0.2]), 1.3),
    (np.array([0.0, -0.6, -0.8]), 3.0, np.array([0.3, -1.8, -2.4]), 1.8),
]
SPHERE_R = 4.0


def _hash3(ix, iy, iz, seed):
    h = (ix.astype(np.uint32) * np.uint32(73856093)) ^ (iy.astype(np.uint32) * np.uint32(19349663)) ^ (
        iz.astype(np.uint32) * np.uint32(83492791)
    )
    h ^= np.uint32(seed)
    h = (h ^ (h >> np.uint32(13))) * np.uint32(1274126177)
    h ^= h >> np.uint32(16)
    return (h & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 0xFFFFFF)


def value_noise(p, freq, seed):
    q = (p * freq).astype(np.float32)
    i0 = np.floor(q)
    f = q - i0
    f = f * f * (3 - 2 * f)
    i0 = i0.astype(np.int64) & 0xFFFFF
    out = np.zeros(p.shape[:-1], dtype=np.float32)
    for dz in (0, 1):
        wz = f[..., 2] if dz else 1 - f[..., 2]
        for dy in (0, 1):
            wy = f[..., 1] if dy else 1 - f[..., 1]
            for dx in (0, 1):
                wx = f[..., 0] if dx else 1 - f[..., 0]
                out += wx * wy * wz * _hash3(i0[..., 0] + dx, i0[..., 1] + dy, i0[..., 2] + dz, seed)
    return out


def texture(p, seed=360):
    """4-octave value noise hashed from world position -> 3 channels in [0.08, 0.92]."""
    chans = []
    for c in range(3):
        v = np.zeros(p.shape[:-1], dtype=np.float32)
        amp, tot = 1.0, 0.0
        for o in range(4):
            v += amp * value_noise(p, 6.0 * (2.1**o), seed * 31 + c * 7 + o)
            tot += amp
            amp *= 0.6
        chans.append(0.08 + 0.84 * v / tot)
    return np.stack(chans, -1)


def intersect(origin, d, shift):
    """Nearest hit of rays origin + t d with the scene translated by `shift`.
    Returns (t, hit point in scene coordinates, is_plane)."""
    o = origin - shift
    b = d @ o
    c = float(o @ o) - SPHERE_R**2
    t = -b + np.sqrt(np.maximum(b * b - c, 0.0))
    is_plane = np.zeros(t.shape, dtype=bool)
    for n, cc, centre, half in PLANES:
        denom = d @ n
        tp = (cc - float(o @ n)) / np.where(np.abs(denom) > 1e-9, denom, 1e-9)
        hit = o + tp[..., None] * d
        inside = (tp > 0) & (np.max(np.abs(hit - centre), axis=-1) < half) & (tp < t)
        t = np.where(inside, tp, t)
        is_plane |= inside
    return t, o + t[..., None] * d, is_plane


def render_camera(cam, w, h, frame=0, seed=360):
    """-> (bgr u16 [h,w,3], true disparity f32 [h,w], plane mask u8 [h,w], background disparity f32)."""
    d, _ = pixel_rays(cam, w, h)
    shift = np.array([0.02 * frame, 0.0, 0.0])  # scene translated 2 cm / frame
    origin = np.asarray(cam["origin"], dtype=np.float64)
    t, p, is_plane = intersect(origin, d, shift)
    rgb = texture(p.astype(np.float32), seed)
    bgr = np.clip(np.rint(rgb[..., ::-1] * 65535.0), 0, 65535).astype(np.uint16)
    # background-only (sphere) disparity for the foreground-mask path
    o = origin - shift
    b = d @ o
    tb = -b + np.sqrt(np.maximum(b * b - (float(o @ o) - SPHERE_R**2), 0.0))
    return bgr, (1.0 / t).astype(np.float32), is_plane.astype(np.uint8), (1.0 / tb).astype(np.float32)


# ---------------------------------------------------------------- pyramid (area average)
def _area_matrix(ssize, dsize):
    scale = ssize / dsize
    m = np.zeros((dsize, ssize), dtype=np.float64)
    for dx in range(dsize):
        a, b = dx * scale, min((dx + 1) * scale, ssize)
        s0, s1 = int(math.floor(a)), int(math.ceil(b))
        for sx in range(s0, s1):
            m[dx, sx] = max(0.0, min(b, sx + 1) - max(a, sx))
        m[dx] /= m[dx].sum()
    return m


def resize_area(img, dw, dh):
    """Area-average downsample (cv2.INTER_AREA semantics up to rounding), any channel count."""
    h, w = img.shape[:2]
    if (w, h) == (dw, dh):
        return img.copy()
    my, mx = _area_matrix(h, dh), _area_matrix(w, dw)
    x = img.astype(np.float64)
    if x.ndim == 2:
        return my @ x @ mx.T
    return np.einsum("yh,hwc,xw->yxc", my, x, mx, optimize=True)


def build_pyramid(bgr, sizes):
    out = []
    for (w, h) in sizes:
        r = resize_area(bgr, w, h)
        out.append(np.clip(np.rint(r), 0, 65535).astype(np.uint16))
    return out


def make_frame(rig, sizes, frame=0, seed=360, with_masks=False):
    """Render every camera of `rig` and build its pyramid.
    -> dict(color[level][cam] u16, truth[cam] f32 at level 0, masks[level][cam], bg_disp[level][cam])"""
    cams = rig["cameras"]
    w0, h0 = sizes[0]
    color = [[None] * len(cams) for _ in sizes]
    masks = [[None] * len(cams) for _ in sizes]
    bgd = [[None] * len(cams) for _ in sizes]
    truth = []
    for ci, cam in enumerate(cams):
        bgr, disp, is_plane, bg = render_camera(cam, w0, h0, frame, seed)
        truth.append(disp)
        for li, im in enumerate(build_pyramid(bgr, sizes)):
            color[li][ci] = im
        if with_masks:
            for li, (w, h) in enumerate(sizes):
                m = resize_area(is_plane.astype(np.float64) * 255.0, w, h)
                masks[li][ci] = (m > 127).astype(np.uint8)  # resize.py threshold=127, CvUtil.h:235-239
                bgd[li][ci] = resize_area(bg, w, h).astype(np.float32)
    return {"color": color, "truth": truth, "masks": masks if with_masks else None,
            "bg_disp": bgd if with_masks else None, "sizes": sizes}


def config(name):
    """BASELINE.json configs -> (n_cams, resolution, widths)."""
    return {
        "cfg1": (4, 512, WIDTHS),
        "cfg2": (16, 2048, WIDTHS),
        "cfg4": (24, 4096, [4096] + WIDTHS),
        "tiny": (4, 96, [96, 64, 48]),
        "small": (6, 160, [160, 100, 64]),
    }[name]


def write_dataset(root, rig, frames, sizes, seed=360, with_masks=False):
    """Write the reference's on-disk layout under `root` (PNG writer: facebook360_dep_amd.imageio)."""
    from . import imageio as dio

    os.makedirs(os.path.join(root, "rigs"), exist_ok=True)
    with open(os.path.join(root, "rigs", "rig_calibrated.json"), "w") as f:
        json.dump(rig, f, indent=2)
    for fi in frames:
        fr = make_frame(rig, sizes, fi, seed + fi, with_masks)
        name = "%06d" % fi
        for li in range(len(sizes)):
            for ci, cam in enumerate(rig["cameras"]):
                d = os.path.join(root, "video", "color_levels", "level_%d" % li, cam["id"])
                os.makedirs(d, exist_ok=True)
                dio.write_png16(os.path.join(d, name + ".png"), fr["color"][li][ci])
                if with_masks:
                    d = os.path.join(root, "video", "foreground_masks_levels", "level_%d" % li, cam["id"])
                    os.makedirs(d, exist_ok=True)
                    dio.write_png8(os.path.join(d, name + ".png"), fr["masks"][li][ci] * 255)
                    d = os.path.join(root, "background", "disparity_levels", "level_%d" % li, cam["id"])
                    os.makedirs(d, exist_ok=True)
                    dio.write_pfm(os.path.join(d, "000000.pfm"), fr["bg_disp"][li][ci])
