"""Deterministic synthetic rig + imagery generator.

Writes / returns data in the reference's layout: rig JSON as `Camera::loadRig` reads it
(source/util/Camera.cpp:30-75,251-258), colour as 16-bit BGR per pyramid level
(`video/color_levels/level_N/<cam>/<frame>.png`, source/util/ImageTypes.h:23), level sizes
from scripts/render/config.py:46 + resize.py:71-74.

numpy + torch (rendering only); independent of both the HIP path and the oracle.
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
# Rendering runs on torch tensors (CPU threads here, the GPU when one is present): plumbing only —
# the generated images are *inputs* to both the HIP path and the oracle.
def _device(device=None):
    import torch

    if device is not None:
        return torch.device(device)
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _undistort(y, d, iters=12):
    x = y.clone()
    for _ in range(iters):
        x2 = x * x
        f = x * (1 + x2 * (d[0] + x2 * (d[1] + x2 * d[2]))) - y
        df = 1 + x2 * (3 * d[0] + x2 * (5 * d[1] + x2 * 7 * d[2]))
        x = x - f / df
    return x


def pixel_rays(cam, w, h, device=None):
    """Unit ray directions (rig space) for the pixel centres of a w x h image of `cam`."""
    import torch

    dev = _device(device)
    res = cam["resolution"]
    fx, fy = cam["focal"][0] * w / res[0], cam["focal"][1] * h / res[1]
    px, py = cam["principal"][0] * w / res[0], cam["principal"][1] * h / res[1]
    xs = (torch.arange(w, dtype=torch.float64, device=dev) + 0.5 - px) / fx
    ys = (torch.arange(h, dtype=torch.float64, device=dev) + 0.5 - py) / fy
    sy, sx = torch.meshgrid(ys, xs, indexing="ij")
    norm = torch.sqrt(sx * sx + sy * sy)
    r = _undistort(norm, cam.get("distortion", (0, 0, 0)))
    kind = cam.get("type", "FTHETA")  # sensor radius -> angle from the axis (Camera.h:344-378)
    if kind == "FTHETA":
        theta = r
    elif kind == "RECTILINEAR":
        theta = torch.atan(r)
    elif kind == "EQUISOLID":
        theta = 2 * torch.asin((r / 2).clamp(max=1.0))
    elif kind == "ORTHOGRAPHIC":
        theta = torch.asin(r.clamp(max=1.0))
    else:
        raise ValueError("unknown camera type %r" % kind)
    s = torch.where(norm > 0, torch.sin(theta) / norm.clamp_min(1e-300), torch.zeros_like(norm))
    unit = torch.stack([s * sx, s * sy, -torch.cos(theta)], -1)
    R = torch.tensor([cam["right"], cam["up"], [-v for v in cam["forward"]]], dtype=torch.float64, device=dev)
    # R^T * unit, written element-wise (BLAS gemv/gemm rejects the 16.8 M-row case of config 4)
    return unit[..., 0:1] * R[0] + unit[..., 1:2] * R[1] + unit[..., 2:3] * R[2]


# ---------------------------------------------------------------- scene
PLANES = [  # (unit normal n, offset c: n.x = c, centre, half-size) — three inset planes 1.5–3 m
    ((1.0, 0.0, 0.0), 1.5, (1.5, 0.2, 0.1), 0.9),
    ((-0.6, 0.8, 0.0), 2.2, (-1.32, 1.76, -0.2), 1.3),
    ((0.0, -0.6, -0.8), 3.0, (0.3, -1.8, -2.4), 1.8),
]
SPHERE_R = 4.0


def _hash3(ix, iy, iz, seed):
    m = 0xFFFFFFFF
    h = ((ix * 73856093) & m) ^ ((iy * 19349663) & m) ^ ((iz * 83492791) & m) ^ (seed & m)
    h = ((h ^ (h >> 13)) * 1274126177) & m
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).to(dtype=__import__("torch").float32) * (1.0 / 0xFFFFFF)


def value_noise(p, freq, seed):
    import torch

    q = (p * freq).to(torch.float32)
    i0 = torch.floor(q)
    f = q - i0
    f = f * f * (3 - 2 * f)
    i0 = i0.to(torch.int64) & 0xFFFFF
    out = torch.zeros(p.shape[:-1], dtype=torch.float32, device=p.device)
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
    import torch

    octs = [value_noise(p, 6.0 * (2.1**o), seed * 31 + o) for o in range(5)]
    chans = []
    for c in range(3):
        v = torch.zeros_like(octs[0])
        amp, tot = 1.0, 0.0
        for o in range(4):
            # each channel mixes a different rotation of the octave stack, so channels decorrelate
            v += amp * octs[(o + c) % 5] if o else amp * octs[c]
            tot += amp
            amp *= 0.6
        chans.append(0.08 + 0.84 * v / tot)
    return torch.stack(chans, -1)


def intersect(origin, d, shift):
    """Nearest hit of rays origin + t d with the scene translated by `shift`.
    Returns (t, hit point in scene coordinates, is_plane, t of the background sphere alone)."""
    import torch

    # Scalars in Python floats and 3-term dot products written out element-wise: BLAS dot products and vectorised
    # reductions round differently from one host CPU to the next, and the frames must not depend on the host.
    ol = [float(origin[i]) - float(shift[i]) for i in range(3)]
    o = torch.tensor(ol, dtype=torch.float64, device=d.device)

    def dot3(v):
        return d[..., 0] * v[0] + d[..., 1] * v[1] + d[..., 2] * v[2]

    b = dot3(ol)
    c = (ol[0] * ol[0] + ol[1] * ol[1] + ol[2] * ol[2]) - SPHERE_R**2
    tb = -b + torch.sqrt((b * b - c).clamp_min(0.0))
    t = tb.clone()
    is_plane = torch.zeros(t.shape, dtype=torch.bool, device=d.device)
    for n, cc, centre, half in PLANES:
        on = ol[0] * n[0] + ol[1] * n[1] + ol[2] * n[2]
        centre = torch.tensor(centre, dtype=torch.float64, device=d.device)
        denom = dot3(n)
        denom = torch.where(denom.abs() > 1e-9, denom, torch.full_like(denom, 1e-9))
        tp = (cc - on) / denom
        hit = o + tp[..., None] * d
        inside = (tp > 0) & ((hit - centre).abs().amax(dim=-1) < half) & (tp < t)
        t = torch.where(inside, tp, t)
        is_plane |= inside
    return t, o + t[..., None] * d, is_plane, tb


def render_camera(cam, w, h, frame=0, seed=360, device=None, as_numpy=True):
    """-> (bgr u16 [h,w,3], true disparity f32 [h,w], plane mask u8 [h,w], background disparity f32)."""
    import torch

    d = pixel_rays(cam, w, h, device)
    shift = (0.02 * frame, 0.0, 0.0)  # scene translated 2 cm / frame
    t, p, is_plane, tb = intersect(cam["origin"], d, shift)
    rgb = texture(p.to(torch.float32), seed)
    bgr = torch.clamp(torch.round(rgb.flip(-1) * 65535.0), 0, 65535).to(torch.int32)
    out = (bgr, (1.0 / t).to(torch.float32), is_plane.to(torch.uint8), (1.0 / tb).to(torch.float32))
    if as_numpy:
        return (out[0].cpu().numpy().astype(np.uint16), out[1].cpu().numpy(), out[2].cpu().numpy(),
                out[3].cpu().numpy())
    return out


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
    """Area-average downsample (cv2.INTER_AREA semantics up to rounding); numpy or torch in,
    same kind out; any channel count."""
    import torch

    is_np = isinstance(img, np.ndarray)
    x = torch.from_numpy(np.ascontiguousarray(img)) if is_np else img
    h, w = x.shape[:2]
    if (w, h) == (dw, dh):
        return img.copy() if is_np else img.clone()
    my = torch.from_numpy(_area_matrix(h, dh)).to(x.device)
    mx = torch.from_numpy(_area_matrix(w, dw)).to(x.device)
    x = x.to(torch.float64)
    if x.ndim == 2:
        r = my @ x @ mx.T
    else:
        r = torch.einsum("yh,hwc,xw->yxc", my, x, mx)
    return r.cpu().numpy() if is_np else r


def build_pyramid(bgr, sizes):
    import torch

    is_np = isinstance(bgr, np.ndarray)
    x = torch.from_numpy(bgr.astype(np.int32)) if is_np else bgr
    out = []
    for (w, h) in sizes:
        r = torch.clamp(torch.round(resize_area(x, w, h).to(torch.float64)), 0, 65535)
        out.append(r.cpu().numpy().astype(np.uint16))
    return out


def make_frame(rig, sizes, frame=0, seed=360, with_masks=False, device=None):
    """Render every camera of `rig` and build its pyramid.
    -> dict(color[level][cam] u16, truth[cam] f32 at level 0, masks[level][cam], bg_disp[level][cam])"""
    cams = rig["cameras"]
    w0, h0 = sizes[0]
    color = [[None] * len(cams) for _ in sizes]
    masks = [[None] * len(cams) for _ in sizes]
    bgd = [[None] * len(cams) for _ in sizes]
    truth = []
    for ci, cam in enumerate(cams):
        bgr, disp, is_plane, bg = render_camera(cam, w0, h0, frame, seed, device, as_numpy=False)
        truth.append(disp.cpu().numpy())
        for li, im in enumerate(build_pyramid(bgr, sizes)):
            color[li][ci] = im
        if with_masks:
            for li, (w, h) in enumerate(sizes):
                m = resize_area(is_plane.to(bgr.dtype) * 255, w, h)
                masks[li][ci] = (m > 127).cpu().numpy().astype(np.uint8)  # resize.py threshold, CvUtil.h:235-239
                bgd[li][ci] = resize_area(bg, w, h).cpu().numpy().astype(np.float32)
    return {"color": color, "truth": truth, "masks": masks if with_masks else None,
            "bg_disp": bgd if with_masks else None, "sizes": sizes}


def config(name):
    """BASELINE.json configs -> (n_cams, resolution, widths)."""
    return {
        "cfg1": (4, 512, WIDTHS),
        "cfg2": (16, 2048, WIDTHS),
        "cfg2s": (16, 512, WIDTHS),  # config 2's rig at a quarter of the resolution (profiling passes)
        "cfg4": (24, 4096, [4096] + WIDTHS),
        "tiny": (4, 96, [96, 64, 48]),
        "small": (6, 160, [160, 100, 64]),
    }[name]


def write_dataset(root, rig, frames, sizes, seed=360, with_masks=False):
    """Write the reference's on-disk layout under `root` (PNG writer: facebook360_dep_amd.imageio). The files are
    encoded on a thread pool (zlib releases the GIL): a 16-camera 2048^2 frame is 128 PNGs over all levels."""
    from concurrent.futures import ThreadPoolExecutor

    from . import imageio as dio

    os.makedirs(os.path.join(root, "rigs"), exist_ok=True)
    with open(os.path.join(root, "rigs", "rig_calibrated.json"), "w") as f:
        json.dump(rig, f, indent=2)
    jobs = []
    with ThreadPoolExecutor(max_workers=min(32, (os.cpu_count() or 1))) as pool:
        for fi in frames:
            fr = make_frame(rig, sizes, fi, seed + fi, with_masks)
            name = "%06d" % fi
            for li in range(len(sizes)):
                for ci, cam in enumerate(rig["cameras"]):
                    d = os.path.join(root, "video", "color_levels", "level_%d" % li, cam["id"])
                    os.makedirs(d, exist_ok=True)
                    jobs.append(pool.submit(dio.write_png16, os.path.join(d, name + ".png"), fr["color"][li][ci]))
                    if with_masks:
                        d = os.path.join(root, "video", "foreground_masks_levels", "level_%d" % li, cam["id"])
                        os.makedirs(d, exist_ok=True)
                        jobs.append(pool.submit(dio.write_png8, os.path.join(d, name + ".png"), fr["masks"][li][ci] * 255))
                        if fi == frames[0]:  # one static background frame (--background_frame=000000)
                            d = os.path.join(root, "background", "disparity_levels", "level_%d" % li, cam["id"])
                            os.makedirs(d, exist_ok=True)
                            jobs.append(pool.submit(dio.write_pfm, os.path.join(d, "000000.pfm"), fr["bg_disp"][li][ci]))
        for j in jobs:
            j.result()
