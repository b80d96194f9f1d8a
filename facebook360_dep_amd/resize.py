#!/usr/bin/env python
"""Pyramid builder — mirrors the reference's scripts/render/resize.py (resize_camera :51-85,
resize_frames :94-133, flags :205-214) on top of the C-ABI: every frame is area-resized
(cv2.INTER_AREA semantics) to the fixed pyramid widths on the GPU and written to
<dst_dir>/level_<L>/<camera>/<frame>.<ext>. Masks are thresholded like resize.py's `threshold`.

    python -m facebook360_dep_amd.resize --src_dir=.../video/color --dst_dir=.../video/color_levels \
        --rig=.../rigs/rig.json --first=000000 --last=000002 [--threshold=127]
"""
import argparse
import json
import os
import sys

import numpy as np

from . import derp, imageio
from .synth import WIDTHS  # scripts/render/config.py:46


def get_frame_path(src_dir, camera, frame):
    cam_dir = os.path.join(src_dir, camera)
    files = sorted(f for f in os.listdir(cam_dir) if not f.startswith("."))
    if not files:
        raise Exception(f"No files in {cam_dir}")
    ext = os.path.splitext(files[0])[1]
    return os.path.join(cam_dir, frame + ext)


def level_sizes(rig_width, rig_height):
    """(width, height) of every pyramid level for a rig resolution — resize.py:71-74."""
    ratio = rig_height / rig_width
    out = []
    for width in WIDTHS:
        height = round(ratio * width)
        height += height % 2
        out.append((width, height))
    return out


def resize_camera(g, src_dir, dst_dir, camera, rig_resolution, frame, threshold):
    """resize.py:51-85. `g` is a derp.Derp context (any rig: only derp_resize_area is used)."""
    original_file = get_frame_path(src_dir, camera, frame)
    if not os.path.isfile(original_file):
        raise Exception(f"Non-existent file for resize: {original_file}")
    frame_fn = os.path.basename(original_file)
    ext = os.path.splitext(frame_fn)[1]
    # cv2.imread(original_file, cv2.IMREAD_UNCHANGED): whatever format the directory holds (resize.py:66-70)
    img = imageio.read_pfm(original_file) if ext == ".pfm" else imageio.read_image(original_file)
    if img.ndim == 3 and img.shape[2] == 4:
        img = img[..., :3]
    # cv2.imwrite(new_file, scaled) keeps the source's container (resize.py:82-85): PNG, TIFF and 8-bit JPEG (quality 95,
    # byte for byte libjpeg's output) stay what they are; BMP and PNM sources are written as PNG under a .png name.
    jpeg = ext.lower() in (".jpg", ".jpeg", ".jpe") and img.dtype == np.uint8
    out_ext = ext if jpeg or ext.lower() in (".pfm", ".png", ".tif", ".tiff") else ".png"
    frame_fn = os.path.splitext(frame_fn)[0] + out_ext
    for level, (width, height) in enumerate(level_sizes(rig_resolution[0], rig_resolution[1])):
        new_file = os.path.join(dst_dir, f"level_{level}", camera, frame_fn)
        os.makedirs(os.path.dirname(new_file), exist_ok=True)
        if width > img.shape[1] or height > img.shape[0]:
            # cv2.resize would enlarge here (INTER_AREA falls back to bilinear when upsampling); the
            # pipeline only runs on frames at least as large as level 0, so this level is skipped.
            continue
        if img.dtype == np.uint8 and img.ndim == 3:  # 8-bit colour: the GPU path works on 16-bit texels
            scaled = g.resize_area(img.astype(np.uint16), width, height).astype(np.uint8)
        else:
            scaled = g.resize_area(img, width, height)
        if threshold is not None:
            scaled = np.where(scaled > threshold, 255, 0).astype(scaled.dtype)  # cv2.threshold(.., 255, THRESH_BINARY)
        if ext == ".pfm":
            imageio.write_pfm(new_file, scaled)
        elif out_ext.lower() in (".tif", ".tiff"):
            imageio.write_tiff(new_file, scaled)
        elif jpeg:
            imageio.write_jpeg(new_file, scaled)
        elif scaled.dtype == np.uint16:
            imageio.write_png16(new_file, scaled)
        else:
            imageio.write_png8(new_file, scaled)


def resize_frames(src_dir, dst_dir, rig, first, last, threshold=None, device=0):
    """resize.py:94-133 (the reference fans out over a process pool; here one GPU context does all)."""
    g = derp.Derp(rig["cameras"], device=device)
    try:
        for frame in range(int(first), int(last) + 1):
            for camera in rig["cameras"]:
                resize_camera(g, src_dir, dst_dir, camera["id"], camera["resolution"], "%06d" % frame, threshold)
    finally:
        g.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--src_dir", required=True, help="path to the source directory")
    ap.add_argument("--dst_dir", required=True, help="path to the destination directory")
    ap.add_argument("--rig", required=True, help="path to the rig JSON")
    ap.add_argument("--first", default="", help="first frame (default: first in the source directory)")
    ap.add_argument("--last", default="", help="last frame (default: last in the source directory)")
    ap.add_argument("--threshold", type=int, default=None, help="binary threshold (masks: 127)")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    with open(args.rig) as f:
        rig = json.load(f)
    cameras_rig = sorted(c["id"] for c in rig["cameras"])
    cameras_dir = sorted(d for d in os.listdir(args.src_dir) if os.path.isdir(os.path.join(args.src_dir, d)))
    if not cameras_dir:
        print(f"No cameras found in {args.src_dir}")
        return 1
    if cameras_rig != cameras_dir:
        print(f"Cameras from rig differ from cameras in source directory: {cameras_rig} vs {cameras_dir}")
        return 1
    frames = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(args.src_dir, cameras_dir[0]))
                    if os.path.isfile(os.path.join(args.src_dir, cameras_dir[0], f)))
    if not frames:
        print(f"No frames found in {args.src_dir}/{cameras_dir[0]}")
        return 1
    first = args.first or frames[0]
    last = args.last or frames[-1]
    resize_frames(args.src_dir, args.dst_dir, rig, first, last, args.threshold, args.device)
    return 0


if __name__ == "__main__":
    sys.exit(main())
