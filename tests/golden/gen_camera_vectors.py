#!/usr/bin/env python3
"""Generates tests/golden/camera_vectors.json by IMPORTING the reference's Python camera
(/root/reference/scripts/util/camera.py) in the authoring container. The reference never
travels; only this script and the vectors it wrote are committed.

Cameras: the 16 FTHETA cameras of res/test/rigs/rig.json (3-term distortion, fov pi/2) and
res/test/cameras/rectilinear.json. Only operations where camera.py agrees with Camera.h are
recorded (no is_outside_image_circle, no ORTHOGRAPHIC, no non-orthonormal ftheta.json).
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "scripts", "util"))
from camera import Camera  # noqa: E402

rng = np.random.default_rng(360)
out = {"generator": "scripts/util/camera.py @ /root/reference", "cameras": []}
rig = json.load(open(os.path.join(REF, "res/test/rigs/rig.json")))["cameras"]
rect = json.load(open(os.path.join(REF, "res/test/cameras/rectilinear.json")))
for cj in rig + [rect]:
    cam = Camera(json_string=json.dumps(cj))
    rec = {"json": cj, "world_to_pixel": [], "pixel_to_world": [], "distort": [], "undistort": []}
    rec["distortion_max"] = float(cam.get_distortion_max())
    fwd = cam.forward()
    for _ in range(24):
        # points in front of the camera, within ~60 degrees of the axis
        d = fwd + 0.9 * rng.normal(size=3) * 0.6
        p = cam.position + d / np.linalg.norm(d) * rng.uniform(0.5, 50.0)
        sees, pix = cam.sees(p)
        if pix is None:
            continue
        rec["world_to_pixel"].append({"point": p.tolist(), "pixel": np.asarray(pix).tolist(), "sees": bool(sees)})
    for _ in range(24):
        pix = np.array([rng.uniform(0.2, 0.8) * cam.resolution[0], rng.uniform(0.2, 0.8) * cam.resolution[1]])
        depth = float(rng.uniform(0.5, 100.0))
        w = cam.pixel_to_world(pix, depth)
        rec["pixel_to_world"].append({"pixel": pix.tolist(), "depth": depth, "point": np.asarray(w).tolist()})
    for r in np.linspace(0.0, 1.5, 7):
        rec["distort"].append([float(r), float(cam.distort(float(r)))])
        rec["undistort"].append([float(r), float(cam.undistort(float(r)))])
    out["cameras"].append(rec)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "camera_vectors.json")
json.dump(out, open(dst, "w"), indent=0)
print("wrote", dst, os.path.getsize(dst), "bytes")
