#!/usr/bin/env python
"""Writes the JPEG vectors of tests/test_image_codecs.py::test_jpeg_committed_vectors_match_libjpeg_turbo.

Each <name>.jpg is ENCODED by Pillow's libjpeg-turbo from a synthetic picture and DECODED again by the same library;
expected.json keeps the shape and the CRC-32 of the decoded 8-bit samples (R, G, B order; one channel for gray). The
library is the one OpenCV's imread links (libjpeg-turbo, default settings: integer "islow" IDCT, fancy upsampling), so
the CRCs are what `cv::imread(path, IMREAD_UNCHANGED)` hands the reference for these files. Run once where Pillow is
installed:  python tests/golden/codecs/gen_codec_vectors.py
"""
import json
import os
import sys
import zlib

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from test_image_codecs import scene  # noqa: E402

expected = {}


def keep(name, im, **kw):
    path = os.path.join(HERE, name + ".jpg")
    im.save(path, **kw)
    a = np.asarray(Image.open(path))
    a = a.reshape(a.shape[0], a.shape[1], -1)
    expected[name + ".jpg"] = {"shape": list(a.shape), "crc32": zlib.crc32(a.astype(np.uint8).tobytes()), "encoder": kw}


for (w, h) in ((37, 29), (64, 48), (7, 5)):
    rgb = Image.fromarray(scene(w, h, 3))
    gray = Image.fromarray(scene(w, h, 1)[..., 0])
    for sub, tag in ((0, "444"), (1, "422"), (2, "420")):
        keep("rgb_%dx%d_%s_q85" % (w, h, tag), rgb, quality=85, subsampling=sub)
        keep("rgb_%dx%d_%s_q60_progressive" % (w, h, tag), rgb, quality=60, subsampling=sub, progressive=True)
    keep("rgb_%dx%d_420_q95_restart" % (w, h), rgb, quality=95, subsampling=2, restart_marker_blocks=2, optimize=True)
    keep("gray_%dx%d_q75" % (w, h), gray, quality=75)
    keep("gray_%dx%d_q40_progressive" % (w, h), gray, quality=40, progressive=True)
keep("rgb_64x48_411_q80", Image.fromarray(scene(64, 48, 3)), quality=80, subsampling="4:1:1")
keep("rgb_64x48_adobe_rgb_q90", Image.fromarray(scene(64, 48, 3)), quality=90, keep_rgb=True)
noise = Image.fromarray(np.random.default_rng(1).integers(0, 256, (45, 52, 3)).astype(np.uint8))
keep("noise_52x45_420_q10_progressive", noise, quality=10, subsampling=2, progressive=True)
keep("noise_52x45_444_q100", noise, quality=100, subsampling=0)
# the ENCODER's vectors: CRC-32 of the file libjpeg-turbo writes for a synthetic picture (scene(w, h, c) of the test module)
# at a quality; the test encodes the same picture with derp_jpeg_encode and must produce the same bytes
encoder = {}
for (w, h, c, q) in ((37, 29, 3, 95), (64, 48, 1, 95), (7, 5, 3, 75), (100, 75, 3, 95), (17, 9, 3, 30), (16, 16, 1, 100)):
    a = scene(w, h, c)
    path = os.path.join(HERE, "_tmp.jpg")
    Image.fromarray(a if c == 3 else a[..., 0]).save(path, quality=q)
    data = open(path, "rb").read()
    os.remove(path)
    encoder["%dx%dx%d_q%d" % (w, h, c, q)] = {"bytes": len(data), "crc32": zlib.crc32(data)}
json.dump(encoder, open(os.path.join(HERE, "encoder_expected.json"), "w"), indent=1, sort_keys=True)
json.dump(expected, open(os.path.join(HERE, "expected.json"), "w"), indent=1, sort_keys=True)
print(len(expected), "vectors; libjpeg", features.version("jpg"), "turbo:", features.check_feature("libjpeg_turbo"))
