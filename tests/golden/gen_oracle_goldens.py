#!/usr/bin/env python3
"""Freezes the oracle's output on the 'tiny' synthetic rig (4 cameras, 96x96, 3 levels) into
tests/golden/oracle_tiny.npz: inputs are regenerated deterministically by the test; the file holds
the expected level-0 / level-2 disparities with and without foreground masks, table samples, and the
sibling stages (rephotography render + SSIM / NCC maps + mean score, foreground mask, layer compositing), and the
level-0 result of a 3-frame sequence run through the per-level temporal-filter schedule (BASELINE config 3).
Run from the repo root:  python tests/golden/gen_oracle_goldens.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from facebook360_dep_amd import synth  # noqa: E402
from tests import common  # noqa: E402


def run():
    n, res, widths = synth.config("tiny")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)
    frame = synth.make_frame(rig, sizes, with_masks=True, device="cpu")
    out = {}
    cnt = {}
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, counters=cnt, partial_coverage=True)
    out["plain_l0"] = np.stack(ref[0])
    out["plain_l2"] = np.stack(ref[2])
    out["plain_counters"] = np.array([[cnt[l]["n_cost"], cnt[l]["n_pair"]] for l in sorted(cnt)], dtype=np.int64)
    ref = common.oracle_pyramid(rig, sizes, frame, res, res, partial_coverage=True, use_foreground_masks=True)
    out["fg_l0"] = np.stack(ref[0])
    L = common.oracle_level(rig, sizes, frame, 1, res, res, partial_coverage=True)
    L.reproject_colors()
    out["warp_1_0"] = L.proj(1, 0, "warp")
    out["color_1_0"] = L.proj(1, 0, "color")
    out["bias_1_0"] = L.proj(1, 0, "bias")
    out["variance_2"] = L.variance(2)
    out["fov_3"] = L.fov_mask(3)
    out["input_color_l1_cam0"] = frame["color"][1][0]
    # sibling stages: rephotography score, foreground masks, layer compositing
    from oracle import oracle_lib as O

    R = O.Rig(rig["cameras"]).normalize()
    cols = frame["color"][0]
    disps = [np.asarray(d, dtype=np.float32) for d in out["plain_l0"]]
    rendered = O.rephotograph(R, 1, cols, disps)
    out["rephoto_cam1"] = rendered
    mask = (np.isfinite(disps[1]) & (disps[1] > 0)).astype(np.uint8)
    x = cols[1].astype(np.float32) * (np.float32(1.0) / np.float32(65535.0)) * mask[..., None]
    score = O.compute_ssim(x, rendered[..., :3], 1)
    out["ssim_cam1"] = score
    out["ncc_cam1_r2"] = O.compute_ssim(x, rendered[..., :3], 2, 0, 0, 1)
    out["mssim_cam1"] = np.array(O.average_score(score, mask), dtype=np.float64)
    # the rephotography renderer (CanopyScene::cubemap): camera 1 alone and all the others, seen from camera 1
    centre = rig["cameras"][1]["origin"]
    out["canopy_only_cam1"] = O.canopy_cubemap(R, cols, disps, [0, 1, 0, 0], centre, 32)
    out["canopy_others_cam1"] = O.canopy_cubemap(R, cols, disps, [1, 0, 1, 1], centre, 32)
    frame1 = synth.make_frame(rig, sizes, frame=3, device="cpu")
    out["fgmask_cam0"] = O.generate_foreground_mask(cols[0], frame1["color"][0][0], 1, 0.04, 4)
    out["layers_cam0"] = O.layer_disparities(np.where(frame["masks"][0][0] == 1, disps[0], 0).astype(np.float32),
                                             frame["bg_disp"][0][0])
    # BASELINE config 3's schedule in miniature: 3 frames, per-level temporal filter seeding the next level
    from facebook360_dep_amd import sequence

    seq = common.OracleSequence(rig, sizes, res, 0, 2, threads=-1)
    sequence.run_schedule(seq, list(range(len(sizes) - 1, -1, -1)), 0, 2, 0, 1)
    out["seq3_l0"] = np.stack([seq.disp[t][0].numpy() for t in range(3)])
    out["seq3_raw_l2_frame1"] = seq.raw[(1, len(sizes) - 1)]
    out["seq3_input_color_l1_cam0_frame2"] = seq.frames[2]["color"][1][0]
    return out


if __name__ == "__main__":
    data = run()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_tiny.npz")
    np.savez_compressed(dst, **data)
    print("wrote", dst, os.path.getsize(dst), "bytes")
