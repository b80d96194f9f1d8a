"""Frame-parallel sequence schedule: one frame per GPU, one process per GPU.

Reproduces the per-level barrier of the reference's render pipeline
(scripts/render/pipeline.py:364-408): for level L, coarse to fine,
    DerpCLI(level L) on every frame  ->  TemporalBilateralFilter(level L) over [t-R, t+R]
    ->  "Transfer": the filtered level overwrites disparity_levels/level_L  ->  level L-1.
The reference moves the +-R frames of raw level-L disparity through the filesystem
(TemporalBilateralFilter.cpp:139-160); here they move with one all_gather per level over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
Compute is injected (HIP library on the GPU box), so the schedule itself is testable on CPU.
"""
import torch


def temporal_window(t, first, last, radius):
    """populateMinMaxFrame (TemporalBilateralFilter.cpp:96-119): frames that exist in
    [t - radius, t + radius], i.e. the window clamped to the sequence [first, last]."""
    return max(first, t - radius), min(last, t + radius)


def all_gather(x, world, dist=None):
    """x: tensor on this rank -> tensor [world, *x.shape] holding every rank's x (rank order)."""
    x = x.contiguous()
    if world == 1 or dist is None:
        return x[None].clone()
    view = x
    if x.dtype not in (torch.float32, torch.uint8):  # collectives move bytes; u16 has no native support
        view = x.view(torch.uint8)
    out = torch.empty((world,) + tuple(view.shape), dtype=view.dtype, device=view.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, view)
    else:
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, view)
    return out.view(x.dtype) if view.dtype != x.dtype else out


def run_level_schedule(rank, world, levels, process_level, level_views, temporal_filter, write_back, dist=None,
                       time_radius=2, first_frame=0):
    """Drive one frame (index first_frame + rank) through `levels` (coarse -> fine).

    process_level(level)                      runs the depth path of this rank's frame at `level`
    level_views(level) -> (disp, guide, mask) this rank's raw disparity [D,h,w] f32, colour guide and
                                              fov&fg mask [D,h,w] u8 as torch tensors
    temporal_filter(level, guides, disps, masks, offset) -> filtered [D,h,w]
                                              guides/disps/masks: lists over the window's frames
    write_back(level, filtered)               the "Transfer" step
    """
    t = first_frame + rank
    lo, hi = temporal_window(t, first_frame, first_frame + world - 1, time_radius)
    for level in levels:
        process_level(level)
        disp, guide, mask = level_views(level)
        all_disp = all_gather(disp, world, dist)
        all_guide = all_gather(guide, world, dist)
        all_mask = all_gather(mask, world, dist)
        idx = [f - first_frame for f in range(lo, hi + 1)]
        filtered = temporal_filter(level, [all_guide[i] for i in idx], [all_disp[i] for i in idx],
                                   [all_mask[i] for i in idx], t - lo)
        write_back(level, filtered)
    return lo, hi
