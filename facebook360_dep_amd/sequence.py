"""Frame-parallel sequence schedule: one frame per GPU, one process per GPU.

Reproduces the per-level barrier of the reference's render pipeline
(scripts/render/pipeline.py:364-408): for level L, coarse to fine,
    DerpCLI(level L) on every frame  ->  TemporalBilateralFilter(level L) over [t-R, t+R]
    ->  "Transfer": the filtered level overwrites disparity_levels/level_L  ->  level L-1.
The reference moves the +-R frames through the filesystem (TemporalBilateralFilter.cpp:139-160).
Here the only data that crosses ranks inside the level loop is the raw level-L disparity of the
+-R neighbour frames, sent point to point (isend / irecv; backend "nccl" = RCCL, where each
neighbour is one direct xGMI link) — colour guides and masks of the neighbour frames are *inputs*
and are fetched once before the loop (`neighbour_exchange` on the whole pyramid).
Compute is injected (HIP library on the GPU box, the oracle in the CPU tests), so the schedule
itself is testable with gloo.
"""
import torch


def temporal_window(t, first, last, radius):
    """populateMinMaxFrame (TemporalBilateralFilter.cpp:96-119): frames that exist in
    [t - radius, t + radius], i.e. the window clamped to the sequence [first, last]."""
    return max(first, t - radius), min(last, t + radius)


def _as_bytes(x):
    return x if x.dtype in (torch.float32, torch.uint8) else x.view(torch.uint8)


MODE = "p2p"  # "p2p": isend/irecv with the neighbour ranks only; "allgather": one collective, then slice


def neighbour_exchange(x, rank, world, dist=None, radius=2):
    """Send this rank's tensor to the ranks whose window contains it and receive theirs.
    -> list of tensors for ranks lo..hi in order (own tensor included, not copied)."""
    lo, hi = temporal_window(rank, 0, world - 1, radius)
    if world == 1 or dist is None:
        return [x]
    x = x.contiguous()
    view = _as_bytes(x)
    if MODE == "allgather":
        out = torch.empty((world,) + tuple(view.shape), dtype=view.dtype, device=view.device)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, view)
        else:
            dist.all_gather([out[r] for r in range(world)], view)
        return [x if r == rank else (out[r].view(x.dtype) if out.dtype != x.dtype else out[r]) for r in range(lo, hi + 1)]
    recv = {}
    ops = []
    for r in range(lo, hi + 1):
        if r == rank:
            continue
        recv[r] = torch.empty_like(view)
        ops.append(dist.P2POp(dist.irecv, recv[r], r))
        ops.append(dist.P2POp(dist.isend, view, r))
    for work in dist.batch_isend_irecv(ops):
        work.wait()
    out = []
    for r in range(lo, hi + 1):
        if r == rank:
            out.append(x)
        else:
            out.append(recv[r].view(x.dtype) if recv[r].dtype != x.dtype else recv[r])
    return out


def run_level_schedule(rank, world, levels, process_level, disparity_view, static_window, temporal_filter, write_back,
                       dist=None, time_radius=2):
    """Drive this rank's frame (frame index = rank) through `levels` (coarse -> fine).

    process_level(level)                   runs the depth path of this rank's frame at `level`
    disparity_view(level) -> tensor        this rank's raw level disparity [D, h, w] f32
    static_window(level) -> (guides, masks) lists over the window's frames (lo..hi), fetched beforehand
    temporal_filter(level, guides, disps, masks, offset) -> filtered [D, h, w]
    write_back(level, filtered)            the "Transfer" step
    """
    lo, hi = temporal_window(rank, 0, world - 1, time_radius)
    for level in levels:
        process_level(level)
        disps = neighbour_exchange(disparity_view(level), rank, world, dist, time_radius)
        guides, masks = static_window(level)
        filtered = temporal_filter(level, guides, disps, masks, rank - lo)
        write_back(level, filtered)
    return lo, hi
