"""Shared helpers of the parity tests: the oracle-side pipeline (DerpCLI's level loop,
DerpCLI.cpp:220-323) and comparison metrics. Test infrastructure only."""
import os

import numpy as np

from oracle import oracle_lib as O


def oracle_rigs(rig, dst_ids=None):
    cams = rig["cameras"]
    dst = cams if dst_ids is None else [c for i in dst_ids for c in cams if c["id"] == i]
    rs = O.Rig(cams).normalize()
    rd = O.Rig(dst).normalize()
    ids = [c["id"] for c in cams]
    return rs, rd, [ids.index(c["id"]) for c in dst]


def oracle_level(rig, sizes, frame, level, w_full, h_full, prev=None, dst_ids=None, **opts):
    """Build one oracle PyramidLevel with inputs set, projections precomputed and (when `prev`
    holds level+1 disparities) the between-level upsample applied. Returns the Level."""
    rs, rd, d2s = oracle_rigs(rig, dst_ids)
    w, h = sizes[level]
    p = O.make_params(level, len(sizes), w, h, w_full, h_full, **opts)
    L = O.Level(rs, rd, d2s, p)
    use_fg = bool(opts.get("use_foreground_masks"))
    for s in range(rs.n):
        L.set_src(s, frame["color"][level][s], frame["masks"][level][s] if use_fg else None)
    for d in range(rd.n):
        if use_fg:
            L.set_dst(d, bg=frame["bg_disp"][level][d2s[d]])
    L.precompute_projections()
    if prev is not None:
        for d in range(rd.n):
            if use_fg:
                up = O.upsample_disparity(rd, d, prev[d], w, h, frame["bg_disp"][level][d2s[d]],
                                          frame["masks"][level + 1][d2s[d]], frame["masks"][level][d2s[d]])
            else:
                up = O.upsample_disparity(rd, d, prev[d], w, h)
            L.set_dst(d, disparity=up)
    L._keep = (rs, rd)
    return L


def oracle_pyramid(rig, sizes, frame, w_full, h_full, level_end=0, dst_ids=None, counters=None, **opts):
    """-> {level: [disparity per dst]} for every level, coarsest to `level_end`."""
    out = {}
    prev = None
    for level in range(len(sizes) - 1, level_end - 1, -1):
        L = oracle_level(rig, sizes, frame, level, w_full, h_full, prev, dst_ids, **opts)
        L.process()
        prev = [L.get_dst(d)[0] for d in range(L.D)]
        out[level] = prev
        if counters is not None:
            c = L.counters()
            counters[level] = c
    return out


def compare_disparity(got, ref, tol=1e-4):
    """-> (#pixels off by more than `tol` relative or with mismatching NaN-ness, max relative error
    over the pixels that are finite in both)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    nan_mismatch = np.isnan(got) != np.isnan(ref)
    both = np.isfinite(got) & np.isfinite(ref)
    inf_same = np.isinf(got) & np.isinf(ref) & (np.sign(got) == np.sign(ref))
    rel = np.zeros(got.shape)
    den = np.maximum(np.abs(ref[both]), 1e-30)
    rel[both] = np.abs(got[both] - ref[both]) / den
    other = ~both & ~inf_same & ~(np.isnan(got) & np.isnan(ref))
    bad = int(nan_mismatch.sum() + (rel > tol).sum() + (other & ~nan_mismatch).sum())
    return bad, float(rel.max()) if rel.size else 0.0


def bit_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype == np.float32:
        return int((a.view(np.uint32) != b.view(np.uint32)).sum() - ((np.isnan(a) & np.isnan(b)).sum()
                   - ((a.view(np.uint32) == b.view(np.uint32)) & np.isnan(a)).sum()))
    return int((a != b).sum())


class OracleSequence:
    """`sequence.run_schedule` backend with the CPU oracle as compute: what a rank holds for its owned
    frames plus the halo buffers it receives into. Test infrastructure (the product backend is
    `sequence.SequenceRunner` over the HIP library)."""

    def __init__(self, rig, sizes, res, first, last, rank=0, world=1, radius=2, partition=0, threads=2,
                 use_foreground_masks=False, frames=None, partial_coverage=True, **opts):
        import torch

        from facebook360_dep_amd import sequence, synth

        self.rig, self.sizes, self.res = rig, sizes, res
        self.first, self.last, self.rank, self.world, self.radius = first, last, rank, world, radius
        self.threads, self.opts, self.use_fg, self.partition = threads, opts, use_foreground_masks, partition
        self.partial_coverage = partial_coverage
        self.n = len(rig["cameras"])
        self.owned = sequence.owned_frames(first, last, world, rank, partition)
        self.halo = sequence.halo_frames(first, last, world, rank, radius, partition) if world > 1 else []
        # `frames` = {frame number: synth.make_frame dict} when the caller already rendered them
        self.frames = {t: (frames[t] if frames is not None else
                           synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu",
                                            with_masks=use_foreground_masks)) for t in self.owned}
        # [frame][level] tensors; owned colour comes from the rendered frame, halo buffers start empty
        self.color, self.disp, self.fg = {}, {}, {}
        for t in self.owned + self.halo:
            self.color[t], self.disp[t], self.fg[t] = {}, {}, {}
            for level, (w, h) in enumerate(sizes):
                if t in self.frames:
                    self.color[t][level] = torch.from_numpy(np.ascontiguousarray(np.stack(self.frames[t]["color"][level])))
                    if use_foreground_masks:
                        self.fg[t][level] = torch.from_numpy(np.ascontiguousarray(np.stack(self.frames[t]["masks"][level])))
                else:
                    self.color[t][level] = torch.zeros((self.n, h, w, 3), dtype=torch.uint16)
                    if use_foreground_masks:
                        self.fg[t][level] = torch.zeros((self.n, h, w), dtype=torch.uint8)
                self.disp[t][level] = torch.zeros((self.n, h, w), dtype=torch.float32)
        self.fov = {}
        self.raw = {}  # [(frame, level)] unfiltered level result, kept for the tests

    # ---- run_schedule backend
    def compute(self, level):
        for t in self.owned:
            prev = None
            if level + 1 < len(self.sizes):
                prev = [self.disp[t][level + 1][d].numpy() for d in range(self.n)]
            L = oracle_level(self.rig, self.sizes, self.frames[t], level, self.res, self.res, prev,
                             partial_coverage=self.partial_coverage, threads=self.threads,
                             use_foreground_masks=self.use_fg,
                             **self.opts)
            L.process()
            for d in range(self.n):
                self.disp[t][level][d] = __import__("torch").from_numpy(L.get_dst(d)[0])
            self.raw[(t, level)] = self.disp[t][level].numpy().copy()
            if level not in self.fov:
                self.fov[level] = np.stack([L.fov_mask(d) for d in range(self.n)])

    def result_crc(self, level=0):
        """Same definition as sequence.SequenceRunner.result_crc (bench.py's `result_crc`)."""
        from facebook360_dep_amd import sequence

        out = {}
        for t in self.owned:
            crc = 0
            for d in range(self.n):
                crc = sequence.disparity_crc(self.disp[t][level][d].numpy(), crc)
            out[t] = crc
        return out

    def tensor(self, frame, level, kind):
        src = {0: self.color, 1: self.fg, 2: self.disp}[kind][frame][level]
        return src.view(__import__("torch").uint8).reshape(-1)

    def scratch(self, level, kind):
        import torch

        return torch.empty_like(self.tensor(self.owned[0] if self.owned else self.halo[0], level, kind))

    def before_exchange(self):
        pass

    def after_exchange(self):
        pass

    def filter(self, level):
        import torch

        from facebook360_dep_amd import sequence

        out = {}
        for t in self.owned:
            lo, hi = sequence.temporal_window(t, self.first, self.last, self.radius)
            res = []
            for d in range(self.n):
                masks = []
                for u in range(lo, hi + 1):
                    m = self.fov[level][d]
                    if self.use_fg:
                        m = m & self.fg[u][level][d].numpy()
                    masks.append(m)
                # weights (b, g, b): TemporalBilateralFilter.cpp:176-178
                res.append(O.temporal_filter([self.color[u][level][d].numpy() for u in range(lo, hi + 1)],
                                             [self.disp[u][level][d].numpy() for u in range(lo, hi + 1)], masks,
                                             t - lo, 0.01, O.temporal_space_radius(level), 0.5, 1.0, 0.5,
                                             threads=self.threads))
            out[t] = torch.from_numpy(np.stack(res))
        for t in self.owned:  # "Transfer" after every owned frame is filtered
            self.disp[t][level].copy_(out[t])

    def exchange_inputs(self, dist, mode="p2p"):
        from facebook360_dep_amd import sequence

        transfers = sequence.plan(self.first, self.last, self.world, self.radius, self.partition)
        for level in range(len(self.sizes)):
            for kind in [0] + ([1] if self.use_fg else []):
                sequence.exchange(transfers, self.rank, lambda f: self.tensor(f, level, kind), dist, mode,
                                  lambda: self.scratch(level, kind))


# ---- observed-count baseline -------------------------------------------------------------------------
# Where a test tolerates last-ulp libm differences (glibc on the host vs OCML on the device), the count it
# OBSERVED on the MI355X is pinned in tests/golden/gpu_observed_baseline.json and asserted for equality, so a
# regression from 0 to "still under the tolerance" is caught. DERP_RECORD_BASELINE=1 records instead
# (into gpurun_out/gpu_observed_baseline.json, to be reviewed and copied into tests/golden/).
_BASELINE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_observed_baseline.json")
_RECORD_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                            "gpu_observed_baseline.json")


def observed(name, value):
    import json

    if os.environ.get("DERP_RECORD_BASELINE"):
        os.makedirs(os.path.dirname(_RECORD_PATH), exist_ok=True)
        rec = {}
        if os.path.exists(_RECORD_PATH):
            with open(_RECORD_PATH) as f:
                rec = json.load(f)
        rec[name] = value
        with open(_RECORD_PATH, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
        return
    with open(_BASELINE_PATH) as f:
        base = json.load(f)
    assert name in base, "no committed baseline for %r (run once with DERP_RECORD_BASELINE=1)" % name
    assert value == base[name], "%s: observed %r, committed baseline %r" % (name, value, base[name])
