"""Shared helpers of the parity tests: the oracle-side pipeline (DerpCLI's level loop,
DerpCLI.cpp:220-323) and comparison metrics. Test infrastructure only."""
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
