#!/usr/bin/env python
"""Turn the files tools/profile_round.sh left in gpurun_out/ into the committed profiles/ set:
trimmed kernel stats (derp:: kernels + one aggregate line for the synthetic-input generator),
per-kernel PMC summaries, the bench lines, and profiles/hbm_traffic.json (read by bench.py).
FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM (gfx950 reports half the bytes; confirmed here on
k_ping_pong_commit, whose reads are 3 x 4 B per pixel); WRITE_SIZE is taken as is (matches the known
9 B/px of the same kernel). Counters are in KB."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
src, dst = "gpurun_out", "profiles"
rows = list(csv.reader(open(os.path.join(src, tag + "_kernel_stats_full.csv"))))
hdr, body = rows[0], rows[1:]
keep = [r for r in body if "derp::" in r[0]]  # templates print as "void derp::k<...>(...)"
other = [r for r in body if "derp::" not in r[0]]
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr)
    for r in keep:
        w.writerow(r)
    w.writerow(["(non-derp kernels: torch synthetic-input rendering / copies, outside the timed region)",
                sum(int(r[1]) for r in other), sum(int(r[2]) for r in other), "", "", "", "", ""])
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = json.load(open(os.path.join(src, "%s_pmc_%s.json" % (tag, c))))
    d = {k: v for k, v in d.items() if "derp::" in k}
    json.dump(d, open(os.path.join(dst, "%s_pmc_%s.json" % (tag, c)), "w"), indent=1, sort_keys=True)
    pm[c] = d
for f in ("_bench.json", "_bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, tag + f), os.path.join(dst, tag + f))


def kb(counter, kernel, field):
    for k, v in pm[counter].items():
        if kernel in k:
            return v[counter][field]
    return 0.0


tr = {}
path = os.path.join(dst, "hbm_traffic.json")
if os.path.exists(path):
    tr = json.load(open(path))
fetch = 2.0 * 1024.0 * kb("FETCH_SIZE", "k_ping_pong(", "max")
write = 1024.0 * kb("WRITE_SIZE", "k_ping_pong(", "max")
tr[cfg] = {
    "source": "%s_pmc_FETCH_SIZE.json / %s_pmc_WRITE_SIZE.json (rocprofv3 --pmc, separate passes)" % (tag, tag),
    "ping_pong_level0_fetch_bytes": fetch,
    "ping_pong_level0_write_bytes": write,
    "ping_pong_level0_bytes_per_launch": fetch + write,
    "corrections": "FETCH_SIZE KB x2 (gfx950 half-count), WRITE_SIZE KB x1",
    "all_levels_fetch_bytes": {n: 2.0 * 1024.0 * kb("FETCH_SIZE", n, "sum") for n in
                               ("k_ping_pong(", "k_random_proposals", "k_reproject", "k_proj_warp", "k_joint_bilateral",
                                "k_blur3_u16", "k_masked_median", "k_brute_costs")},
}
json.dump(tr, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(tr[cfg], indent=1))
