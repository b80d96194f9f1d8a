#!/usr/bin/env python
"""Turn the files tools/profile_round.sh left in gpurun_out/ into the committed profiles/ set:
trimmed kernel stats (derp:: kernels + one aggregate line for the synthetic-input generator), per-kernel PMC
summaries (HBM bytes AND the SQ issue counters), the bench lines, and profiles/valu_roofline.json, which
bench.py reads for `roofline`.

Counter handling, per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE is doubled (gfx950 reports half the
bytes; confirmed here on k_ping_pong_commit, whose reads are 3 x 4 B per pixel), WRITE_SIZE is taken as is;
both are in KB. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x 4 = cycles), summed over the
chip. GRBM_GUI_ACTIVE counts cycles and is summed over the 8 XCDs (checked: value / 8 / launch duration =
2.3-2.4 GHz), so the chip-busy cycles of a launch are GRBM_GUI_ACTIVE / 8. "level-0 launch" = the dispatch with the largest value of each counter
(the finest level is by far the biggest launch of a kernel).
usage: tools/make_profiles.py <tag> <config>"""
import csv
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("derp_kernels.h", "derp_capi.hip", "derp_camera.h", "gcc_algos.h", "derp_sequence.h")


def kernel_sources_sha256():
    """Hash of the translation unit the counters were collected on; bench.py refuses to combine them with the
    timings of other sources."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "facebook360_dep_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
src, dst = "gpurun_out", "profiles"
N_SIMD = 1024
N_XCD = 8
stats_path = os.path.join(src, tag + "_kernel_stats_full.csv")
rows = list(csv.reader(open(stats_path))) if os.path.exists(stats_path) else [["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]]
hdr, body = rows[0], rows[1:]
keep = [r for r in body if "derp::" in r[0]]  # templates print as "void derp::k<...>(...)"
other = [r for r in body if "derp::" not in r[0]]
if body:
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(hdr)
        for r in keep:
            w.writerow(r)
        w.writerow(["(non-derp kernels: torch synthetic-input rendering / copies / RCCL, outside the depth path)",
                    sum(int(r[1]) for r in other), sum(int(r[2]) for r in other), "", "", "", "", ""])
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ISSUE", "SQ_INSTS", "L2", "VALU_F32", "VALU_F64"):
    path = os.path.join(src, "%s_pmc_%s.json" % (tag, c))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    d = {k: v for k, v in d.items() if "derp::" in k}
    json.dump(d, open(os.path.join(dst, "%s_pmc_%s.json" % (tag, c)), "w"), indent=1, sort_keys=True)
    pm[c] = d
for f in ("_bench.json", "_bench_under_rocprof.json"):
    if os.path.exists(os.path.join(src, tag + f)):
        shutil.copy(os.path.join(src, tag + f), os.path.join(dst, tag + f))


def names_of(kernel):
    """A cost kernel exists under two register budgets (k_ping_pong / k_ping_pong_w3, derp_kernels.h): a run launches one of
    them, chosen by the rig's camera count — look for the name given, then for its _w3 twin."""
    return [kernel, kernel.replace("(", "_w3(") if kernel.endswith("(") else kernel + "_w3"]


def val(group, kernel, counter, field):
    for name in names_of(kernel):
        for k, v in pm.get(group, {}).items():
            if name in k and counter in v and v[counter].get("max", 0.0) > 0:
                return v[counter][field]
    return 0.0


def kernel_view(kernel):
    fetch = 2.0 * 1024.0 * val("FETCH_SIZE", kernel, "FETCH_SIZE", "max")
    write = 1024.0 * val("WRITE_SIZE", kernel, "WRITE_SIZE", "max")
    g = lambda c: val("SQ_ISSUE", kernel, c, "max")  # noqa: E731
    wave, valu, gui = g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU"), g("GRBM_GUI_ACTIVE")
    out = {"hbm_fetch_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write,
           "hbm_bytes_per_launch": fetch + write}
    if wave:
        out["valu_busy_cycles_per_launch"] = 4.0 * valu
        out["wave_cycle_shares"] = {
            "valu": round(valu / wave, 4), "scalar": round(g("SQ_ACTIVE_INST_SCA") / wave, 4),
            "lds": round(g("SQ_ACTIVE_INST_LDS") / wave, 4), "wait_any": round(g("SQ_WAIT_ANY") / wave, 4),
            "wait_inst_any": round(g("SQ_WAIT_INST_ANY") / wave, 4)}
        if gui:
            # cycles the launch kept the chip busy; each of the 1024 SIMDs could have a VALU instruction active in
            # every one of them
            busy = gui / N_XCD
            out["valu_busy_frac"] = round(4.0 * valu / (N_SIMD * busy), 4)
            out["waves_per_simd_avg"] = round(4.0 * wave / (N_SIMD * busy), 3)
            out["chip_busy_cycles_per_launch"] = busy
    i = lambda c: val("SQ_INSTS", kernel, c, "max")  # noqa: E731
    if i("SQ_INSTS_VALU"):
        # second numerator: VALU wave-instructions x 4 cycles (the issue time of a 64-lane instruction on a 16-lane
        # pipe). SQ_ACTIVE_INST_VALU (above) is summed over WAVES: with three or more resident waves per SIMD two of
        # them can have an instruction in the pipe in the same cycle, which is how a "busy fraction" above 1 arises
        # for kernels like k_joint_bilateral — both are upper estimates of pipe occupancy there, and neither prices
        # the 8- / 16-pass instructions (fp64 transcendentals, packed fp32) at their true cost.
        out["valu_insts_x4_cycles_per_launch"] = 4.0 * i("SQ_INSTS_VALU")
        if gui:
            out["valu_insts_x4_frac"] = round(4.0 * i("SQ_INSTS_VALU") / (N_SIMD * gui / N_XCD), 4)
        out["insts_per_launch"] = {"valu": i("SQ_INSTS_VALU"), "salu": i("SQ_INSTS_SALU"), "lds": i("SQ_INSTS_LDS"),
                                   "vmem_rd": i("SQ_INSTS_VMEM_RD")}
        out["lds_bank_conflict_over_idx_active"] = round(i("SQ_LDS_BANK_CONFLICT") / max(i("SQ_LDS_IDX_ACTIVE"), 1.0), 4)
        out["valu_lane_utilisation"] = round(i("SQ_THREAD_CYCLES_VALU") / max(64.0 * val("SQ_ISSUE", kernel, "SQ_ACTIVE_INST_VALU", "max"), 1.0), 4)
    return out


sys.path.insert(0, os.path.join(ROOT, "tools"))
import valu_model  # noqa: E402

UBENCH = os.path.join(dst, "r04_ubench.jsonl")


def max_ms(kernel):
    """duration of the longest dispatch of `kernel` in the kernel trace (= its level-0 launch)"""
    for name in names_of(kernel):
        for r in keep:
            if name in r[0]:
                try:
                    return float(r[hdr.index("MaxNs")]) / 1e6
                except (ValueError, IndexError):
                    return None
    return None


def issue_model(kernel, view):
    """tools/valu_model.py on the typed VALU counters of the level-0 launch: SIMD issue cycles per launch."""
    if "VALU_F32" not in pm or "VALU_F64" not in pm or not os.path.exists(UBENCH):
        return None
    counts, used = {}, kernel
    for name in names_of(kernel):
        for grp in ("VALU_F32", "VALU_F64"):
            for k, v in pm[grp].items():
                if name in k and v.get("SQ_INSTS_VALU", v.get("SQ_WAVE_CYCLES", {})).get("max", 0.0) > 0:
                    for c, x in v.items():
                        counts[c.replace("SQ_INSTS_VALU_", "").replace("SQ_INSTS_VALU", "VALU")] = x["max"]
        if counts.get("VALU"):
            used = name
            break
    if not counts.get("VALU"):
        return None
    waves = view.get("waves_per_simd_avg") or 3
    costs = valu_model.class_costs(UBENCH, waves)
    m = valu_model.issue_cycles(counts, valu_model.static_mix(used), costs)
    out = {"issue_cycles_per_launch": m["cycles_upper"], "issue_cycles_if_simple_ops_coissue": m["cycles_lower"],
           "instructions_by_class": m["by_class"], "instructions_by_counter": m["dynamic_by_counter"],
           "class_split_sensitivity": m["class_split_sensitivity"], "kernel_symbol_priced": used,
           "class_cycles_used": costs}
    ms = max_ms(kernel)
    if ms:
        avail = N_SIMD * 2.4e9 * ms * 1e-3
        out["trace_max_ms"] = ms
        out["issue_frac_at_2.4GHz"] = round(m["cycles_upper"] / avail, 4)
        out["issue_frac_if_simple_ops_coissue"] = round(m["cycles_lower"] / avail, 4)
    return out


path = os.path.join(dst, "valu_roofline.json")
tr = json.load(open(path)) if os.path.exists(path) else {}
pp = kernel_view("k_ping_pong(")
entry = {
    "kernel_sources_sha256": kernel_sources_sha256(),
    "source": "%s_pmc_{FETCH_SIZE,WRITE_SIZE,SQ_ISSUE,SQ_INSTS}.json (rocprofv3 --pmc, one pass per group, on "
              "`bench.py --steps 1 --warmup 0`); %s_kernel_stats.csv for durations" % (tag, tag),
    "corrections": "FETCH_SIZE KB x2 (gfx950 half-count), WRITE_SIZE KB x1; SQ quad-cycles x4",
    "ping_pong_level0_valu_busy_cycles_per_launch": pp.get("valu_busy_cycles_per_launch"),
    "ping_pong_level0_valu_busy_frac": pp.get("valu_busy_frac"),
    "ping_pong_level0_valu_insts_x4_cycles_per_launch": pp.get("valu_insts_x4_cycles_per_launch"),
    "ping_pong_level0_waves_per_simd": pp.get("waves_per_simd_avg"),
    "ping_pong_level0_wave_cycle_shares": pp.get("wave_cycle_shares"),
    "ping_pong_level0_hbm_bytes_per_launch": pp.get("hbm_bytes_per_launch"),
    "kernels_level0_launch": {n: kernel_view(n) for n in
                              ("k_ping_pong(", "k_random_proposals", "k_reproject_bias", "k_proj_warp_inv", "k_blur3_u16",
                               "k_joint_bilateral", "k_temporal", "k_proj_warp(", "k_brute_costs")},
    "all_launches_fetch_bytes": {n: 2.0 * 1024.0 * val("FETCH_SIZE", n, "FETCH_SIZE", "sum") for n in
                                 ("k_ping_pong(", "k_random_proposals", "k_reproject_bias", "k_proj_warp_inv", "k_proj_warp(",
                                  "k_joint_bilateral", "k_blur3_u16", "k_masked_median", "k_brute_costs", "k_temporal")},
}
for n, view in entry["kernels_level0_launch"].items():
    im = issue_model(n, view)
    if im:
        view["issue_model"] = im
for short, kname in (("random", "k_random_proposals"),):  # the same fields for the second cost kernel of the step
    v = entry["kernels_level0_launch"][kname]
    im = v.get("issue_model")
    entry[short + "_level0_hbm_bytes_per_launch"] = v.get("hbm_bytes_per_launch")
    entry[short + "_level0_waves_per_simd"] = v.get("waves_per_simd_avg")
    entry[short + "_level0_wave_cycle_shares"] = v.get("wave_cycle_shares")
    if im:
        entry[short + "_level0_issue_cycles_per_launch"] = im["issue_cycles_per_launch"]
        entry[short + "_level0_issue_cycles_if_simple_ops_coissue"] = im["issue_cycles_if_simple_ops_coissue"]
        entry[short + "_level0_class_cycles"] = im["class_cycles_used"]
    l2 = pm.get("L2", {})
    for k, x in l2.items():
        if kname in k and "TCC_REQ_sum" in x and x["TCC_REQ_sum"]["max"] > 0:
            req, miss = x["TCC_REQ_sum"]["max"], x.get("TCC_MISS_sum", {}).get("max", 0.0)
            entry[short + "_level0_l2"] = {"requests": req, "misses": miss, "hit_rate": round(1.0 - miss / max(req, 1.0), 4)}
ppm = entry["kernels_level0_launch"]["k_ping_pong("].get("issue_model")
if ppm:
    entry["ping_pong_level0_issue_cycles_per_launch"] = ppm["issue_cycles_per_launch"]
    entry["ping_pong_level0_issue_cycles_if_simple_ops_coissue"] = ppm["issue_cycles_if_simple_ops_coissue"]
    entry["ping_pong_level0_class_cycles"] = ppm["class_cycles_used"]
    entry["issue_model"] = ("tools/valu_model.py: typed VALU counters of the level-0 launch (%s_pmc_VALU_F32/_F64.json) priced at the "
                            "per-class issue intervals of tools/valu_ubench.hip (r04_ubench.jsonl)" % tag)
# every derp:: kernel of the PMC run (`bench.py --steps 1`: one step = the whole sequence)
entry["whole_step_hbm_fetch_bytes"] = 2.0 * 1024.0 * sum(v["FETCH_SIZE"]["sum"] for v in pm.get("FETCH_SIZE", {}).values())
entry["whole_step_hbm_write_bytes"] = 1024.0 * sum(v["WRITE_SIZE"]["sum"] for v in pm.get("WRITE_SIZE", {}).values())
# effective clock of the profiled level-0 launch: GRBM cycles / its duration in the kernel trace (max duration row)
for r in keep:
    if ("k_ping_pong(" in r[0] or "k_ping_pong_w3(" in r[0]) and pp.get("chip_busy_cycles_per_launch"):
        try:
            max_ns = float(r[hdr.index("MaxNs")])
            entry["ping_pong_level0_trace_max_ms"] = max_ns / 1e6
        except (ValueError, IndexError):
            pass
tr[cfg] = entry
json.dump(tr, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in entry.items() if k.startswith("ping_pong")}, indent=1))
