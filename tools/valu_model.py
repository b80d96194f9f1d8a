#!/usr/bin/env python
"""Issue-cycle model of a kernel's VALU work on gfx950 — the numerator of bench.py's `roofline.frac`.

Round 3 charged every VALU instruction 4 cycles (or summed SQ_ACTIVE_INST_VALU over the resident waves), which read
above 1 for kernels with many waves of cheap instructions. This model prices instruction CLASSES at the issue
intervals measured on this chip by tools/valu_ubench.hip (profiles/r04_ubench.jsonl), at the kernel's own number
of waves per SIMD:

  S   v_add / v_sub / v_mul_f32, v_mov_b32, v_and / v_or / v_xor_b32, v_lshrrev_b32, v_add_u32   2.08 (2.8 at 3 waves)
  F   every other 32-bit VALU instruction (fma, cvt, trunc, min / max, compares, cndmask, lshl, bfe, mul_lo ...)   4.1-4.2
  D   fp64 add / mul / fma / div_* / ldexp, packed fp32 (v_pk_*), 64-bit integer ops                    4.3-4.4
  T32 / T64   transcendentals: rcp / sqrt / exp f32 8.1; rcp / rsq / sqrt f64 16.0

A SIMD has two 16-lane halves: D instructions hold both for ~4.3 cycles, F instructions hold one for ~4.1 while
the other half can run an S instruction of another wave (measured: v_trunc + v_add per wave pair = 4.10 cycles,
not 6.2), and an S instruction alone takes ~2.1. Hence two figures:

  cycles_upper  every instruction at its own stand-alone interval              (what `roofline.frac` uses: <= 1)
  cycles_lower  S instructions hidden behind F instructions where there are enough of them

Dynamic counts come from rocprofv3's typed counters on the level-0 launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32,
_CVT, _{ADD,MUL,FMA,TRANS}_F64, _INT32, _INT64, SQ_INSTS_VALU); which counter sees which instruction was measured
by running the micro-benchmark under the same counters (profiles/r04_ubench_pmc_*.json): a packed instruction
counts once in ADD / MUL / FMA_F32, v_sub in ADD, v_fmac / v_div_fmas in FMA, add / sub / mul_lo / bfe / lshl_add in
INT32, 64-bit shifts / mad_u64 in INT64; moves, logic, shifts, compares, cndmask, min / max, trunc are in no typed
counter ("other" = SQ_INSTS_VALU - sum of the typed ones). Inside a counter the split between classes (packed vs
plain, S vs F) is taken from the kernel's STATIC instruction mix (the disassembly of the build flags in
__graft_entry__.HIP_FLAGS) — the hot loops dominate both.
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

S_OPS = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
         "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32"}
T32_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
T64_OPS = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
# which typed counter an instruction lands in (measured, profiles/r04_ubench_pmc_*.json)
COUNTER_OF = [
    (re.compile(r"^v_(pk_)?(add|sub|subrev)_f32"), "ADD_F32"), (re.compile(r"^v_(pk_)?mul_f32"), "MUL_F32"),
    (re.compile(r"^v_(pk_)?(fma|fmac|mad|mac)_f32|^v_div_fmas_f32"), "FMA_F32"),
    (re.compile(r"^v_cvt_"), "CVT"), (re.compile(r"^v_(add|sub)_f64"), "ADD_F64"), (re.compile(r"^v_mul_f64"), "MUL_F64"),
    (re.compile(r"^v_(fma|fmac)_f64|^v_div_fmas_f64"), "FMA_F64"),
    (re.compile(r"^v_(add|sub|subrev)(_co)?_u32|^v_addc|^v_subb|^v_mul_(lo|hi)_(u|i)32|^v_bfe_|^v_lshl_add_u32|^v_add3_u32|"
                r"^v_mad_(u|i)32|^v_add_lshl_u32|^v_mul_u32_u24|^v_mad_u32_u24"), "INT32"),
    (re.compile(r"^v_lshlrev_b64|^v_lshrrev_b64|^v_ashrrev_i64|^v_mad_(u|i)64|^v_lshl_add_u64"), "INT64"),
]


def klass(op):
    if op in T64_OPS:
        return "T64"
    if op in T32_OPS:
        return "T32"
    if op.startswith("v_pk_") or op.endswith("_f64") and not op.startswith("v_cmp") and not op.startswith("v_cvt") \
            or re.match(r"^v_(lshlrev_b64|lshrrev_b64|ashrrev_i64|mad_(u|i)64|lshl_add_u64)", op):
        return "D"
    if op in S_OPS:
        return "S"
    return "F"


def counter_of(op):
    if op in T32_OPS:
        return "TRANS_F32"
    if op in T64_OPS:
        return "TRANS_F64"
    for rx, name in COUNTER_OF:
        if rx.match(op):
            return name
    return "OTHER"


def _column(ubench_path, w):
    rows = [json.loads(ln) for ln in open(ubench_path) if ln.startswith("{")]
    t = {}
    for r in rows:
        if r.get("mode") == "throughput" and r.get("waves_per_simd") == w:
            t[r["op"]] = r["ticks_per_inst_max"]
    pick = lambda names: sum(t[n] for n in names) / len(names)  # noqa: E731
    return {"S": pick(["add_f32", "mul_f32", "mov_b32", "and_b32", "add_u32"]),
            "F": pick(["fma_f32", "cvt_f32_u32_sdwa", "trunc_f32", "cmp_lt_f32", "max_f32", "mul_lo_u32"]),
            "D": pick(["add_f64", "mul_f64", "fma_f64", "pk_mul_f32", "pk_add_f32", "lshlrev_b64"]),
            "T32": pick(["rcp_f32", "sqrt_f32", "exp_f32"]), "T64": pick(["rcp_f64", "rsq_f64", "sqrt_f64"])}


def class_costs(ubench_path, waves):
    """-> {class: cycles per wave64 instruction on one SIMD} at the kernel's MEASURED average of `waves` resident waves
    per SIMD: the two neighbouring columns of the micro-benchmark (1..4 waves) mixed in the proportion that gives that
    average — 2.91 waves = 91 % of the time three waves (plain fp32 2.80 cycles), 9 % two (2.08). Round 4 took the
    rounded column, i.e. the worst one for a three-wave kernel."""
    w = min(max(float(waves), 1.0), 4.0)
    lo = int(w) if w < 4 else 3
    f = w - lo
    a, b = _column(ubench_path, lo), _column(ubench_path, lo + 1)
    out = {k: (1 - f) * a[k] + f * b[k] for k in a}
    out["waves_per_simd"] = round(w, 3)
    out["waves_columns_mixed"] = [lo, lo + 1, round(1 - f, 3), round(f, 3)]
    return out


_asm = {}


def device_asm():
    """Device assembly of the library with the flags the product is built with, cold paths of the cost kernels
    compiled out (DERP_MIX_HOT_ONLY: the tap-by-tap SSD fallback, the non-FTHETA camera types)."""
    if "text" not in _asm:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g

        flags = [f for f in g.HIP_FLAGS if f not in ("-shared", "-fPIC")]
        out = "/tmp/derp_device_hot.s"
        subprocess.check_call([g.HIPCC] + flags + ["-DDERP_MIX_HOT_ONLY=1", "-S", "--cuda-device-only", "-o", out,
                                                   os.path.join(g.CSRC, "derp_capi.hip")], stderr=subprocess.DEVNULL)
        _asm["text"] = open(out).read()
    return _asm["text"]


def static_mix(kernel, trips=8.0):
    """Loop-weighted static VALU mix of the first kernel whose mangled symbol contains `kernel`: an instruction in a
    block the compiler marks "in Loop: ... Depth=d" weighs trips^d (the cost kernels run ~6.5 sources per
    candidate and 8-9 candidates per pixel; the filters walk windows of >= 9 taps)."""
    text = device_asm()
    # (k_ping_pong must not find k_ping_pong_w3 or k_ping_pong_commit: the name is followed by the mangled parameter list, "E...")
    m = re.search(r"^(_ZN4derp\d+%sE\w*):" % re.escape(kernel.rstrip("(<")), text, re.M) or \
        re.search(r"^(_Z\w*%s\w*):" % re.escape(kernel.rstrip("(<")), text, re.M)
    if not m:
        return Counter()
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    ops = Counter()
    depth = 0
    for ln in body.splitlines():
        if re.match(r"^(\.LBB|; %bb\.)", ln):
            d = re.search(r"Depth=(\d+)", ln)
            depth = int(d.group(1)) if d else 0
            continue
        mm = re.match(r"^\s+(v_[a-z0-9_]+)", ln)
        if mm:
            ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", mm.group(1))] += trips ** depth
    return ops


def issue_cycles(counts, mix, costs):
    """counts: {typed counter: dynamic count, "VALU": total}; mix: static mnemonic Counter; costs: class_costs().
    -> dict(cycles_upper, cycles_lower, by_class {class: dynamic instruction count}, ...)"""
    typed = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32",
             "INT64"]
    dyn = {k: float(counts.get(k, 0.0)) for k in typed}
    dyn["OTHER"] = max(float(counts.get("VALU", 0.0)) - sum(dyn.values()), 0.0)
    # static class shares inside each counter
    share = {}
    for op, n in mix.items():
        share.setdefault(counter_of(op), Counter())[klass(op)] += n
    by_class = Counter()
    for ctr, n in dyn.items():
        sh = share.get(ctr)
        if not sh:
            sh = Counter({"T32": 1} if ctr == "TRANS_F32" else {"T64": 1} if ctr == "TRANS_F64" else
                         {"D": 1} if ctr.endswith("F64") or ctr == "INT64" else {"F": 1})
        tot = float(sum(sh.values()))
        for c, k in sh.items():
            by_class[c] += n * k / tot
    upper = sum(by_class[c] * costs[c] for c in by_class)
    hidden = min(by_class["S"], by_class["F"])
    lower = upper - hidden * costs["S"]
    # The split between S and F INSIDE a typed counter (and inside "other") is the static-mix heuristic above, not a
    # measurement: how much the result depends on it = the same pricing with 10 % of the S + F instructions moved from one
    # class to the other.
    sf = by_class["S"] + by_class["F"]
    shift = 0.1 * sf * (costs["F"] - costs["S"])
    sens = {"shift": "10 % of the S + F instructions re-classed", "cycles_upper_if_more_F": upper + shift,
            "cycles_upper_if_more_S": upper - shift, "relative": round(shift / upper, 4) if upper else None}
    return {"cycles_upper": upper, "cycles_lower": lower, "by_class": {c: by_class[c] for c in sorted(by_class)},
            "class_split_sensitivity": sens,
            "dynamic_by_counter": dyn,
            "static_class_share_by_counter": {c: {k: round(v / float(sum(sh.values())), 3) for k, v in sh.items()}
                                              for c, sh in sorted(share.items())}}


# The reference's arithmetic per (cost call, source) pair, for the ALGORITHMIC numerator (wasted instructions must not
# count as achievement): computeSSD (DerpUtil.cpp:126-162) = 27 + 3 truncating bilerps of 4 mul + 3 add (CvUtil.h:83-103),
# 27 x (db, dn, two squares, two accumulations) -> 30 x 7 + 27 x 6 = 372 plain fp32 operations, 30 float -> ushort
# truncations and 4 x (27 + 3) ushort -> float conversions of distinct taps = 16 x 3 + 4 x 3 = 60 after sharing;
# worldToSrcPoint / Camera::sees (DerpUtil.cpp:56-73, Camera.h:184-190,301-341) = 87 fp64 operations per projected
# source as compiled here (3 of them transcendental: one sqrt, two divisions' reciprocals).
ALG_PER_PAIR = {"S": 372.0, "F": 90.0, "D": 84.0, "T64": 3.0}


def algorithmic_cycles(n_pair_executed, costs):
    return n_pair_executed * sum(ALG_PER_PAIR[c] * costs[c] for c in ALG_PER_PAIR)


if __name__ == "__main__":
    k = sys.argv[1] if len(sys.argv) > 1 else "k_ping_pong"
    mix = static_mix(k)
    tot = float(sum(mix.values()))
    print(k, "loop-weighted static VALU mix (shares):")
    by = Counter()
    for op, n in mix.items():
        by[(counter_of(op), klass(op))] += n
    for key, n in sorted(by.items()):
        print("  %-10s %-3s %6.3f" % (key[0], key[1], n / tot))
