#!/usr/bin/env python
"""bench.py — depth Mpix/s (full pyramid, 16-camera 2048^2 rig): BASELINE.json's metric.

Workload (the same for every --gpus N, so 1/2/4/8 is ONE scaling curve — strong scaling): BASELINE
config 3, an 8-frame sequence of the 16-camera 2048^2 synthetic rig with the per-level temporal filter
(scripts/render/pipeline.py:364-408). A step = the whole sequence, coarse to fine: for every level, every
frame's processLevel (projection warps built once per level, colour tables per frame, colour reprojection, brute force, random
proposals, ping-pong, bilateral, median, FOV mask, upsample hand-off), the exchange of the halo frames' raw
level disparity between ranks, the temporal filter and the write-back. Frames are sharded over the ranks in
contiguous chunks (8/N frames per GPU); the exchange is RCCL send/recv issued by the library on its own
stream (fallbacks: torch.distributed point-to-point, then broadcast). Inputs are resident in HBM when the
timed region starts.

At N = 1 the same run also times BASELINE config 2 (one frame, no temporal filter) and reports it as
`config2_single_frame`, next to `roofline` (dominant kernel = level-0 ping-pong; the kernel is bound by
VALU issue, not HBM — see DESIGN.md §6) and `cpu_baseline` (the CPU oracle on the host cores).

Prints ONE JSON line on rank 0 (driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2.4 GHz peak engine clock; HBM3E 8.0 TB/s spec
HBM_PEAK_GBS = 8000.0
N_SIMD = 256 * 4
PEAK_CLOCK_GHZ = 2.4
VALU_PEAK_GCYC = N_SIMD * PEAK_CLOCK_GHZ  # SIMD-cycles available per second (x 1e9)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", help="rig: cfg1 | cfg2 | cfg4 | small | tiny (default: BASELINE config 2/3's rig)")
    ap.add_argument("--frames", type=int, default=8, help="frames of the sequence (BASELINE config 3: 8)")
    ap.add_argument("--temporal", type=int, default=1, help="0 = no temporal filter (frames are then independent replicas)")
    ap.add_argument("--partition", default="block", choices=["block", "cyclic"])
    ap.add_argument("--synth-device", default="cuda",
                    help="where the synthetic frames are rendered: cuda (default, fast) | cpu (bit-identical to the frames "
                         "the CPU tests render, so result_crc can be compared with the oracle's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "full", "sample"],
                    help="the CPU oracle on frame 0 of the workload: full = every level measured (about 2.5 minutes for "
                         "16 x 2048^2 on 16 CPUs), sample = levels 9..2 measured and levels 1-0 extrapolated by pixel "
                         "count (about 10 s), auto = full when this process may use >= 16 CPUs")
    ap.add_argument("--no-single-frame", action="store_true", help="skip the config-2 single-frame leg at N = 1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--exchange", default="rccl,torch,broadcast",
                    help="transports to try for the halo exchange, in order")
    ap.add_argument("--cache-warp-tables", type=int, default=0,
                    help="1 = keep the rig-only projection warps across steps too (default 0: every step rebuilds them "
                         "once per level; the frames of a level on one rank share them)")
    return ap.parse_args()


def kernel_sources_sha256():
    """Same definition as tools/make_profiles.py: the sources the device code is built from."""
    import hashlib

    h = hashlib.sha256()
    for name in ("derp_kernels.h", "derp_capi.hip", "derp_camera.h", "gcc_algos.h", "derp_sequence.h"):
        with open(os.path.join(ROOT, "facebook360_dep_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def usable_cpus():
    """CPUs this process may actually burn: affinity mask and cgroup quota (a container that sees 256 hardware
    threads may be allowed 16 CPUs' worth of time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def b_alg(n_cost, n_pair):
    """logical gather bytes of the cost loop."""
    return 64.0 * n_cost + 272.0 * n_pair


def cpu_baseline(rig, sizes, frame, res, n_cams, mode="auto"):
    """the CPU oracle ("port") on this host's cores. Config 1 in full; frame 0 of the bench rig either
    in full (every level measured: `extrapolated` false) or, on a host with few CPUs, with levels 9..2 measured and
    levels 1-0 extrapolated from level 2's time per pixel. Returns the cpu_baseline object."""
    import numpy as np  # noqa: F401

    from facebook360_dep_amd import synth
    from tests import common

    threads = os.cpu_count() or 1
    cores = usable_cpus()
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"unit": "Mpix/s", "cores": cores, "kind": "port", "cpu_model": model,
           "threads": threads,
           "cores_note": "%d worker threads (one per visible hardware thread) on %d usable CPUs (scheduler affinity and "
                         "cgroup CPU quota of this container)" % (threads, cores)}
    # --- config 1 in full (4 x 512^2, 8 levels)
    n1, r1, w1 = synth.config("cfg1")
    rig1 = synth.make_rig(n1, r1)
    sizes1 = synth.level_sizes(r1, r1, w1)
    frame1 = synth.make_frame(rig1, sizes1)
    t0 = time.time()
    common.oracle_pyramid(rig1, sizes1, frame1, r1, r1, partial_coverage=True, threads=-1)
    t1 = time.time() - t0
    out["config1_full"] = {"value": round(n1 * r1 * r1 / t1 / 1e6, 4), "seconds": round(t1, 2),
                           "workload": "BASELINE config 1 in full: 4 x 512^2, %d levels" % len(sizes1)}
    # --- the bench rig: coarse levels measured, the finest extrapolated
    full = mode == "full" or (mode == "auto" and cores >= 16)
    first_measured = 0 if full else next((lv for lv, (w, h) in enumerate(sizes) if w <= 512), len(sizes) - 1)
    t_levels = {}
    prev = None
    for level in range(len(sizes) - 1, first_measured - 1, -1):
        t0 = time.time()
        L = common.oracle_level(rig, sizes, frame, level, res, res, prev, partial_coverage=int(n_cams <= 4), threads=-1)
        L.process()
        prev = [L.get_dst(d)[0] for d in range(L.D)]
        t_levels[level] = time.time() - t0
    measured = sum(t_levels.values())
    w, h = sizes[first_measured]
    per_px = t_levels[first_measured] / (w * h)
    extra = sum(per_px * sizes[lv][0] * sizes[lv][1] for lv in range(first_measured))
    total = measured + extra
    w0, h0 = sizes[0]
    out["value"] = round(n_cams * w0 * h0 / total / 1e6, 4)
    out["sample"] = ("frame 0 of the bench workload (%d cameras, %dx%d): levels %d..%d measured in %.1f s on %d CPUs; "
                     "levels %d..0 EXTRAPOLATED from level %d's time per pixel (+%.1f s) -> %.1f s per frame. "
                     "Not the timed workload itself: one frame, no temporal filter."
                     % (n_cams, w0, h0, len(sizes) - 1, first_measured, measured, cores, first_measured - 1,
                        first_measured, extra, total)) if first_measured > 0 else (
        "frame 0 of the bench workload (%d cameras, %dx%d) in full, all %d levels measured in this run: %.1f s on %d CPUs "
        "(%d threads). Not the timed workload itself: one frame, no temporal filter." % (n_cams, w0, h0, len(sizes), measured,
                                                                                        cores, threads))
    out["extrapolated"] = first_measured > 0
    out["measured_seconds"] = round(measured, 2)
    out["level_seconds"] = {str(lv): round(t, 3) for lv, t in sorted(t_levels.items())}
    # the same frame run in full in an earlier round (tools/oracle_full_frame.py, committed under profiles/): quoted
    # beside an extrapolated value so that the two can be compared
    once = os.path.join(ROOT, "profiles", "oracle_full_frame.json")
    if not full and os.path.exists(once):
        try:
            with open(once) as f:
                out["full_frame_measured_once"] = json.load(f)
        except Exception:  # noqa: BLE001
            pass
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (no launcher: RANK / WORLD_SIZE unset) with N > 1: start the N ranks
    here, one process per GPU — the environment `python -m torch.distributed.run --nproc-per-node N` would give them —
    and let rank 0 print the one JSON line. Refuses (non-zero exit, nothing on stdout) when the node has fewer than N
    devices: a line that says n_gpus 1 for a --gpus 8 request would be taken for a scaling measurement."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()
    single = bool(os.environ.get("DERP_BENCH_SINGLE_DEVICE"))  # developer check of the N > 1 path on a 1-GPU box
    if have < args.gpus and not single:
        raise SystemExit("bench.py: --gpus %d asked for, %d visible device(s): refusing to measure fewer GPUs than "
                         "requested" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = []
    try:
        # a rank that dies leaves the others in a collective: take the whole job down with it
        while len(rcs) < len(procs):
            rcs = [p.poll() for p in procs]
            if any(rc not in (None, 0) for rc in rcs):
                break
            rcs = [rc for rc in rcs if rc is not None]
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None and any(q.poll() not in (None, 0) for q in procs):
                p.kill()
        rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %s" % ", ".join("rank %d rc %d" % b for b in bad))


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL peer-memory handles need it here
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return launch_ranks(args)
    # stdout carries the one JSON line and nothing else: whatever a library prints there (Gloo's "Rank 0 is connected"
    # banner, a runtime warning) goes to stderr — file descriptor 1 is pointed at 2 for the run, the line is written to
    # the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np  # noqa: F401
    import torch

    from facebook360_dep_amd import derp, sequence, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:  # never print a line whose n_gpus is not what --gpus asked for
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    if os.environ.get("DERP_BENCH_SINGLE_DEVICE"):  # developer check of the N>1 path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    temporal = bool(args.temporal)

    n_cams, res, widths = synth.config(args.config)
    rig = synth.make_rig(n_cams, res)
    sizes = synth.level_sizes(res, res, widths)
    n_levels = len(sizes)
    first, last = 0, args.frames - 1
    partition = sequence.BLOCK if args.partition == "block" else sequence.CYCLIC

    g = derp.Derp(rig["cameras"], device=local_rank, partial_coverage=int(n_cams <= 4),
                  rebuild_warp_tables=int(not args.cache_warp_tables))
    g.set_pyramid(sizes, res, res)
    runner = sequence.SequenceRunner(g, first, last, rank, world, do_temporal_filter=int(temporal), partition=partition)
    upload_s, frame0 = 0.0, None
    for t in runner.owned:  # every rank renders and uploads only the frames it owns
        frame = synth.make_frame(rig, sizes, frame=t, seed=360 + t, device=args.synth_device)
        t0 = time.time()
        runner.upload_frame(t, frame)
        upload_s += time.time() - t0
        if t == 0:
            frame0 = frame
    upload_bytes = sum(w * h for (w, h) in sizes) * n_cams * 6 * len(runner.owned)

    transport = "local"
    if world > 1 and temporal:
        transport = runner.attach_best(dist, prefer=tuple(args.exchange.split(",")),
                                       log=lambda m: print("bench: " + m, file=sys.stderr))
        runner.exchange_inputs()  # colour guides of the halo frames: inputs, fetched once, outside the timed region
        g.synchronize()
        torch.cuda.synchronize()

    def step():
        runner.run()

    def fence():
        g.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    g.profile_reset()
    g.profile_enable(True)
    runner.stats_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    g.profile_enable(False)

    w0, h0 = sizes[0]
    total_mpix = args.frames * args.steps * n_cams * w0 * h0 / 1e6  # every frame of the sequence, all ranks
    value = total_mpix / dt

    # ---- per-rank measurements of the timed region
    pp = g.profile_query("ping_pong", 0)
    launches = max(pp["launches"], 1)
    kernel_ms = pp["ms"] / launches
    memo = g.profile_memoised("ping_pong", 0) / launches
    # Levels whose frames ran on overlapping work lanes (derp_seq_level_compute at the coarse levels): their per-stage spans
    # overlap in time, so those levels are reported by their wall on the context's stream ("lanes_wall") — in
    # level_ms_per_frame and as stage "coarse_levels_on_lanes" — and left out of the per-stage sums, which therefore still
    # add up to the step (the temporal stage of every level is exclusive and stays where it was).
    lanes_wall = [g.profile_query("lanes_wall", lv)["ms"] for lv in range(n_levels)]
    laned = [lv for lv in range(n_levels) if lanes_wall[lv] > 0]
    stage_ms = {s: round(sum(g.profile_query(s, lv)["ms"] for lv in range(n_levels)
                             if lv not in laned or s == "temporal") / args.steps, 3) for s in derp.STAGES}
    stage_ms["coarse_levels_on_lanes"] = round(sum(lanes_wall) / args.steps, 3)
    level_ms = [round(((lanes_wall[lv] + g.profile_query("temporal", lv)["ms"]) if lv in laned else
                       sum(g.profile_query(s, lv)["ms"] for s in derp.STAGES)) / args.steps / max(len(runner.owned), 1), 3)
                for lv in range(n_levels)]
    cnt = g.counters()
    # ---- what was computed: CRC-32 of every frame's level-0 (filtered) disparity, gathered to rank 0, so that an
    # N-GPU line can be checked against the 1-GPU line (tests/golden/bench_result_crc.json holds the N = 1 values)
    crc = {str(t): "%08x" % v for t, v in runner.result_crc().items()}
    transports = [transport]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (crc, transport))
        crc = {k: v for (c, _) in gathered for k, v in c.items()}
        transports = [t for (_, t) in gathered]
    crc = {k: crc[k] for k in sorted(crc, key=int)}
    # the N = 1 values of the default workload are committed (tests/golden/bench_result_crc.json, recorded on the
    # MI355X and re-asserted by test_config3_full_size_two_ranks_equal_one): say in the line itself whether this
    # run — whatever its --gpus N — computed those depth maps. null = no committed values for this workload.
    crc_matches = None
    if args.config == "cfg2" and args.frames == 8 and temporal and args.synth_device == "cuda":
        try:
            with open(os.path.join(ROOT, "tests", "golden", "bench_result_crc.json")) as f:
                crc_matches = (crc == json.load(f)["cfg2_8"])
        except Exception:  # noqa: BLE001
            crc_matches = None
    xs = runner.stats()
    exch = {"bytes_received_per_step": xs["bytes_received"] // max(args.steps, 1),
            "bytes_sent_per_step": xs["bytes_sent"] // max(args.steps, 1),
            "ms_per_step_on_stream": round(xs["exchange_ms"] / max(args.steps, 1), 3),
            # of which the compute stream stood still for (the rest ran beside the filter of the frames whose windows are
            # local to the rank); hidden_frac = 1 - exposed / on_stream
            "ms_per_step_exposed": round(xs["exchange_exposed_ms"] / max(args.steps, 1), 3),
            "hidden_frac": (round(1.0 - xs["exchange_exposed_ms"] / xs["exchange_ms"], 3) if xs["exchange_ms"] > 0 else None)}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, exch)
        exch = {"per_rank": gathered,
                "bytes_received_per_step": sum(e["bytes_received_per_step"] for e in gathered)}

    # ---- roofline of the dominant kernel: level-0 ping-pong (one launch per frame per step)
    prof, stale = {}, None
    ppath = os.path.join(ROOT, "profiles", "valu_roofline.json")
    if os.path.exists(ppath):
        try:
            with open(ppath) as f:
                prof = json.load(f).get(args.config, {})
        except Exception:  # noqa: BLE001
            prof = {}
    # the counter numerators were collected on ONE version of the kernels: with any other, cycles and time would
    # come from different programs, so the counter-derived fields are withheld (tools/profile_round.sh re-collects)
    here = kernel_sources_sha256()
    if prof and os.environ.get("DERP_LIB"):  # a developer's variant library (tools/variants.sh): not the profiled binary
        stale = "DERP_LIB selects another library than the one the counters were collected on: counter-derived fields withheld"
        prof = {}
    if prof and prof.get("kernel_sources_sha256") != here:
        stale = ("profiles/valu_roofline.json was collected on kernel sources %s..., this tree is %s...: counter-derived "
                 "fields withheld" % (str(prof.get("kernel_sources_sha256"))[:12], here[:12]))
        prof = {}
    def kernel_roofline(name, stage, key, memo_evals):
        """SURVEY 8(d) for one cost kernel's level-0 launch. `frac` is ACHIEVEMENT: the VALU issue cycles the reference's
        own arithmetic needs for the (cost call, source) pairs the launch executed / the cycles the chip had during the
        launch. `issue_frac` is ACTIVITY: the instruction stream the kernel really issued, priced the same way (low =
        every plain fp32 / move / logic instruction co-issues beside a 4-cycle instruction of another wave, high = none
        does). `hbm_frac` = bytes through the memory-side counters / launch time / 8 TB/s."""
        q = g.profile_query(stage, 0)
        n_l = max(q["launches"], 1)
        ms = q["ms"] / n_l
        sec = ms * 1e-3
        n_cost_l, n_pair_l = q["n_cost"] / n_l, q["n_pair"] / n_l
        exec_frac = 1.0 - memo_evals / n_cost_l if n_cost_l else 1.0  # evaluations served from the memo execute no pairs
        cc = prof.get(key + "_level0_class_cycles")
        hi, lo = prof.get(key + "_level0_issue_cycles_per_launch"), prof.get(key + "_level0_issue_cycles_if_simple_ops_coissue")
        traffic = prof.get(key + "_level0_hbm_bytes_per_launch")
        out = {"kernel": "%s @ level 0" % name, "bound": "valu", "unit": "G SIMD issue-cycles/s", "peak": VALU_PEAK_GCYC,
               "achieved": None, "frac": None, "traffic": traffic, "kernel_ms": round(ms, 3), "launches_timed": q["launches"]}
        if cc and sec > 0:
            per_pair = 372.0 * cc["S"] + 90.0 * cc["F"] + 84.0 * cc["D"] + 3.0 * cc["T64"]  # tools/valu_model.py ALG_PER_PAIR
            ach = n_pair_l * exec_frac * per_pair / 64.0 / sec / 1e9
            out["achieved"] = round(ach, 1)
            out["frac"] = round(ach / VALU_PEAK_GCYC, 4)
            out["algorithmic_cycles_per_pair"] = round(per_pair, 1)
            out["class_cycles"] = {k: round(v, 3) for k, v in cc.items() if isinstance(v, float)}
            out["waves_per_simd"] = prof.get(key + "_level0_waves_per_simd")
        if hi and lo and sec > 0:
            out["issue_frac"] = {"low": round(lo / sec / 1e9 / VALU_PEAK_GCYC, 4), "high": round(hi / sec / 1e9 / VALU_PEAK_GCYC, 4)}
        if traffic and sec > 0:
            out["hbm_traffic_GBps"] = round(traffic / sec / 1e9, 1)
            out["hbm_frac"] = round(traffic / sec / 1e9 / HBM_PEAK_GBS, 4)
        if prof.get(key + "_level0_l2"):
            out["l2"] = prof[key + "_level0_l2"]
        out["wave_issue_breakdown"] = prof.get(key + "_level0_wave_cycle_shares")
        alg_bytes = b_alg(n_cost_l, n_pair_l) * exec_frac
        out["logical_gathers"] = {  # SURVEY 8(d)'s B_alg on executed evaluations: L1 / L2 serve most of it, so this is no bound
            "bytes_per_launch_executed": alg_bytes, "n_cost_per_launch": n_cost_l, "n_pair_per_launch": n_pair_l,
            "memoised_cost_evals_per_launch": memo_evals,
            "rate_over_hbm_peak_NOT_A_BOUND": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 4) if sec > 0 else None}
        return out

    roofline = kernel_roofline("k_ping_pong", "ping_pong", "ping_pong", memo)
    roofline["note"] = (
        "frac = (executed (cost call, source) pairs of one level-0 launch) x (VALU issue cycles the reference's arithmetic "
        "needs per pair: 372 plain fp32, 90 conversions / truncations, 84 fp64, 3 fp64 transcendental — tools/valu_model.py "
        "ALG_PER_PAIR — priced per class at the intervals tools/valu_ubench.hip measured on this chip, at the kernel's "
        "measured waves per SIMD) / 64 lanes / this run's HIP-event launch time / (%d SIMDs x %.1f GHz). issue_frac prices the "
        "instructions the kernel really issued (rocprofv3 typed VALU counters) the same way: [all simple ops co-issue, none "
        "does]. The kernel gathers from L1 / L2-resident tables: HBM is not its bound (hbm_frac)." % (N_SIMD, PEAK_CLOCK_GHZ))
    roofline["stale"] = stale
    roofline["random_proposals"] = kernel_roofline("k_random_proposals", "random_proposals", "random", 0.0)
    roofline["whole_step_algorithmic_GBps"] = round(b_alg(cnt["n_cost"], cnt["n_pair"]) / dt / 1e9, 1)
    # measured HBM bytes of a whole step (all kernels; the committed PMC passes ran the default
    # sequence on one GPU, so only quoted for that shape) beside the compulsory floor: every source level read
    # once per destination + disparity / masks read and written once
    whole_fetch, whole_write = prof.get("whole_step_hbm_fetch_bytes"), prof.get("whole_step_hbm_write_bytes")
    px_all = sum(w * h for (w, h) in sizes)
    floor_bytes = args.frames * px_all * (n_cams * n_cams * 6 + n_cams * (4 + 4 + 1 + 1))
    if whole_fetch and world == 1 and args.frames == 8:
        step_s = dt / args.steps
        roofline["whole_step_hbm"] = {
            "fetch_bytes": whole_fetch, "write_bytes": whole_write,
            "GBps": round((whole_fetch + whole_write) / step_s / 1e9, 1),
            "frac_of_peak": round((whole_fetch + whole_write) / step_s / 1e9 / HBM_PEAK_GBS, 4),
            "compulsory_floor_bytes": floor_bytes}

    frames_here = len(runner.owned)
    out = {
        "metric": "depth Mpix/s per GPU (full pyramid, 16-cam 2048^2 rig); % HBM-read roofline",
        "value": round(value, 3),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32+f64",  # photometry in f32 over u16 texels; camera geometry in f64
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE config %s: %d-camera %dx%d synthetic rig, %d-frame sequence, full %d-level pyramid per "
                         "frame, %s; the same sequence for every --gpus N (%s partition, %d frame(s) on this rank)"
                         % ({"cfg2": "3" if args.frames > 1 else "2", "cfg4": "4", "cfg1": "1"}.get(args.config, args.config),
                            n_cams, res, res, args.frames, n_levels,
                            "per-level temporal filter (+-2 frames)" if temporal else "no temporal filter",
                            args.partition, frames_here)),
            "name": "cfg3" if (args.config == "cfg2" and args.frames == 8 and temporal) else args.config,
            "rig": args.config,
            "cameras": n_cams,
            "resolution": [res, res],
            "levels": [list(s) for s in sizes],
            "frames": args.frames,
            "temporal_filter": temporal,
            "halo_transport": transport,
            "halo_exchange": exch,
            "warp_tables": ("cached across steps" if args.cache_warp_tables else
                            "rebuilt once per level per step, shared by the frames of the level on a rank (rig-only data)"),
            "parallelism": "frames x%d" % world,
        },
        "per_gpu_value": round(value / world, 3),
        # SURVEY 8(d), for completeness: every pyramid level's pixels, not only the finest level's
        "pyramid_value": round(value * sum(w * h for (w, h) in sizes) / (w0 * h0), 3),
        "ms_per_frame": round(dt / args.steps / args.frames * 1e3 * world, 3),
        "roofline": roofline,
        "stage_ms_per_step": stage_ms,
        "level_ms_per_frame": level_ms,  # all stages of one level of one frame (this rank), finest level first
        "levels_on_work_lanes": laned,   # their frames overlap: level time = wall / frames, stages under "coarse_levels_on_lanes"
        # CRC-32 (zlib) of each frame's level-0 disparity after the last step, all destinations in rig order, raw
        # float32 bytes: identical for every --gpus N and --partition if the sharded run computed the same depth maps
        "result_crc": crc,
        "result_crc_matches_n1": crc_matches,
        "halo_transport_per_rank": transports,
        "input_upload": {"bytes": upload_bytes, "seconds": round(upload_s, 3),
                         "note": "host->HBM staging of this rank's colour pyramids, outside the timed region"},
        "device": g.device_name(),
    }

    if world == 1 and not args.no_single_frame:
        # BASELINE config 2: one frame (slot 0), no temporal filter — DerpCLI's own level loop
        g.select_frame(0)
        g.process_pyramid()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.process_pyramid()
        fence()
        d2 = time.perf_counter() - t0
        out["config2_single_frame"] = {
            "value": round(args.steps * n_cams * w0 * h0 / 1e6 / d2, 3), "unit": "Mpix/s",
            "ms_per_frame": round(d2 / args.steps * 1e3, 3),
            "workload": "BASELINE config 2: %d-camera %dx%d rig, single frame, full %d-level pyramid, no temporal filter"
                        % (n_cams, res, res, n_levels)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and frame0 is not None:
        out["cpu_baseline"] = cpu_baseline(rig, sizes, frame0, res, n_cams, args.cpu_baseline)
    runner.close()
    g.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
