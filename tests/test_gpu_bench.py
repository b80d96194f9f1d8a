"""bench.py end to end on small configurations: the single-process line, and the N = 2 launch exactly as the
driver does it (torch.distributed.run, one rank per process) with both ranks on cuda:0 — RCCL refuses two
ranks on one device, so this exercises the transport fall-back chain (rccl -> torch p2p -> broadcast) over
gloo, the halo exchange and the max-over-ranks timing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_oracle_crc = {}


def _oracle_crc_small(frames=4):
    """result_crc of bench.py's `small` workload as the CPU oracle computes it on THIS host (the synthetic frames
    are rendered with torch CPU kernels and BLAS dot products, whose last bits depend on the host CPU, so the
    values are computed here rather than committed; the full-size workload's CRCs, rendered on the GPU, are
    committed in tests/golden/bench_result_crc.json and checked by test_config3_full_size_two_ranks_equal_one)."""
    if frames not in _oracle_crc:
        from facebook360_dep_amd import sequence, synth
        from tests import common

        n, res, widths = synth.config("small")
        rig = synth.make_rig(n, res)
        sizes = synth.level_sizes(res, res, widths)
        seq = common.OracleSequence(rig, sizes, res, 0, frames - 1, threads=-1, partial_coverage=False)
        sequence.run_schedule(seq, list(range(len(sizes) - 1, -1, -1)), 0, frames - 1, 0, 1)
        _oracle_crc[frames] = {str(t): "%08x" % v for t, v in seq.result_crc().items()}
    return _oracle_crc[frames]


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_small(built):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "small", "--frames", "4", "--steps", "1",
                        "--warmup", "1", "--synth-device", "cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = _line(p)
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["unit"] == "Mpix/s" and out["value"] > 0
    assert out["config"]["frames"] == 4 and out["config"]["temporal_filter"] is True
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["kernel_ms"] > 0
    assert out["config2_single_frame"]["value"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["config1_full"]["value"] > 0
    assert out["stage_ms_per_step"]["temporal"] > 0
    # stdout carries the line and nothing else (whatever a library prints goes to stderr)
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # the `small` rig's levels all run on work lanes: reported by their wall, and the stage times still add up to the step
    assert out["levels_on_work_lanes"] and out["stage_ms_per_step"]["coarse_levels_on_lanes"] > 0
    assert sum(out["stage_ms_per_step"].values()) <= 1.02 * out["ms_per_step"]
    assert out["config"]["halo_exchange"]["ms_per_step_exposed"] == 0.0  # one rank: nothing is exchanged
    # what was computed: the CPU oracle's CRCs for this workload
    assert out["result_crc"] == _oracle_crc_small(4) and out["halo_transport_per_rank"] == ["local"]


def test_bench_two_ranks_one_gpu(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DERP_BENCH_SINGLE_DEVICE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "small", "--frames", "4",
                        "--backend", "gloo", "--synth-device", "cpu"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    out = _line(p)
    assert out["n_gpus"] == 2 and out["config"]["frames"] == 4
    assert out["config"]["halo_transport"] in ("torch", "broadcast", "rccl")
    # 4 frames on 2 ranks, radius 2: each rank receives the two frames across the chunk boundary at every level
    n, res = 6, 160
    px = sum(w * h for (w, h) in out["config"]["levels"])
    assert out["config"]["halo_exchange"]["bytes_received_per_step"] == 4 * px * n * 4
    # the sharded run computed the 1-GPU run's (= the oracle's) depth maps, frame by frame
    assert out["result_crc"] == _oracle_crc_small(4)
    assert len(out["halo_transport_per_rank"]) == 2 and len(set(out["halo_transport_per_rank"])) == 1


def test_bench_plain_command_shards_itself_or_fails(built):
    """The driver's command is `python bench.py --gpus N` with no launcher around it: bench.py must start the N ranks
    itself (here: two ranks on the one GPU, DERP_BENCH_SINGLE_DEVICE) and print n_gpus = N — and on a box with fewer
    than N devices it must FAIL, never print a 1-GPU line for a --gpus N request."""
    import torch

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "small",
           "--frames", "4", "--backend", "gloo", "--synth-device", "cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, DERP_BENCH_SINGLE_DEVICE="1"))
    out = _line(p)
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "frames x2"
    assert out["result_crc"] == _oracle_crc_small(4)
    if torch.cuda.device_count() < 2:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
        assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout[-500:]
        assert "refusing" in p.stderr


@pytest.mark.parametrize("world,partition", [(2, 0), (4, 0), (4, 1), (8, 0)])
def test_result_crc_of_emulated_ranks_equals_one_rank(built, world, partition):
    """bench.py's workload shape (`small` rig, temporal filter) on `world` ranks emulated on one GPU: the
    per-frame CRCs every rank would print equal the CPU oracle's (4 frames) and, for the 8-frame sequence,
    the 1-rank run's."""
    from facebook360_dep_amd import derp, sequence, synth

    n, res, widths = synth.config("small")
    rig = synth.make_rig(n, res)
    sizes = synth.level_sizes(res, res, widths)

    def run(first, last, w):
        made = []
        for rank in range(w):
            g = derp.Derp(rig["cameras"])
            g.set_pyramid(sizes, res, res)
            r = sequence.SequenceRunner(g, first, last, rank, w, partition=partition)
            for t in r.owned:
                r.upload_frame(t, synth.make_frame(rig, sizes, frame=t, seed=360 + t, device="cpu"))
            made.append((g, r))
        if w == 1:
            made[0][1].run()
            made[0][0].synchronize()
        else:
            sequence.run_loopback([r for (_, r) in made], len(sizes) - 1)
        crc = {}
        for (g, r) in made:
            crc.update({str(t): "%08x" % v for t, v in r.result_crc().items()})
            g.close()
        return crc

    if world <= 4:
        assert run(0, 3, world) == _oracle_crc_small(4)
    assert run(0, 7, world) == run(0, 7, 1)
