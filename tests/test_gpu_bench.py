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


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_small(built):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "small", "--frames", "4", "--steps", "1",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = _line(p)
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and out["unit"] == "Mpix/s" and out["value"] > 0
    assert out["config"]["frames"] == 4 and out["config"]["temporal_filter"] is True
    assert out["roofline"]["bound"] == "valu" and out["roofline"]["kernel_ms"] > 0
    assert out["config2_single_frame"]["value"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["config1_full"]["value"] > 0
    assert out["stage_ms_per_step"]["temporal"] > 0


def test_bench_two_ranks_one_gpu(built):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DERP_BENCH_SINGLE_DEVICE="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "1", "--warmup", "1", "--config", "small", "--frames", "4",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    out = _line(p)
    assert out["n_gpus"] == 2 and out["config"]["frames"] == 4
    assert out["config"]["halo_transport"] in ("torch", "broadcast", "rccl")
    # 4 frames on 2 ranks, radius 2: each rank receives the two frames across the chunk boundary at every level
    n, res = 6, 160
    px = sum(w * h for (w, h) in out["config"]["levels"])
    assert out["config"]["halo_exchange"]["bytes_received_per_step"] == 4 * px * n * 4
