"""The executables' input parsers (cli_common.h: rig JSON, OpenEXR, PFM) on files that are damaged the way a fuzzer
damages them (the cases below are what 300 000 random mutations under AddressSanitizer found or came close to): every
one must end like any other input error of the executables — a "Check failed" line and exit status 1 — not with a
read past the buffer, a stack overflow or an out-of-memory abort."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("parsers") / "parser_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", exe, os.path.join(ROOT, "tests", "native", "parser_main.cpp"), "-lz", "-ldl"])
    return exe


def run(exe, kind, path):
    p = subprocess.run([exe, kind, path], capture_output=True, text=True, timeout=60)
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-1500:]
    return p


def test_rig_json(harness, tmp_path):
    from facebook360_dep_amd import synth

    good = json.dumps(synth.make_rig(4, 96), indent=1)
    path = str(tmp_path / "rig.json")
    open(path, "w").write(good)
    p = run(harness, "json", path)
    assert p.returncode == 0 and p.stdout.split() == ["ok", "4"]
    cut = good.index('"id"') + 8
    for text in (good[:cut],                      # ends inside a string
                 good[:cut] + "\\",               # ... on a backslash
                 good[:cut] + "\\u12",            # ... inside a \u escape
                 good[: len(good) // 2],          # ends inside an array
                 "[" * 100000,                    # recursion depth
                 '{"cameras": [{"id": }]}', '{"cameras": [1, 2,, 3]}', "", "{", '{"a"', '{"a":', "nul", "-", "1e999999"):
        open(path, "w").write(text)
        p = run(harness, "json", path)
        assert p.returncode == 1 and ("Check failed" in p.stderr or "parse error" in p.stderr or "missing key" in p.stderr), (text[:40], p.stderr[-300:])


def test_exr_and_pfm(harness, tmp_path):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_exr import _py_exr

    img = np.random.default_rng(0).normal(0, 1, (37, 29)).astype(np.float32)
    path = str(tmp_path / "a.exr")
    _py_exr(path, img, 2)
    assert run(harness, "exr", path).stdout.split() == ["ok", "29", "37"]
    good = open(path, "rb").read()
    at = good.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    for window in ((0, 0, 2 ** 31 - 1, 2 ** 31 - 1), (0, 0, 900000, 900000), (-2 ** 31, -2 ** 31, 2 ** 31 - 1, 5), (0, 0, 28, 10 ** 6)):
        bad = bytearray(good)
        bad[at: at + 16] = struct.pack("<iiii", *window)
        open(path, "wb").write(bytes(bad))
        p = run(harness, "exr", path)
        assert p.returncode == 1 and "OpenEXR" in p.stderr, (window, p.stderr[-300:])
    for cut in (len(good) - 50, 400, 60, 9):
        open(path, "wb").write(good[:cut])
        assert run(harness, "exr", path).returncode == 1
    path = str(tmp_path / "a.pfm")
    for header in (b"Pf\n-3 5\n-1.0\n", b"Pf\n70000 70000\n-1.0\n", b"Pf\n5 5\n1.0\n", b"Pf\n5\n", b"PF\n5 5\n-1.0\n", b"Pf\n5 5\n-1.0\n" + b"\0" * 50):
        open(path, "wb").write(header + b"\0" * 16)
        p = run(harness, "pfm", path)
        assert p.returncode == 1 and "pfm" in p.stderr, (header, p.stderr[-300:])
    open(path, "wb").write(b"Pf\n5 5\n-1.0\n" + b"\0" * 100)
    assert run(harness, "pfm", path).stdout.split() == ["ok", "5", "5"]


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus N` without a launcher starts its own N ranks; with fewer than N devices (none in the CPU
    container) it must exit non-zero and print no JSON line — a `--gpus 8` request must never yield an `n_gpus: 1`
    line. A launcher whose WORLD_SIZE disagrees with --gpus is refused the same way."""
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DERP_BENCH_SINGLE_DEVICE")}
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "tiny"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode != 0 and "refusing" in p.stderr and "{" not in p.stdout
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "tiny"],
                       capture_output=True, text=True, timeout=300, env=dict(env, RANK="0", WORLD_SIZE="1"))
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr and "{" not in p.stdout
