"""cli_common.h's IoPool on the CPU: result files (class 1) and staging copies (class 0) must not wait behind the input
decodes (class 2) — not in the queue (priorities) and not for a worker either (two express workers of a large pool never
take class-2 jobs; a small pool has none and the urgent job waits for the first worker to come free)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("iopool") / "iopool_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "iopool_main.cpp"), "-lz", "-ldl"])
    return exe


def test_urgent_jobs_find_a_worker_while_the_pool_is_decoding(harness):
    lat1, lat0, ran = subprocess.check_output([harness, "12"], text=True, timeout=60).split()
    assert int(ran) == 48  # every class-2 job still ran (on the ten general workers)
    assert float(lat1) < 90 and float(lat0) < 90, (lat1, lat0)  # not the 180 ms the first general worker needs to come free


def test_small_pools_keep_every_worker_general(harness):
    lat1, lat0, ran = subprocess.check_output([harness, "4"], text=True, timeout=60).split()
    assert int(ran) == 16
    assert 120 < float(lat1) < 390 and 120 < float(lat0) < 390, (lat1, lat0)  # behind a running job (180 ms), ahead of the queued ones
