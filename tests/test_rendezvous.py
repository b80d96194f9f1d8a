"""cli/rendezvous.h on the CPU: the start-up rendezvous of DerpSequence's ranks through a shared directory — with the
leftovers of a crashed job in place (a token with a matching go file, halo files), ranks that arrive late, and the
all / none / mixed agreement every rank must read identically. The ranks are processes of a small harness
(tests/native/rendezvous_main.cpp) built here with g++."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("rv") / "rendezvous_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "rendezvous_main.cpp"), "-lz", "-ldl"])
    return exe


def _launch(exe, d, world, delays, oks, chatter_ms=None):
    more = [str(chatter_ms)] if chatter_ms else []
    procs = [subprocess.Popen([exe, d, str(r), str(world), str(delays[r]), str(oks[r])] + more, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=60)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("rendezvous hung")
        assert p.returncode == 0, e[-2000:]
        outs.append(o.split())
    return outs


@pytest.mark.parametrize("delays", [(0, 0, 0), (300, 0, 0), (0, 300, 50), (0, 0, 400)])
def test_rendezvous_ignores_a_dead_jobs_leftovers(harness, tmp_path, delays):
    d = str(tmp_path / ".derp_seq")
    os.makedirs(os.path.join(d, "halo"))
    with open(os.path.join(d, "token"), "w") as f:
        f.write("dead-job")
    with open(os.path.join(d, "go.dead-job"), "w") as f:
        f.write("some-old-word\n")
    for r in range(3):
        with open(os.path.join(d, "ready.dead-job.%d" % r), "w") as f:
            f.write("some-old-word")
        with open(os.path.join(d, "step.dead-job.%d" % r), "w") as f:
            f.write("0")
    with open(os.path.join(d, "halo", "L0_k2_f000001_to1.bin"), "wb") as f:
        f.write(b"\0" * 1024)
    outs = _launch(harness, d, 3, delays, (1, 1, 1))
    tokens = {o[0] for o in outs}
    assert len(tokens) == 1 and "dead-job" not in tokens
    assert [o[1] for o in outs] == ["1", "1", "1"]
    assert [o[2] for o in outs] == ["0", "0", "0"]  # nobody saw the dead job's halo files after the rendezvous
    assert not os.path.exists(d)  # rank 0 removed it after every rank said bye (halo leftovers included)
    assert os.listdir(str(tmp_path)) == []


def test_wipe_of_a_large_dead_directory_while_the_other_ranks_are_already_answering(harness, tmp_path):
    """The GPU suite once failed with "halo file of another launch (stale?)": ranks 1.. answer the dead job's token
    inside the directory while rank 0 deletes it; libstdc++'s remove_all gives up at the first entry that vanished
    between readdir and unlink (a rank's temporary file, renamed meanwhile — measured: 350 of 3000 wipes under such
    chatter stop with ENOENT, 1 of them before the halo files) and the dead job's halo files survive. Rank 0 now
    renames the directory out of sight before deleting it, and halo files carry the token in their names. The old
    failure is too rare to reproduce on demand; this guards the new wipe: 4000 dead files, the other ranks
    re-publishing their answers while rank 0 wipes, no leftovers and no grave directory afterwards."""
    for attempt in range(3):
        d = str(tmp_path / ("run%d" % attempt) / ".derp_seq")
        os.makedirs(os.path.join(d, "halo"))
        with open(os.path.join(d, "token"), "w") as f:
            f.write("dead-job")
        for i in range(4000):
            with open(os.path.join(d, "halo", "L0_k2_f%06d_to1.bin" % i), "wb") as f:
                f.write(b"\0" * 64)
        outs = _launch(harness, d, 4, (40, 0, 0, 0), (1, 1, 1, 1), chatter_ms=400)
        assert len({o[0] for o in outs}) == 1 and [o[2] for o in outs] == ["0"] * 4, outs
        assert os.listdir(os.path.dirname(d)) == []


@pytest.mark.parametrize("oks,want", [((1, 1), "1"), ((0, 0), "0"), ((1, 0), "-1"), ((0, 1), "-1")])
def test_rendezvous_agreement_is_the_same_on_every_rank(harness, tmp_path, oks, want):
    d = str(tmp_path / ".derp_seq")
    outs = _launch(harness, d, 2, (0, 100), oks)
    assert [o[1] for o in outs] == [want, want] and outs[0][0] == outs[1][0]


def test_two_launches_in_a_row_get_different_tokens(harness, tmp_path):
    d = str(tmp_path / ".derp_seq")
    a = _launch(harness, d, 2, (0, 0), (1, 1))
    b = _launch(harness, d, 2, (0, 0), (1, 1))
    assert a[0][0] != b[0][0]
