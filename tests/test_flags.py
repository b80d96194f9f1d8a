"""cli_common.h's gflags look-alike on the CPU — the process boundary is flags in, exit status out (scripts/render/
worker.py:66-107 builds these command lines), so the parser has to take what gflags takes: --name=value, --name value,
-name, --bool / --nobool / --bool=false, --flagfile (one flag per line, '#' comments, bare names and unknown flags
tolerated there, as res/test/derp_cli.flags relies on), and refuse what gflags refuses with exit status 1."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("flags") / "flags_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "flags_main.cpp"),
                           "-lz", "-ldl"])
    return exe


def parse(exe, *args):
    p = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=30)
    return p.returncode, dict(line.split("=", 1) for line in p.stdout.splitlines() if "=" in line), p.stderr


def test_forms_gflags_accepts(harness, tmp_path):
    rc, v, _ = parse(harness)
    assert rc == 0 and v == {"input_root": "", "cameras": "all", "threads": "-1", "level_start": "9", "sigma": "0.01",
                             "partial_coverage": "0", "do_median_filter": "1"}
    rc, v, _ = parse(harness, "--input_root=/a b/c", "--threads", "4", "-level_start=0x10", "--sigma", "2.5e-3", "--partial_coverage",
                     "--nodo_median_filter", "--cameras=")
    assert rc == 0 and v["input_root"] == "/a b/c" and v["threads"] == "4" and v["level_start"] == "16" and float(v["sigma"]) == 0.0025
    assert v["partial_coverage"] == "1" and v["do_median_filter"] == "0" and v["cameras"] == ""
    rc, v, _ = parse(harness, "--threads=010")
    assert rc == 0 and v["threads"] == "10"  # decimal unless it starts with 0x, like gflags
    rc, v, _ = parse(harness, "--partial_coverage=false", "--do_median_filter=1", "--partial_coverage", "stray_positional")
    assert rc == 0 and v["partial_coverage"] == "1" and v["do_median_filter"] == "1"  # the later occurrence wins; bools take no separate value
    ff = tmp_path / "a.flags"
    ff.write_text("# comment\n\n--input_root\n--threads=7\n  --sigma=0.5  \n--unknown_in_a_shared_file=1\n--nopartial_coverage\n--cameras=cam0,cam1\n")
    rc, v, _ = parse(harness, "--partial_coverage", "--flagfile=" + str(ff), "--threads=3")
    assert rc == 0 and v["threads"] == "3" and float(v["sigma"]) == 0.5 and v["partial_coverage"] == "0" and v["cameras"] == "cam0,cam1"
    rc, v, _ = parse(harness, "--flagfile", str(ff))
    assert rc == 0 and v["threads"] == "7"
    rc, v, err = parse(harness, "--log_dir=" + str(tmp_path / "logs"), "--threads=2")
    assert rc == 0 and os.path.exists(str(tmp_path / "logs" / "flags_main.INFO")) and "--threads=2" in err  # SystemUtil.cpp:78-97


def test_what_gflags_refuses(harness, tmp_path):
    for args, word in ((["--bogus=1"], "unknown command line flag"), (["--threads=four"], "illegal value"), (["--threads=1.5"], "illegal value"),
                       (["--threads=99999999999"], "illegal value"), (["--sigma=abc"], "illegal value"), (["--threads"], "missing its argument"),
                       (["--flagfile=" + str(tmp_path / "nope.flags")], "can't open flagfile")):
        rc, _, err = parse(harness, *args)
        assert rc == 1 and word in err, (args, err[-200:])
    p = subprocess.run([harness, "--helpxml"], capture_output=True, text=True)
    assert p.returncode == 0 and "<name>sigma</name>" in p.stdout and "<default>0.01</default>" in p.stdout
    p = subprocess.run([harness, "--help"], capture_output=True, text=True)
    assert p.returncode == 0 and "-level_start (an int) type: int32 default: 9" in p.stdout
