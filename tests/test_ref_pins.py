"""Reference-derived pins of the process boundary (tests/golden/gen_ref_pins.py generated the fixtures by
running the reference's own Python on the reference's own sources at authoring time):
flag tables of the six executables, pyramid level sizes, and the PFM container. No GPU needed: flags are
parsed (and --helpxml answered) before any device is touched."""
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "facebook360_dep_amd", "bin")


def _gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _cxx_literal(text):
    """A scraped C++ default / description -> its value: adjacent string literals concatenate."""
    parts = re.findall(r'"((?:[^"\\]|\\.)*)"', text)
    return "".join(parts).replace('\\"', '"').replace("\\n", "\n") if parts else None


def _helpxml(binary):
    p = subprocess.run([os.path.join(BIN, binary), "--helpxml"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    flags = {}
    for m in re.finditer(r"<flag><file>.*?</file><name>(.*?)</name><meaning>(.*?)</meaning><default>(.*?)</default>"
                         r"<current>.*?</current><type>(.*?)</type></flag>", p.stdout, re.S):
        unesc = lambda v: v.replace("&lt;", "<").replace("&gt;", ">").replace("&amp;", "&")  # noqa: E731
        flags[m.group(1)] = dict(meaning=unesc(m.group(2)), default=unesc(m.group(3)), type=m.group(4))
    return flags


@pytest.mark.parametrize("binary,ref_binary", [
    ("DerpCLI", "DerpCLI"), ("TemporalBilateralFilter", "TemporalBilateralFilter"),
    ("UpsampleDisparity", "UpsampleDisparity"), ("LayerDisparities", "LayerDisparities"),
    ("GenerateForegroundMasks", "GenerateForegroundMasks"), ("ComputeRephotographyErrors", "ComputeRephotographyErrors"),
    ("DerpSequence", "DerpCLI")])  # the fused sequence driver takes every DerpCLI flag unchanged
def test_flag_tables_match_the_reference(built, binary, ref_binary):
    """Every DEFINE_* of the reference binary exists here with the same name, type, default and
    description (scripts read them: system_util.py:123-176, res/flags/*.flags)."""
    ref = _gold("ref_flags.json")[ref_binary]["flags"]
    mine = _helpxml(binary)
    assert len(ref) >= 9
    type_of = {"string": "string", "integer": "int32", "float": "double", "boolean": "bool"}
    for f in ref:
        name = f["name"]
        assert name in mine, "%s lacks --%s" % (binary, name)
        got = mine[name]
        assert got["type"] == type_of[f["type"]], (name, got["type"], f["type"])
        if f["type"] == "string":
            assert got["default"] == _cxx_literal(f["default"]), (name, got["default"], f["default"])
        elif f["type"] == "boolean":
            assert (got["default"] == "true") == bool(f["default"]), (name, got["default"], f["default"])
        else:
            assert float(got["default"]) == float(f["default"]), (name, got["default"], f["default"])
        want = _cxx_literal(f["descr"])
        # this build may append a bracketed remark ("[accepted; the GPU path ignores it]") to a description
        assert got["meaning"] == want or got["meaning"].startswith(want + " ["), (name, got["meaning"], want)
    # flags this build adds are marked as extensions or are glog's own
    ref_names = {f["name"] for f in ref}
    for name, got in mine.items():
        if name not in ref_names:
            assert "[extension" in got["meaning"] or got["meaning"].startswith("glog:"), (binary, name)


def test_derp_sequence_filter_flags_default_like_temporal_bilateral_filter(built):
    ref = {f["name"]: f for f in _gold("ref_flags.json")["TemporalBilateralFilter"]["flags"]}
    mine = _helpxml("DerpSequence")
    for name in ("sigma", "space_radius", "time_radius", "weight_b", "weight_g", "weight_r"):
        assert float(mine[name]["default"]) == float(ref[name]["default"]), name


def test_level_sizes_match_config_and_resize_py():
    from facebook360_dep_amd import resize, synth

    gold = _gold("ref_level_sizes.json")
    assert synth.WIDTHS == gold["widths"]  # scripts/render/config.py:46
    for case in gold["cases"]:
        w, h = case["rig_resolution"]
        want = [tuple(s) for s in case["sizes"] if s[0] <= w]
        assert synth.level_sizes(w, h) == want, case["rig_resolution"]
        assert resize.level_sizes(w, h) == [tuple(s) for s in case["sizes"]], case["rig_resolution"]


def test_pfm_container(tmp_path):
    """cv_util::writeCvMat32FC1ToPFM / readCvMat32FC1FromPFM (CvUtil.cpp:39-73): 'Pf', 'w h', '-1.0', then
    row-major little-endian floats, TOP row first (not the bottom-up order of the Netpbm convention)."""
    from facebook360_dep_amd import imageio as dio

    gold = _gold("ref_pfm.json")
    w, h = gold["width"], gold["height"]
    vals = np.array([np.nan if v is None else v for v in gold["values_row_major_top_first"]], np.float32).reshape(h, w)
    path = str(tmp_path / "x.pfm")
    dio.write_pfm(path, vals)
    raw = open(path, "rb").read()
    assert raw.hex() == gold["file_hex"]
    assert raw.startswith(gold["header"].encode())
    back = dio.read_pfm(path)
    assert back.shape == (h, w) and np.array_equal(back[0], vals[0]) and np.isnan(back[1, 2])
    assert struct.unpack("<f", raw[len(gold["header"]):len(gold["header"]) + 4])[0] == vals[0, 0]
