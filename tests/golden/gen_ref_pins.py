"""Authoring-time generator (needs /root/reference; never runs on the GPU box or in the test suite).

Pins everything about the path's process boundary that the reference tree itself can state without a
C++ build, by RUNNING the reference's own Python where that is possible:

  ref_flags.json      name / type / default / description of every gflags DEFINE_* of the six binaries on
                      the path, extracted with the reference's own scraper — `get_flags` of
                      scripts/util/system_util.py:123-176 is loaded from the file (ast) and executed on the
                      reference's .cpp sources, exactly as the reference's UI / pipeline does to learn a
                      binary's flags.
  ref_level_sizes.json  pyramid level sizes for several rig resolutions: WIDTHS is read by importing
                      scripts/render/config.py, and the height rule is the two statements of
                      scripts/render/resize.py:72-73, extracted from the file and executed.
  ref_pfm.json        the PFM container of cv_util::writeCvMat32FC1ToPFM (source/util/CvUtil.cpp:39-49):
                      the header literals are scraped from the source; the expected bytes of a 5x3 ramp are
                      header + row-major little-endian floats, top row first.

Usage: python tests/golden/gen_ref_pins.py   (from the repo root)
"""
import ast
import importlib.util
import json
import os
import re
import struct

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

BINARIES = {
    "DerpCLI": "source/depth_estimation/DerpCLI.cpp",
    "TemporalBilateralFilter": "source/depth_estimation/TemporalBilateralFilter.cpp",
    "UpsampleDisparity": "source/depth_estimation/UpsampleDisparity.cpp",
    "LayerDisparities": "source/depth_estimation/LayerDisparities.cpp",
    "GenerateForegroundMasks": "source/render/GenerateForegroundMasks.cpp",
    "ComputeRephotographyErrors": "source/render/ComputeRephotographyErrors.cpp",
}


def load_get_flags():
    """The reference's own `get_flags`, cut out of system_util.py (the module itself imports packages this
    image lacks) and executed with the two modules it uses."""
    path = os.path.join(REF, "scripts/util/system_util.py")
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_flags")
    ns = {"os": os, "re": re}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["get_flags"]


def gen_flags():
    get_flags = load_get_flags()
    out = {}
    for name, rel in BINARIES.items():
        out[name] = {"source": rel, "flags": get_flags(os.path.join(REF, rel))}
    return out


def gen_level_sizes():
    spec = importlib.util.spec_from_file_location("ref_config", os.path.join(REF, "scripts/render/config.py"))
    cfg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfg)
    src = open(os.path.join(REF, "scripts/render/resize.py")).read()
    m = re.search(r"ratio = rig_resolution\[1\] / rig_resolution\[0\]\n\s+for level, width in enumerate\(config\.WIDTHS\):\n"
                  r"(\s+height = [^\n]+\n\s+height \+= [^\n]+\n)", src)
    body = "\n".join(line.strip() for line in m.group(1).strip().splitlines())
    out = {"widths": list(cfg.WIDTHS), "height_rule": body.splitlines(), "cases": []}
    for res in [(2048, 2048), (4096, 4096), (512, 512), (3840, 2160), (1000, 751), (2048, 1025), (96, 96), (7680, 4320)]:
        ratio = res[1] / res[0]
        sizes = []
        for width in cfg.WIDTHS:
            ns = {"ratio": ratio, "width": width, "round": round}
            exec(body, ns)
            sizes.append([width, ns["height"]])
        out["cases"].append({"rig_resolution": list(res), "sizes": sizes})
    return out


def gen_pfm():
    src = open(os.path.join(REF, "source/util/CvUtil.cpp")).read()
    fn = src[src.index("void writeCvMat32FC1ToPFM"):src.index("cv::Mat_<float> readCvMat32FC1FromPFM")]
    lits = re.findall(r'file << (.*?);', fn)
    assert lits == ['"Pf\\n"', 'width << " " << height << "\\n"', '"-1.0\\n"'], lits
    assert "file.write((char*)m.ptr(), width * height * sizeof(float))" in fn  # row-major as stored: top row first
    w, h = 5, 3
    vals = [float(y * 10 + x) + 0.25 for y in range(h) for x in range(w)]
    vals[7] = float("nan")
    header = "Pf\n%d %d\n-1.0\n" % (w, h)
    payload = struct.pack("<%df" % (w * h), *vals)
    return {"width": w, "height": h, "header": header, "statements": lits,
            "values_row_major_top_first": [None if v != v else v for v in vals],
            "file_hex": (header.encode() + payload).hex()}


if __name__ == "__main__":
    for name, data in (("ref_flags.json", gen_flags()), ("ref_level_sizes.json", gen_level_sizes()),
                       ("ref_pfm.json", gen_pfm())):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", name)
