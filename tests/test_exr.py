"""--output_formats=exr (PyramidLevel.h:515-516 -> cv::imwrite of a CV_32FC1): the executables' OpenEXR writer
(cli/cli_common.h, write_exr_f32), their reader (read_exr_f32: the .exr files sort before the .pfm of the same frame, so
the sibling binaries' first-extension lookup picks them, as cv::imread would read them in the reference) and the
reader in facebook360_dep_amd/imageio.py, on the CPU: sizes that end in a
short last block of scan lines, NaN, data that deflate cannot shrink (stored raw), and the header's fixed fields.
Both ends follow the published OpenEXR file layout; no OpenEXR build exists in this image to pin them to (parity
unpinned, like every OpenCV codec of SURVEY 8c)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "%s/facebook360_dep_amd/cli/cli_common.h"
int main(int argc, char** argv) {
  if (std::string(argv[1]) == "read") {  // read <in.exr> <out.pfm>: the executables' reader on a foreign file
    int rw = 0, rh = 0;
    const std::vector<float> img = cli::load_float(argv[2], rw, rh);
    cli::write_pfm(argv[3], img.data(), rw, rh);
    return 0;
  }
  const int w = atoi(argv[2]), h = atoi(argv[3]), noise = atoi(argv[4]);
  std::vector<float> m((size_t)w * h);
  uint32_t s = 12345;
  for (size_t i = 0; i < m.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    if (noise) {
      memcpy(&m[i], &s, 4);  // arbitrary bit patterns (NaNs with payloads among them): deflate cannot shrink these
    } else {
      m[i] = (i %% 7 == 0) ? NAN : (float)(std::sin(i * 0.001) * 0.5 + (i %% 13) * 1e-3);
    }
  }
  cli::write_exr_f32(argv[1], m.data(), w, h);
  cli::write_pfm(std::string(argv[1]) + ".pfm", m.data(), w, h);
  // and back through the executables' own reader (what load_float / image_size do for a .exr input)
  int rw = 0, rh = 0, sw = 0, sh = 0;
  const std::vector<float> back = cli::load_float(argv[1], rw, rh);
  if (!cli::image_size(argv[1], sw, sh) || sw != w || sh != h || rw != w || rh != h ||
      memcmp(back.data(), m.data(), m.size() * 4) != 0) {
    fprintf(stderr, "C++ EXR reader disagrees with the writer\\n");
    return 2;
  }
  cli::write_pfm(std::string(argv[1]) + ".back.pfm", back.data(), rw, rh);
  return 0;
}
'''


@pytest.fixture(scope="module")
def exr_tool(tmp_path_factory):
    d = tmp_path_factory.mktemp("exr")
    src = d / "t.cpp"
    src.write_text(SRC % ROOT)
    exe = str(d / "t")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe, "-lz",
                           "-lpthread"])
    return exe


@pytest.mark.parametrize("w,h,noise", [(97, 35, 0), (64, 16, 0), (5, 1, 0), (33, 48, 1)])
def test_exr_writer_against_reader(exr_tool, tmp_path, w, h, noise):
    from facebook360_dep_amd import imageio as dio

    path = str(tmp_path / "a.exr")
    subprocess.check_call([exr_tool, path, str(w), str(h), str(noise)])
    got, want = dio.read_exr(path), dio.read_pfm(path + ".pfm")
    assert got.shape == (h, w) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(dio.read_pfm(path + ".back.pfm").view(np.uint32), want.view(np.uint32))  # the C++ reader
    data = open(path, "rb").read()
    assert data[:8] == b"\x76\x2f\x31\x01\x02\x00\x00\x00"  # magic, version 2, single-part scan-line
    assert b"channels\0chlist\0" in data and b"compression\0compression\0\x01\0\0\0\x03" in data  # ZIP, 16-line blocks
    i = data.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    assert struct.unpack_from("<4i", data, i) == (0, 0, w - 1, h - 1)
    if noise:  # stored raw: chunk size = lines * w * 4
        blocks = (h + 15) // 16
        hdr_end = data.index(b"screenWindowWidth\0float\0") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1
        off0, = struct.unpack_from("<Q", data, hdr_end)
        assert off0 == hdr_end + 8 * blocks
        y, size = struct.unpack_from("<ii", data, off0)
        assert y == 0 and size == 16 * w * 4


def _py_exr(path, img, compression):
    """A second, independent encoder (NONE = 0 or ZIPS = 2: one scan line per chunk) for the reader tests."""
    import zlib

    h, w = img.shape

    def attr(name, typ, value):
        return name + b"\0" + typ + b"\0" + struct.pack("<i", len(value)) + value

    hdr = b"\x76\x2f\x31\x01\x02\x00\x00\x00"
    hdr += attr(b"channels", b"chlist", b"Y\0" + struct.pack("<i", 2) + b"\0\0\0\0" + struct.pack("<ii", 1, 1) + b"\0")
    hdr += attr(b"compression", b"compression", bytes([compression]))
    hdr += attr(b"dataWindow", b"box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    hdr += attr(b"displayWindow", b"box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    hdr += attr(b"lineOrder", b"lineOrder", b"\0")
    hdr += attr(b"pixelAspectRatio", b"float", struct.pack("<f", 1.0))
    hdr += attr(b"screenWindowCenter", b"v2f", struct.pack("<ff", 0.0, 0.0))
    hdr += attr(b"screenWindowWidth", b"float", struct.pack("<f", 1.0)) + b"\0"
    chunks = []
    for y in range(h):
        raw = img[y].astype("<f4").tobytes()
        body = raw
        if compression == 2:
            b = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int64)
            t[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
            z = zlib.compress(t.astype(np.uint8).tobytes())
            if len(z) < len(raw):
                body = z
        chunks.append(struct.pack("<ii", y, len(body)) + body)
    off = len(hdr) + 8 * h
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("compression", [0, 2])
def test_exr_reader_on_other_encodings(exr_tool, tmp_path, compression):
    """Uncompressed and ZIPS (one line per chunk) files from an independent encoder: both readers return the floats."""
    from facebook360_dep_amd import imageio as dio

    rng = np.random.default_rng(3)
    img = (rng.random((19, 41)) * 2).astype(np.float32)
    img[::4, ::3] = np.nan
    img[5] = 0.25  # a line deflate shrinks
    path = str(tmp_path / "b.exr")
    _py_exr(path, img, compression)
    assert np.array_equal(dio.read_exr(path).view(np.uint32), img.view(np.uint32))
    subprocess.check_call([exr_tool, "read", path, path + ".pfm"])
    assert np.array_equal(dio.read_pfm(path + ".pfm").view(np.uint32), img.view(np.uint32))
