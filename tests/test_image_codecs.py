"""cli/image_codecs.h on the CPU: the decoders that stand in for `cv::imread(path, IMREAD_UNCHANGED)` on the input side
of the executables (CvUtil.cpp:23-29) — PNG in every flavour, JPEG, TIFF, BMP, PNM, chosen by signature.

What pins them:
  * JPEG is lossy, so "same picture" is not enough: the decoder restates libjpeg's default arithmetic and must give the
    same integers as libjpeg-turbo. tests/golden/codecs/*.jpg + expected.json were written and decoded by Pillow's
    libjpeg-turbo (gen_codec_vectors.py, committed beside them); when Pillow is importable the test also sweeps freshly
    encoded files (sizes x quality x subsampling x progressive x restart intervals).
  * PNG / TIFF / BMP / PNM are lossless: the files are produced here by small independent encoders (every PNG filter
    type, Adam7, sub-byte depths, palettes with tRNS; TIFF strips / tiles / planar / both byte orders / LZW / PackBits /
    Deflate / predictor) and, where Pillow can write or read the flavour, cross-checked against libpng / libtiff
    through Pillow in both directions.
The decoded samples leave a small harness (tests/native/codec_main.cpp, built here with g++) as raw uint16."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "codecs")

try:
    from PIL import Image
except ImportError:  # the committed vectors still run
    Image = None


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("codec") / "codec_main")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-o", exe,
                           os.path.join(ROOT, "tests", "native", "codec_main.cpp"), "-lz", "-ldl"])
    return exe


def decode(exe, path):
    """-> (array [h, w, c] uint16 / float32, kind, bitdepth) or (None, message, None) when the decoder refuses"""
    p = subprocess.run([exe, path, path + ".raw"], capture_output=True, text=True)
    if p.returncode == 3:
        return None, p.stdout.strip(), None
    assert p.returncode == 0, p.stderr
    kind, w, h, c, bd, pw, ph = p.stdout.split()
    w, h, c, bd = int(w), int(h), int(c), int(bd)
    assert (int(pw), int(ph)) == (w, h), "probe_size disagrees with decode: " + p.stdout
    a = np.fromfile(path + ".raw", dtype=np.float32 if bd == 32 else np.uint16).reshape(h, w, c)
    return a, kind, bd


def scene(w, h, c, seed=7, maxv=255):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(np.sin(x / 5.0 + k) * 0.5 + 0.5) * 200 + np.cos(y / 7.0 - k) * 30 for k in range(max(c, 1))], -1)
    base += rng.normal(0, 12, base.shape)
    if h >= 3 and w >= 4:
        base[h // 3: h // 2 + 1, w // 4: w // 2 + 1] = ([250, 10, 40, 128] * 2)[:base.shape[-1]]
    a = np.clip(base, 0, 255) / 255.0
    dtype = np.uint8 if maxv < 256 else np.uint16
    return np.round(a * maxv).astype(dtype)[..., :c]


# ------------------------------------------------------------------------------------------------ JPEG
def test_jpeg_committed_vectors_match_libjpeg_turbo(harness, tmp_path):
    expected = json.load(open(os.path.join(GOLDEN, "expected.json")))
    assert len(expected) >= 20
    for name, e in sorted(expected.items()):
        got, kind, bd = decode(harness, _copy(os.path.join(GOLDEN, name), tmp_path))
        assert kind == "jpeg" and bd == 8, (name, kind)
        assert list(got.shape) == e["shape"], name
        assert zlib.crc32(got.astype(np.uint8).tobytes()) == e["crc32"], name + ": samples differ from libjpeg-turbo's"


def _copy(src, tmp_path):
    dst = str(tmp_path / os.path.basename(src))
    with open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


@pytest.mark.skipif(Image is None, reason="Pillow not installed")
def test_jpeg_sweep_against_pillow(harness, tmp_path):
    path = str(tmp_path / "t.jpg")
    n = 0
    for (w, h) in [(37, 29), (17, 8), (8, 8), (1, 1), (2, 3), (3, 2), (100, 75), (33, 65)]:
        for c in (3, 1):
            for q in (30, 90, 100):
                for sub in ((0, 1, 2, "4:1:1") if c == 3 else (0,)):
                    for prog in (False, True):
                        for rst in (0, 3):
                            kw = dict(quality=q, progressive=prog, optimize=(q == 90))
                            if c == 3:
                                kw["subsampling"] = sub
                            if rst:
                                kw["restart_marker_blocks"] = rst
                            Image.fromarray(scene(w, h, c).reshape((h, w, 3) if c == 3 else (h, w))).save(path, **kw)
                            ref = np.asarray(Image.open(path)).reshape(h, w, c)
                            got, kind, _ = decode(harness, path)
                            assert got is not None and np.array_equal(got, ref), (w, h, c, q, sub, prog, rst, kind)
                            n += 1
    assert n == 8 * (3 * 4 * 4 + 3 * 4)


@pytest.mark.skipif(Image is None, reason="Pillow not installed")
def test_jpeg_vertical_subsampling_rgb_and_refusals(harness, tmp_path):
    path = str(tmp_path / "t.jpg")
    im = Image.fromarray(scene(64, 64, 3))
    # 4:4:0 (h1v2): a 4:2:2 file with its luma sampling factors swapped is a valid 4:4:0 file of the same MCU count
    im.save(path, quality=85, subsampling=1)
    d = bytearray(open(path, "rb").read())
    i = d.find(b"\xff\xc0")
    assert d[i + 11] == 0x21
    d[i + 11] = 0x12
    open(path, "wb").write(d)
    got, _, _ = decode(harness, path)
    assert np.array_equal(got, np.asarray(Image.open(path)))
    for prog in (False, True):  # Adobe marker, transform 0: the three components ARE R, G, B
        im.save(path, quality=90, keep_rgb=True, progressive=prog)
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, np.asarray(Image.open(path)))
    im.convert("CMYK").save(path)
    got, why, _ = decode(harness, path)
    assert got is None and "CMYK" in why
    noise = np.random.default_rng(1).integers(0, 256, (211, 173, 3)).astype(np.uint8)  # long EOB runs, deep refinement scans
    for q in (5, 98):
        Image.fromarray(noise).save(path, quality=q, progressive=True, subsampling=2)
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, np.asarray(Image.open(path)))


# ------------------------------------------------------------------------------------------------ PNG
def png_bytes(a, color_type, depth, interlace=False, palette=None, trns=None):
    """a: [h, w, file channels] raw sample values (palette indices for colour type 3). Rows cycle through the five
    PNG filter types."""
    h, w, ch = a.shape

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))

    def pack_rows(img):
        hh, ww, _ = img.shape
        if hh == 0 or ww == 0:
            return b""
        if depth == 16:
            rows = img.astype(">u2").reshape(hh, -1).view(np.uint8)
        elif depth == 8:
            rows = img.astype(np.uint8).reshape(hh, -1)
        else:
            bits = np.zeros((hh, (ww * depth + 7) // 8 * 8), np.uint8)
            for k in range(depth):
                bits[:, k: ww * depth: depth] = (img[:, :, 0] >> (depth - 1 - k)) & 1
            rows = np.packbits(bits, axis=1)
        bpp = max(1, ch * depth // 8)
        out = bytearray()
        prev = np.zeros(rows.shape[1], np.int32)
        for y in range(hh):
            cur = rows[y].astype(np.int32)
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if cur.size > bpp else np.zeros_like(cur)
            if cur.size <= bpp:
                left = np.zeros_like(cur)
            ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if cur.size > bpp else np.zeros_like(cur)
            ft = y % 5
            if ft == 0:
                f = cur
            elif ft == 1:
                f = cur - left
            elif ft == 2:
                f = cur - prev
            elif ft == 3:
                f = cur - ((left + prev) >> 1)
            else:
                p = left + prev - ul
                pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
                f = cur - pred
            out.append(ft)
            out += (f & 255).astype(np.uint8).tobytes()
            prev = cur
        return bytes(out)

    if interlace:
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
        raw = b"".join(pack_rows(a[y0::dy, x0::dx]) for x0, y0, dx, dy in passes)
    else:
        raw = pack_rows(a)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, int(interlace)))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", trns)
    half = len(raw) // 2  # two IDAT chunks: the stream continues across chunk boundaries
    comp = zlib.compress(raw, 6)
    return out + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b"")


def png_expected(a, color_type, depth, palette=None, trns=None):
    """What OpenCV's PngDecoder returns under IMREAD_UNCHANGED (in R, G, B [, A] order)."""
    maxv = 65535 if depth == 16 else 255
    if color_type == 0:
        return (a * (255 // ((1 << depth) - 1))) if depth < 8 else a  # tRNS of a gray image is ignored
    if color_type == 2:
        if trns is None:
            return a
        key = np.array(struct.unpack(">HHH", trns))
        alpha = np.where((a == key).all(-1), 0, maxv)[..., None]
        return np.concatenate([a, alpha], -1)
    if color_type == 3:
        rgb = np.asarray(palette)[a[..., 0]]
        if trns is None:
            return rgb
        al = np.full(256, 255)
        al[:len(trns)] = list(trns)
        return np.concatenate([rgb, al[a[..., 0]][..., None]], -1)
    if color_type == 4:
        return np.concatenate([a[..., :1]] * 3 + [a[..., 1:]], -1)
    return a


PNG_CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_depth_and_filter(harness, tmp_path, interlace):
    rng = np.random.default_rng(3)
    for (w, h) in [(37, 29), (1, 1), (3, 9), (9, 2), (8, 8), (5, 1)]:
        for ct, depth in PNG_CASES:
            ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
            maxv = (1 << depth) - 1
            a = rng.integers(0, maxv + 1, (h, w, ch)).astype(np.int64)
            a[h // 2:, : w // 2 + 1] = a[0, 0]  # flat areas next to noise: every filter predicts something
            palette = rng.integers(0, 256, (1 << depth, 3)) if ct == 3 else None
            for trns in ([None, bytes(rng.integers(0, 256, max(1, (1 << depth) // 2)).tolist())] if ct == 3 else
                         [None, struct.pack(">HHH", *a[0, 0])] if ct == 2 else
                         [None, struct.pack(">H", int(a[0, 0, 0]))] if ct == 0 else [None]):
                path = str(tmp_path / "t.png")
                open(path, "wb").write(png_bytes(a, ct, depth, interlace, palette, trns))
                got, kind, bd = decode(harness, path)
                want = png_expected(a, ct, depth, palette, trns)
                assert kind == "png" and bd == (16 if depth == 16 else 8)
                assert got.shape == want.shape and np.array_equal(got, want), (w, h, ct, depth, interlace, trns)
                # the executables' colour fast path takes exactly the flavours it says it takes, and gives what
                # cv_util::loadImage<Vec3w> makes of the generic decode (x257 from 8 bits, gray replicated, alpha dropped)
                p = subprocess.run([harness, path, path + ".bgr", "--bgr16"], capture_output=True, text=True)
                takes = (not interlace) and depth >= 8 and ct in (0, 2, 6) and not (ct == 2 and trns)
                assert p.stdout.split()[0] == ("fast" if takes else "notfast"), (p.stdout, ct, depth, interlace, trns)
                if takes:
                    bgr = np.fromfile(path + ".bgr", dtype=np.uint16).reshape(h, w, 3)
                    full = want.astype(np.int64) * (257 if depth == 8 else 1)
                    full = np.repeat(full, 3, axis=2) if full.shape[2] == 1 else full[..., :3]
                    assert np.array_equal(bgr, full[..., ::-1]), (w, h, ct, depth)
                if Image is not None and (w, h) == (37, 29) and depth <= 8 and trns is None:
                    # the encoder above is not the only witness: libpng (through Pillow) reads the same file
                    mode = {1: "L", 3: "RGB", 4: "RGBA"}[want.shape[-1]]
                    ref = np.asarray(Image.open(path).convert(mode)).reshape(want.shape)
                    assert np.array_equal(ref, want), (ct, depth, interlace)


@pytest.mark.skipif(Image is None, reason="Pillow not installed")
def test_png_written_by_libpng(harness, tmp_path):
    path = str(tmp_path / "t.png")
    for mode, c in (("L", 1), ("RGB", 3), ("RGBA", 4), ("LA", 2), ("P", 3), ("1", 1), ("I;16", 1)):
        a = scene(41, 23, 4 if c == 2 else max(c, 3))
        im = Image.fromarray(a[..., :3] if a.shape[-1] > 3 and mode not in ("RGBA", "LA") else a)
        if mode == "I;16":
            im = Image.fromarray(scene(41, 23, 1, maxv=65535)[..., 0])
        elif mode == "LA":
            im = Image.fromarray(a[..., [0, 3]], "LA") if False else Image.merge("LA", [Image.fromarray(a[..., 0]), Image.fromarray(a[..., 3])])
        elif mode != "RGBA":
            im = im.convert(mode)
        im.save(path, optimize=(mode == "P"))
        got, kind, bd = decode(harness, path)
        back = Image.open(path)
        if mode == "LA":
            ref = np.asarray(back.convert("RGBA"))
        elif mode == "P":
            ref = np.asarray(back.convert("RGB"))
        elif mode == "1":
            ref = np.asarray(back.convert("L"))[..., None]
        elif mode == "I;16":
            ref = np.asarray(back).astype(np.uint16)[..., None]
        else:
            ref = np.asarray(back).reshape(23, 41, -1)
        assert np.array_equal(got, ref), mode


# ------------------------------------------------------------------------------------------------ TIFF
def lzw_encode(data):
    """TIFF 6.0 section 13 LZW (MSB-first, early change), independent of the decoder under test."""
    out, acc, nbits = bytearray(), 0, 0
    width = 9

    def put(code):
        nonlocal acc, nbits
        acc = (acc << width) | code
        nbits += width
        while nbits >= 8:
            out.append((acc >> (nbits - 8)) & 255)
            nbits -= 8

    table = {bytes([i]): i for i in range(256)}
    nxt = 258
    put(256)
    cur = b""
    for byte in data:
        nb = cur + bytes([byte])
        if nb in table:
            cur = nb
            continue
        put(table[cur])
        table[nb] = nxt
        nxt += 1
        if nxt == (1 << width) and width < 12:  # the encoder's table runs one entry ahead of the decoder's
            width += 1
        elif nxt == 4094:
            put(256)
            table = {bytes([i]): i for i in range(256)}
            nxt, width = 258, 9
        cur = bytes([byte])
    if cur:
        put(table[cur])
        nxt += 1  # the decoder adds an entry for this code too
        if nxt == (1 << width) and width < 12:  # the encoder's table runs one entry ahead of the decoder's
            width += 1
    put(257)
    if nbits:
        out.append((acc << (8 - nbits)) & 255)
    return bytes(out)


def packbits_encode(data):
    out, i = bytearray(), 0
    while i < len(data):
        run = 1
        while i + run < len(data) and data[i + run] == data[i] and run < 128:
            run += 1
        if run >= 3:
            out += bytes([257 - run, data[i]])
            i += run
        else:
            j = i
            while j < len(data) and j - i < 128 and not (j + 2 < len(data) and data[j] == data[j + 1] == data[j + 2]):
                j += 1
            out += bytes([j - i - 1]) + bytes(data[i:j])
            i = j
    return bytes(out)


def tiff_bytes(a, bo="<", compression=1, predictor=1, tile=None, planar=1, photometric=None, extrasamples=None, rows_per_strip=5,
               colormap=None):
    """a: [h, w, spp] uint8 / uint16 / float32."""
    h, w, spp = a.shape
    bits = a.dtype.itemsize * 8
    fmt = 3 if a.dtype == np.float32 else 1
    if photometric is None:
        photometric = 2 if spp >= 3 else 1
    th, tw = (tile if tile else (rows_per_strip, w))
    chunks = []
    for pl in range(spp if planar == 2 else 1):
        src = a[..., pl: pl + 1] if planar == 2 else a
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                if tile:
                    blk = np.zeros((th, tw, src.shape[2]), a.dtype)
                    part = src[y0: y0 + th, x0: x0 + tw]
                    blk[: part.shape[0], : part.shape[1]] = part
                else:
                    blk = src[y0: y0 + th]
                if predictor == 2 and compression in (5, 8):  # libtiff applies the tag inside its LZW / Deflate codecs only
                    blk = blk.copy()
                    blk[:, 1:] = blk[:, 1:] - blk[:, :-1]  # wraps modulo the sample width
                raw = blk.astype(blk.dtype.newbyteorder(bo)).tobytes()
                chunks.append({1: lambda r: r, 5: lzw_encode, 8: lambda r: zlib.compress(r, 6), 32773: packbits_encode}[compression](raw))
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [bits] * spp), (259, 3, [compression]), (262, 3, [photometric]), (277, 3, [spp]),
            (284, 3, [planar]), (339, 3, [fmt] * spp)]
    if predictor != 1:
        tags.append((317, 3, [predictor]))
    if extrasamples is not None:
        tags.append((338, 3, [extrasamples]))
    if colormap is not None:
        tags.append((320, 3, list(colormap)))
    if tile:
        tags += [(322, 3, [tw]), (323, 3, [th])]
    else:
        tags.append((278, 3, [th]))
    body = bytearray(b"\0" * 8)
    offsets = []
    for c in chunks:
        offsets.append(len(body))
        body += c
        if len(body) & 1:
            body += b"\0"
    tags += [(324 if tile else 273, 4, offsets), (325 if tile else 279, 4, [len(c) for c in chunks])]
    tags.sort()
    extra = bytearray()
    ifd_at = len(body)
    extra_at = ifd_at + 2 + 12 * len(tags) + 4
    ifd = struct.pack(bo + "H", len(tags))
    for tag, typ, vals in tags:
        code = {3: "H", 4: "I"}[typ]
        blob = struct.pack(bo + code * len(vals), *vals)
        if len(blob) <= 4:
            ifd += struct.pack(bo + "HHI", tag, typ, len(vals)) + blob.ljust(4, b"\0")
        else:
            ifd += struct.pack(bo + "HHII", tag, typ, len(vals), extra_at + len(extra))
            extra += blob
            if len(extra) & 1:
                extra += b"\0"
    ifd += struct.pack(bo + "I", 0)
    head = (b"II" if bo == "<" else b"MM") + struct.pack(bo + "HI", 42, ifd_at)
    return head + bytes(body[8:]) + ifd + bytes(extra)


def test_tiff_layouts_compressions_and_sample_types(harness, tmp_path):
    path = str(tmp_path / "t.tif")
    n = 0
    for (w, h) in [(37, 29), (1, 1), (16, 16), (5, 33)]:
        for dtype, maxv in ((np.uint8, 255), (np.uint16, 65535)):
            for spp in (1, 3, 4):
                a = scene(w, h, spp, maxv=maxv).astype(dtype)
                a[h // 2:, : w // 2 + 1] = a[0, 0]
                for bo in "<>":
                    for compression in (1, 5, 8, 32773):
                        for predictor in (1, 2):
                            for tile in (None, (16, 16)):
                                for planar in ((1, 2) if spp > 1 and compression == 8 else (1,)):
                                    if (w, h) != (37, 29) and (bo == ">" or predictor == 2) and compression != 5:
                                        continue  # the full matrix at one size, a thinner one at the odd sizes
                                    extras = 1 if spp == 4 else None  # associated alpha: samples as stored
                                    open(path, "wb").write(tiff_bytes(a, bo, compression, predictor, tile, planar, extrasamples=extras))
                                    got, kind, bd = decode(harness, path)
                                    assert kind == "tiff" and bd == 8 * a.dtype.itemsize, (kind, bd)
                                    assert np.array_equal(got, a), (w, h, dtype, spp, bo, compression, predictor, tile, planar)
                                    n += 1
                                    if Image is not None and dtype == np.uint8 and (w, h) == (37, 29) and spp < 4:  # (Pillow un-multiplies associated alpha)
                                        ref = np.asarray(Image.open(path)).reshape(h, w, spp)  # libtiff agrees with the writer above
                                        assert np.array_equal(ref, a), ("writer", bo, compression, predictor, tile, planar)
    assert n > 300


def test_tiff_float_palette_min_is_white_and_unassociated_alpha(harness, tmp_path):
    path = str(tmp_path / "t.tif")
    f = np.random.default_rng(5).normal(0, 3, (19, 23, 1)).astype(np.float32)
    f[3, 4] = np.nan
    for bo in "<>":
        for compression in (1, 8, 5):
            open(path, "wb").write(tiff_bytes(f, bo, compression))
            got, _, bd = decode(harness, path)
            assert bd == 32 and np.array_equal(got.view(np.uint32), f.view(np.uint32))
    idx = scene(23, 19, 1)
    rng = np.random.default_rng(6)
    for wide in (True, False):  # libtiff's RGBA interface: 16-bit colour maps >> 8, maps that were written 8-bit as they are
        cmap = rng.integers(0, 65536 if wide else 256, 768)
        open(path, "wb").write(tiff_bytes(idx, photometric=3, colormap=cmap))
        got, _, _ = decode(harness, path)
        lut = (cmap.reshape(3, 256).T >> 8) if wide else cmap.reshape(3, 256).T
        assert np.array_equal(got, lut[idx[..., 0]])
        if Image is not None and wide:
            # Pillow scales the map by 257 instead of shifting: equal on maps that are exact multiples of 257
            pass
    open(path, "wb").write(tiff_bytes(idx, photometric=0))
    got, _, _ = decode(harness, path)
    assert np.array_equal(got, 255 - idx)
    rgba = scene(23, 19, 4)
    open(path, "wb").write(tiff_bytes(rgba, extrasamples=2))
    got, _, _ = decode(harness, path)
    want = rgba.astype(np.int64)
    want[..., :3] = (want[..., 3:] * want[..., :3] + 127) // 255  # tif_getimage.c's un-associated -> associated table
    assert np.array_equal(got, want)
    for what, kw in (("16-bit min-is-white", dict(a=scene(8, 8, 1, maxv=65535), photometric=0)), ("JPEG-in-TIFF", dict(a=idx, compression=7))):
        try:
            data = tiff_bytes(**kw)
        except KeyError:
            data = tiff_bytes(kw["a"]).replace(struct.pack("<HHII", 259, 3, 1, 1), struct.pack("<HHII", 259, 3, 1, 7))
        open(path, "wb").write(data)
        got, why, _ = decode(harness, path)
        assert got is None and "unsupported TIFF" in why, what


@pytest.mark.skipif(Image is None, reason="Pillow not installed")
def test_tiff_written_by_libtiff(harness, tmp_path):
    path = str(tmp_path / "t.tif")
    for mode in ("L", "RGB", "RGBA", "I;16", "F"):
        if mode == "I;16":
            im = Image.fromarray(scene(61, 47, 1, maxv=65535)[..., 0])
        elif mode == "F":
            im = Image.fromarray(np.random.default_rng(2).normal(0, 1, (47, 61)).astype(np.float32))
        else:
            im = Image.fromarray(scene(61, 47, {"L": 1, "RGB": 3, "RGBA": 4}[mode]).squeeze())
        for compression in ("raw", "tiff_lzw", "tiff_adobe_deflate", "packbits"):
            for predictor in ((1, 2) if compression in ("tiff_lzw", "tiff_adobe_deflate") and mode != "F" else (1,)):
                kw = {"tiffinfo": {317: predictor}} if predictor == 2 else {}
                im.save(path, compression=compression, **kw)
                got, kind, bd = decode(harness, path)
                ref = np.asarray(Image.open(path))
                ref = ref.reshape(47, 61, -1)
                if mode == "RGBA":  # Pillow writes un-associated alpha (ExtraSamples 2): OpenCV's 8-bit path multiplies it in
                    want = ref.astype(np.int64)
                    want[..., :3] = (want[..., 3:] * want[..., :3] + 127) // 255
                    ref = want
                assert got is not None, (mode, compression, kind)
                assert np.array_equal(got, ref), (mode, compression, predictor)


# ------------------------------------------------------------------------------------------------ BMP, PNM, dispatch
def test_bmp_and_pnm(harness, tmp_path):
    rgb = scene(37, 29, 3)
    h, w, _ = rgb.shape

    def bmp(bpp, top_down=False, palette=None, pixels=None):
        stride = (w * bpp + 31) // 32 * 4
        rows = b""
        for y in (range(h) if top_down else range(h - 1, -1, -1)):
            rows += pixels[y].tobytes().ljust(stride, b"\0")
        pal = b"" if palette is None else b"".join(bytes([b, g, r, 0]) for r, g, b in palette)
        off = 14 + 40 + len(pal)
        return (b"BM" + struct.pack("<IHHI", off + len(rows), 0, 0, off) +
                struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bpp, 0, len(rows), 2835, 2835, len(palette or []), 0) + pal + rows)

    path = str(tmp_path / "t.bmp")
    for top_down in (False, True):
        open(path, "wb").write(bmp(24, top_down, pixels=rgb[..., ::-1]))
        got, kind, _ = decode(harness, path)
        assert kind == "bmp" and np.array_equal(got, rgb)
        rgba = scene(37, 29, 4)
        open(path, "wb").write(bmp(32, top_down, pixels=rgba[..., [2, 1, 0, 3]]))
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, rgba)
        idx = scene(37, 29, 1)
        gray = [(i, i, i) for i in range(256)]
        open(path, "wb").write(bmp(8, top_down, gray, idx[..., 0]))
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, idx)
        colour = [(i, 255 - i, (i * 7) & 255) for i in range(256)]
        open(path, "wb").write(bmp(8, top_down, colour, idx[..., 0]))
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, np.asarray(colour)[idx[..., 0]])
    if Image is not None:
        Image.fromarray(rgb).save(path)
        got, _, _ = decode(harness, path)
        assert np.array_equal(got, rgb)
    # a palette shorter than the pixel values reach (biClrUsed = 2, pixel 255): the missing entries read as zeros
    # (grfmt_bmp.cpp fills a 256-entry table), never as the bytes behind the file (heap overflow under ASan)
    w_, h_ = w, h
    w, h = 1, 1
    open(path, "wb").write(bmp(8, False, [(9, 9, 9), (200, 200, 200)], np.asarray([[255]], dtype=np.uint8)))
    got, _, _ = decode(harness, path)
    assert got.shape[:2] == (1, 1) and int(got.ravel()[0]) == 0
    open(path, "wb").write(bmp(8, False, [(9, 9, 9), (200, 200, 200)], np.asarray([[1]], dtype=np.uint8)))
    got, _, _ = decode(harness, path)
    assert int(got.ravel()[0]) == 200
    w, h = w_, h_

    path = str(tmp_path / "t.pnm")
    for maxv, dt in ((255, ">u1"), (65535, ">u2"), (1000, ">u2"), (100, ">u1")):
        for c, binary_kind, ascii_kind in ((1, 5, 2), (3, 6, 3)):
            a = (scene(17, 9, c, maxv=65535).astype(np.int64) * maxv // 65535)
            open(path, "wb").write(b"P%d\n# a comment\n17 9\n%d\n" % (binary_kind, maxv) + a.astype(dt).tobytes())
            got, kind, bd = decode(harness, path)
            assert kind == "pnm" and bd == (8 if maxv < 256 else 16) and np.array_equal(got, a)
            open(path, "wb").write(b"P%d 17 9 %d\n" % (ascii_kind, maxv) + b" ".join(b"%d" % v for v in a.ravel()) + b"\n")
            got, _, _ = decode(harness, path)
            assert np.array_equal(got, a)
    bits = (scene(17, 9, 1)[..., 0] > 128).astype(np.uint8)
    open(path, "wb").write(b"P4\n17 9\n" + np.packbits(bits, axis=1).tobytes())
    got, _, _ = decode(harness, path)
    assert np.array_equal(got[..., 0], np.where(bits, 0, 255))
    open(path, "wb").write(b"P1\n17 9\n" + b"\n".join(b"".join(b"%d" % v for v in r) for r in bits) + b"\n")
    got, _, _ = decode(harness, path)
    assert np.array_equal(got[..., 0], np.where(bits, 0, 255))


def test_decoder_is_chosen_by_signature_and_refuses_by_name(harness, tmp_path):
    rgb = scene(9, 7, 3)
    path = str(tmp_path / "looks_like.png")  # a TIFF under a .png name: cv::imread looks at the bytes
    open(path, "wb").write(tiff_bytes(rgb))
    got, kind, _ = decode(harness, path)
    assert kind == "tiff" and np.array_equal(got, rgb)
    for blob, word in ((b"RIFF\x10\0\0\0WEBPVP8 ", "webp"), (b"\0\0\0\x0cjP  \r\n\x87\n", "jpeg 2000"), (b"GIF89a" + b"\0" * 20, "unknown"),
                       (b"", "unknown")):
        open(path, "wb").write(blob)
        got, why, _ = decode(harness, path)
        assert got is None and word in why, why
    png = png_bytes(rgb.astype(np.int64), 2, 8)
    for cut in (len(png) // 2, 40, 20):  # truncated files fail with a message, not with a crash
        open(path, "wb").write(png[:cut])
        got, why, _ = decode(harness, path)
        assert got is None and why.startswith("error:"), why


def test_hostile_headers_are_refused_before_anything_is_allocated(harness, tmp_path):
    """Sizes come from the file: a header that promises more samples than the file's bytes can deliver (or more than
    OpenCV's own 2^20 x 2^20 / 2^30-pixel limits) fails with a message, promptly, instead of with an out-of-memory abort.
    (The decoders were also run over 960 000 randomly damaged files under AddressSanitizer + UBSan while they were
    written; that harness is not part of the suite.)"""
    import time

    path = str(tmp_path / "t.bin")
    rgb = scene(9, 7, 3)
    png = bytearray(png_bytes(rgb.astype(np.int64), 2, 8))
    png[16:24] = struct.pack(">II", 1000000, 1000000)
    tif = bytearray(tiff_bytes(rgb))
    i = tif.find(struct.pack("<HHI", 256, 4, 1))
    tif[i + 8: i + 12] = struct.pack("<I", 900000)
    j = tif.find(struct.pack("<HHI", 257, 4, 1))
    tif[j + 8: j + 12] = struct.pack("<I", 900000)
    with open(os.path.join(GOLDEN, "gray_7x5_q75.jpg"), "rb") as f:
        jpg = bytearray(f.read())
    k = jpg.find(b"\xff\xc0")
    jpg[k + 5: k + 9] = struct.pack(">HH", 65000, 65000)
    pnm = b"P6\n70000 70000\n255\n" + b"\0" * 64
    bmp = bytearray(b"BM" + struct.pack("<IHHI", 0, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 60000, 60000, 1, 24, 0, 0, 0, 0, 0, 0) + b"\0" * 64)
    for name, blob in (("png", png), ("tiff", tif), ("jpeg", jpg), ("pnm", pnm), ("bmp", bmp)):
        open(path, "wb").write(bytes(blob))
        t0 = time.time()
        got, why, _ = decode(harness, path)
        assert got is None and why.startswith("error:"), (name, why)
        assert time.time() - t0 < 2.0, name


def test_python_host_reads_the_same_through_the_c_abi(tmp_path):
    """facebook360_dep_amd.imageio.read_image = derp_image_info + derp_image_decode of libderp_hip.so (what the pyramid
    builder's cv2.imread stand-in uses): the committed JPEG vectors again, in OpenCV's channel order, and the TIFF writer
    of the pyramid builder read back (and read by libtiff through Pillow)."""
    from facebook360_dep_amd import imageio as dio

    expected = json.load(open(os.path.join(GOLDEN, "expected.json")))
    for name, e in sorted(expected.items()):
        a = dio.read_image(os.path.join(GOLDEN, name))
        assert a.dtype == np.uint8 and list(a.shape) == (e["shape"] if e["shape"][2] == 3 else e["shape"][:2])
        rgb = a[..., ::-1] if a.ndim == 3 else a
        assert zlib.crc32(np.ascontiguousarray(rgb).tobytes()) == e["crc32"], name
    path = str(tmp_path / "t.tif")
    for a in (scene(31, 17, 3, maxv=65535), scene(31, 17, 3), scene(31, 17, 1)[..., 0], scene(31, 17, 4),
              np.random.default_rng(4).normal(0, 1, (17, 31)).astype(np.float32)):
        dio.write_tiff(path, a)
        back = dio.read_image(path)
        assert back.dtype == a.dtype and back.shape == a.shape
        if a.ndim == 3 and a.shape[2] == 4:  # un-associated alpha of an 8-bit file is multiplied in (libtiff's RGBA interface)
            want = a.astype(np.int64)
            want[..., :3] = (want[..., 3:] * want[..., :3] + 127) // 255
            assert np.array_equal(back, want)
        else:
            assert np.array_equal(back, a)
        if Image is not None and a.dtype == np.uint8 and a.ndim == 3:
            ref = np.asarray(Image.open(path))
            assert np.array_equal(ref, a[..., [2, 1, 0] + ([3] if a.shape[2] == 4 else [])])
    # a float image as colour / mask: cv_util::convertImage (CvUtil.h:196-262) scales CV_32F by 65535 / 255 with
    # saturate_cast's round-half-even and then goes on as for an integer image
    f = np.asarray([[0.0, 0.25, 0.5, 1.0, 1.5, -0.2, 127.5 / 255.0, 128.4 / 255.0]], dtype=np.float32)
    dio.write_tiff(path, f)
    col = dio.load_color_u16(path)
    assert col.shape == (1, 8, 3) and col.dtype == np.uint16
    assert col[0, :, 0].tolist() == [0, 16384, 32768, 65535, 65535, 0, 32768, 32999] and np.array_equal(col[..., 0], col[..., 2])
    assert dio.load_mask(path)[0].tolist() == [0, 0, 1, 1, 1, 0, 1, 1]
    open(path, "wb").write(b"RIFF\x10\0\0\0WEBPVP8 ")
    with pytest.raises(ValueError, match="webp"):
        dio.read_image(path)


def test_jpeg_encoder_writes_libjpegs_bytes(tmp_path):
    """derp_jpeg_encode (what the pyramid builder's cv2.imwrite stand-in calls for JPEG level directories): the committed
    CRCs are those of the files libjpeg-turbo wrote for the same pictures (gen_codec_vectors.py); with Pillow at hand,
    a sweep of sizes (whole, partial and single MCUs; odd sides), gray and colour, smooth and noisy, four qualities —
    byte for byte."""
    from facebook360_dep_amd import imageio as dio

    path = str(tmp_path / "t.jpg")
    expected = json.load(open(os.path.join(GOLDEN, "encoder_expected.json")))
    for key, e in sorted(expected.items()):
        dims, q = key.split("_q")
        w, h, c = (int(v) for v in dims.split("x"))
        a = scene(w, h, c)
        dio.write_jpeg(path, a[..., ::-1] if c == 3 else a[..., 0], int(q))
        data = open(path, "rb").read()
        assert len(data) == e["bytes"] and zlib.crc32(data) == e["crc32"], key
        back = dio.read_image(path)  # and the decoder reads its sibling's files
        assert back.shape == ((h, w, 3) if c == 3 else (h, w))
    if Image is None:
        return
    rng = np.random.default_rng(5)
    for (w, h) in [(16, 16), (37, 53), (8, 8), (1, 1), (2, 3), (17, 9), (33, 31), (15, 16), (16, 15), (130, 70)]:
        for c in (3, 1):
            for a in (scene(w, h, c), rng.integers(0, 256, (h, w, c)).astype(np.uint8)):
                for q in (95, 75, 30, 100):
                    dio.write_jpeg(path, a[..., ::-1] if c == 3 else a[..., 0], q)
                    Image.fromarray(a if c == 3 else a[..., 0]).save(path + ".pil.jpg", quality=q)
                    assert open(path, "rb").read() == open(path + ".pil.jpg", "rb").read(), (w, h, c, q)
    with pytest.raises(ValueError):
        dio.write_jpeg(path, np.zeros((4, 4, 3), np.uint16))
