"""On-disk formats of the depth path, numpy + zlib only.

PFM exactly as cv_util::writeCvMat32FC1ToPFM / readCvMat32FC1FromPFM do it
(source/util/CvUtil.cpp:39-73): header "Pf\\n<W> <H>\\n-1.0\\n", raw little-endian float32 rows
written TOP-to-bottom (no flip). PNG: 8/16-bit gray / RGB(A), returned in OpenCV's BGR order
(cv::imread IMREAD_UNCHANGED, CvUtil.cpp:23-29).
"""
import struct
import zlib

import numpy as np


def write_pfm(path, m):
    m = np.ascontiguousarray(m, dtype="<f4")
    h, w = m.shape
    with open(path, "wb") as f:
        f.write(b"Pf\n")
        f.write(("%d %d\n" % (w, h)).encode())
        f.write(b"-1.0\n")
        f.write(m.tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        data = f.read()
    nl1 = data.index(b"\n")
    if data[:nl1] != b"Pf":
        raise ValueError("expected 'Pf' in 1-channel .pfm file header: %s" % path)
    nl2 = data.index(b"\n", nl1 + 1)
    w, h = map(int, data[nl1 + 1:nl2].split())
    nl3 = data.index(b"\n", nl2 + 1)
    if float(data[nl2 + 1:nl3]) > 0:
        raise ValueError("only little endian .pfm files supported: %s" % path)
    return np.frombuffer(data, dtype="<f4", count=w * h, offset=nl3 + 1).reshape(h, w).copy()


def read_exr(path):
    """Single-part scan-line OpenEXR with one FLOAT channel (what cv::imwrite writes for a CV_32FC1 disparity and
    what this build's executables write for --output_formats=exr): NO / ZIPS / ZIP compression."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x76\x2f\x31\x01":
        raise ValueError("not an OpenEXR file: %s" % path)
    version, = struct.unpack_from("<I", data, 4)
    if version & 0xFF != 2 or version & 0x1A00:  # tiled / multi-part / deep
        raise ValueError("unsupported OpenEXR flavour %#x: %s" % (version, path))
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos)
        name = data[pos:e].decode()
        e2 = data.index(b"\0", e + 1)
        typ = data[e + 1:e2].decode()
        size, = struct.unpack_from("<i", data, e2 + 1)
        attrs[name] = (typ, data[e2 + 5:e2 + 5 + size])
        pos = e2 + 5 + size
    pos += 1
    ch = attrs["channels"][1]
    e = ch.index(b"\0")
    ptype, = struct.unpack_from("<i", ch, e + 1)
    if ch[e + 17:] != b"\0" or ptype != 2:
        raise ValueError("expected exactly one FLOAT channel: %s" % path)
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines = {0: 1, 2: 1, 3: 16}.get(comp)
    if lines is None:
        raise ValueError("unsupported compression %d: %s" % (comp, path))
    blocks = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % blocks, data, pos)
    out = np.empty((h, w), dtype=np.float32)
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, off)
        n = min(lines, y1 - y + 1)
        raw = n * w * 4
        buf = data[off + 8:off + 8 + size]
        if comp != 0 and size < raw:
            t = np.frombuffer(zlib.decompress(buf), dtype=np.uint8).astype(np.int64)
            t = ((np.cumsum(t - 128) + 128) & 0xFF).astype(np.uint8)  # undo the predictor
            half = (raw + 1) // 2
            px = np.empty(raw, dtype=np.uint8)
            px[0::2], px[1::2] = t[:half], t[half:]
            buf = px.tobytes()
        out[y - y0:y - y0 + n] = np.frombuffer(buf, dtype="<f4", count=n * w).reshape(n, w)
    return out


def _chunk(tag, payload):
    return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)


def _write_png(path, arr, bitdepth):
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, ch = arr.shape
    color_type = {1: 0, 3: 2, 4: 6}[ch]
    if ch >= 3:  # BGR(A) -> RGB(A)
        arr = arr[..., [2, 1, 0] + ([3] if ch == 4 else [])]
    dt = ">u2" if bitdepth == 16 else "u1"
    raw = np.ascontiguousarray(arr.astype(dt)).reshape(h, -1).view(np.uint8)
    rows = np.concatenate([np.zeros((h, 1), dtype=np.uint8), raw], axis=1)  # filter type 0
    png = b"\x89PNG\r\n\x1a\n"
    png += _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bitdepth, color_type, 0, 0, 0))
    png += _chunk(b"IDAT", zlib.compress(rows.tobytes(), 3))
    png += _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def write_png16(path, arr):
    _write_png(path, arr, 16)


def write_png8(path, arr):
    _write_png(path, arr, 8)


def read_png(path):
    """-> uint8 / uint16 array [h, w] or [h, w, 3|4] in BGR(A) order."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG: %s" % path)
    pos, idat, ihdr = 8, [], None
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, bitdepth, color_type, _, _, interlace = ihdr
    if interlace or bitdepth not in (8, 16) or color_type not in (0, 2, 6):
        raise ValueError("unsupported PNG flavour: %s" % path)
    ch = {0: 1, 2: 3, 6: 4}[color_type]
    bpp = ch * bitdepth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        ft = raw[y, 0]
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for i in range(bpp, w * bpp):
                cur[i] = (cur[i] + cur[i - bpp]) & 255
        elif ft == 3:
            cur = line.copy()
            for i in range(w * bpp):
                left = cur[i - bpp] if i >= bpp else 0
                cur[i] = (cur[i] + ((left + prev[i]) >> 1)) & 255
        elif ft == 4:
            cur = line.copy()
            for i in range(w * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (cur[i] + pr) & 255
        else:
            raise ValueError("bad PNG filter")
        out[y] = cur
        prev = cur
    arr = out.view(">u2").astype(np.uint16) if bitdepth == 16 else out
    arr = arr.reshape(h, w, ch)
    if ch >= 3:
        arr = arr[..., [2, 1, 0] + ([3] if ch == 4 else [])]
    return np.ascontiguousarray(arr[..., 0] if ch == 1 else arr)


def read_image(path):
    """What cv2.imread(path, cv2.IMREAD_UNCHANGED) returns (scripts/render/resize.py:66-70) for a PNG / JPEG / TIFF /
    BMP / PNM file — uint8 / uint16 / float32, [h, w] or [h, w, 3|4] in B, G, R [, A] order — decoded by the C-ABI's
    derp_image_decode (cli/image_codecs.h; the decoder is chosen by the file's signature, JPEG samples are
    libjpeg-turbo's integers). Raises ValueError with the decoder's message for anything it refuses."""
    import ctypes as C

    from . import derp

    lib = derp.lib()
    lib.derp_image_last_error.restype = C.c_char_p
    with open(path, "rb") as f:
        data = f.read()
    w, h, ch, bd = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if lib.derp_image_info(data, C.c_size_t(len(data)), C.byref(w), C.byref(h), C.byref(ch), C.byref(bd)):
        raise ValueError("failed to load image: %s (%s)" % (path, lib.derp_image_last_error().decode()))
    out = np.empty((h.value, w.value, ch.value), dtype=np.float32 if bd.value == 32 else np.uint16)
    if lib.derp_image_decode(data, C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)):
        raise ValueError("failed to load image: %s (%s)" % (path, lib.derp_image_last_error().decode()))
    if bd.value == 8:
        out = out.astype(np.uint8)
    return np.ascontiguousarray(out[..., 0] if ch.value == 1 else out)


def write_jpeg(path, arr, quality=95):
    """cv2.imwrite(path_with_a_jpeg_extension, arr) with OpenCV's defaults: uint8 [h, w] or [h, w, 3] (B, G, R) through
    the C-ABI's derp_jpeg_encode — byte for byte what libjpeg-turbo writes at that quality (baseline, 4:2:0)."""
    import ctypes as C

    from . import derp

    a = np.ascontiguousarray(arr)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("JPEG holds 8-bit gray or 3-channel images")
    lib = derp.lib()
    lib.derp_image_last_error.restype = C.c_char_p
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else 3
    size = C.c_size_t(0)
    cap = w * h * ch * 4 + 4096  # a coefficient costs at most a 16-bit code + 11 bits: under four bytes per sample
    buf = (C.c_ubyte * cap)()
    if lib.derp_jpeg_encode(a.ctypes.data_as(C.c_void_p), w, h, ch, int(quality), buf, C.c_size_t(cap), C.byref(size)):
        raise ValueError("failed to save image: %s (%s)" % (path, lib.derp_image_last_error().decode()))
    with open(path, "wb") as f:
        f.write(memoryview(buf)[: size.value])


def write_tiff(path, arr):
    """A baseline little-endian TIFF, one Deflate-compressed strip: uint8 / uint16 [h, w] or [h, w, 3|4] in B, G, R [, A]
    order (stored R, G, B [, A]; a fourth sample is declared as un-associated alpha), float32 [h, w]. Lossless, so
    which TIFF flavour cv2.imwrite would have picked (LZW) does not matter to a reader."""
    a = np.asarray(arr)
    if a.ndim == 2:
        a = a[..., None]
    h, w, spp = a.shape
    if spp >= 3:
        a = a[..., [2, 1, 0] + ([3] if spp == 4 else [])]
    fmt = 3 if a.dtype == np.float32 else 1
    bits = a.dtype.itemsize * 8
    strip = zlib.compress(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes(), 6)
    tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [bits] * spp), (259, 3, [8]), (262, 3, [2 if spp >= 3 else 1]), (273, 4, [8]),
            (277, 3, [spp]), (278, 4, [h]), (279, 4, [len(strip)]), (284, 3, [1]), (339, 3, [fmt] * spp)]
    if spp == 4:
        tags.append((338, 3, [2]))
    tags.sort()
    body = strip + (b"\0" if len(strip) & 1 else b"")
    ifd_at = 8 + len(body)
    extra_at = ifd_at + 2 + 12 * len(tags) + 4
    ifd, extra = struct.pack("<H", len(tags)), b""
    for tag, typ, vals in tags:
        blob = struct.pack("<" + {3: "H", 4: "I"}[typ] * len(vals), *vals)
        if len(blob) <= 4:
            ifd += struct.pack("<HHI", tag, typ, len(vals)) + blob.ljust(4, b"\0")
        else:
            ifd += struct.pack("<HHII", tag, typ, len(vals), extra_at + len(extra))
            extra += blob + (b"\0" if len(blob) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_at) + body + ifd + struct.pack("<I", 0) + extra)


def load_color_u16(path):
    """cv_util::loadImage<Vec3w> (CvUtil.h:226-284): cv::imread(IMREAD_UNCHANGED) of any format it reads, depth -> 16U
    (x257 from 8-bit), channels -> BGR (gray replicated, alpha dropped)."""
    a = read_image(path)
    if a.dtype == np.float32:  # convertTo(CV_16U, 65535 / 1.0): saturate_cast of the value rounded half to even
        a = np.clip(np.rint(np.nan_to_num(a.astype(np.float64), nan=0.0) * 65535.0), 0, 65535).astype(np.uint16)
        if a.ndim == 3 and a.shape[2] == 1:  # (multi-channel floats go through the channel handling below, like convertImage)
            a = a[..., 0]
    if a.dtype == np.uint8:
        a = a.astype(np.uint16) * 257
    if a.ndim == 2:
        a = np.repeat(a[..., None], 3, axis=2)
    return np.ascontiguousarray(a[..., :3])


def load_mask(path):
    """cv_util::loadImage<bool> (CvUtil.h:226-262): 8-bit, threshold > 127 -> 1 per channel, then (3 / 4 channels)
    COLOR_BGR[A]2GRAY of the 0 / 1 values, whose rounded fixed-point weights give 1 exactly when the GREEN channel's bit
    is set (G alone rounds to 1, B + R together to 0)."""
    a = read_image(path)
    if a.dtype == np.float32:  # convertTo(CV_8U, 255 / 1.0), then the threshold
        a = np.clip(np.rint(np.nan_to_num(a.astype(np.float64), nan=0.0) * 255.0), 0, 255).astype(np.uint8)
        if a.ndim == 3 and a.shape[2] == 1:
            a = a[..., 0]
    if a.ndim == 3:
        a = a[..., 1]
    if a.dtype == np.uint16:
        a = np.rint(a.astype(np.float64) * (255.0 / 65535.0)).astype(np.uint8)
    return (a > 127).astype(np.uint8)
