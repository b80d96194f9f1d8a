// Raster decoders of the executables' input side: what `cv::imread(path, cv::IMREAD_UNCHANGED)` hands
// cv_util::loadImage (CvUtil.cpp:23-29, CvUtil.h:268-284) for the file kinds a colour / mask / disparity directory may
// hold — PNG (every colour type, bit depth and Adam7), JPEG (baseline + progressive Huffman, 8-bit), TIFF (strips and
// tiles; none / LZW / Deflate / PackBits; predictor 2; 8 / 16-bit unsigned, 32-bit float), BMP, PNM. Like OpenCV the
// decoder is chosen by the file's signature, not by its extension. Self-contained C++17 + zlib (the reference links
// OpenCV's imgcodecs, i.e. libpng / libjpeg-turbo / libtiff): written from the formats' specifications; the JPEG
// path restates libjpeg's default decompression arithmetic (the "islow" 13-bit integer IDCT, triangle-filter
// "fancy" chroma upsampling, 16-bit fixed-point YCbCr -> RGB) because a lossy format is only a drop-in when the
// samples are the same integers — pinned against Pillow's libjpeg-turbo in tests/test_image_codecs.py.
// Errors are exceptions (codecs::Error); cli_common.h turns them into the glog-style fatal line with the file name.
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace codecs {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline void need(bool ok, const char* what) {
  if (!ok) {
    throw Error(what);
  }
}

// channels 1 / 3 / 4 in FILE order (R, G, B [, A]) — callers swap to OpenCV's BGR; bitdepth 8 or 16 (8-bit samples are
// stored widened, not scaled) or 32 = float samples in `f32` (TIFF SampleFormat 3 only)
struct Raster {
  int w = 0, h = 0, channels = 0, bitdepth = 0;
  std::vector<uint16_t> px;
  std::vector<float> f32;
};

struct Bytes {
  const unsigned char* d;
  size_t n;
  void span(size_t off, size_t len, const char* what) const { need(off <= n && len <= n - off, what); }
};
// OpenCV's own limits (CV_IO_MAX_IMAGE_WIDTH / _HEIGHT 2^20, CV_IO_MAX_IMAGE_PIXELS 2^30), and a plausibility bound
// every decoder applies before it allocates: a header may not promise more samples than `perByte` per byte of
// compressed data can deliver (deflate expands at most 1032 : 1, a TIFF LZW code at most 4 KB, a JPEG block of 64 pixels needs
// a bit or two) — a corrupt or hostile header fails with a message instead of with an out-of-memory abort.
inline void plausible(int64_t w, int64_t h, size_t bytesPromised, size_t bytesPresent, size_t perByte) {
  need(w > 0 && h > 0 && w <= (1 << 20) && h <= (1 << 20) && w * h <= ((int64_t)1 << 30), "image size outside 1 .. 2^20 x 2^20, 2^30 pixels");
  need(bytesPromised / perByte <= bytesPresent + 1024, "corrupt image: the header promises more data than the file can hold");
}

inline uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline uint32_t be16(const unsigned char* p) { return (uint32_t(p[0]) << 8) | p[1]; }

// zlib streams are inflated by libdeflate when the system has it (libdeflate.so.0 — libtiff's own dependency on this
// image; 1.7 x zlib's inflate on 16-bit camera PNGs, and PNG inflation is what DerpSequence's disk-to-disk time is
// made of on a 16-CPU box), found with dlopen so that nothing links against it; zlib otherwise, and whenever libdeflate
// objects to a stream (it insists on the exact size and a valid Adler-32; libpng / libtiff are more forgiving).
// DERP_NO_LIBDEFLATE=1 forces zlib.
struct LibDeflate {
  void* (*alloc)() = nullptr;
  int (*zlibDecompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
  void (*release)(void*) = nullptr;
  LibDeflate() {
    const char* off = getenv("DERP_NO_LIBDEFLATE");
    if (off && *off && *off != '0') {
      return;
    }
    if (void* lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL)) {
      alloc = reinterpret_cast<void* (*)()>(dlsym(lib, "libdeflate_alloc_decompressor"));
      zlibDecompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(lib, "libdeflate_zlib_decompress"));
      release = reinterpret_cast<void (*)(void*)>(dlsym(lib, "libdeflate_free_decompressor"));
      if (!alloc || !zlibDecompress || !release) {
        alloc = nullptr;
      }
    }
  }
  static const LibDeflate& get() {
    static const LibDeflate one;
    return one;
  }
};
// inflates exactly `expect` bytes into `out` (caller-owned, uninitialised is fine)
inline void inflate_into(const unsigned char* src, size_t n, unsigned char* out, size_t expect, const char* what) {
  const LibDeflate& ld = LibDeflate::get();
  if (ld.alloc) {
    struct Holder {
      void* d = nullptr;
      void (*release)(void*) = nullptr;
      ~Holder() {
        if (d) {
          release(d);
        }
      }
    };
    static thread_local Holder h;
    if (!h.d) {
      h.d = ld.alloc();
      h.release = ld.release;
    }
    size_t got = 0;
    if (h.d && ld.zlibDecompress(h.d, src, n, out, expect, &got) == 0 && got == expect) {
      return;
    }
  }
  z_stream z;
  memset(&z, 0, sizeof z);
  need(inflateInit(&z) == Z_OK, what);
  z.next_in = const_cast<Bytef*>(src);
  z.avail_in = (uInt)n;
  z.next_out = out;
  z.avail_out = (uInt)expect;
  const int rc = inflate(&z, Z_FINISH);
  const size_t got = expect - z.avail_out;
  inflateEnd(&z);
  // libpng / libtiff accept a stream that fills the expected size even when trailing bytes follow
  need((rc == Z_STREAM_END || rc == Z_OK || rc == Z_BUF_ERROR) && got == expect, what);
}
inline std::vector<unsigned char> inflate_all(const unsigned char* src, size_t n, size_t expect, const char* what) {
  std::vector<unsigned char> out(expect);
  inflate_into(src, n, out.data(), expect, what);
  return out;
}

// ================================================================================================ PNG
// PNG (ISO/IEC 15948). What OpenCV's PngDecoder asks libpng for under IMREAD_UNCHANGED (grfmt_png.cpp readHeader /
// readData): palette -> RGB, gray below 8 bits -> scaled to 8, tRNS -> alpha for palette / RGB (ignored for gray),
// gray + alpha -> 4 channels, 16-bit kept.
struct PngInfo {
  int w = 0, h = 0, depth = 0, colorType = -1, interlace = 0;
};
inline bool png_header(const Bytes& b, PngInfo& info) {
  if (b.n < 33 || memcmp(b.d, "\x89PNG\r\n\x1a\n", 8) != 0 || memcmp(b.d + 12, "IHDR", 4) != 0) {
    return false;
  }
  info.w = (int)be32(b.d + 16);
  info.h = (int)be32(b.d + 20);
  info.depth = b.d[24];
  info.colorType = b.d[25];
  info.interlace = b.d[28];
  return info.w > 0 && info.h > 0;
}
inline Raster decode_png(const Bytes& b) {
  PngInfo info;
  need(png_header(b, info), "not a PNG file");
  const int ct = info.colorType, depth = info.depth;
  const bool okDepth = ct == 0   ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                       : ct == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                                 : ((ct == 2 || ct == 4 || ct == 6) && (depth == 8 || depth == 16));
  need(okDepth && info.interlace <= 1, "unsupported PNG flavour");
  std::vector<unsigned char> idat, plte, trns;
  size_t pos = 8;
  while (pos + 12 <= b.n) {
    const uint32_t len = be32(b.d + pos);
    b.span(pos + 8, (size_t)len + 4, "truncated PNG chunk");
    const unsigned char* body = b.d + pos + 8;
    if (!memcmp(b.d + pos + 4, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(b.d + pos + 4, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!memcmp(b.d + pos + 4, "tRNS", 4)) {
      trns.assign(body, body + len);
    } else if (!memcmp(b.d + pos + 4, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  const int fileCh = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : 4;
  const int bitsPerPixel = fileCh * depth;
  const int bpp = std::max(1, bitsPerPixel / 8);  // filter unit
  const int W = info.w, H = info.h;
  struct Pass {
    int x0, y0, dx, dy;
  };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole = {0, 0, 1, 1};
  const int nPass = info.interlace ? 7 : 1;
  size_t total = 0;
  for (int p = 0; p < nPass; ++p) {
    const Pass& ps = info.interlace ? adam7[p] : whole;
    const int pw = (W - ps.x0 + ps.dx - 1) / ps.dx, ph = (H - ps.y0 + ps.dy - 1) / ps.dy;
    if (pw > 0 && ph > 0) {
      total += (size_t)ph * (1 + ((size_t)pw * bitsPerPixel + 7) / 8);
    }
  }
  plausible(W, H, total, idat.size(), 1032);
  const std::vector<unsigned char> raw = inflate_all(idat.data(), idat.size(), total, "corrupt PNG (inflate)");

  const bool palette = ct == 3;
  need(!palette || plte.size() >= 3, "PNG palette missing");
  const bool trnsAlpha = !trns.empty() && (ct == 3 || (ct == 2 && trns.size() >= 6));
  Raster img;
  img.w = W;
  img.h = H;
  img.bitdepth = depth == 16 ? 16 : 8;
  img.channels = ct == 0 ? 1 : (ct == 4 || ct == 6 || trnsAlpha) ? 4 : 3;
  img.px.assign((size_t)W * H * img.channels, 0);
  const unsigned maxv = depth == 16 ? 65535u : 255u;
  const unsigned tr = ct == 2 && trnsAlpha ? be16(trns.data()) : 0, tg = ct == 2 && trnsAlpha ? be16(trns.data() + 2) : 0,
                 tb = ct == 2 && trnsAlpha ? be16(trns.data() + 4) : 0;

  size_t at = 0;
  std::vector<unsigned char> cur, prev;
  for (int p = 0; p < nPass; ++p) {
    const Pass& ps = info.interlace ? adam7[p] : whole;
    const int pw = (W - ps.x0 + ps.dx - 1) / ps.dx, ph = (H - ps.y0 + ps.dy - 1) / ps.dy;
    if (pw <= 0 || ph <= 0) {
      continue;
    }
    const size_t stride = ((size_t)pw * bitsPerPixel + 7) / 8;
    cur.assign(stride, 0);
    prev.assign(stride, 0);
    for (int yy = 0; yy < ph; ++yy) {
      const unsigned char* line = raw.data() + at;
      at += stride + 1;
      const unsigned char* in = line + 1;
      const size_t B = (size_t)bpp;
      switch (line[0]) {  // one tight loop per PNG filter type
        case 0:
          memcpy(cur.data(), in, stride);
          break;
        case 1:
          for (size_t i = 0; i < B && i < stride; ++i) {
            cur[i] = in[i];
          }
          for (size_t i = B; i < stride; ++i) {
            cur[i] = (unsigned char)(in[i] + cur[i - B]);
          }
          break;
        case 2:
          for (size_t i = 0; i < stride; ++i) {
            cur[i] = (unsigned char)(in[i] + prev[i]);
          }
          break;
        case 3:
          for (size_t i = 0; i < B && i < stride; ++i) {
            cur[i] = (unsigned char)(in[i] + (prev[i] >> 1));
          }
          for (size_t i = B; i < stride; ++i) {
            cur[i] = (unsigned char)(in[i] + ((cur[i - B] + prev[i]) >> 1));
          }
          break;
        case 4:
          for (size_t i = 0; i < B && i < stride; ++i) {
            cur[i] = (unsigned char)(in[i] + prev[i]);
          }
          for (size_t i = B; i < stride; ++i) {
            const int a = cur[i - B], bb = prev[i], c = prev[i - B];
            const int pp = a + bb - c, pa = abs(pp - a), pb = abs(pp - bb), pc = abs(pp - c);
            cur[i] = (unsigned char)(in[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c)));
          }
          break;
        default:
          throw Error("bad PNG filter type");
      }
      const int y = ps.y0 + yy * ps.dy;
      if (!info.interlace && depth >= 8 && !palette && !trnsAlpha && ct != 4) {
        // the common case (what the pipeline's own resize step writes): whole rows of 8 / 16-bit samples as they are
        uint16_t* o = &img.px[(size_t)y * W * img.channels];
        const int nv = W * img.channels;
        if (depth == 16) {
          for (int i = 0; i < nv; ++i) {
            o[i] = uint16_t((cur[2 * i] << 8) | cur[2 * i + 1]);
          }
        } else {
          for (int i = 0; i < nv; ++i) {
            o[i] = cur[i];
          }
        }
        prev.swap(cur);
        continue;
      }
      for (int xx = 0; xx < pw; ++xx) {
        const int x = ps.x0 + xx * ps.dx;
        uint16_t* o = &img.px[((size_t)y * W + x) * img.channels];
        unsigned s[4] = {0, 0, 0, 0};
        if (depth == 16) {
          for (int c = 0; c < fileCh; ++c) {
            s[c] = be16(&cur[((size_t)xx * fileCh + c) * 2]);
          }
        } else if (depth == 8) {
          for (int c = 0; c < fileCh; ++c) {
            s[c] = cur[(size_t)xx * fileCh + c];
          }
        } else {  // 1 / 2 / 4 bits, one channel, leftmost pixel in the high bits
          const size_t bit = (size_t)xx * depth;
          s[0] = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
        }
        switch (ct) {
          case 0:
            o[0] = (uint16_t)(depth < 8 ? s[0] * (255u / ((1u << depth) - 1)) : s[0]);
            break;
          case 2:
            o[0] = s[0], o[1] = s[1], o[2] = s[2];
            if (trnsAlpha) {
              o[3] = (uint16_t)((s[0] == tr && s[1] == tg && s[2] == tb) ? 0 : maxv);
            }
            break;
          case 3:
            need((size_t)s[0] * 3 + 2 < plte.size(), "PNG palette index out of range");
            o[0] = plte[s[0] * 3], o[1] = plte[s[0] * 3 + 1], o[2] = plte[s[0] * 3 + 2];
            if (trnsAlpha) {
              o[3] = s[0] < trns.size() ? trns[s[0]] : 255;
            }
            break;
          case 4:
            o[0] = o[1] = o[2] = s[0], o[3] = s[1];
            break;
          default:
            o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
        }
      }
      prev.swap(cur);
    }
  }
  return img;
}

// The pipeline's own flavour — non-interlaced 8 / 16-bit gray, RGB or RGBA without tRNS, which is what resize.py's
// cv2.imwrite leaves in color_levels/ — straight into the caller's interleaved B, G, R uint16 buffer (8-bit x 257 =
// convertTo(CV_16U, 65535 / 255); gray replicated; alpha dropped: cv_util::loadImage<Vec3w>, CvUtil.h:226-262): rows are
// unfiltered in place in the inflated buffer and converted once, no intermediate raster. false = not that flavour
// (nothing written; use decode_png). `w` / `h` return the file's size; a mismatch with expectW / expectH (when
// given, > 0) is reported by the caller.
inline bool png_fast_bgr16(const Bytes& b, uint16_t* out, int expectW, int expectH, int& w, int& h) {
  PngInfo info;
  if (!png_header(b, info) || info.interlace || (info.depth != 8 && info.depth != 16) ||
      (info.colorType != 0 && info.colorType != 2 && info.colorType != 6)) {
    return false;
  }
  const unsigned char* idat = nullptr;
  size_t idatLen = 0;
  std::vector<unsigned char> joined;
  int nIdat = 0;
  size_t pos = 8;
  while (pos + 12 <= b.n) {
    const uint32_t len = be32(b.d + pos);
    b.span(pos + 8, (size_t)len + 4, "truncated PNG chunk");
    if (!memcmp(b.d + pos + 4, "IDAT", 4)) {
      if (nIdat++ == 0) {
        idat = b.d + pos + 8;
        idatLen = len;
      } else {
        if (nIdat == 2) {
          joined.assign(idat, idat + idatLen);
        }
        joined.insert(joined.end(), b.d + pos + 8, b.d + pos + 8 + len);
      }
    } else if (!memcmp(b.d + pos + 4, "tRNS", 4) && info.colorType == 2) {
      return false;
    } else if (!memcmp(b.d + pos + 4, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (nIdat > 1) {
    idat = joined.data();
    idatLen = joined.size();
  }
  w = info.w;
  h = info.h;
  if ((expectW > 0 && w != expectW) || (expectH > 0 && h != expectH)) {
    return true;  // the caller compares sizes before it looks at `out`
  }
  const int ch = info.colorType == 0 ? 1 : info.colorType == 2 ? 3 : 4, bytes = info.depth / 8;
  const size_t B = (size_t)ch * bytes, stride = (size_t)w * B, total = (size_t)h * (stride + 1);
  plausible(w, h, total, idatLen, 1032);
  std::unique_ptr<unsigned char[]> raw(new unsigned char[total]);  // not value-initialised: inflate fills every byte
  inflate_into(idat, idatLen, raw.get(), total, "corrupt PNG (inflate)");
  const std::vector<unsigned char> zeros(stride, 0);
  const unsigned char* prev = zeros.data();
  for (int y = 0; y < h; ++y) {
    unsigned char* line = raw.get() + (size_t)y * (stride + 1);
    unsigned char* cur = line + 1;
    switch (line[0]) {
      case 0:
        break;
      case 1:
        for (size_t i = B; i < stride; ++i) {
          cur[i] = (unsigned char)(cur[i] + cur[i - B]);
        }
        break;
      case 2:
        for (size_t i = 0; i < stride; ++i) {
          cur[i] = (unsigned char)(cur[i] + prev[i]);
        }
        break;
      case 3:
        for (size_t i = 0; i < B && i < stride; ++i) {
          cur[i] = (unsigned char)(cur[i] + (prev[i] >> 1));
        }
        for (size_t i = B; i < stride; ++i) {
          cur[i] = (unsigned char)(cur[i] + ((cur[i - B] + prev[i]) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < B && i < stride; ++i) {
          cur[i] = (unsigned char)(cur[i] + prev[i]);
        }
        for (size_t i = B; i < stride; ++i) {
          const int a = cur[i - B], bb = prev[i], c = prev[i - B];
          const int pp = a + bb - c, pa = abs(pp - a), pb = abs(pp - bb), pc = abs(pp - c);
          cur[i] = (unsigned char)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c)));
        }
        break;
      default:
        throw Error("bad PNG filter type");
    }
    uint16_t* o = out + (size_t)y * w * 3;
    if (bytes == 2) {
      if (ch == 1) {
        for (int x = 0; x < w; ++x) {
          o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]);
        }
      } else {
        for (int x = 0; x < w; ++x) {
          const unsigned char* s = cur + (size_t)x * B;
          o[3 * x] = (uint16_t)((s[4] << 8) | s[5]);
          o[3 * x + 1] = (uint16_t)((s[2] << 8) | s[3]);
          o[3 * x + 2] = (uint16_t)((s[0] << 8) | s[1]);
        }
      }
    } else if (ch == 1) {
      for (int x = 0; x < w; ++x) {
        o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = (uint16_t)(cur[x] * 257);
      }
    } else {
      for (int x = 0; x < w; ++x) {
        const unsigned char* s = cur + (size_t)x * B;
        o[3 * x] = (uint16_t)(s[2] * 257);
        o[3 * x + 1] = (uint16_t)(s[1] * 257);
        o[3 * x + 2] = (uint16_t)(s[0] * 257);
      }
    }
    prev = cur;
  }
  return true;
}

// ================================================================================================ JPEG
// ITU T.81 baseline / extended-sequential / progressive Huffman decoding, 8-bit samples, 1 or 3 components. The sample
// reconstruction follows libjpeg's defaults (dct_method JDCT_ISLOW, do_fancy_upsampling, jdcolor's tables), which is
// what cv::imread gets from libjpeg(-turbo): jidctint.c's 13-bit constants, jdsample.c's h2v1 / h2v2 / h1v2 triangle
// filters with its alternating rounding biases, jdcolor.c's FIX(1.40200) ... 16-bit tables.
struct JpegComponent {
  int id = 0, h = 1, v = 1, tq = 0;
  int blocksW = 0, blocksH = 0;  // padded to whole MCUs of an interleaved scan
  int width = 0, height = 0;     // downsampled_width / _height: ceil(W * h / hmax)
  std::vector<int16_t> coef;     // [blocksH][blocksW][64], natural order
  int quant[64];
  bool quantLatched = false;
  int dcTable = 0, acTable = 0, pred = 0;
};
struct JpegHuff {
  bool present = false;
  uint8_t look[512];       // 9-bit fast path: code length (0 = longer than 9)
  uint8_t lookSym[512];
  int maxcode[18], valptr[17], mincode[17];
  uint8_t vals[256];
  void build(const uint8_t* counts, const uint8_t* symbols, int total) {
    present = true;
    memcpy(vals, symbols, (size_t)total);
    memset(look, 0, sizeof look);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      need(code + counts[len - 1] <= (1 << len), "corrupt JPEG Huffman table (over-subscribed)");
      for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
        if (len <= 9) {
          const int first = code << (9 - len);
          for (int f = 0; f < (1 << (9 - len)); ++f) {
            look[first + f] = (uint8_t)len;
            lookSym[first + f] = symbols[k];
          }
        }
      }
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};
struct JpegBits {
  const unsigned char* d;
  size_t n, pos;
  uint64_t acc = 0;
  int cnt = 0;
  bool hitMarker = false;
  void fill() {
    while (cnt <= 48) {
      unsigned byte = 0;
      if (!hitMarker && pos < n) {
        byte = d[pos];
        if (byte == 0xff) {
          if (pos + 1 < n && d[pos + 1] == 0) {
            pos += 2;
          } else {
            hitMarker = true;  // the entropy-coded segment ends here: feed zeros (jdhuff.c does the same)
            byte = 0;
          }
        } else {
          ++pos;
        }
      }
      acc = (acc << 8) | byte;
      cnt += 8;
    }
  }
  int peek(int nb) {
    if (cnt < nb) {
      fill();
    }
    return (int)((acc >> (cnt - nb)) & ((1u << nb) - 1));
  }
  void skip(int nb) { cnt -= nb; }
  int get(int nb) {
    if (nb == 0) {
      return 0;
    }
    const int v = peek(nb);
    cnt -= nb;
    return v;
  }
  int decode(const JpegHuff& t) {
    const int p = peek(9);
    if (t.look[p]) {
      cnt -= t.look[p];
      return t.lookSym[p];
    }
    int code = peek(16), len = 10;
    for (; len <= 16; ++len) {
      const int c = code >> (16 - len);
      if (t.maxcode[len] >= 0 && c <= t.maxcode[len] && c >= t.mincode[len]) {
        cnt -= len;
        return t.vals[t.valptr[len] + c - t.mincode[len]];
      }
    }
    throw Error("corrupt JPEG data: bad Huffman code");
  }
  static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
  void restart() {  // byte-align, step over the RSTn marker
    acc = 0;
    cnt = 0;
    hitMarker = false;
    while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7)) {
      ++pos;
    }
    if (pos + 1 < n) {
      pos += 2;
    }
  }
};
static const uint8_t kJpegZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                             6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                             39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct JpegInfo {
  int w = 0, h = 0, ncomp = 0, precision = 0;
  bool progressive = false;
};
// walks the marker segments up to the frame header
inline bool jpeg_header(const Bytes& b, JpegInfo& info) {
  if (b.n < 4 || b.d[0] != 0xff || b.d[1] != 0xd8) {
    return false;
  }
  size_t pos = 2;
  while (pos + 4 <= b.n) {
    if (b.d[pos] != 0xff) {
      ++pos;
      continue;
    }
    const int m = b.d[pos + 1];
    if (m == 0xff) {
      ++pos;
      continue;
    }
    if (m == 0xd8 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) {
      pos += 2;
      continue;
    }
    const size_t len = be16(b.d + pos + 2);
    if (m >= 0xc0 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      if (pos + 2 + len > b.n || len < 8) {
        return false;
      }
      info.precision = b.d[pos + 4];
      info.h = (int)be16(b.d + pos + 5);
      info.w = (int)be16(b.d + pos + 7);
      info.ncomp = b.d[pos + 9];
      info.progressive = m == 0xc2;
      return info.w > 0 && info.h > 0;
    }
    pos += 2 + len;
  }
  return false;
}

inline void jpeg_idct_islow(const int16_t* coef, const int* quant, uint8_t* out, int stride) {
  // jidctint.c (CONST_BITS 13, PASS1_BITS 2); without its all-zero-AC shortcuts, which produce the same integers
  const int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  int64_t ws[64];
  auto descale = [](int64_t x, int nb) { return (x + ((int64_t)1 << (nb - 1))) >> nb; };
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 8; ++i) {
      int64_t in[8];
      for (int k = 0; k < 8; ++k) {
        in[k] = pass == 0 ? (int64_t)coef[k * 8 + i] * quant[k * 8 + i] : ws[i * 8 + k];
      }
      int64_t z2 = in[2], z3 = in[6];
      int64_t z1 = (z2 + z3) * F0541;
      int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
      z2 = in[0];
      z3 = in[4];
      int64_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = in[7], tmp1 = in[5], tmp2 = in[3], tmp3 = in[1];
      z1 = tmp0 + tmp3;
      z2 = tmp1 + tmp2;
      z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1175;
      tmp0 *= F0298, tmp1 *= F2053, tmp2 *= F3072, tmp3 *= F1501;
      z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
      z3 += z5;
      z4 += z5;
      tmp0 += z1 + z3;
      tmp1 += z2 + z4;
      tmp2 += z2 + z3;
      tmp3 += z1 + z4;
      const int64_t r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
      if (pass == 0) {
        for (int k = 0; k < 8; ++k) {
          ws[k * 8 + i] = descale(r[k], 13 - 2);
        }
      } else {
        for (int k = 0; k < 8; ++k) {
          // range_limit[x & RANGE_MASK] of jdmaster.c's post-IDCT table (centre 128, wraps beyond +-512)
          const int x = (int)(descale(r[k], 13 + 2 + 3) & 1023);
          out[i * stride + k] = (uint8_t)(x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896);
        }
      }
    }
  }
}

inline Raster decode_jpeg(const Bytes& b) {
  JpegInfo info;
  need(jpeg_header(b, info), "not a JPEG file");
  int qt[4][64];
  bool qtSeen[4] = {false, false, false, false};
  JpegHuff dc[4], ac[4];
  std::vector<JpegComponent> comps;
  int W = 0, H = 0, hmax = 1, vmax = 1, restartInterval = 0, adobeTransform = -1;
  bool jfif = false, progressive = false, sawSof = false, done = false;
  size_t pos = 2;
  while (!done && pos + 4 <= b.n) {
    if (b.d[pos] != 0xff) {
      ++pos;
      continue;
    }
    const int m = b.d[pos + 1];
    if (m == 0xff) {
      ++pos;
      continue;
    }
    if (m == 0xd9) {
      break;
    }
    if (m == 0xd8 || m == 0x01 || m == 0x00 || (m >= 0xd0 && m <= 0xd7)) {
      pos += 2;
      continue;
    }
    const size_t len = be16(b.d + pos + 2);
    b.span(pos + 2, len, "truncated JPEG segment");
    need(len >= 2, "corrupt JPEG segment");
    const unsigned char* s = b.d + pos + 4;
    const size_t sl = len - 2;
    switch (m) {
      case 0xe0:
        jfif = jfif || (sl >= 5 && !memcmp(s, "JFIF", 5));
        break;
      case 0xee:
        if (sl >= 12 && !memcmp(s, "Adobe", 5)) {
          adobeTransform = s[11];
        }
        break;
      case 0xdb:
        for (size_t i = 0; i < sl;) {
          const int pq = s[i] >> 4, tq = s[i] & 15;
          need(tq < 4 && pq <= 1 && i + 1 + 64 * (pq + 1) <= sl, "corrupt JPEG quantisation table");
          for (int k = 0; k < 64; ++k) {
            qt[tq][kJpegZigzag[k]] = pq ? (int)be16(s + i + 1 + 2 * k) : s[i + 1 + k];
          }
          qtSeen[tq] = true;
          i += 1 + 64 * (pq + 1);
        }
        break;
      case 0xc4:
        for (size_t i = 0; i < sl;) {
          need(i + 17 <= sl, "corrupt JPEG Huffman table");
          const int tc = s[i] >> 4, th = s[i] & 15;
          int total = 0;
          for (int k = 0; k < 16; ++k) {
            total += s[i + 1 + k];
          }
          need(tc <= 1 && th < 4 && total <= 256 && i + 17 + total <= sl, "corrupt JPEG Huffman table");
          (tc ? ac : dc)[th].build(s + i + 1, s + i + 17, total);
          i += 17 + total;
        }
        break;
      case 0xdd:
        need(sl >= 2, "corrupt JPEG restart interval");
        restartInterval = (int)be16(s);
        break;
      case 0xc0:
      case 0xc1:
      case 0xc2: {
        need(!sawSof, "JPEG with more than one frame");
        sawSof = true;
        progressive = m == 0xc2;
        need(sl >= 6 && s[0] == 8, "unsupported JPEG sample precision (8-bit only)");
        H = (int)be16(s + 1);
        W = (int)be16(s + 3);
        const int nc = s[5];
        need(W > 0 && H > 0 && (nc == 1 || nc == 3) && sl >= (size_t)6 + 3 * nc,
             nc == 4 ? "unsupported JPEG colour space (CMYK / YCCK)" : "unsupported JPEG frame");
        comps.resize(nc);
        for (int c = 0; c < nc; ++c) {
          comps[c].id = s[6 + 3 * c];
          comps[c].h = s[7 + 3 * c] >> 4;
          comps[c].v = s[7 + 3 * c] & 15;
          comps[c].tq = s[8 + 3 * c];
          need(comps[c].h >= 1 && comps[c].h <= 4 && comps[c].v >= 1 && comps[c].v <= 4 && comps[c].tq < 4, "corrupt JPEG frame header");
          hmax = std::max(hmax, comps[c].h);
          vmax = std::max(vmax, comps[c].v);
        }
        if (nc == 1) {  // a single-component frame is never interleaved: its sampling factors do not matter
          comps[0].h = comps[0].v = hmax = vmax = 1;
        }
        plausible(W, H, (size_t)W * H, b.n, 1024);  // a flat image costs libjpeg 2 bits per 8 x 8 block: 256 px per byte
        const int mcusX = (W + 8 * hmax - 1) / (8 * hmax), mcusY = (H + 8 * vmax - 1) / (8 * vmax);
        for (auto& c : comps) {
          c.blocksW = mcusX * c.h;
          c.blocksH = mcusY * c.v;
          c.width = (W * c.h + hmax - 1) / hmax;
          c.height = (H * c.v + vmax - 1) / vmax;
          c.coef.assign((size_t)c.blocksW * c.blocksH * 64, 0);
        }
        break;
      }
      case 0xc3:
      case 0xc5:
      case 0xc6:
      case 0xc7:
      case 0xc9:
      case 0xca:
      case 0xcb:
      case 0xcd:
      case 0xce:
      case 0xcf:
        throw Error("unsupported JPEG process (lossless / hierarchical / arithmetic coding)");
      case 0xda: {
        need(sawSof && sl >= 1, "JPEG scan before the frame header");
        const int ns = s[0];
        need(ns >= 1 && ns <= (int)comps.size() && sl >= (size_t)4 + 2 * ns, "corrupt JPEG scan header");
        JpegComponent* sc[4];
        for (int i = 0; i < ns; ++i) {
          sc[i] = nullptr;
          for (auto& c : comps) {
            if (c.id == s[1 + 2 * i]) {
              sc[i] = &c;
            }
          }
          need(sc[i] != nullptr, "JPEG scan names an unknown component");
          sc[i]->dcTable = s[2 + 2 * i] >> 4;
          sc[i]->acTable = s[2 + 2 * i] & 15;
          need(sc[i]->dcTable < 4 && sc[i]->acTable < 4, "corrupt JPEG scan header");
          if (!sc[i]->quantLatched) {  // jdinput.c latch_quant_tables: the table in force at the component's first scan
            need(qtSeen[sc[i]->tq], "JPEG quantisation table missing");
            memcpy(sc[i]->quant, qt[sc[i]->tq], sizeof qt[0]);
            sc[i]->quantLatched = true;
          }
          sc[i]->pred = 0;
        }
        const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
        if (progressive) {
          need(Ss <= Se && Se <= 63 && (Ss == 0 ? Se == 0 : ns == 1) && Al <= 13, "corrupt progressive JPEG scan");
        }
        const int sS = progressive ? Ss : 0, sE = progressive ? Se : 63;
        for (int i = 0; i < ns; ++i) {
          need((sS > 0 || Ah != 0 || dc[sc[i]->dcTable].present) && (sE == 0 || ac[sc[i]->acTable].present),
               "JPEG Huffman table missing");
        }
        JpegBits br{b.d, b.n, pos + 2 + len};
        int eobrun = 0;
        // a single-component scan walks the component's own blocks (its unpadded extent), an interleaved one MCUs
        const int mcusX = ns == 1 ? (sc[0]->width + 7) / 8 : (W + 8 * hmax - 1) / (8 * hmax);
        const int mcusY = ns == 1 ? (sc[0]->height + 7) / 8 : (H + 8 * vmax - 1) / (8 * vmax);
        int untilRestart = restartInterval;
        for (int my = 0; my < mcusY; ++my) {
          for (int mx = 0; mx < mcusX; ++mx) {
            if (restartInterval && untilRestart == 0) {
              br.restart();
              for (int i = 0; i < ns; ++i) {
                sc[i]->pred = 0;
              }
              eobrun = 0;
              untilRestart = restartInterval;
            }
            --untilRestart;
            for (int i = 0; i < ns; ++i) {
              JpegComponent& c = *sc[i];
              const int bh = ns == 1 ? 1 : c.h, bv = ns == 1 ? 1 : c.v;
              for (int by = 0; by < bv; ++by) {
                for (int bx = 0; bx < bh; ++bx) {
                  int16_t* blk = &c.coef[((size_t)(my * bv + by) * c.blocksW + (mx * bh + bx)) * 64];
                  if (!progressive) {
                    const int t = br.decode(dc[c.dcTable]);
                    need(t <= 15, "corrupt JPEG data: bad DC size");
                    c.pred += t ? JpegBits::extend(br.get(t), t) : 0;
                    blk[0] = (int16_t)c.pred;
                    for (int k = 1; k < 64;) {
                      const int rs = br.decode(ac[c.acTable]), r = rs >> 4, sz = rs & 15;
                      if (sz == 0) {
                        if (r != 15) {
                          break;
                        }
                        k += 16;
                        continue;
                      }
                      k += r;
                      blk[kJpegZigzag[k]] = (int16_t)JpegBits::extend(br.get(sz), sz);
                      ++k;
                    }
                  } else if (Ss == 0) {
                    if (Ah == 0) {
                      const int t = br.decode(dc[c.dcTable]);
                      need(t <= 15, "corrupt JPEG data: bad DC size");
                      c.pred += t ? JpegBits::extend(br.get(t), t) : 0;
                      blk[0] = (int16_t)(c.pred * (1 << Al));
                    } else if (br.get(1)) {
                      blk[0] = (int16_t)(blk[0] | (1 << Al));
                    }
                  } else if (Ah == 0) {
                    if (eobrun > 0) {
                      --eobrun;
                    } else {
                      for (int k = Ss; k <= Se; ++k) {
                        const int rs = br.decode(ac[c.acTable]), r = rs >> 4, sz = rs & 15;
                        if (sz) {
                          k += r;
                          blk[kJpegZigzag[k]] = (int16_t)(JpegBits::extend(br.get(sz), sz) * (1 << Al));
                        } else if (r == 15) {
                          k += 15;
                        } else {
                          eobrun = (1 << r) + (r ? br.get(r) : 0) - 1;
                          break;
                        }
                      }
                    }
                  } else {  // AC refinement (jdphuff.c decode_mcu_AC_refine)
                    const int p1 = 1 << Al, m1 = -(1 << Al);
                    int k = Ss;
                    auto refine = [&](int16_t& v) {
                      if (br.get(1) && (v & p1) == 0) {
                        v = (int16_t)(v + (v >= 0 ? p1 : m1));
                      }
                    };
                    if (eobrun == 0) {
                      for (; k <= Se; ++k) {
                        const int rs = br.decode(ac[c.acTable]);
                        int r = rs >> 4, sz = rs & 15, val = 0;
                        if (sz) {
                          val = br.get(1) ? p1 : m1;
                        } else if (r != 15) {
                          eobrun = (1 << r) + (r ? br.get(r) : 0);
                          break;
                        }
                        do {
                          int16_t& v = blk[kJpegZigzag[k]];
                          if (v != 0) {
                            refine(v);
                          } else if (--r < 0) {
                            break;
                          }
                          ++k;
                        } while (k <= Se);
                        if (val) {
                          blk[kJpegZigzag[k]] = (int16_t)val;
                        }
                      }
                    }
                    if (eobrun > 0) {
                      for (; k <= Se; ++k) {
                        int16_t& v = blk[kJpegZigzag[k]];
                        if (v != 0) {
                          refine(v);
                        }
                      }
                      --eobrun;
                    }
                  }
                }
              }
            }
          }
        }
        pos = br.pos;  // the bit reader stops in front of the marker that ended the segment
        continue;
      }
      default:
        break;
    }
    pos += 2 + len;
  }
  need(sawSof, "JPEG without a frame header");
  for (auto& c : comps) {
    need(c.quantLatched, "JPEG component without a scan");
  }

  // ---- samples: IDCT per block, then (for chroma) upsampling to the full grid
  struct Plane {
    int w, h;  // padded to whole blocks
    std::vector<uint8_t> s;
  };
  std::vector<Plane> planes(comps.size());
  for (size_t ci = 0; ci < comps.size(); ++ci) {
    JpegComponent& c = comps[ci];
    Plane& p = planes[ci];
    p.w = c.blocksW * 8;
    p.h = c.blocksH * 8;
    p.s.resize((size_t)p.w * p.h);
    for (int by = 0; by < c.blocksH; ++by) {
      for (int bx = 0; bx < c.blocksW; ++bx) {
        jpeg_idct_islow(&c.coef[((size_t)by * c.blocksW + bx) * 64], c.quant, &p.s[(size_t)by * 8 * p.w + bx * 8], p.w);
      }
    }
  }
  Raster img;
  img.w = W;
  img.h = H;
  img.bitdepth = 8;
  img.channels = (int)comps.size();
  img.px.resize((size_t)W * H * img.channels);
  std::vector<std::vector<uint8_t>> full(comps.size());
  for (size_t ci = 0; ci < comps.size(); ++ci) {
    const JpegComponent& c = comps[ci];
    const Plane& p = planes[ci];
    std::vector<uint8_t>& o = full[ci];
    o.resize((size_t)W * H);
    const int hx = hmax / c.h, vx = vmax / c.v;
    need(hmax % c.h == 0 && vmax % c.v == 0, "unsupported JPEG sampling factors");
    const int cw = c.width, chh = c.height;
    auto row = [&](int y) { return &p.s[(size_t)std::min(std::max(y, 0), chh - 1) * p.w]; };  // jdmainct.c: edge rows replicated
    if (hx == 1 && vx == 1) {
      for (int y = 0; y < H; ++y) {
        memcpy(&o[(size_t)y * W], row(y), (size_t)W);
      }
    } else if (hx == 2 && vx == 1 && cw > 2) {  // h2v1_fancy_upsample
      std::vector<uint8_t> line((size_t)cw * 2);
      for (int y = 0; y < H; ++y) {
        const uint8_t* in = row(y);
        line[0] = in[0];
        line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int x = 1; x < cw - 1; ++x) {
          line[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
          line[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
        }
        line[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
        line[2 * cw - 1] = in[cw - 1];
        memcpy(&o[(size_t)y * W], line.data(), (size_t)W);
      }
    } else if (hx == 1 && vx == 2) {  // h1v2_fancy_upsample
      for (int y = 0; y < H; ++y) {
        const uint8_t* in0 = row(y >> 1);
        const uint8_t* in1 = (y & 1) ? row((y >> 1) + 1) : row((y >> 1) - 1);
        const int bias = (y & 1) ? 2 : 1;
        for (int x = 0; x < W; ++x) {
          o[(size_t)y * W + x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
        }
      }
    } else if (hx == 2 && vx == 2 && cw > 2) {  // h2v2_fancy_upsample
      std::vector<uint8_t> line((size_t)cw * 2);
      for (int y = 0; y < H; ++y) {
        const uint8_t* in0 = row(y >> 1);
        const uint8_t* in1 = (y & 1) ? row((y >> 1) + 1) : row((y >> 1) - 1);
        int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
        line[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
        line[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
        for (int x = 1; x < cw - 1; ++x) {
          nextcol = in0[x + 1] * 3 + in1[x + 1];
          line[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          line[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
        }
        line[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        line[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        memcpy(&o[(size_t)y * W], line.data(), (size_t)W);
      }
    } else {  // int_upsample / h2v1_upsample / h2v2_upsample: replication
      for (int y = 0; y < H; ++y) {
        const uint8_t* in = row(y / vx);
        for (int x = 0; x < W; ++x) {
          o[(size_t)y * W + x] = in[x / hx];
        }
      }
    }
  }
  if (comps.size() == 1) {
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      img.px[i] = full[0][i];
    }
    return img;
  }
  // jdapimin.c default_decompress_parms: JFIF -> YCbCr; Adobe transform 0 -> RGB, 1 -> YCbCr; neither marker: RGB if the
  // component ids spell 'R' 'G' 'B', else YCbCr
  const bool ycc = jfif ? true : adobeTransform >= 0 ? adobeTransform != 0 : !(comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
  auto clamp8 = [](int v) { return v < 0 ? 0 : v > 255 ? 255 : v; };
  for (size_t i = 0; i < (size_t)W * H; ++i) {
    const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
    if (!ycc) {
      img.px[3 * i] = (uint16_t)y, img.px[3 * i + 1] = full[1][i], img.px[3 * i + 2] = full[2][i];
      continue;
    }
    // jdcolor.c build_ycc_rgb_table: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 32768, arithmetic right shifts
    const int r = y + (int)((91881 * (int64_t)cr + 32768) >> 16);
    const int g = y + (int)((-22554 * (int64_t)cb + 32768 + -46802 * (int64_t)cr) >> 16);
    const int bl = y + (int)((116130 * (int64_t)cb + 32768) >> 16);
    img.px[3 * i] = (uint16_t)clamp8(r), img.px[3 * i + 1] = (uint16_t)clamp8(g), img.px[3 * i + 2] = (uint16_t)clamp8(bl);
  }
  return img;
}

// ---- JPEG encoder: what cv::imwrite(".jpg", 8-bit image) leaves behind with its defaults (quality 95, baseline,
// 4:2:0, the standard's Huffman tables, no restart markers) — libjpeg's compressor restated: jccolor.c's 16-bit RGB ->
// YCbCr tables, jcsample.c's h2v2 box filter with its alternating 1 / 2 rounding bias over edge-replicated rows,
// jfdctint.c's "islow" forward DCT, jcdctmgr.c's round-half-up quantisation, jccoefct.c's dummy blocks (zero AC, the
// neighbour's DC) beyond the image, jchuff.c's entropy coder; the tables are ITU T.81 Annex K's. The pyramid builder
// uses it so that a JPEG source directory yields JPEG level directories like scripts/render/resize.py:82-85 does;
// byte-identical to libjpeg-turbo's output (tests/test_image_codecs.py compares with Pillow's encoder).
static const uint8_t kJpegStdLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                                          69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                                          81, 104, 113, 92, 49, 64,  78,  87,  103, 121, 120, 101, 72, 92, 95,  98,  112, 100, 103, 99};
static const uint8_t kJpegStdChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                            99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                            99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t kJpegStdBits[4][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},      // DC luminance
                                            {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125},    // AC luminance
                                            {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},      // DC chrominance
                                            {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}};   // AC chrominance
static const uint8_t kJpegStdDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kJpegStdAcLuma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
static const uint8_t kJpegStdAcChroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

inline void jpeg_fdct_islow(int* d) {  // jfdctint.c, in place on 64 level-shifted samples
  const int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  auto descale = [](int64_t x, int nb) { return (int)((x + ((int64_t)1 << (nb - 1))) >> nb); };
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 8; ++i) {
      int* p = pass == 0 ? d + 8 * i : d + i;
      const int st = pass == 0 ? 1 : 8;
      const int64_t t0 = p[0] + p[7 * st], t7 = p[0] - p[7 * st], t1 = p[st] + p[6 * st], t6 = p[st] - p[6 * st], t2 = p[2 * st] + p[5 * st],
                    t5 = p[2 * st] - p[5 * st], t3 = p[3 * st] + p[4 * st], t4 = p[3 * st] - p[4 * st];
      const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
      const int lo = pass == 0 ? 13 - 2 : 13 + 2;
      p[0] = pass == 0 ? (int)((t10 + t11) * 4) : descale(t10 + t11, 2);
      p[4 * st] = pass == 0 ? (int)((t10 - t11) * 4) : descale(t10 - t11, 2);
      int64_t z1 = (t12 + t13) * F0541;
      p[2 * st] = descale(z1 + t13 * F0765, lo);
      p[6 * st] = descale(z1 + t12 * (-F1847), lo);
      z1 = t4 + t7;
      int64_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
      const int64_t z5 = (z3 + z4) * F1175;
      const int64_t m4 = t4 * F0298, m5 = t5 * F2053, m6 = t6 * F3072, m7 = t7 * F1501;
      z1 *= -F0899, z2 *= -F2562, z3 *= -F1961, z4 *= -F0390;
      z3 += z5;
      z4 += z5;
      p[7 * st] = descale(m4 + z1 + z3, lo);
      p[5 * st] = descale(m5 + z2 + z4, lo);
      p[3 * st] = descale(m6 + z2 + z3, lo);
      p[st] = descale(m7 + z1 + z4, lo);
    }
  }
}

// px: 8-bit samples, interleaved R, G, B (channels 3) or gray (channels 1), row-major
inline std::vector<unsigned char> encode_jpeg(const uint8_t* px, int W, int H, int channels, int quality = 95) {
  need((channels == 1 || channels == 3) && W > 0 && H > 0 && W < 65536 && H < 65536, "JPEG encoder: 1 or 3 channels, sides below 65536");
  quality = std::min(std::max(quality, 1), 100);
  const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;  // jpeg_quality_scaling
  int qt[2][64];
  for (int t = 0; t < 2; ++t) {
    for (int i = 0; i < 64; ++i) {
      const long v = ((long)(t ? kJpegStdChromaQ : kJpegStdLumaQ)[i] * scale + 50) / 100;
      qt[t][i] = (int)std::min(std::max(v, 1L), 255L);  // force_baseline
    }
  }
  struct Table {
    uint16_t code[256];
    uint8_t size[256];
  } huff[4];
  const uint8_t* vals[4] = {kJpegStdDcVals, kJpegStdAcLuma, kJpegStdDcVals, kJpegStdAcChroma};
  for (int t = 0; t < 4; ++t) {
    memset(huff[t].size, 0, sizeof huff[t].size);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      for (int i = 0; i < kJpegStdBits[t][len - 1]; ++i, ++k, ++code) {
        huff[t].code[vals[t][k]] = (uint16_t)code;
        huff[t].size[vals[t][k]] = (uint8_t)len;
      }
      code <<= 1;
    }
  }
  std::vector<unsigned char> out;
  auto put16 = [&](int v) {
    out.push_back((unsigned char)(v >> 8));
    out.push_back((unsigned char)v);
  };
  out.push_back(0xff), out.push_back(0xd8);
  static const unsigned char jfif[] = {0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  out.insert(out.end(), jfif, jfif + sizeof jfif);
  for (int t = 0; t < (channels == 3 ? 2 : 1); ++t) {
    out.push_back(0xff), out.push_back(0xdb);
    put16(67);
    out.push_back((unsigned char)t);
    for (int k = 0; k < 64; ++k) {
      out.push_back((unsigned char)qt[t][kJpegZigzag[k]]);
    }
  }
  out.push_back(0xff), out.push_back(0xc0);
  put16(8 + 3 * channels);
  out.push_back(8);
  put16(H);
  put16(W);
  out.push_back((unsigned char)channels);
  for (int c = 0; c < channels; ++c) {
    out.push_back((unsigned char)(c + 1));
    out.push_back((unsigned char)(c == 0 && channels == 3 ? 0x22 : 0x11));
    out.push_back((unsigned char)(c ? 1 : 0));
  }
  for (int t = 0; t < (channels == 3 ? 4 : 2); ++t) {
    int n = 0;
    for (int i = 0; i < 16; ++i) {
      n += kJpegStdBits[t][i];
    }
    out.push_back(0xff), out.push_back(0xc4);
    put16(19 + n);
    out.push_back((unsigned char)(((t & 1) << 4) | (t >> 1)));
    out.insert(out.end(), kJpegStdBits[t], kJpegStdBits[t] + 16);
    out.insert(out.end(), vals[t], vals[t] + n);
  }
  out.push_back(0xff), out.push_back(0xda);
  put16(6 + 2 * channels);
  out.push_back((unsigned char)channels);
  for (int c = 0; c < channels; ++c) {
    out.push_back((unsigned char)(c + 1));
    out.push_back((unsigned char)(c ? 0x11 : 0x00));
  }
  out.push_back(0), out.push_back(63), out.push_back(0);

  // ---- component planes: colour conversion, then (chroma) the 2 x 2 box over edge-replicated rows and columns
  const int hs = channels == 3 ? 2 : 1;  // luma sampling factor = MCU size / 8
  const int mcusX = (W + 8 * hs - 1) / (8 * hs), mcusY = (H + 8 * hs - 1) / (8 * hs);
  struct Comp {
    int w, h;        // downsampled_width / _height
    int bw, bh;      // width_in_blocks / height_in_blocks (real blocks)
    int pw, ph;      // plane size: padded to whole MCUs
    std::vector<uint8_t> s;
  } comp[3];
  for (int c = 0; c < channels; ++c) {
    const int f = c == 0 ? hs : 1;
    comp[c].w = (W * f + hs - 1) / hs;
    comp[c].h = (H * f + hs - 1) / hs;
    comp[c].bw = (comp[c].w + 7) / 8;
    comp[c].bh = (comp[c].h + 7) / 8;
    comp[c].pw = mcusX * f * 8;
    comp[c].ph = mcusY * f * 8;
    comp[c].s.assign((size_t)comp[c].pw * comp[c].ph, 0);
  }
  // full-resolution Y / Cb / Cr rows with the right edge replicated to the chroma blocks' extent (expand_right_edge)
  // and, for an odd height, the last row repeated to complete its row group (expand_bottom_edge)
  const int fullW = channels == 3 ? std::max(comp[0].bw * 8, comp[1].bw * 16) : comp[0].bw * 8;
  const int fullH = channels == 3 ? ((H + 1) & ~1) : H;
  std::vector<uint8_t> ycc[3];
  for (int c = 0; c < channels; ++c) {
    ycc[c].resize((size_t)fullW * fullH);
  }
  for (int y = 0; y < fullH; ++y) {
    const uint8_t* row = px + (size_t)std::min(y, H - 1) * W * channels;
    for (int x = 0; x < fullW; ++x) {
      const uint8_t* s = row + (size_t)std::min(x, W - 1) * channels;
      if (channels == 1) {
        ycc[0][(size_t)y * fullW + x] = s[0];
        continue;
      }
      const int64_t r = s[0], g = s[1], b = s[2];
      ycc[0][(size_t)y * fullW + x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
      ycc[1][(size_t)y * fullW + x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
      ycc[2][(size_t)y * fullW + x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
    }
  }
  for (int c = 0; c < channels; ++c) {
    Comp& k = comp[c];
    const bool sub = channels == 3 && c > 0;
    const int rows = sub ? fullH / 2 : fullH, cols = k.bw * 8;
    for (int y = 0; y < k.ph; ++y) {
      uint8_t* o = &k.s[(size_t)y * k.pw];
      const int sy = std::min(y, rows - 1);  // below the image: the last (downsampled) row again
      if (!sub) {
        memcpy(o, &ycc[c][(size_t)sy * fullW], (size_t)cols);
      } else {
        const uint8_t* a = &ycc[c][(size_t)(2 * sy) * fullW];
        const uint8_t* bb = a + fullW;
        int bias = 1;
        for (int x = 0; x < cols; ++x) {
          o[x] = (uint8_t)((a[2 * x] + a[2 * x + 1] + bb[2 * x] + bb[2 * x + 1] + bias) >> 2);
          bias ^= 3;
        }
      }
    }
  }

  // ---- entropy coding, MCU by MCU
  uint64_t acc = 0;
  int nacc = 0;
  auto emit = [&](unsigned code, int size) {
    acc = (acc << size) | (code & ((1u << size) - 1));
    nacc += size;
    while (nacc >= 8) {
      const unsigned char byte = (unsigned char)(acc >> (nacc - 8));
      out.push_back(byte);
      if (byte == 0xff) {
        out.push_back(0);
      }
      nacc -= 8;
    }
  };
  int lastDc[3] = {0, 0, 0};
  int prevDcOfBlock = 0;  // DC of the block coded before (jccoefct.c's dummy blocks copy it)
  for (int my = 0; my < mcusY; ++my) {
    for (int mx = 0; mx < mcusX; ++mx) {
      for (int c = 0; c < channels; ++c) {
        const Comp& k = comp[c];
        const int f = c == 0 ? hs : 1;
        for (int by = 0; by < f; ++by) {
          for (int bx = 0; bx < f; ++bx) {
            const int blockX = mx * f + bx, blockY = my * f + by;
            int blk[64];
            const bool real = blockX < k.bw && blockY < k.bh;
            if (real) {
              for (int y = 0; y < 8; ++y) {
                const uint8_t* srow = &k.s[(size_t)(blockY * 8 + y) * k.pw + blockX * 8];
                for (int x = 0; x < 8; ++x) {
                  blk[8 * y + x] = srow[x] - 128;
                }
              }
              jpeg_fdct_islow(blk);
              const int* q = qt[c ? 1 : 0];
              for (int i = 0; i < 64; ++i) {
                const int div = q[i] * 8;
                const int v = blk[i];
                blk[i] = v < 0 ? -((-v + (div >> 1)) / div) : (v + (div >> 1)) / div;
              }
            } else {  // beyond the component's real blocks: zero AC, the DC of the block coded just before
              memset(blk, 0, sizeof blk);
              blk[0] = prevDcOfBlock;
            }
            prevDcOfBlock = blk[0];
            const Table& dc = huff[c ? 2 : 0];
            const Table& ac = huff[c ? 3 : 1];
            int diff = blk[0] - lastDc[c];
            lastDc[c] = blk[0];
            int t = diff < 0 ? -diff : diff, t2 = diff < 0 ? diff - 1 : diff, nbits = 0;
            while (t) {
              ++nbits;
              t >>= 1;
            }
            emit(dc.code[nbits], dc.size[nbits]);
            if (nbits) {
              emit((unsigned)t2, nbits);
            }
            int run = 0;
            for (int kz = 1; kz < 64; ++kz) {
              const int v = blk[kJpegZigzag[kz]];
              if (v == 0) {
                ++run;
                continue;
              }
              while (run > 15) {
                emit(ac.code[0xf0], ac.size[0xf0]);
                run -= 16;
              }
              t = v < 0 ? -v : v;
              t2 = v < 0 ? v - 1 : v;
              nbits = 0;
              while (t) {
                ++nbits;
                t >>= 1;
              }
              emit(ac.code[(run << 4) + nbits], ac.size[(run << 4) + nbits]);
              emit((unsigned)t2, nbits);
              run = 0;
            }
            if (run > 0) {
              emit(ac.code[0], ac.size[0]);
            }
          }
        }
      }
    }
  }
  if (nacc) {
    emit(0x7f, 8 - nacc);  // pad the last byte with ones
  }
  out.push_back(0xff), out.push_back(0xd9);
  return out;
}

// ================================================================================================ TIFF
// TIFF 6.0 (+ the Deflate / SampleFormat supplements), classic (not BigTIFF), first directory. What OpenCV's TiffDecoder
// returns under IMREAD_UNCHANGED: 8-bit images go through libtiff's RGBA interface (palette expanded, min-is-white
// inverted, un-associated alpha multiplied into the colours), 16-bit and float ones are the file's samples as stored.
struct TiffEntry {
  int type = 0;
  uint32_t count = 0;
  size_t valuePos = 0;  // where the values are (inside the entry when they fit in 4 bytes)
};
struct TiffFile {
  Bytes b;
  bool le = true;
  uint32_t u16(size_t p) const {
    b.span(p, 2, "truncated TIFF file");
    return le ? (b.d[p] | (b.d[p + 1] << 8)) : ((b.d[p] << 8) | b.d[p + 1]);
  }
  uint32_t u32(size_t p) const {
    b.span(p, 4, "truncated TIFF file");
    return le ? (b.d[p] | (b.d[p + 1] << 8) | (b.d[p + 2] << 16) | ((uint32_t)b.d[p + 3] << 24)) : be32(b.d + p);
  }
  std::vector<std::pair<int, TiffEntry>> dir;
  bool open() {
    if (b.n < 8) {
      return false;
    }
    if (!memcmp(b.d, "II\x2a\x00", 4)) {
      le = true;
    } else if (!memcmp(b.d, "MM\x00\x2a", 4)) {
      le = false;
    } else {
      return false;
    }
    const size_t ifd = u32(4);
    const uint32_t n = u16(ifd);
    for (uint32_t i = 0; i < n; ++i) {
      const size_t e = ifd + 2 + 12 * (size_t)i;
      TiffEntry t;
      const int tag = (int)u16(e);
      t.type = (int)u16(e + 2);
      t.count = u32(e + 4);
      static const int size[] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8};
      const size_t bytes = (size_t)(t.type >= 1 && t.type <= 12 ? size[t.type] : 1) * t.count;
      t.valuePos = bytes <= 4 ? e + 8 : (size_t)u32(e + 8);
      dir.emplace_back(tag, t);
    }
    return true;
  }
  const TiffEntry* find(int tag) const {
    for (const auto& e : dir) {
      if (e.first == tag) {
        return &e.second;
      }
    }
    return nullptr;
  }
  uint32_t value(const TiffEntry& t, uint32_t i) const {
    need(i < t.count, "TIFF tag with too few values");
    switch (t.type) {
      case 1:
      case 6:
      case 7:
        b.span(t.valuePos + i, 1, "truncated TIFF file");
        return b.d[t.valuePos + i];
      case 3:
      case 8:
        return u16(t.valuePos + 2 * (size_t)i);
      case 4:
      case 9:
        return u32(t.valuePos + 4 * (size_t)i);
      default:
        throw Error("unsupported TIFF tag type");
    }
  }
  uint32_t get(int tag, uint32_t dflt, uint32_t i = 0) const {
    const TiffEntry* t = find(tag);
    return t && i < t->count ? value(*t, i) : dflt;
  }
};
inline bool tiff_header(const Bytes& b, int& w, int& h) {
  TiffFile f{b};
  try {
    if (!f.open()) {
      return false;
    }
    w = (int)f.get(256, 0);
    h = (int)f.get(257, 0);
  } catch (const Error&) {
    return false;
  }
  return w > 0 && h > 0;
}
inline void tiff_lzw(const unsigned char* src, size_t n, std::vector<unsigned char>& out, size_t expect) {
  // TIFF 6.0 section 13: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, the width grows one
  // code EARLY (when the next free entry is 511 / 1023 / 2047)
  struct Entry {
    int prev;
    uint16_t len;
    uint8_t first, last;
  };
  std::vector<Entry> tab(4096);
  for (int i = 0; i < 256; ++i) {
    tab[i] = {-1, 1, (uint8_t)i, (uint8_t)i};
  }
  out.assign(expect, 0);
  size_t o = 0;
  uint32_t acc = 0;
  int have = 0, bits = 9, next = 258, old = -1;
  size_t p = 0;
  for (;;) {
    while (have < bits && p < n) {
      acc = (acc << 8) | src[p++];
      have += 8;
    }
    if (have < bits) {
      break;
    }
    const int code = (int)((acc >> (have - bits)) & ((1u << bits) - 1));
    have -= bits;
    if (code == 257) {
      break;
    }
    if (code == 256) {
      bits = 9;
      next = 258;
      old = -1;
      continue;
    }
    int emit = code;
    if (old < 0) {
      need(code < 256, "corrupt TIFF LZW data");
    } else {
      need(code <= next && next < 4096, "corrupt TIFF LZW data");
      tab[next] = {old, (uint16_t)(tab[old].len + 1), tab[old].first, code < next ? tab[code].first : tab[old].first};
      ++next;
    }
    const int len = tab[emit].len;
    if (o + len > expect) {  // libtiff stops at the strip's size
      int skip = (int)(o + len - expect), e = emit;
      for (int i = 0; i < skip; ++i) {
        e = tab[e].prev;
      }
      for (size_t q = expect; q-- > o; e = tab[e].prev) {
        out[q] = tab[e].last;
      }
      o = expect;
      break;
    }
    {
      int e = emit;
      for (size_t q = o + len; q-- > o; e = tab[e].prev) {
        out[q] = tab[e].last;
      }
    }
    o += len;
    old = code;
    if (next + 1 >= (1 << bits) && bits < 12) {
      ++bits;
    }
    if (o == expect) {
      break;
    }
  }
  need(o == expect, "short TIFF LZW data");
}
inline Raster decode_tiff(const Bytes& b) {
  TiffFile f{b};
  need(f.open(), "not a TIFF file");
  const int W = (int)f.get(256, 0), H = (int)f.get(257, 0);
  need(W > 0 && H > 0 && W <= (1 << 20) && H <= (1 << 20), "TIFF ImageWidth / ImageLength outside 1 .. 2^20");
  const int spp = (int)f.get(277, 1), bits = (int)f.get(258, 1), compression = (int)f.get(259, 1), photometric = (int)f.get(262, 1);
  const int planar = (int)f.get(284, 1), predictor = (int)f.get(317, 1), format = (int)f.get(339, 1), fill = (int)f.get(266, 1);
  const int orientation = (int)f.get(274, 1);
  for (int s = 1; s < spp; ++s) {
    need((int)f.get(258, (uint32_t)bits, (uint32_t)s) == bits, "unsupported TIFF: samples of different widths");
  }
  need(fill == 1 && orientation == 1, "unsupported TIFF: FillOrder / Orientation other than 1");
  const bool isFloat = format == 3;
  need((bits == 8 && format == 1) || (bits == 16 && format == 1) || (bits == 32 && isFloat),
       "unsupported TIFF sample type (8 / 16-bit unsigned or 32-bit float only)");
  need(spp >= 1 && spp <= 4 && (!isFloat || spp == 1), "unsupported TIFF SamplesPerPixel");
  need(compression == 1 || compression == 5 || compression == 8 || compression == 32946 || compression == 32773,
       "unsupported TIFF compression (none / LZW / Deflate / PackBits only)");
  need(predictor == 1 || (predictor == 2 && !isFloat), "unsupported TIFF predictor");
  need(photometric == 0 || photometric == 1 || photometric == 2 || (photometric == 3 && bits == 8 && spp == 1),
       "unsupported TIFF photometric interpretation");
  need(photometric != 0 || bits == 8, "unsupported TIFF: min-is-white beyond 8 bits");
  need((photometric == 2) == (spp >= 3), "unsupported TIFF: SamplesPerPixel does not fit the photometric interpretation");
  need(spp != 2, "unsupported TIFF: gray + alpha");
  const bool tiled = f.find(322) != nullptr;
  const int tw = tiled ? (int)f.get(322, 0) : W, th = tiled ? (int)f.get(323, 0) : (int)std::min<uint32_t>(f.get(278, (uint32_t)H), (uint32_t)H);
  need(tw > 0 && th > 0 && tw <= (1 << 20) && th <= (1 << 20), "corrupt TIFF tile / strip size");
  const TiffEntry* offs = f.find(tiled ? 324 : 273);
  const TiffEntry* counts = f.find(tiled ? 325 : 279);
  need(offs && counts, "TIFF without strip / tile offsets");
  const int across = (W + tw - 1) / tw, down = (H + th - 1) / th;
  const int planes = planar == 2 ? spp : 1, chunkSpp = planar == 2 ? 1 : spp;
  const int64_t nChunks = (int64_t)across * down * planes;  // bounded before any 32-bit arithmetic uses it
  need(nChunks <= INT32_MAX / 2, "corrupt TIFF: too many strips / tiles");
  need(nChunks <= (int64_t)offs->count && counts->count >= offs->count, "TIFF with too few strips / tiles");
  const int bytesPer = bits / 8;
  {
    size_t present = 0;
    for (uint32_t i = 0; i < (uint32_t)nChunks; ++i) {
      present += f.value(*counts, i);
    }
    need(present <= b.n, "TIFF strips / tiles larger than the file");
    plausible(W, H, (size_t)W * H * spp * bytesPer, present, compression == 1 ? 1 : 4096);
    need((int64_t)tw * th <= ((int64_t)1 << 30) && (size_t)tw * th * spp * bytesPer / 4096 <= b.n + 1024, "corrupt TIFF tile / strip size");
  }

  // samples as stored, interleaved, native order
  std::vector<uint16_t> px;
  std::vector<float> fl;
  if (isFloat) {
    fl.assign((size_t)W * H, 0.f);
  } else {
    px.assign((size_t)W * H * spp, 0);
  }
  std::vector<unsigned char> chunk, tmp;
  for (int pl = 0; pl < planes; ++pl) {
    for (int ty = 0; ty < down; ++ty) {
      for (int tx = 0; tx < across; ++tx) {
        const uint32_t idx = (uint32_t)((pl * down + ty) * across + tx);
        const size_t off = f.value(*offs, idx), cnt = f.value(*counts, idx);
        b.span(off, cnt, "TIFF strip / tile outside the file");
        const int rows = tiled ? th : std::min(th, H - ty * th);  // a tile is always whole, the last strip is short
        const size_t rowBytes = (size_t)tw * chunkSpp * bytesPer, expect = rowBytes * rows;
        const unsigned char* src = b.d + off;
        if (compression == 1) {
          need(cnt >= expect, "short TIFF strip / tile");
          chunk.assign(src, src + expect);
        } else if (compression == 5) {
          tiff_lzw(src, cnt, chunk, expect);
        } else if (compression == 32773) {
          chunk.assign(expect, 0);
          size_t o = 0, p = 0;
          while (o < expect && p < cnt) {
            const int nn = (signed char)src[p++];
            if (nn >= 0) {
              const size_t len = std::min((size_t)nn + 1, expect - o);
              need(p + len <= cnt, "corrupt TIFF PackBits data");
              memcpy(&chunk[o], src + p, len);
              p += (size_t)nn + 1;
              o += len;
            } else if (nn != -128) {
              need(p < cnt, "corrupt TIFF PackBits data");
              const size_t len = std::min((size_t)(1 - nn), expect - o);
              memset(&chunk[o], src[p++], len);
              o += len;
            }
          }
          need(o == expect, "short TIFF PackBits data");
        } else {
          chunk = inflate_all(src, cnt, expect, "corrupt TIFF Deflate data");
        }
        for (int r = 0; r < rows; ++r) {
          const int y = ty * th + r;
          if (y >= H) {
            break;
          }
          const unsigned char* line = &chunk[rowBytes * r];
          const int nv = tw * chunkSpp;
          if (isFloat) {
            for (int x = 0; x < tw && tx * tw + x < W; ++x) {
              uint32_t u = f.le ? (line[4 * x] | (line[4 * x + 1] << 8) | (line[4 * x + 2] << 16) | ((uint32_t)line[4 * x + 3] << 24)) : be32(line + 4 * x);
              memcpy(&fl[(size_t)y * W + tx * tw + x], &u, 4);
            }
            continue;
          }
          tmp.resize((size_t)nv * 2);
          uint16_t* v = reinterpret_cast<uint16_t*>(tmp.data());
          for (int i = 0; i < nv; ++i) {
            v[i] = bits == 8 ? line[i] : (uint16_t)(f.le ? (line[2 * i] | (line[2 * i + 1] << 8)) : ((line[2 * i] << 8) | line[2 * i + 1]));
          }
          if (predictor == 2 && (compression == 5 || compression == 8 || compression == 32946)) {
            // horizontal differencing, per sample, modulo the sample width; libtiff knows the tag only inside its LZW and
            // Deflate codecs — an uncompressed or PackBits file that carries it is read as stored
            const unsigned mask = bits == 8 ? 0xffu : 0xffffu;
            for (int i = chunkSpp; i < nv; ++i) {
              v[i] = (uint16_t)((v[i] + v[i - chunkSpp]) & mask);
            }
          }
          for (int x = 0; x < tw && tx * tw + x < W; ++x) {
            for (int c = 0; c < chunkSpp; ++c) {
              px[((size_t)y * W + tx * tw + x) * spp + (planar == 2 ? pl : c)] = v[x * chunkSpp + c];
            }
          }
        }
      }
    }
  }
  Raster img;
  img.w = W;
  img.h = H;
  if (isFloat) {
    img.channels = 1;
    img.bitdepth = 32;
    img.f32.swap(fl);
    return img;
  }
  img.bitdepth = bits;
  if (photometric == 3) {  // palette through libtiff's RGBA interface: 16-bit map entries >> 8 unless the map is 8-bit already
    const TiffEntry* cmap = f.find(320);
    need(cmap && cmap->count >= 3 * 256, "TIFF palette missing");
    bool wide = false;
    for (uint32_t i = 0; i < 3 * 256; ++i) {
      wide = wide || f.value(*cmap, i) >= 256;
    }
    img.channels = 3;
    img.px.resize((size_t)W * H * 3);
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      for (int c = 0; c < 3; ++c) {
        const uint32_t e = f.value(*cmap, (uint32_t)(c * 256 + px[i]));
        img.px[3 * i + c] = (uint16_t)(wide ? e >> 8 : e);
      }
    }
    return img;
  }
  img.channels = spp;
  img.px.swap(px);
  if (photometric == 0) {
    for (auto& v : img.px) {
      v = (uint16_t)(255 - v);
    }
  }
  if (bits == 8 && spp == 4 && f.get(338, 0) == 2) {  // tif_getimage.c putRGBUAcontig8bittile: colour = (a * c + 127) / 255
    for (size_t i = 0; i < (size_t)W * H; ++i) {
      const unsigned a = img.px[4 * i + 3];
      for (int c = 0; c < 3; ++c) {
        img.px[4 * i + c] = (uint16_t)((a * img.px[4 * i + c] + 127) / 255);
      }
    }
  }
  return img;
}

// ================================================================================================ BMP
// Windows bitmaps as OpenCV's BmpDecoder reads them: BITMAPINFOHEADER (or larger) with 8-bit palette, 24-bit or
// 32-bit uncompressed pixels (BI_RGB / BI_BITFIELDS with the standard masks); bottom-up unless the height is negative.
struct BmpInfo {
  int w = 0, h = 0, bpp = 0, compression = 0;
  bool topDown = false;
  size_t dataOffset = 0, headerSize = 0;
};
inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline bool bmp_header(const Bytes& b, BmpInfo& info) {
  if (b.n < 54 || b.d[0] != 'B' || b.d[1] != 'M') {
    return false;
  }
  info.dataOffset = le32(b.d + 10);
  info.headerSize = le32(b.d + 14);
  if (info.headerSize < 40) {
    return false;
  }
  info.w = (int32_t)le32(b.d + 18);
  const int32_t hh = (int32_t)le32(b.d + 22);
  info.topDown = hh < 0;
  info.h = hh < 0 ? -hh : hh;
  info.bpp = b.d[28] | (b.d[29] << 8);
  info.compression = (int)le32(b.d + 30);
  return info.w > 0 && info.h > 0;
}
inline Raster decode_bmp(const Bytes& b) {
  BmpInfo info;
  need(bmp_header(b, info), "not a BMP file");
  need((info.bpp == 8 || info.bpp == 24 || info.bpp == 32) && (info.compression == 0 || (info.compression == 3 && info.bpp == 32)),
       "unsupported BMP flavour (8-bit palette, 24-bit, 32-bit uncompressed only)");
  const size_t stride = (((size_t)info.w * info.bpp + 31) / 32) * 4;
  plausible(info.w, info.h, stride * info.h, b.n, 1);
  b.span(info.dataOffset, stride * info.h, "truncated BMP file");
  Raster img;
  img.w = info.w;
  img.h = info.h;
  img.bitdepth = 8;
  // the palette as the pixels index it: 256 entries, the file's biClrUsed of them, zeros beyond (grfmt_bmp.cpp reads
  // the entries into a zero-filled 256-entry table) — a pixel value above biClrUsed must not reach past the file
  unsigned char pal[256 * 4] = {0};
  bool grayPalette = true;
  if (info.bpp == 8) {
    uint32_t used = le32(b.d + 46);
    used = used == 0 || used > 256 ? 256 : used;
    b.span(14 + (size_t)info.headerSize, (size_t)used * 4, "truncated BMP palette");
    memcpy(pal, b.d + 14 + info.headerSize, (size_t)used * 4);
    for (uint32_t i = 0; i < used; ++i) {  // grfmt_bmp.cpp IsColorPalette: gray only when every entry has b == g == r
      grayPalette = grayPalette && pal[4 * i] == pal[4 * i + 1] && pal[4 * i] == pal[4 * i + 2];
    }
  }
  img.channels = info.bpp == 8 ? (grayPalette ? 1 : 3) : info.bpp == 24 ? 3 : 4;
  img.px.resize((size_t)info.w * info.h * img.channels);
  for (int y = 0; y < info.h; ++y) {
    const unsigned char* line = b.d + info.dataOffset + stride * (info.topDown ? y : info.h - 1 - y);
    uint16_t* o = &img.px[(size_t)y * info.w * img.channels];
    for (int x = 0; x < info.w; ++x) {
      if (info.bpp == 8) {
        const unsigned char* e = pal + 4 * line[x];
        if (grayPalette) {
          o[x] = e[0];
        } else {
          o[3 * x] = e[2], o[3 * x + 1] = e[1], o[3 * x + 2] = e[0];
        }
      } else if (info.bpp == 24) {
        o[3 * x] = line[3 * x + 2], o[3 * x + 1] = line[3 * x + 1], o[3 * x + 2] = line[3 * x];
      } else {
        o[4 * x] = line[4 * x + 2], o[4 * x + 1] = line[4 * x + 1], o[4 * x + 2] = line[4 * x], o[4 * x + 3] = line[4 * x + 3];
      }
    }
  }
  return img;
}

// ================================================================================================ PNM
// Netpbm P1..P6 as OpenCV's PxMDecoder reads them: maxval < 256 -> 8-bit, else 16-bit big-endian, samples not rescaled;
// bitmaps: 0 -> 255, 1 -> 0.
struct PnmInfo {
  int kind = 0, w = 0, h = 0, maxval = 1;
  size_t data = 0;
};
inline bool pnm_header(const Bytes& b, PnmInfo& info) {
  if (b.n < 7 || b.d[0] != 'P' || b.d[1] < '1' || b.d[1] > '6') {
    return false;
  }
  info.kind = b.d[1] - '0';
  size_t p = 2;
  auto number = [&](int& out) {
    for (;;) {
      while (p < b.n && isspace(b.d[p])) {
        ++p;
      }
      if (p < b.n && b.d[p] == '#') {
        while (p < b.n && b.d[p] != '\n') {
          ++p;
        }
        continue;
      }
      break;
    }
    if (p >= b.n || !isdigit(b.d[p])) {
      return false;
    }
    long v = 0;
    while (p < b.n && isdigit(b.d[p]) && v < (1 << 28)) {
      v = v * 10 + (b.d[p++] - '0');
    }
    out = (int)v;
    return true;
  };
  if (!number(info.w) || !number(info.h)) {
    return false;
  }
  if (info.kind != 1 && info.kind != 4 && !number(info.maxval)) {
    return false;
  }
  info.data = p + 1;  // exactly one whitespace byte ends the header
  return info.w > 0 && info.h > 0 && info.maxval > 0 && info.maxval < 65536;
}
inline Raster decode_pnm(const Bytes& b) {
  PnmInfo info;
  need(pnm_header(b, info), "not a PNM file");
  Raster img;
  img.w = info.w;
  img.h = info.h;
  img.channels = (info.kind == 3 || info.kind == 6) ? 3 : 1;
  img.bitdepth = info.maxval < 256 ? 8 : 16;
  const size_t nv = (size_t)info.w * info.h * img.channels;
  plausible(info.w, info.h, info.kind == 4 ? nv / 8 : info.kind == 1 ? nv : info.kind >= 5 ? nv * (img.bitdepth / 8) : nv * 2, b.n - info.data, 1);
  img.px.resize(nv);
  if (info.kind == 4) {
    const size_t stride = ((size_t)info.w + 7) / 8;
    b.span(info.data, stride * info.h, "truncated PBM file");
    for (int y = 0; y < info.h; ++y) {
      for (int x = 0; x < info.w; ++x) {
        img.px[(size_t)y * info.w + x] = ((b.d[info.data + stride * y + (x >> 3)] >> (7 - (x & 7))) & 1) ? 0 : 255;
      }
    }
  } else if (info.kind >= 5) {
    const int bytes = img.bitdepth / 8;
    b.span(info.data, nv * bytes, "truncated PNM file");
    for (size_t i = 0; i < nv; ++i) {
      img.px[i] = bytes == 1 ? b.d[info.data + i] : (uint16_t)be16(b.d + info.data + 2 * i);
    }
  } else {
    size_t p = info.data - 1;
    for (size_t i = 0; i < nv; ++i) {
      while (p < b.n && !isdigit(b.d[p])) {
        if (b.d[p] == '#') {
          while (p < b.n && b.d[p] != '\n') {
            ++p;
          }
        } else {
          ++p;
        }
      }
      need(p < b.n, "truncated PNM file");
      long v = 0;
      if (info.kind == 1) {
        v = b.d[p++] - '0';  // bitmap digits need no separator
        img.px[i] = v ? 0 : 255;
        continue;
      }
      while (p < b.n && isdigit(b.d[p]) && v < 65536) {
        v = v * 10 + (b.d[p++] - '0');
      }
      img.px[i] = (uint16_t)std::min<long>(v, 65535);
    }
  }
  return img;
}

// ================================================================================================ dispatch
inline const char* sniff(const Bytes& b) {
  if (b.n >= 8 && !memcmp(b.d, "\x89PNG\r\n\x1a\n", 8)) {
    return "png";
  }
  if (b.n >= 3 && b.d[0] == 0xff && b.d[1] == 0xd8 && b.d[2] == 0xff) {
    return "jpeg";
  }
  if (b.n >= 4 && (!memcmp(b.d, "II\x2a\x00", 4) || !memcmp(b.d, "MM\x00\x2a", 4))) {
    return "tiff";
  }
  if (b.n >= 2 && b.d[0] == 'B' && b.d[1] == 'M') {
    return "bmp";
  }
  if (b.n >= 3 && b.d[0] == 'P' && b.d[1] >= '1' && b.d[1] <= '6' && isspace(b.d[2])) {
    return "pnm";
  }
  if (b.n >= 12 && !memcmp(b.d, "RIFF", 4) && !memcmp(b.d + 8, "WEBP", 4)) {
    return "webp (unsupported)";
  }
  if (b.n >= 12 && (!memcmp(b.d + 4, "jP  ", 4) || !memcmp(b.d, "\xff\x4f\xff\x51", 4))) {
    return "jpeg 2000 (unsupported)";
  }
  if (b.n >= 4 && !memcmp(b.d, "\x76\x2f\x31\x01", 4)) {
    return "exr";
  }
  return "unknown";
}
inline Raster decode(const unsigned char* d, size_t n) {
  const Bytes b{d, n};
  const std::string kind = sniff(b);
  if (kind == "png") {
    return decode_png(b);
  }
  if (kind == "jpeg") {
    return decode_jpeg(b);
  }
  if (kind == "tiff") {
    return decode_tiff(b);
  }
  if (kind == "bmp") {
    return decode_bmp(b);
  }
  if (kind == "pnm") {
    return decode_pnm(b);
  }
  throw Error(("unsupported image format: " + kind).c_str());
}
// geometry AND sample layout from the headers alone (what decode() would return): false when the bytes are not an
// image this library reads. PNG needs the chunk list up to the first IDAT (a tRNS chunk adds the alpha channel), BMP
// its palette, TIFF its directory.
inline bool probe_info(const unsigned char* d, size_t n, int& w, int& h, int& channels, int& bitdepth) {
  const Bytes b{d, n};
  const std::string kind = sniff(b);
  try {
    if (kind == "png") {
      PngInfo i;
      if (!png_header(b, i)) {
        return false;
      }
      bool trns = false;
      for (size_t pos = 8; pos + 12 <= b.n;) {
        const uint32_t len = be32(b.d + pos);
        if (!memcmp(b.d + pos + 4, "tRNS", 4)) {
          trns = len > 0;
        } else if (!memcmp(b.d + pos + 4, "IDAT", 4) || !memcmp(b.d + pos + 4, "IEND", 4)) {
          break;
        }
        pos += 12 + (size_t)len;
      }
      w = i.w, h = i.h;
      bitdepth = i.depth == 16 ? 16 : 8;
      channels = i.colorType == 0 ? 1 : (i.colorType == 4 || i.colorType == 6 || (trns && (i.colorType == 2 || i.colorType == 3))) ? 4 : 3;
      return true;
    }
    if (kind == "jpeg") {
      JpegInfo i;
      if (!jpeg_header(b, i)) {
        return false;
      }
      w = i.w, h = i.h, channels = i.ncomp, bitdepth = 8;
      return true;
    }
    if (kind == "tiff") {
      TiffFile f{b};
      if (!f.open()) {
        return false;
      }
      w = (int)f.get(256, 0), h = (int)f.get(257, 0);
      const int bits = (int)f.get(258, 1);
      bitdepth = bits;
      channels = f.get(262, 1) == 3 ? 3 : (int)f.get(277, 1);
      return w > 0 && h > 0;
    }
    if (kind == "bmp") {
      BmpInfo i;
      if (!bmp_header(b, i)) {
        return false;
      }
      w = i.w, h = i.h, bitdepth = 8;
      channels = i.bpp == 24 ? 3 : i.bpp == 32 ? 4 : 1;
      if (i.bpp == 8) {
        uint32_t used = le32(b.d + 46);
        used = used == 0 || used > 256 ? 256 : used;
        b.span(14 + i.headerSize, (size_t)used * 4, "truncated BMP palette");
        const unsigned char* pal = b.d + 14 + i.headerSize;
        for (uint32_t k = 0; k < used; ++k) {
          if (pal[4 * k] != pal[4 * k + 1] || pal[4 * k] != pal[4 * k + 2]) {
            channels = 3;
          }
        }
      }
      return true;
    }
    if (kind == "pnm") {
      PnmInfo i;
      if (!pnm_header(b, i)) {
        return false;
      }
      w = i.w, h = i.h;
      channels = (i.kind == 3 || i.kind == 6) ? 3 : 1;
      bitdepth = i.maxval < 256 ? 8 : 16;
      return true;
    }
  } catch (const Error&) {
  }
  return false;
}
inline bool probe_size(const unsigned char* d, size_t n, int& w, int& h) {
  const Bytes b{d, n};
  const std::string kind = sniff(b);
  if (kind == "png") {
    PngInfo i;
    if (png_header(b, i)) {
      w = i.w, h = i.h;
      return true;
    }
  } else if (kind == "jpeg") {
    JpegInfo i;
    if (jpeg_header(b, i)) {
      w = i.w, h = i.h;
      return true;
    }
  } else if (kind == "tiff") {
    return tiff_header(b, w, h);
  } else if (kind == "bmp") {
    BmpInfo i;
    if (bmp_header(b, i)) {
      w = i.w, h = i.h;
      return true;
    }
  } else if (kind == "pnm") {
    PnmInfo i;
    if (pnm_header(b, i)) {
      w = i.w, h = i.h;
      return true;
    }
  }
  return false;
}

}  // namespace codecs
