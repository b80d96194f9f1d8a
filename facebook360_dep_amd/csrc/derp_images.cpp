// Host-side entry points of the C-ABI (include/derp_hip.h, "raster inputs"): the decoders of cli/image_codecs.h for
// callers that are not the C++ executables — the Python pyramid builder (facebook360_dep_amd/resize.py, the mirror of
// scripts/render/resize.py:66-70, which reads its sources with cv2.imread(path, cv2.IMREAD_UNCHANGED)). No device
// code and no HIP call: a second translation unit of libderp_hip.so.
#include <string>

#include "../../include/derp_hip.h"
#include "../cli/image_codecs.h"

static thread_local std::string g_image_error;

extern "C" const char* derp_image_last_error(void) { return g_image_error.c_str(); }

extern "C" int derp_image_info(const void* bytes, size_t n, int* w, int* h, int* channels, int* bitdepth) {
  int iw = 0, ih = 0, ic = 0, ib = 0;  // from the headers alone: nothing is decoded here
  if (!codecs::probe_info(static_cast<const unsigned char*>(bytes), n, iw, ih, ic, ib)) {
    g_image_error = std::string("unsupported image format: ") + codecs::sniff(codecs::Bytes{static_cast<const unsigned char*>(bytes), n});
    return 1;
  }
  if (w) *w = iw;
  if (h) *h = ih;
  if (channels) *channels = ic;
  if (bitdepth) *bitdepth = ib;
  return 0;
}

extern "C" int derp_image_decode(const void* bytes, size_t n, void* out, size_t out_bytes) {
  try {
    const codecs::Raster r = codecs::decode(static_cast<const unsigned char*>(bytes), n);
    const size_t count = (size_t)r.w * r.h * r.channels;
    if (r.bitdepth == 32) {
      codecs::need(out_bytes == count * sizeof(float), "output buffer size does not match the image");
      memcpy(out, r.f32.data(), out_bytes);
      return 0;
    }
    codecs::need(out_bytes == count * sizeof(uint16_t), "output buffer size does not match the image");
    uint16_t* o = static_cast<uint16_t*>(out);
    if (r.channels == 1) {
      memcpy(o, r.px.data(), out_bytes);
    } else {  // file order R, G, B [, A] -> OpenCV's B, G, R [, A]
      for (size_t i = 0; i < (size_t)r.w * r.h; ++i) {
        const uint16_t* s = &r.px[i * r.channels];
        uint16_t* d = o + i * r.channels;
        d[0] = s[2], d[1] = s[1], d[2] = s[0];
        if (r.channels == 4) {
          d[3] = s[3];
        }
      }
    }
    return 0;
  } catch (const std::exception& e) {
    g_image_error = e.what();
    return 1;
  }
}

extern "C" int derp_jpeg_encode(const void* pixels, int w, int h, int channels, int quality, void* out, size_t cap, size_t* size) {
  try {
    codecs::need(pixels && size && (channels == 1 || channels == 3) && w > 0 && h > 0, "derp_jpeg_encode: 8-bit gray or B, G, R pixels");
    const uint8_t* src = static_cast<const uint8_t*>(pixels);
    std::vector<uint8_t> rgb;
    if (channels == 3) {  // OpenCV's order in, the file's order inside
      rgb.resize((size_t)w * h * 3);
      for (size_t i = 0; i < (size_t)w * h; ++i) {
        rgb[3 * i] = src[3 * i + 2], rgb[3 * i + 1] = src[3 * i + 1], rgb[3 * i + 2] = src[3 * i];
      }
      src = rgb.data();
    }
    const std::vector<unsigned char> j = codecs::encode_jpeg(src, w, h, channels, quality);
    *size = j.size();
    if (out && cap >= j.size()) {
      memcpy(out, j.data(), j.size());
      return 0;
    }
    g_image_error = "derp_jpeg_encode: output buffer too small (the size needed is returned)";
    return out ? 1 : 0;  // a call without a buffer asks for the size
  } catch (const std::exception& e) {
    g_image_error = e.what();
    return 1;
  }
}
