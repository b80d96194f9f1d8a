// Test harness (CPU only) for cli/image_codecs.h. usage: codec_main <image file> [<raw output>]
// Prints "<kind> <w> <h> <channels> <bitdepth> <probe_w> <probe_h>" and writes the samples (uint16 little-endian,
// file channel order; float32 for bitdepth 32) to <raw output>. With a third argument --bgr16: png_fast_bgr16 instead. A file the decoder refuses: prints "error: <why>", exit 3.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>

#include "../../facebook360_dep_amd/cli/image_codecs.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    return 2;
  }
  std::ifstream f(argv[1], std::ios::binary);
  const std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (argc > 3 && std::string(argv[3]) == "--bgr16") {  // the executables' colour fast path: B, G, R uint16, or "notfast"
    try {
      codecs::PngInfo info;
      if (!codecs::png_header(codecs::Bytes{data.data(), data.size()}, info)) {
        printf("notfast\n");
        return 0;
      }
      std::vector<uint16_t> out((size_t)info.w * info.h * 3, 0xabcd);
      int w = 0, h = 0;
      if (!codecs::png_fast_bgr16(codecs::Bytes{data.data(), data.size()}, out.data(), 0, 0, w, h)) {
        printf("notfast\n");
        return 0;
      }
      printf("fast %d %d\n", w, h);
      std::ofstream o(argv[2], std::ios::binary);
      o.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * 2));
      return 0;
    } catch (const codecs::Error& e) {
      printf("error: %s\n", e.what());
      return 3;
    }
  }
  try {
    const codecs::Raster r = codecs::decode(data.data(), data.size());
    int pw = -1, ph = -1, qw = -1, qh = -1, qc = -1, qb = -1;
    codecs::probe_size(data.data(), data.size(), pw, ph);
    if (!codecs::probe_info(data.data(), data.size(), qw, qh, qc, qb) || qw != r.w || qh != r.h || qc != r.channels || qb != r.bitdepth) {
      printf("error: probe_info says %d x %d x %d, %d bits; decode %d x %d x %d, %d bits\n", qw, qh, qc, qb, r.w, r.h, r.channels, r.bitdepth);
      return 4;
    }
    printf("%s %d %d %d %d %d %d\n", codecs::sniff(codecs::Bytes{data.data(), data.size()}), r.w, r.h, r.channels, r.bitdepth, pw, ph);
    if (argc > 2) {
      std::ofstream o(argv[2], std::ios::binary);
      if (r.bitdepth == 32) {
        o.write(reinterpret_cast<const char*>(r.f32.data()), (std::streamsize)(r.f32.size() * 4));
      } else {
        o.write(reinterpret_cast<const char*>(r.px.data()), (std::streamsize)(r.px.size() * 2));
      }
    }
  } catch (const codecs::Error& e) {
    printf("error: %s\n", e.what());
    return 3;
  }
  return 0;
}
