// GenerateForegroundMasks — drop-in for source/render/GenerateForegroundMasks.cpp (the producer of
// the masks DerpCLI consumes): same flags (:44-56), writes <foreground_masks>/<cam>/<frame>.png
// (8-bit, 0 / 255). Compute = derp_resize_area (INTER_AREA downscale to --width) +
// derp_generate_foreground_mask.
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
   - Generates foreground masks by comparing each frame against a background frame.

   - Example:
     ./GenerateForegroundMasks \
     --first=000000 \
     --last=000000 \
     --rig=/path/to/rigs/rig.json \
     --color=/path/to/video/color \
     --background_color=/path/to/background/color \
     --foreground_masks=/path/to/video/output
)";

static std::vector<uint16_t> load_resized(derp_ctx* ctx, const fs::path& path, int outW, int outH) {
  int w, h;
  std::vector<uint16_t> img = load_color_bgr16(path, w, h);
  if (w == outW && h == outH) {
    return img;  // cv_util::resizeImage returns the image itself when the size matches
  }
  std::vector<uint16_t> out((size_t)outW * outH * 3);
  DERP_OK(ctx, derp_resize_area(ctx, 0, img.data(), w, h, out.data(), outW, outH));
  return out;
}

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("background_color", "", "path to input background color images (required)");
  F.str("background_frame", "000000", "background frame (lexical)");
  F.i32("blur_radius", 1, "Gaussian blur radius (0 = no blur)");
  F.str("cameras", "", "comma-separated cameras to render (empty for all)");
  F.str("color", "", "path to input color images (required)");
  F.str("first", "", "first frame to process (lexical) (required)");
  F.str("foreground_masks", "", "path to output foreground masks (required)");
  F.str("last", "", "last frame to process (lexical) (required)");
  F.i32("morph_closing_size", 4, "Morphological closing size (0 = no closing)");
  F.str("rig", "", "path to camera rig .json (required)");
  F.i32("threads", -1, "number of threads (-1 = max allowed, 0 = no threading) [accepted; the GPU path ignores it]");
  F.dbl("threshold", 0.04, "foreground/background RGB L2-norm threshold [0..1]");
  F.i32("width", 2048, "optional downscaled output width");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  CHECK_MSG(F.s("color") != "", "color");
  CHECK_MSG(F.s("rig") != "", "rig");
  CHECK_MSG(F.s("background_color") != "", "background_color");
  CHECK_MSG(F.s("foreground_masks") != "", "foreground_masks");
  CHECK_MSG(F.s("first") != "", "first");
  CHECK_MSG(F.s("last") != "", "last");
  CHECK_MSG(F.s("background_frame") != "", "background_frame");
  CHECK_MSG(F.i("width") > 0, "width > 0");
  CHECK_MSG(F.i("blur_radius") >= 0, "blur_radius >= 0");
  CHECK_MSG(F.d("threshold") >= 0, "threshold >= 0");
  CHECK_MSG(F.i("morph_closing_size") >= 0, "morph_closing_size >= 0");
  const std::vector<derp_camera_desc> rigAll = load_rig(F.s("rig"));
  const std::vector<derp_camera_desc> rig = filter_destinations(rigAll, F.s("cameras"));
  CHECK_MSG(!rig.empty(), "rig.size() > 0");
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rigAll.data(), (int)rigAll.size(), rig.data(), (int)rig.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  // output size from the first background image (GenerateForegroundMasks.cpp:86-91)
  int bw, bh;
  CHECK_MSG(image_size(image_path(F.s("background_color"), rig[0].id, F.s("background_frame")), bw, bh),
            "cannot read background image");
  const int outW = std::min(bw, F.i("width"));
  const int outH = (int)lrint(outW * bh / float(bw));
  std::vector<std::vector<uint16_t>> background(rig.size());
  for (size_t i = 0; i < rig.size(); ++i) {
    background[i] = load_resized(ctx, image_path(F.s("background_color"), rig[i].id, F.s("background_frame")), outW, outH);
  }
  verify_image_paths(F.s("color"), rig, F.s("first"), F.s("last"));
  for (const auto& cam : rig) {
    fs::create_directories(fs::path(F.s("foreground_masks")) / cam.id);
  }
  for (int f = std::stoi(F.s("first")); f <= std::stoi(F.s("last")); ++f) {
    const std::string frame = zero_pad(f);
    LOG_INFO("Processing frame " + frame + "...");
    for (size_t i = 0; i < rig.size(); ++i) {
      const std::vector<uint16_t> color = load_resized(ctx, image_path(F.s("color"), rig[i].id, frame), outW, outH);
      std::vector<uint8_t> mask((size_t)outW * outH);
      DERP_OK(ctx, derp_generate_foreground_mask(ctx, background[i].data(), color.data(), outW, outH, F.i("blur_radius"),
                                                 (float)F.d("threshold"), F.i("morph_closing_size"), mask.data()));
      size_t count = 0;
      std::vector<uint16_t> px(mask.size());
      for (size_t k = 0; k < mask.size(); ++k) {
        px[k] = mask[k] ? 255 : 0;  // imwrite(255.0f * mask)
        count += mask[k];
      }
      LOG_INFO(fmt("foreground amount: %.2f%%", 100.0 * count / mask.size()));
      write_png(fs::path(F.s("foreground_masks")) / rig[i].id / (frame + ".png"), px.data(), outW, outH, 1, 8);
    }
  }
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
