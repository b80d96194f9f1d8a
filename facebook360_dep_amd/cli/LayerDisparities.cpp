// LayerDisparities — drop-in for source/depth_estimation/LayerDisparities.cpp: same flags (:35-43);
// writes <output>/disparity/<cam>/<frame>.png (8-bit, as cv::imwrite converts the float image).
// Compute = derp_layer_disparities.
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
   - Layers foreground disparity atop background disparity assuming nans to correspond to locations
   without valid disparities.

   - Example:
     ./LayerDisparities \
     --rig=/path/to/rigs/rig.json \
     --background_disp=/path/to/background/disparity \
     --foreground_disp=/path/to/output/disparity \
     --output=/path/to/output \
     --first=000000 \
     --last=000000
)";

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("background_disp", "", "path to background disparity directory (required)");
  F.str("background_frame", "000000", "background frame to process (lexical)");
  F.str("cameras", "", "destination cameras");
  F.str("first", "000000", "first frame to process (lexical)");
  F.str("foreground_disp", "", "path to foreground disparity directory (required)");
  F.str("last", "000000", "last frame to process (lexical)");
  F.str("output", "", "path to output disparity directory");
  F.str("rig", "", "path to camera rig .json (required)");
  F.i32("threads", -1, "number of threads (-1 = auto, 0 = none) [accepted; the GPU path ignores it]");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  CHECK_MSG(F.s("rig") != "", "rig");
  CHECK_MSG(F.s("background_disp") != "", "background_disp");
  CHECK_MSG(F.s("foreground_disp") != "", "foreground_disp");
  CHECK_MSG(F.s("first") <= F.s("last"), "first <= last");
  const std::vector<derp_camera_desc> rigSrc = load_rig(F.s("rig"));
  const std::vector<derp_camera_desc> rigDst = filter_destinations(rigSrc, F.s("cameras"));
  CHECK_MSG(!rigDst.empty(), "no destination cameras!");
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rigSrc.data(), (int)rigSrc.size(), rigDst.data(), (int)rigDst.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  std::vector<std::vector<float>> bg(rigDst.size());
  std::vector<std::pair<int, int>> bgSize(rigDst.size());
  for (size_t i = 0; i < rigDst.size(); ++i) {
    bg[i] = load_float(image_path(F.s("background_disp"), rigDst[i].id, F.s("background_frame")), bgSize[i].first,
                       bgSize[i].second);
  }
  for (int f = std::stoi(F.s("first")); f <= std::stoi(F.s("last")); ++f) {
    const std::string frame = zero_pad(f);
    for (size_t i = 0; i < rigDst.size(); ++i) {
      int w, h;
      const std::vector<float> fg = load_float(image_path(F.s("foreground_disp"), rigDst[i].id, frame), w, h);
      CHECK_MSG(w == bgSize[i].first && h == bgSize[i].second, "Background and foreground images must be of the same size!");
      std::vector<uint8_t> out(fg.size());
      DERP_OK(ctx, derp_layer_disparities(ctx, fg.data(), bg[i].data(), fg.size(), out.data()));
      const fs::path dir = fs::path(F.s("output")) / "disparity" / rigDst[i].id;
      fs::create_directories(dir);
      std::vector<uint16_t> px(out.begin(), out.end());
      write_png(dir / (frame + ".png"), px.data(), w, h, 1, 8);
    }
  }
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
