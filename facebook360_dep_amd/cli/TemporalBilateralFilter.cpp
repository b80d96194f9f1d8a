// TemporalBilateralFilter — drop-in for source/depth_estimation/TemporalBilateralFilter.cpp:
// same flags (:40-59), inputs and output layout
// (<output_root>/disparity_time_filtered_levels/level_L/<cam>/<frame>.pfm). Compute = derp_temporal_filter.
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
  - Runs temporal filter across disparity frames using corresponding color frames as guides.

  - Example:
    ./TemporalBilateralFilter \
    --input_root=/path/to/ \
    --output_root=/path/to/output \
    --rig=/path/to/rigs/rig.json \
    --first=000000 \
    --last=000000
)";

// populateMinMaxFrame (TemporalBilateralFilter.cpp:96-119)
static void populate_min_max(const std::string& dir, int level, const std::string& camId, int cur, int radius,
                             int& first, int& last) {
  const fs::path levelDir = fs::path(dir) / ("level_" + std::to_string(level)) / camId;
  const std::string ext = first_extension(levelDir);
  int lf = INT32_MAX, ll = 0;
  for (int f = cur - radius; f <= cur + radius; ++f) {
    if (f >= 0 && fs::exists(levelDir / (zero_pad(f) + ext))) {
      lf = std::min(f, lf);
      ll = std::max(f, ll);
    }
  }
  first = std::max(lf, first);
  last = std::min(ll, last);
}

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("color", "", "color directory");
  F.str("cameras", "", "destination cameras");
  F.str("disparity", "", "disparity directory");
  F.str("first", "000000", "first frame to process (lexical)");
  F.str("foreground_masks", "", "foreground masks directory");
  F.str("input_root", "", "output root directory (required)");
  F.str("last", "000000", "last frame to process (lexical)");
  F.i32("level", 0, "pyramid level being processed");
  F.str("output_formats", "", "saved formats, comma separated (exr, png, pfm supported)");
  F.str("output_root", "", "output root directory (required)");
  F.i32("resolution", 2048, "8192, 4096, 2048, 1024, 512, 256");
  F.str("rig", "", "path to camera rig .json (required)");
  F.dbl("sigma", 0.01, "spatio-temporal smoothing");
  F.i32("space_radius", -1, "space filtering radius");
  F.i32("threads", -1, "number of threads (-1 = auto, 0 = none) [accepted; the GPU path ignores it]");
  F.i32("time_radius", 2, "temporal filtering radius");
  F.boolean("use_foreground_masks", false, "use pre-computed foreground masks");
  F.dbl("weight_b", 0.5, "Blue channel weight");
  F.dbl("weight_g", 1.0, "Green channel weight");
  F.dbl("weight_r", 1.0, "Red channel weight");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  CHECK_MSG(F.s("rig") != "", "rig");
  CHECK_MSG(F.s("input_root") != "", "input_root");
  CHECK_MSG(F.s("output_root") != "", "output_root");
  if (F.s("color").empty()) {
    F.set("color", F.s("input_root") + "/video/color_levels");
  }
  if (F.s("foreground_masks").empty()) {
    F.set("foreground_masks", F.s("input_root") + "/video/foreground_masks_levels");
  }
  if (F.s("disparity").empty()) {
    F.set("disparity", F.s("output_root") + "/disparity_levels");
  }
  CHECK_MSG(F.i("time_radius") >= 0, "time_radius >= 0");
  const std::vector<derp_camera_desc> rigSrc = load_rig(F.s("rig"));
  const std::vector<derp_camera_desc> rigDst = filter_destinations(rigSrc, F.s("cameras"));
  CHECK_MSG(!rigDst.empty(), "no destination cameras!");
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rigSrc.data(), (int)rigSrc.size(), rigDst.data(), (int)rigDst.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  const int level = F.i("level");
  const bool useFg = F.b("use_foreground_masks");
  auto levelDir = [&](const std::string& base) { return fs::path(base) / ("level_" + std::to_string(level)); };
  const bool savePng = F.s("output_formats").find("png") != std::string::npos;
  const bool saveExr = F.s("output_formats").find("exr") != std::string::npos;

  for (int cur = std::stoi(F.s("first")); cur <= std::stoi(F.s("last")); ++cur) {  // filterFrame, :121-184
    int first = 0, last = INT32_MAX;
    populate_min_max(F.s("color"), level, rigDst[0].id, cur, F.i("time_radius"), first, last);
    populate_min_max(F.s("disparity"), level, rigDst[0].id, cur, F.i("time_radius"), first, last);
    if (useFg) {
      populate_min_max(F.s("foreground_masks"), level, rigDst[0].id, cur, F.i("time_radius"), first, last);
    }
    CHECK_MSG(first <= cur && cur <= last, fmt("frame %06d has no complete colour/disparity inputs", cur));
    const int n = last - first + 1;
    LOG_INFO("Filtering images...");
    for (size_t cam = 0; cam < rigDst.size(); ++cam) {
      std::vector<std::vector<uint16_t>> colors(n);
      std::vector<std::vector<float>> disps(n);
      std::vector<std::vector<uint8_t>> masks(n);
      int w = 0, h = 0;
      for (int t = 0; t < n; ++t) {
        const std::string frame = zero_pad(first + t);
        colors[t] = load_color_bgr16(image_path(levelDir(F.s("color")), rigDst[cam].id, frame), w, h);
        int w2, h2;
        disps[t] = load_float(image_path(levelDir(F.s("disparity")), rigDst[cam].id, frame), w2, h2);
        CHECK_MSG(w2 == w && h2 == h, "colour / disparity size mismatch");
        masks[t].assign((size_t)w * h, 1);
        DERP_OK(ctx, derp_fov_mask(ctx, (int)cam, w, h, masks[t].data()));  // generateFovMasks, :150-151
        if (useFg) {
          const std::vector<uint8_t> fg = load_mask(image_path(levelDir(F.s("foreground_masks")), rigDst[cam].id, frame), w2, h2);
          for (size_t i = 0; i < fg.size(); ++i) {
            masks[t][i] &= fg[i];
          }
        }
      }
      // spaceRadius (:165-168): max(ceil(1 * 0.9^level), 1)
      const float scale = std::pow(0.9f, level);
      const int spaceRadius = F.i("space_radius") == -1 ? (int)std::max(std::ceil(1 * scale), 1.0f) : F.i("space_radius");
      std::vector<const uint16_t*> gp(n);
      std::vector<const float*> dp(n);
      std::vector<const uint8_t*> mp(n);
      for (int t = 0; t < n; ++t) {
        gp[t] = colors[t].data();
        dp[t] = disps[t].data();
        mp[t] = masks[t].data();
      }
      std::vector<float> out((size_t)w * h);
      // weights passed as (b, g, b) — reference quirk kept (TemporalBilateralFilter.cpp:176-178)
      DERP_OK(ctx, derp_temporal_filter(ctx, gp.data(), dp.data(), mp.data(), n, w, h, cur - first, (float)F.d("sigma"),
                                        spaceRadius, (float)F.d("weight_b"), (float)F.d("weight_g"),
                                        (float)F.d("weight_b"), out.data()));
      const fs::path dir = fs::path(F.s("output_root")) / "disparity_time_filtered_levels" /
          ("level_" + std::to_string(level)) / rigDst[cam].id;
      fs::create_directories(dir);
      write_pfm(dir / (zero_pad(cur) + ".pfm"), out.data(), w, h);
      if (savePng) {
        write_disparity_png(dir / (zero_pad(cur) + ".png"), out.data(), w, h);
      }
      if (saveExr) {
        write_exr_f32(dir / (zero_pad(cur) + ".exr"), out.data(), w, h);
      }
    }
  }
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
