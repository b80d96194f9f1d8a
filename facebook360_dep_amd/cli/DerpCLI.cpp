// DerpCLI — drop-in for the reference's source/depth_estimation/DerpCLI.cpp: same flag names and
// defaults (DerpCLI.cpp:40-67), same inputs (rig JSON, <color>/level_L/<cam>/<frame>.<ext>, masks,
// background disparities, previous-level PFMs) and the same outputs
// (<output_root>/disparity_levels/level_L/<cam>/<frame>.pfm [+ .png]). All computation happens in
// libderp_hip.so through include/derp_hip.h; there is no CPU path.
//
// The reference iterates level-outer / frame-inner (DerpCLI.cpp:220-229); frames are independent
// inside DerpCLI, so this driver goes frame-outer and keeps one frame's pyramid resident in HBM
// from the coarsest requested level to the finest — identical files, no PFM round trip in between.
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
  - Computes disparity maps for a rig of cameras, coarse to fine over a pyramid of levels.

  - Example:
    ./DerpCLI \
    --input_root=/path/to/ \
    --output_root=/path/to/output \
    --rig=/path/to/rigs/rig.json \
    --first=000000 \
    --last=000000
)";

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("background_disp", "", "path to background disparities");
  F.str("background_frame", "000000", "background frame (lexical)");
  F.str("cameras", "", "comma-separated destinations to render (empty for all)");
  F.str("color", "", "path to input color images");
  F.boolean("do_bilateral_filter", true, "apply bilateral filter at each level");
  F.boolean("do_median_filter", true, "apply median filter to disparity at each level");
  F.str("first", "000000", "first frame to process (lexical)");
  F.str("foreground_masks", "", "path to foreground masks");
  F.str("input_root", "", "path to input data (required)");
  F.str("last", "000000", "last frame to process (lexical)");
  F.i32("level_end", -1, "level to end at (-1 = finest)");
  F.i32("level_start", -1, "level to start at (-1 = coarsest)");
  F.dbl("max_depth_m", 1e4, "max depth (m)");
  F.dbl("min_depth_m", .50, "min depth (m)");
  F.i32("mismatches_start_level", -1, "(-1 = no mismatch handling)");
  F.i32("num_levels", -1, "number of levels in the pyramid (-1 = uses highest level)");
  F.str("output_formats", "", "saved formats, comma separated (exr, png, pfm supported)");
  F.str("output_root", "", "path to output directory (required)");
  F.boolean("partial_coverage", false, "set to true if no 360 coverage");
  F.i32("ping_pong_iterations", 1, "number of spatial propagation iterations");
  F.i32("random_proposals", 2, "number of proposed random disparities before propagation");
  F.i32("resolution", 2048, "Output resolution (width in pixels)");
  F.str("rig", "", "path to camera rig .json");
  F.boolean("save_debug_images", false, "if true, save debugging output images");
  F.i32("threads", -1, "number of threads (-1 = auto, 0 = none) [accepted; the GPU path ignores it]");
  F.boolean("use_foreground_masks", false, "use pre-computed foreground masks");
  F.dbl("var_high_thresh", 1e-3, "ignore variances higher than this threshold");
  F.dbl("var_noise_floor", 4e-5, "noise variance floor on original, full-size images");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  Timer total;

  // ---- verifyInputs (DerpCLI.cpp:69-118)
  CHECK_MSG(F.s("input_root") != "", "input_root");
  CHECK_MSG(F.s("output_root") != "", "output_root");
  if (F.i("level_start") >= 0 && F.i("level_end") >= 0) {
    CHECK_MSG(F.i("level_start") >= F.i("level_end"), "level_start >= level_end");
  }
  const std::string inputRoot = F.s("input_root"), outputRoot = F.s("output_root");
  if (F.s("rig").empty()) {
    F.set("rig", inputRoot + "/rigs/rig_calibrated.json");
  }
  if (F.s("color").empty()) {
    F.set("color", inputRoot + "/video/color_levels");
  }
  if (F.s("background_disp").empty()) {
    F.set("background_disp", inputRoot + "/background/disparity_levels");
  }
  if (F.s("foreground_masks").empty()) {
    F.set("foreground_masks", inputRoot + "/video/foreground_masks_levels");
  }
  CHECK_MSG(F.i("random_proposals") >= 0, "random_proposals >= 0");
  CHECK_MSG(F.s("first") <= F.s("last"), "first <= last");
  CHECK_MSG(fs::is_directory(F.s("color")), "No images in " + F.s("color"));
  const bool useFg = F.b("use_foreground_masks");
  if (useFg) {
    CHECK_MSG(fs::is_directory(F.s("background_disp")),
              "Asked to use background but no background disparities found in " + F.s("background_disp"));
    CHECK_MSG(fs::is_directory(F.s("foreground_masks")),
              "Asked to use foreground masks but no foreground masks found in " + F.s("foreground_masks"));
  }
  bool savePng = false;
  {
    std::stringstream ss(F.s("output_formats"));
    std::string f;
    while (std::getline(ss, f, ',')) {
      CHECK_MSG(f.empty() || f == "exr" || f == "png" || f == "pfm", "Invalid output format specified: " + f);
      savePng |= f == "png";
      if (f == "exr") {
        LOG_WARNING("exr output is not supported by this build; pfm is always written");
      }
    }
    if (F.s("output_formats").empty()) {
      LOG_WARNING("No explicit output formats specified. Forcing PFM...");
    }
  }

  // ---- rig (DerpCLI.cpp:185-192)
  const std::vector<derp_camera_desc> rigSrc = load_rig(F.s("rig"));
  CHECK_MSG(!rigSrc.empty(), "no source cameras!");
  const std::vector<derp_camera_desc> rigDst = filter_destinations(rigSrc, F.s("cameras"));
  CHECK_MSG(!rigDst.empty(), "no destination cameras!");
  const int S = (int)rigSrc.size(), D = (int)rigDst.size();
  std::vector<int> dst2src(D, 0);
  for (int d = 0; d < D; ++d) {
    for (int s = 0; s < S; ++s) {
      if (strcmp(rigDst[d].id, rigSrc[s].id) == 0) {
        dst2src[d] = s;
        break;
      }
    }
  }

  // ---- pyramid geometry (DerpCLI.cpp:194-215)
  std::map<int, std::pair<int, int>> sizes;
  pyramid_level_sizes(sizes, F.s("color"));
  pyramid_level_sizes(sizes, fs::path(outputRoot) / "disparity_levels");
  CHECK_MSG(!sizes.empty(), "No pyramid levels found in " + F.s("color"));
  const int numLevels = F.i("num_levels") == -1 ? sizes.rbegin()->first + 1 : F.i("num_levels");
  const int levelStart = F.i("level_start") >= 0 ? F.i("level_start") : numLevels - 1;
  int levelEnd = 0;
  for (const auto& kv : sizes) {  // getLevelEnd, DerpCLI.cpp:158-177
    if (kv.second.first <= F.i("resolution")) {
      levelEnd = kv.first;
      break;
    }
  }
  if (F.i("level_end") >= 0) {
    CHECK_MSG(F.i("level_end") >= levelEnd,
              fmt("Requested end level %d, which is larger than requested resolution (%d)", F.i("level_end"),
                  F.i("resolution")));
  }
  levelEnd = std::max(levelEnd, F.i("level_end"));
  CHECK_MSG(F.i("level_start") <= numLevels, "level_start <= numLevels");
  const int firstFrame = std::stoi(F.s("first")), numFrames = std::stoi(F.s("last")) - firstFrame + 1;
  auto levelDir = [&](const std::string& base, int level) { return fs::path(base) / ("level_" + std::to_string(level)); };
  const fs::path dispLevels = fs::path(outputRoot) / "disparity_levels";
  // verifyInputImagePaths (DerpCLI.cpp:137-156)
  verify_image_paths(levelDir(F.s("color"), levelStart), rigSrc, F.s("first"), F.s("last"));
  if (useFg) {
    verify_image_paths(levelDir(F.s("background_disp"), levelStart), rigDst, F.s("background_frame"),
                       F.s("background_frame"));
    verify_image_paths(levelDir(F.s("foreground_masks"), levelStart), rigDst, F.s("first"), F.s("last"));
  }
  if (levelStart < numLevels - 1) {
    verify_image_paths(levelDir(dispLevels.string(), levelStart + 1), rigDst, F.s("first"), F.s("last"));
  }
  fs::create_directories(outputRoot);
  const int widthFull = (int)rigDst[0].resolution[0], heightFull = (int)rigDst[0].resolution[1];

  // ---- context
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rigSrc.data(), S, rigDst.data(), D) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  derp_options opt;
  derp_options_default(&opt);
  opt.min_depth_m = (float)F.d("min_depth_m");
  opt.max_depth_m = (float)F.d("max_depth_m");
  opt.var_noise_floor = (float)F.d("var_noise_floor");
  opt.var_high_thresh = (float)F.d("var_high_thresh");
  opt.random_proposals = F.i("random_proposals");
  opt.ping_pong_iterations = F.i("ping_pong_iterations");
  opt.mismatches_start_level = F.i("mismatches_start_level");
  opt.do_bilateral_filter = F.b("do_bilateral_filter");
  opt.do_median_filter = F.b("do_median_filter");
  opt.use_foreground_masks = useFg;
  opt.partial_coverage = F.b("partial_coverage");
  opt.rebuild_warp_tables = 0;  // warps depend on rig + level size only: build once, reuse across frames
  DERP_OK(ctx, derp_set_options(ctx, &opt));
  // levels outside [levelEnd, min(levelStart + 1, numLevels - 1)] are declared absent (no HBM spent on them)
  std::vector<int> W(numLevels, 0), H(numLevels, 0);
  const int topLevel = std::min(levelStart + 1, numLevels - 1);
  for (int l = levelEnd; l <= topLevel; ++l) {
    CHECK_MSG(sizes.count(l), fmt("no images found for level %d", l));
    W[l] = sizes[l].first;
    H[l] = sizes[l].second;
  }
  DERP_OK(ctx, derp_set_pyramid(ctx, numLevels, W.data(), H.data(), widthFull, heightFull));

  for (int level = levelStart; level >= levelEnd; --level) {  // createLevelOutputDirs, DerpUtil.cpp:311-330
    for (const auto& cam : rigDst) {
      fs::create_directories(fs::path(outputRoot) / "disparity" / cam.id);
      fs::create_directories(levelDir(dispLevels.string(), level) / cam.id);
      if (F.b("save_debug_images")) {
        for (const char* t : {"cost", "confidence", "mismatches"}) {
          fs::create_directories(levelDir((fs::path(outputRoot) / t).string(), level) / cam.id);
        }
      }
    }
  }

  for (int iFrame = 0; iFrame < numFrames; ++iFrame) {
    const std::string frameName = zero_pad(iFrame + firstFrame);
    Timer frameTimer;
    // ---- inputs of every level this run touches (loadLevelImages, ImageUtil.h:79-94)
    for (int level = levelStart; level >= levelEnd; --level) {
      int w, h;
      for (int s = 0; s < S; ++s) {
        const std::vector<uint16_t> img = load_color_bgr16(image_path(levelDir(F.s("color"), level), rigSrc[s].id, frameName), w, h);
        CHECK_MSG(w == W[level] && h == H[level], fmt("image size mismatch at level %d camera %s", level, rigSrc[s].id));
        DERP_OK(ctx, derp_upload_color(ctx, level, s, img.data()));
        if (useFg) {
          const std::vector<uint8_t> m = load_mask(image_path(levelDir(F.s("foreground_masks"), level), rigSrc[s].id, frameName), w, h);
          CHECK_MSG(w == W[level] && h == H[level], "mask size mismatch");
          DERP_OK(ctx, derp_upload_foreground_mask(ctx, level, s, m.data()));
        }
      }
      if (useFg) {
        for (int d = 0; d < D; ++d) {
          const std::vector<float> bg = load_float(
              image_path(levelDir(F.s("background_disp"), level), rigDst[d].id, F.s("background_frame")), w, h);
          CHECK_MSG(w == W[level] && h == H[level], "background disparity size mismatch");
          DERP_OK(ctx, derp_upload_background_disparity(ctx, level, d, bg.data()));
        }
      }
    }
    if (useFg && levelStart < numLevels - 1) {  // coarse masks feed the masked upsample (DerpCLI.cpp:280-285)
      int w, h;
      for (int s = 0; s < S; ++s) {
        const std::vector<uint8_t> m = load_mask(image_path(levelDir(F.s("foreground_masks"), levelStart + 1), rigSrc[s].id, frameName), w, h);
        DERP_OK(ctx, derp_upload_foreground_mask(ctx, levelStart + 1, s, m.data()));
      }
    }
    if (levelStart < numLevels - 1) {  // resume: previous level from disk (DerpCLI.cpp:287-288)
      int w, h;
      for (int d = 0; d < D; ++d) {
        const std::vector<float> prev = load_float(image_path(levelDir(dispLevels.string(), levelStart + 1), rigDst[d].id, frameName, ".pfm"), w, h);
        CHECK_MSG(w == W[levelStart + 1] && h == H[levelStart + 1], "previous-level disparity size mismatch");
        DERP_OK(ctx, derp_upload_disparity(ctx, levelStart + 1, d, prev.data()));
      }
    }
    // ---- the level loop (DerpCLI.cpp:220-323)
    for (int level = levelStart; level >= levelEnd; --level) {
      LOG_INFO(fmt("Processing %s level %d", frameName.c_str(), level));
      DERP_OK(ctx, derp_process_level(ctx, level));
      DERP_OK(ctx, derp_synchronize(ctx));
      // saveResults (PyramidLevel.h:487-529): PFM always, PNG on request
      std::vector<float> disp((size_t)W[level] * H[level]);
      for (int d = 0; d < D; ++d) {
        DERP_OK(ctx, derp_download_disparity(ctx, level, d, disp.data()));
        const fs::path base = levelDir(dispLevels.string(), level) / rigDst[d].id;
        write_pfm(base / (frameName + ".pfm"), disp.data(), W[level], H[level]);
        if (savePng) {
          write_disparity_png(base / (frameName + ".png"), disp.data(), W[level], H[level]);
        }
        if (F.b("save_debug_images")) {
          std::vector<float> cost(disp.size()), conf(disp.size());
          DERP_OK(ctx, derp_download_cost(ctx, d, cost.data(), conf.data()));
          for (auto& v : cost) {
            v *= 255.0f / 100.0f / 65535.0f * 257.0f;  // kScaleCostPlot, 8-bit range in a 16-bit file
          }
          write_disparity_png(levelDir((fs::path(outputRoot) / "cost").string(), level) / rigDst[d].id / (frameName + ".png"),
                              cost.data(), W[level], H[level]);
          for (auto& v : conf) {
            v *= 255.0f * 100.0f / 65535.0f * 257.0f;  // kScaleConfidencePlot
          }
          write_disparity_png(levelDir((fs::path(outputRoot) / "confidence").string(), level) / rigDst[d].id / (frameName + ".png"),
                              conf.data(), W[level], H[level]);
          std::vector<uint8_t> mm(disp.size());
          DERP_OK(ctx, derp_download_mismatch_mask(ctx, d, mm.data()));
          std::vector<uint16_t> mpx(mm.size());
          for (size_t i = 0; i < mm.size(); ++i) {
            mpx[i] = mm[i] ? 255 : 0;
          }
          write_png(levelDir((fs::path(outputRoot) / "mismatches").string(), level) / rigDst[d].id / (frameName + ".png"),
                    mpx.data(), W[level], H[level], 1, 8);
        }
      }
      LOG_INFO(fmt("-- Elapsed time: %.3fs wall (frame %s, level %d)", frameTimer.s(), frameName.c_str(), level));
    }
  }
  uint64_t nCost = 0, nPair = 0, insufficient = 0;
  derp_get_counters(ctx, &nCost, &nPair, &insufficient);
  if (insufficient) {
    LOG_WARNING(fmt("Insufficient coverage at %llu pixel(s) due to partial coverage or noisy foreground masks",
                    (unsigned long long)insufficient));
  }
  char name[256];
  derp_device_name(ctx, name, sizeof name);
  LOG_INFO(fmt("-- TOTAL: %.3fs wall on %s", total.s(), name));
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
