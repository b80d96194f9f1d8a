// ComputeRephotographyErrors — stands in for source/render/ComputeRephotographyErrors.cpp, the
// reference's quality gate for DerpCLI (scripts/test/test_derp_cli.py:64-100 expects 90 % +- 5 %):
// same flags (:42-50), same log lines ("<cam> MSSIM: R ..%, G ..%, B ..%", "<frame> average ...",
// "TOTAL average MSSIM: R ..%, G ..%, B ..%" as the last line of <log_dir>/<program>.INFO), plots under
// <output>/rephoto/<cam>/<frame>.png.
// Difference, by construction: the reference compares OpenGL cubemaps of disparity meshes centred on
// each camera (CanopyScene); this build has no OpenGL renderer, so both sides live in the camera's own
// image: reference = the camera's colour where its disparity is valid, rendered = derp_rephotograph
// of all the other cameras. The score arithmetic (derp_ssim / derp_average_score) is the reference's.
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
   - Computes rephotography error for a set of frames: every camera is re-rendered from the colour
   and disparity of all the other cameras and compared with what it actually saw (MSSIM or NCC).

   - Example:
     ./ComputeRephotographyErrors \
     --first=000000 \
     --last=000000 \
     --output=/path/to/output \
     --rig=/path/to/rigs/rig.json \
     --color=/path/to/video/color \
     --disparity=/path/to/output/disparity
)";

static std::string format_results(const double* avg) {  // rephoto_util::formatResults
  return fmt("R %.2f%%, G %.2f%%, B %.2f%%", 100 * avg[2], 100 * avg[1], 100 * avg[0]);
}

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("cameras", "", "comma-separated cameras to render (empty for all)");
  F.str("color", "", "path to input color images (required)");
  F.str("disparity", "", "path to disparity images (required)");
  F.str("first", "", "first frame to process (lexical) (required)");
  F.str("last", "", "last frame to process (lexical) (required)");
  F.str("method", "MSSIM", "MSSIM or NCC");
  F.str("output", "", "path to output directory (required)");
  F.str("rig", "", "path to camera rig .json (required)");
  F.i32("stat_radius", 1, "local statistics window radius");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  CHECK_MSG(F.s("color") != "", "color");
  CHECK_MSG(F.s("disparity") != "", "disparity");
  CHECK_MSG(F.s("first") != "", "first");
  CHECK_MSG(F.s("last") != "", "last");
  CHECK_MSG(F.s("output") != "", "output");
  CHECK_MSG(F.s("rig") != "", "rig");
  const std::string method = F.s("method");
  CHECK_MSG(method == "MSSIM" || method == "NCC", ("Invalid method " + method).c_str());
  CHECK_MSG(F.i("stat_radius") > 0, "blurRadius > 0");
  const float abg = method == "MSSIM" ? 1.0f : 0.0f;

  const std::vector<derp_camera_desc> rig = load_rig(F.s("rig"));
  CHECK_MSG(rig.size() > 1, "rig.size() > 1");
  std::vector<std::string> only;
  {
    std::stringstream ss(F.s("cameras"));
    std::string item;
    while (std::getline(ss, item, ',')) {
      if (!item.empty()) {
        only.push_back(item);
      }
    }
  }
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rig.data(), (int)rig.size(), rig.data(), (int)rig.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  const fs::path rephotoDir = fs::path(F.s("output")) / "rephoto";
  for (const auto& cam : rig) {
    fs::create_directories(rephotoDir / cam.id);
  }
  const int first = std::stoi(F.s("first"));
  const int numFrames = std::stoi(F.s("last")) - first + 1;
  CHECK_MSG(numFrames > 0, "numFrames > 0");
  double total[3] = {0, 0, 0};
  for (int iFrame = 0; iFrame < numFrames; ++iFrame) {
    const std::string frame = zero_pad(iFrame + first);
    LOG_INFO("Processing frame " + frame + "...");
    LOG_INFO("Loading color and disparity images...");
    int w = 0, h = 0;
    std::vector<std::vector<float>> disps(rig.size());
    std::vector<std::vector<uint16_t>> colors(rig.size());
    for (size_t i = 0; i < rig.size(); ++i) {
      int dw, dh;
      disps[i] = read_pfm(fs::path(F.s("disparity")) / rig[i].id / (frame + ".pfm"), dw, dh);
      if (i == 0) {
        w = dw;
        h = dh;
      }
      CHECK_MSG(dw == w && dh == h, "disparity sizes differ between cameras");
      int cw, ch;
      std::vector<uint16_t> img = load_color_bgr16(image_path(F.s("color"), rig[i].id, frame), cw, ch);
      if (cw != w || ch != h) {  // loadResizedImages(..., disps[0].size(), INTER_AREA)
        std::vector<uint16_t> out((size_t)w * h * 3);
        DERP_OK(ctx, derp_resize_area(ctx, 0, img.data(), cw, ch, out.data(), w, h));
        img.swap(out);
      }
      colors[i].swap(img);
    }
    const size_t n = (size_t)w * h;
    std::vector<const uint16_t*> cp(rig.size());
    std::vector<const float*> dp(rig.size());
    for (size_t i = 0; i < rig.size(); ++i) {
      cp[i] = colors[i].data();
      dp[i] = disps[i].data();
    }
    DERP_OK(ctx, derp_rephotograph_upload(ctx, cp.data(), dp.data(), w, h));
    double frameScore[3] = {0, 0, 0};
    int used = 0;
    for (size_t i = 0; i < rig.size(); ++i) {
      const std::string camId = rig[i].id;
      if (!only.empty() && std::find(only.begin(), only.end(), camId) == only.end()) {
        continue;
      }
      LOG_INFO("Processing " + frame + " - " + camId + "...");
      std::vector<float> rendered(n * 4);
      DERP_OK(ctx, derp_rephotograph_render(ctx, (int)i, rendered.data()));
      // reference side: own colour in [0, 1]; mask = own disparity valid (the cubemap's alpha > 0)
      std::vector<float> x(n * 3), y(n * 3);
      std::vector<uint8_t> mask(n);
      const float s = 1.0f / 65535.0f;
      for (size_t k = 0; k < n; ++k) {
        const float d = disps[i][k];
        mask[k] = (d > 0) && !std::isinf(d);
        for (int c = 0; c < 3; ++c) {
          x[3 * k + c] = mask[k] ? colors[i][3 * k + c] * s : 0.0f;  // zeroOutNans'd, alpha-less reference
          y[3 * k + c] = rendered[4 * k + c];
        }
      }
      std::vector<float> score(n * 3);
      DERP_OK(ctx, derp_ssim(ctx, x.data(), y.data(), w, h, F.i("stat_radius"), abg, abg, 1.0f, score.data()));
      double avg[3];
      CHECK_MSG(derp_average_score(score.data(), mask.data(), w, h, avg) == 0, "derp_average_score");
      LOG_INFO(camId + " " + method + ": " + format_results(avg));
      for (int c = 0; c < 3; ++c) {
        frameScore[c] += avg[c];
      }
      ++used;
      // plot: reference | rendered (masked) | score, 8 bit, side by side (stackResults without the
      // colour map and the caption)
      std::vector<uint16_t> plot((size_t)3 * w * h * 3);
      auto to8 = [](float v) { return (uint16_t)(v <= 0 ? 0 : v >= 1 ? 255 : lrintf(v * 255.0f)); };
      for (int yy = 0; yy < h; ++yy) {
        for (int xx = 0; xx < w; ++xx) {
          const size_t k = (size_t)yy * w + xx;
          for (int c = 0; c < 3; ++c) {
            const int rgb = 2 - c;  // write_png takes RGB
            plot[((size_t)yy * 3 * w + xx) * 3 + rgb] = to8(x[3 * k + c]);
            plot[((size_t)yy * 3 * w + w + xx) * 3 + rgb] = mask[k] ? to8(y[3 * k + c]) : 0;
            const float sc = score[3 * k + c];
            plot[((size_t)yy * 3 * w + 2 * w + xx) * 3 + rgb] = mask[k] && sc == sc ? to8(sc) : 0;
          }
        }
      }
      write_png(rephotoDir / camId / (frame + ".png"), plot.data(), 3 * w, h, 3, 8);
    }
    const int nCams = !only.empty() ? (int)only.size() : (int)rig.size();
    (void)used;
    for (int c = 0; c < 3; ++c) {
      frameScore[c] /= nCams;
      total[c] += frameScore[c];
    }
    LOG_INFO(frame + " average " + method + ": " + format_results(frameScore));
  }
  for (int c = 0; c < 3; ++c) {
    total[c] /= numFrames;
  }
  LOG_INFO("TOTAL average " + method + ": " + format_results(total));
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
