// ComputeRephotographyErrors — drop-in for source/render/ComputeRephotographyErrors.cpp, the reference's
// quality gate for DerpCLI (scripts/test/test_derp_cli.py:64-100 expects 90 % +- 5 %): same flags (:42-50),
// same log lines ("<cam> MSSIM: R ..%, G ..%, B ..%", "<frame> average ...", "TOTAL average MSSIM: R ..%,
// G ..%, B ..%" as the last line of <log_dir>/<program>.INFO), plots under <output>/rephoto/<cam>/<frame>.png.
// For every camera i, two cubemaps centred on it are rendered from disparity meshes — the camera alone, and all
// the other cameras (generateCubemaps, :77-95) — by derp_canopy_cubemap, a HIP software rasteriser with
// CanopyScene's pipeline (the reference uses OpenGL; what GL leaves implementation-defined is fixed in
// DESIGN.md §8), and compared with the reference's score arithmetic (derp_ssim / derp_average_score) under the
// reference cubemap's alpha mask. The second cubemap pair of the reference (disparity colours) only feeds its
// plot and is not rendered.
#include "derp_job.h"

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
  IoPool pool(-1);
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
    std::vector<int> dws(rig.size()), dhs(rig.size()), cws(rig.size()), chs(rig.size());
    {  // one decode job per file on the I/O pool
      IoBatch loads;
      for (size_t i = 0; i < rig.size(); ++i) {
        loads.add(pool, [&, i] {
          disps[i] = read_pfm(fs::path(F.s("disparity")) / rig[i].id / (frame + ".pfm"), dws[i], dhs[i]);
        });
        loads.add(pool, [&, i] { colors[i] = load_color_bgr16(image_path(F.s("color"), rig[i].id, frame), cws[i], chs[i]); });
      }
      loads.wait();
    }
    w = dws[0];
    h = dhs[0];
    for (size_t i = 0; i < rig.size(); ++i) {
      CHECK_MSG(dws[i] == w && dhs[i] == h, "disparity sizes differ between cameras");
      if (cws[i] != w || chs[i] != h) {  // loadResizedImages(..., disps[0].size(), INTER_AREA)
        std::vector<uint16_t> out((size_t)w * h * 3);
        DERP_OK(ctx, derp_resize_area(ctx, 0, colors[i].data(), cws[i], chs[i], out.data(), w, h));
        colors[i].swap(out);
      }
    }
    std::vector<const uint16_t*> cp(rig.size());
    std::vector<const float*> dp(rig.size());
    for (size_t i = 0; i < rig.size(); ++i) {
      cp[i] = colors[i].data();
      dp[i] = disps[i].data();
    }
    DERP_OK(ctx, derp_rephotograph_upload(ctx, cp.data(), dp.data(), w, h));
    double frameScore[3] = {0, 0, 0};
    int used = 0;
    const int E = h;  // cubeHeight = colors[0].rows (ComputeRephotographyErrors.cpp:126)
    const size_t nc = (size_t)6 * E * E;
    // the cubemaps are hundreds of MB at full size: page-locked buffers (transfers at the PCIe rate), allocated once
    Arena aRef, aRender, aX, aY, aScore;
    aRef.ensure(nc * 4 * sizeof(float));
    aRender.ensure(nc * 4 * sizeof(float));
    aX.ensure(nc * 3 * sizeof(float));
    aY.ensure(nc * 3 * sizeof(float));
    aScore.ensure(nc * 3 * sizeof(float));
    float *cubeRef = static_cast<float*>(aRef.p), *cubeRender = static_cast<float*>(aRender.p);
    float *x = static_cast<float*>(aX.p), *y = static_cast<float*>(aY.p), *score = static_cast<float*>(aScore.p);
    std::vector<uint8_t> mask(nc);
    IoBatch plots;  // the PNG plots are encoded and written behind the next camera's rendering
    for (size_t i = 0; i < rig.size(); ++i) {
      const std::string camId = rig[i].id;
      if (!only.empty() && std::find(only.begin(), only.end(), camId) == only.end()) {
        continue;
      }
      LOG_INFO("Processing " + frame + " - " + camId + "...");
      // cubesRef = generateCubemaps({rig[i]}, ...), cubesRender = generateCubemaps(removeOne(i, rig), ...), both
      // seen from camera i's position (:137-145)
      std::vector<uint8_t> onlyI(rig.size(), 0), allButI(rig.size(), 1);
      onlyI[i] = 1;
      allButI[i] = 0;
      DERP_OK(ctx, derp_canopy_cubemap(ctx, onlyI.data(), rig[i].origin, E, cubeRef));
      DERP_OK(ctx, derp_canopy_cubemap(ctx, allButI.data(), rig[i].origin, E, cubeRender));
      // mask = 255 * (alpha > 0) of the reference cubemap; removeAlpha on both (:147-155)
      parallel_rows(pool, 6 * E, [&](int r0, int r1) {
        for (size_t k = (size_t)r0 * E; k < (size_t)r1 * E; ++k) {
          mask[k] = cubeRef[4 * k + 3] > 0;
          for (int c = 0; c < 3; ++c) {
            x[3 * k + c] = cubeRef[4 * k + c];
            y[3 * k + c] = cubeRender[4 * k + c];
          }
        }
      });
      DERP_OK(ctx, derp_ssim(ctx, x, y, E, 6 * E, F.i("stat_radius"), abg, abg, 1.0f, score));
      double avg[3];
      CHECK_MSG(derp_average_score(score, mask.data(), E, 6 * E, avg) == 0, "derp_average_score");
      LOG_INFO(camId + " " + method + ": " + format_results(avg));
      for (int c = 0; c < 3; ++c) {
        frameScore[c] += avg[c];
      }
      ++used;
      // plot: reference | rendered | score cubemaps side by side, 8 bit (stackResults without the colour map
      // and the caption)
      const int pw = 3 * E, ph = 6 * E;
      auto plot = std::make_shared<std::vector<uint16_t>>((size_t)pw * ph * 3);
      auto to8 = [](float v) { return (uint16_t)(v <= 0 ? 0 : v >= 1 ? 255 : lrintf(v * 255.0f)); };
      parallel_rows(pool, ph, [&](int r0, int r1) {
        uint16_t* out = plot->data();
        for (int yy = r0; yy < r1; ++yy) {
          for (int xx = 0; xx < E; ++xx) {
            const size_t k = (size_t)yy * E + xx;
            for (int c = 0; c < 3; ++c) {
              const int rgb = 2 - c;  // write_png takes RGB
              out[((size_t)yy * pw + xx) * 3 + rgb] = to8(x[3 * k + c]);
              out[((size_t)yy * pw + E + xx) * 3 + rgb] = mask[k] ? to8(y[3 * k + c]) : 0;
              const float sc = score[3 * k + c];
              out[((size_t)yy * pw + 2 * E + xx) * 3 + rgb] = mask[k] && sc == sc ? to8(sc) : 0;
            }
          }
        }
      });
      const fs::path plotPath = rephotoDir / camId / (frame + ".png");
      plots.add(pool, [plot, plotPath, pw, ph] { write_png(plotPath, plot->data(), pw, ph, 3, 8); });
    }
    plots.wait();
    for (Arena* a : {&aRef, &aRender, &aX, &aY, &aScore}) {
      a->release();
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
