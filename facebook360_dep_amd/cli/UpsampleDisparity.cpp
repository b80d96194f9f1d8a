// UpsampleDisparity — drop-in for source/depth_estimation/UpsampleDisparity.cpp: same flags
// (:37-55) and layout (<output>/<cam>/<frame>.pfm). Compute = derp_upsample_disparity +
// derp_joint_bilateral_f32 (BASELINE config 5's combined bilateral upsample).
#include "cli_common.h"

using namespace cli;

static const char* kUsage = R"(
  - Upsamples disparities to a given resolution, optionally refining with a colour-guided
    joint bilateral filter and foreground masks.

  - Example:
    ./UpsampleDisparity \
    --rig=/path/to/rigs/rig.json \
    --disparity=/path/to/output/disparity \
    --output=/path/to/output/disparity_upsample \
    --resolution=2048 \
    --color=/path/to/video/color
)";

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  F.str("background_disp", "", "background disparity directory (output resolution)");
  F.str("background_frame", "000000", "background frame (lexical)");
  F.str("cameras", "", "destination cameras");
  F.str("color", "", "color directory (output resolution)");
  F.str("disparity", "", "disparity directory (input resolution) (required)");
  F.str("first", "000000", "first frame to process (lexical)");
  F.str("foreground_masks_in", "", "(optional) masks directory (input resolution)");
  F.str("foreground_masks_out", "", "(optional) masks directory (output resolution)");
  F.i32("height", -1, "output image height (aspect ratio maintained if unspecified)");
  F.str("last", "000000", "last frame to process (lexical)");
  F.str("output", "", "output directory (required)");
  F.str("output_formats", "", "saved formats, comma separated (exr, png, pfm supported)");
  F.i32("resolution", -1, "output resolution width in pixels (required)");
  F.str("rig", "", "path to camera rig .json");
  F.dbl("sigma", 0.05, "bilateral filter color difference sigma");
  F.i32("threads", -1, "number of threads (-1 = auto, 0 = none) [accepted; the GPU path ignores it]");
  F.dbl("weight_b", 0.5, "bilateral filter blue channel weight");
  F.dbl("weight_g", 0.5, "bilateral filter green channel weight");
  F.dbl("weight_r", 1.0, "bilateral filter red channel weight");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  CHECK_MSG(F.s("disparity") != "", "disparity");
  CHECK_MSG(F.s("output") != "", "output");
  CHECK_MSG(F.i("resolution") != -1, "resolution");
  const std::vector<derp_camera_desc> rigSrc = load_rig(F.s("rig"));
  const std::vector<derp_camera_desc> rigDst = filter_destinations(rigSrc, F.s("cameras"));
  CHECK_MSG(!rigDst.empty(), "no destination cameras!");
  verify_image_paths(F.s("disparity"), rigDst, F.s("first"), F.s("last"));
  derp_ctx* ctx = nullptr;
  if (derp_create(&ctx, F.i("device"), rigSrc.data(), (int)rigSrc.size(), rigDst.data(), (int)rigDst.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  const std::string exts = F.s("output_formats").empty() ? "pfm" : F.s("output_formats");
  int height = F.i("height");
  if (height == -1) {  // UpsampleDisparity.cpp:91-98
    height = (int)std::round(float(rigDst[0].resolution[1]) / rigDst[0].resolution[0] * F.i("resolution"));
    height += height % 2;
  }
  const int wUp = F.i("resolution"), hUp = height;
  const bool useFg = !F.s("foreground_masks_in").empty();

  for (int f = std::stoi(F.s("first")); f <= std::stoi(F.s("last")); ++f) {
    const std::string frame = zero_pad(f);
    for (size_t i = 0; i < rigDst.size(); ++i) {
      int w, h, w2, h2;
      const std::vector<float> disp = load_float(image_path(F.s("disparity"), rigDst[i].id, frame), w, h);
      std::vector<float> bgUp;
      if (!F.s("background_disp").empty()) {
        bgUp = load_float(image_path(F.s("background_disp"), rigDst[i].id, F.s("background_frame")), w2, h2);
        CHECK_MSG(w2 == wUp && h2 == hUp, "background disparity must be at the output resolution");
      }
      std::vector<uint8_t> maskIn((size_t)w * h, 1), maskUp((size_t)wUp * hUp, 1);
      if (useFg) {
        maskIn = load_mask(image_path(F.s("foreground_masks_in"), rigDst[i].id, frame), w2, h2);
        CHECK_MSG(w2 == w && h2 == h, "foreground_masks_in size mismatch");
        CHECK_MSG(!bgUp.empty(), "foreground masks need --background_disp");
      }
      if (!F.s("foreground_masks_out").empty()) {
        maskUp = load_mask(image_path(F.s("foreground_masks_out"), rigDst[i].id, frame), w2, h2);
        CHECK_MSG(w2 == wUp && h2 == hUp, "foreground_masks_out must be at the output resolution");
      }
      std::vector<float> up((size_t)wUp * hUp);
      if (w == wUp && h == hUp && !useFg) {
        up = disp;  // cv::resize to the same size: NaN -> 1e-4, values unchanged
        for (auto& v : up) {
          if (v != v) {
            v = 1e-4f;
          }
        }
      } else {
        DERP_OK(ctx, derp_upsample_disparity(ctx, (int)i, disp.data(), w, h, useFg ? bgUp.data() : nullptr,
                                             useFg ? maskIn.data() : nullptr, useFg ? maskUp.data() : nullptr, wUp, hUp,
                                             useFg, up.data()));
      }
      if (!F.s("color").empty()) {
        // getRadius (UpsampleDisparityLib.cpp:93-96)
        const float scale = float(wUp) / float(w);
        const int radius = (int)(scale * scale + 1);
        LOG_INFO(fmt("Applying filter with radius %d to %dx%d disparity to %s...", radius, wUp, hUp, rigDst[i].id));
        const std::vector<uint16_t> c16 = load_color_bgr16(image_path(F.s("color"), rigDst[i].id, frame), w2, h2);
        // a guide of any size is resized to the output (cv_util::resizeImage, CvUtil.h:139-147): INTER_AREA when it is
        // larger — what the pipeline passes (pipeline.py:410-411) — and OpenCV's bilinear emulation of it when smaller
        std::vector<float> guide(c16.size());
        const float s = 1.0f / 65535.0f;  // loadImage<Vec3f>: convertTo(CV_32F, 1/65535)
        for (size_t k = 0; k < c16.size(); ++k) {
          guide[k] = c16[k] * s;
        }
        if (w2 != wUp || h2 != hUp) {
          // colorUp = cv_util::resizeImage(colors[i], sizeUp): INTER_AREA on Vec3f (UpsampleDisparity.cpp:117,
          // CvUtil.h:139-147)
          std::vector<float> guideUp((size_t)wUp * hUp * 3);
          DERP_OK(ctx, derp_resize_area(ctx, 3, guide.data(), w2, h2, guideUp.data(), wUp, hUp));
          guide.swap(guideUp);
        }
        std::vector<float> filtered(up.size());
        DERP_OK(ctx, derp_joint_bilateral_f32(ctx, up.data(), guide.data(), maskUp.data(), wUp, hUp, radius,
                                              (float)F.d("sigma"), (float)F.d("weight_b"), (float)F.d("weight_g"),
                                              (float)F.d("weight_r"), filtered.data()));
        up.swap(filtered);
      }
      LOG_INFO("Saving output images...");
      const fs::path dir = fs::path(F.s("output")) / rigDst[i].id;
      fs::create_directories(dir);
      std::stringstream ss(exts);
      std::string ext;
      while (std::getline(ss, ext, ',')) {
        if (ext == "pfm" || ext == ".pfm") {
          write_pfm(dir / (frame + ".pfm"), up.data(), wUp, hUp);
        } else if (ext == "png" || ext == ".png") {
          write_disparity_png(dir / (frame + ".png"), up.data(), wUp, hUp);
        } else if (ext == "exr" || ext == ".exr") {
          write_exr_f32(dir / (frame + ".exr"), up.data(), wUp, hUp);
        } else if (!ext.empty()) {
          LOG_WARNING("output format not supported by this build: " + ext);
        }
      }
    }
  }
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
