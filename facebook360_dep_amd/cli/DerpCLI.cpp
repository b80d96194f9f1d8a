// DerpCLI — drop-in for the reference's source/depth_estimation/DerpCLI.cpp: same flag names and
// defaults (DerpCLI.cpp:40-67), same inputs (rig JSON, <color>/level_L/<cam>/<frame>.<ext>, masks,
// background disparities, previous-level PFMs) and the same outputs
// (<output_root>/disparity_levels/level_L/<cam>/<frame>.pfm [+ .png]). All computation happens in
// libderp_hip.so through include/derp_hip.h; there is no CPU path.
//
// The reference iterates level-outer / frame-inner (DerpCLI.cpp:220-229); frames are independent
// inside DerpCLI, so this driver goes frame-outer and keeps one frame's pyramid resident in HBM
// from the coarsest requested level to the finest — identical files, no PFM round trip in between.
// Around the GPU runs an I/O pipeline (derp_job.h): frame f + 1 is decoded by the worker pool into
// page-locked staging memory while frame f computes, and finished levels are written by the pool while
// the next level / frame computes.
#include "derp_job.h"

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
  define_derp_flags(F);
  F.parse(argc, argv);
  Timer total;
  DerpJob J(F);
  J.setup_host();
  const double tHost = total.s();
  J.setup_device(-1);
  const double tDevice = total.s();
  J.create_output_dirs();
  derp_ctx* ctx = J.ctx;
  LOG_INFO(fmt("-- start-up: flags + rig + input check %.3fs, HIP runtime + context %.3fs, output dirs %.3fs", tHost,
               tDevice - tHost, total.s() - tDevice));

  IoPool pool(F.i("threads"));
  FrameStager stager(J, pool);
  LevelWriter writer(J, pool);
  size_t outBytes = 0;
  std::map<int, size_t> offOut;
  for (int level = J.levelStart; level >= J.levelEnd; --level) {
    offOut[level] = outBytes;
    outBytes += J.npx(level) * 4 * J.D;
  }
  double tCompute = 0;
  stager.start_decode(J.firstFrame, 0);
  for (int iFrame = 0; iFrame < J.numFrames; ++iFrame) {
    const std::string frameName = zero_pad(iFrame + J.firstFrame);
    const int parity = iFrame & 1;
    Timer frameTimer;
    stager.wait(parity);  // this frame's inputs are decoded
    if (iFrame + 1 < J.numFrames) {
      stager.start_decode(iFrame + 1 + J.firstFrame, parity ^ 1);  // on the pool, while this frame is on the GPU
    }
    stager.upload(parity);
    writer.begin(parity, outBytes);  // the files of frame f - 2 are on disk: its download buffers are free again
    // ---- the level loop (DerpCLI.cpp:220-323)
    for (int level = J.levelStart; level >= J.levelEnd; --level) {
      LOG_INFO(fmt("Processing %s level %d", frameName.c_str(), level));
      {
        Timer t;
        DERP_OK(ctx, derp_process_level(ctx, level));
        DERP_OK(ctx, derp_synchronize(ctx));
        tCompute += t.s();
      }
      writer.save(parity, offOut[level], level, frameName, {J.dispLevels});
      if (F.b("save_debug_images")) {
        save_debug_images(J, level, frameName);
      }
      LOG_INFO(fmt("-- Elapsed time: %.3fs wall (frame %s, level %d)", frameTimer.s(), frameName.c_str(), level));
    }
  }
  writer.finish();
  LOG_INFO(fmt("-- I/O vs compute over %d frame(s): waited for decode %.3fs, upload %.3fs, compute %.3fs, download %.3fs, "
               "waited for writes %.3fs (%d I/O threads; decode and file writes overlap the compute)",
               J.numFrames, stager.waited, stager.uploading, tCompute, writer.downloading, writer.waited,
               (int)pool.workers.size()));
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
