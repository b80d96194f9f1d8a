// TemporalBilateralFilter — drop-in for source/depth_estimation/TemporalBilateralFilter.cpp:
// same flags (:40-59), inputs and output layout
// (<output_root>/disparity_time_filtered_levels/level_L/<cam>/<frame>.pfm). Compute = derp_temporal_filter.
#include "derp_job.h"

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

// The fast path. filterFrame (TemporalBilateralFilter.cpp:121-184) loads, for every frame it filters and every camera,
// all frames of the window again: with time_radius 2 every colour image and every disparity map of the chunk is decoded
// five times, on one thread (the reference does the same; at 16 x 2048^2 x 8 frames that is 640 PNG inflations and 52 s
// for the finest level in this build's round-4 binary — profiles/r05_pipeline_timing.txt). When the windows are those
// of one contiguous run of frames A..B — every file of [first - R, last + R] that exists, which is what
// populateMinMaxFrame (:96-119) yields whenever no file is missing in the middle — the level runs on the sequence
// driver instead (derp_seq_*, the machinery of bin/DerpSequence): every frame of A..B is decoded ONCE by the worker
// pool (starting before the HIP runtime does), uploaded once, a frame is filtered as soon as its window is in HBM,
// and its files are written by the pool behind the next frames. Same kernel, same windows, same bytes on disk
// (tests/test_gpu_cli.py). Anything else — gaps in the frame numbering — takes the frame-by-frame path below.
static bool run_on_sequence_engine(Flags& F, const std::vector<derp_camera_desc>& rigDst, Timer& total) {
  const int level = F.i("level"), R = F.i("time_radius");
  const int first = std::stoi(F.s("first")), last = std::stoi(F.s("last"));
  const bool useFg = F.b("use_foreground_masks");
  if (first > last) {
    return false;
  }
  // the windows filterFrame would use
  int A = INT32_MAX, B = 0;
  std::vector<std::pair<int, int>> win;
  for (int cur = first; cur <= last; ++cur) {
    int wf = 0, wl = INT32_MAX;
    populate_min_max(F.s("color"), level, rigDst[0].id, cur, R, wf, wl);
    populate_min_max(F.s("disparity"), level, rigDst[0].id, cur, R, wf, wl);
    if (useFg) {
      populate_min_max(F.s("foreground_masks"), level, rigDst[0].id, cur, R, wf, wl);
    }
    if (!(wf <= cur && cur <= wl)) {
      return false;  // the frame-by-frame path reports it
    }
    win.emplace_back(wf, wl);
    A = std::min(A, wf);
    B = std::max(B, wl);
  }
  for (int cur = first; cur <= last; ++cur) {
    if (win[cur - first].first != std::max(cur - R, A) || win[cur - first].second != std::min(cur + R, B)) {
      return false;
    }
  }
  // every file of A..B must exist for every camera (the frame-by-frame path would fail on the first missing one anyway)
  for (int f = A; f <= B; ++f) {
    for (const auto& cam : rigDst) {
      for (const std::string& dir : {F.s("color"), F.s("disparity")}) {
        const fs::path camDir = fs::path(dir) / ("level_" + std::to_string(level)) / cam.id;
        std::error_code ec;
        if (!fs::exists(camDir / (zero_pad(f) + first_extension(camDir)), ec)) {
          return false;
        }
      }
    }
  }
  // DerpJob's view of the same job: one level, frames A..B, raw disparities from --disparity
  Flags G;
  define_derp_flags(G);
  G.program = F.program;
  for (const char* k : {"input_root", "output_root", "rig", "color", "foreground_masks", "cameras", "output_formats", "threads", "device"}) {
    G.set(k, F.s(k));
  }
  G.str("disparity", F.s("disparity"), "");
  G.set("first", zero_pad(A));
  G.set("last", zero_pad(B));
  G.set("level_start", std::to_string(level));
  G.set("level_end", std::to_string(level));
  G.set("resolution", std::to_string(1 << 30));  // the level is named explicitly: no end level from a width
  G.set("use_foreground_masks", useFg ? "true" : "false");
  DerpJob J(G);
  J.filterOnly = true;
  J.setup_host();
  // every frame of A..B stays in HBM for the level (colour 8 B, raw + filtered disparity 8 B, masks 2 B per pixel and
  // camera, + the window-only frames' scratch): a chunk that would not fit comfortably takes the frame-by-frame path,
  // which holds one window at a time. First against a cap that needs no device (DERP_TBF_HBM_BUDGET_GB), then — once the
  // context exists — against what the device really has free.
  const double residentBytes = (double)(B - A + 1) * J.D * (double)J.npx(level) * 20.0;
  {
    const char* e = getenv("DERP_TBF_HBM_BUDGET_GB");
    if (residentBytes > (e ? atof(e) : 128.0) * 1e9) {
      LOG_INFO(fmt("frames %06d..%06d of level %d need %.1f GB resident: filtering frame by frame instead", A, B, level, residentBytes / 1e9));
      return false;
    }
  }
  std::vector<int> owned;
  for (int f = A; f <= B; ++f) {
    owned.push_back(f);
  }
  const int nOwned = (int)owned.size();
  IoPool pool(F.i("threads"));
  FrameStore store(J, pool, owned);
  for (int k = 0; k < nOwned; ++k) {
    store.schedule(k, level);
  }
  store.pump();
  const double tHost = total.s();
  J.setup_device(-1);
  derp_ctx* ctx = J.ctx;
  LOG_INFO(fmt("-- start-up: flags + rig + input check %.3fs, HIP runtime + context %.3fs (images decoding since %.3fs)", tHost,
               total.s() - tHost, tHost));
  {
    uint64_t freeB = 0, totalB = 0;
    DERP_OK(ctx, derp_device_memory(ctx, &freeB, &totalB));
    if (residentBytes > 0.85 * (double)freeB) {  // a smaller or shared device: the window-at-a-time path still fits
      LOG_INFO(fmt("frames %06d..%06d of level %d need %.1f GB resident, the device has %.1f GB free: filtering frame by frame "
                   "instead", A, B, level, residentBytes / 1e9, (double)freeB / 1e9));
      derp_destroy(ctx);
      J.ctx = nullptr;
      return false;
    }
  }
  derp_seq_options so;
  derp_seq_options_default(&so);
  so.time_radius = R;
  so.sigma = (float)F.d("sigma");
  so.weight_b = (float)F.d("weight_b");
  so.weight_g = (float)F.d("weight_g");
  so.weight_r = (float)F.d("weight_r");
  so.space_radius = F.i("space_radius");
  so.use_foreground_masks = useFg;
  so.do_temporal_filter = 1;
  derp_seq* seq = nullptr;
  DERP_OK(ctx, derp_seq_create(&seq, ctx, A, B, 0, 1, &so));
  LevelWriter writer(J, pool);
  store.reserve_bounce(level);
  writer.reserve_ring(3);
  const fs::path outDir = fs::path(F.s("output_root")) / "disparity_time_filtered_levels";
  for (const auto& cam : rigDst) {
    fs::create_directories(DerpJob::levelDir(outDir, level) / cam.id);
  }
  const std::vector<fs::path> dirs{outDir};
  std::vector<char> filtered(nOwned, 0), saved(nOwned, 0);
  double tFilter = 0;
  auto wanted = [&](int j) { return owned[j] >= first && owned[j] <= last; };
  for (int k = 0; k < nOwned; ++k) {
    store.wait(k, level);
    store.hand_over(seq, k, level, true);
    DERP_OK(ctx, derp_seq_level_provided_frame(seq, level, owned[k]));
    std::vector<int> ready;  // filtered in an earlier iteration: written while this frame's filter runs
    for (int j = 0; j < nOwned; ++j) {
      if (filtered[j] && !saved[j]) {
        ready.push_back(j);
      }
    }
    Timer t;
    for (int j = 0; j <= k; ++j) {
      if (wanted(j) && !filtered[j]) {
        const int rc = derp_seq_level_filter_frame(seq, level, owned[j]);
        if (rc == 2) {
          continue;  // a frame of its window is still to come
        }
        DERP_OK(ctx, rc);
        filtered[j] = 1;
      }
    }
    tFilter += t.s();
    for (int j : ready) {
      writer.save_seq(seq, owned[j], level, zero_pad(owned[j]), dirs, true, true);
      saved[j] = 1;
    }
  }
  for (int j = 0; j < nOwned; ++j) {
    if (wanted(j)) {
      CHECK_MSG(filtered[j], fmt("frame %06d could not be filtered", owned[j]));
      if (!saved[j]) {
        writer.save_seq(seq, owned[j], level, zero_pad(owned[j]), dirs, true, true);
      }
    }
  }
  writer.finish();
  LOG_INFO(fmt("-- filter: %d frame(s) of level %d read once each (%d I/O threads): waited for decode %.3fs, filter calls %.3fs, "
               "downloads incl. waiting for the GPU %.3fs, waited for writes %.3fs", nOwned, level, (int)pool.workers.size(),
               store.waited, tFilter, writer.downloading, writer.waited));
  LOG_INFO(fmt("-- TOTAL: %.3fs wall", total.s()));
  derp_seq_destroy(seq);
  derp_destroy(ctx);
  return true;
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
  F.i32("threads", -1, "number of threads (-1 = auto, 0 = none) [here: image decode / file write workers; the filter is the GPU's]");
  F.i32("time_radius", 2, "temporal filtering radius");
  F.boolean("use_foreground_masks", false, "use pre-computed foreground masks");
  F.dbl("weight_b", 0.5, "Blue channel weight");
  F.dbl("weight_g", 1.0, "Green channel weight");
  F.dbl("weight_r", 1.0, "Red channel weight");
  F.i32("device", 0, "HIP device index [extension]");
  F.parse(argc, argv);
  Timer total;
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
  if (!getenv("DERP_TBF_LEGACY") && run_on_sequence_engine(F, rigDst, total)) {
    return EXIT_SUCCESS;
  }
  derp_ctx* ctx = nullptr;
  const double tHost = total.s();
  if (derp_create(&ctx, F.i("device"), rigSrc.data(), (int)rigSrc.size(), rigDst.data(), (int)rigDst.size()) != 0) {
    LOG_FATAL(std::string("derp_create failed: ") + derp_last_error(nullptr));
  }
  LOG_INFO(fmt("-- start-up: flags + rig %.3fs, HIP runtime + context %.3fs", tHost, total.s() - tHost));
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
  LOG_INFO(fmt("-- TOTAL: %.3fs wall", total.s()));
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
