// What DerpCLI's main does around processLevel (source/depth_estimation/DerpCLI.cpp:69-218), shared by the
// DerpCLI and DerpSequence executables: the flag table, input verification, rig and pyramid geometry, the
// HIP context, and the I/O pipeline around the GPU (worker-pool PNG / PFM decode into page-locked staging
// memory, uploads, downloads into page-locked memory, worker-pool file writes).
#pragma once
#include <deque>
#include "cli_common.h"

namespace cli {

// DerpCLI.cpp:40-67 — names, defaults and descriptions are API (scripts scrape them, system_util.py:123-176)
inline void define_derp_flags(Flags& F) {
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
  F.i32("threads", -1,
        "number of threads (-1 = auto, 0 = none) [here: image decode / file write workers; the compute is the GPU's]");
  F.boolean("use_foreground_masks", false, "use pre-computed foreground masks");
  F.dbl("var_high_thresh", 1e-3, "ignore variances higher than this threshold");
  F.dbl("var_noise_floor", 4e-5, "noise variance floor on original, full-size images");
  F.i32("device", 0, "HIP device index [extension]");
}

struct DerpJob {
  Flags& F;
  std::vector<derp_camera_desc> rigSrc, rigDst;
  int S = 0, D = 0;
  std::map<int, std::pair<int, int>> sizes;
  int numLevels = 0, levelStart = 0, levelEnd = 0, firstFrame = 0, numFrames = 0;
  bool useFg = false, savePng = false, saveExr = false;
  std::string inputRoot, outputRoot;
  fs::path dispLevels;
  std::vector<int> W, H;
  int widthFull = 0, heightFull = 0;
  derp_ctx* ctx = nullptr;
  // TemporalBilateralFilter on the sequence engine: ONE level whose raw disparities come from disk (dispLevels =
  // --disparity) instead of being computed; no background disparities, no previous level
  bool filterOnly = false;

  explicit DerpJob(Flags& f) : F(f) {}
  static fs::path levelDir(const fs::path& base, int level) { return base / ("level_" + std::to_string(level)); }
  size_t npx(int level) const { return (size_t)W[level] * H[level]; }

  // verifyInputs + rig + pyramid geometry (DerpCLI.cpp:69-218); `device` overrides --device when >= 0
  void setup(int device = -1) {
    setup_host();
    setup_device(device);
  }
  // everything that needs no GPU: flags, rig, level sizes, input verification — image decoding can start right
  // after it, while setup_device pays for the HIP start-up and the table allocations
  void setup_host() {
    CHECK_MSG(F.s("input_root") != "", "input_root");
    CHECK_MSG(F.s("output_root") != "", "output_root");
    if (F.i("level_start") >= 0 && F.i("level_end") >= 0) {
      CHECK_MSG(F.i("level_start") >= F.i("level_end"), "level_start >= level_end");
    }
    inputRoot = F.s("input_root");
    outputRoot = F.s("output_root");
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
    useFg = F.b("use_foreground_masks");
    if (useFg) {
      CHECK_MSG(filterOnly || fs::is_directory(F.s("background_disp")),
                "Asked to use background but no background disparities found in " + F.s("background_disp"));
      CHECK_MSG(fs::is_directory(F.s("foreground_masks")),
                "Asked to use foreground masks but no foreground masks found in " + F.s("foreground_masks"));
    }
    {
      std::stringstream ss(F.s("output_formats"));
      std::string f;
      while (std::getline(ss, f, ',')) {
        CHECK_MSG(f.empty() || f == "exr" || f == "png" || f == "pfm", "Invalid output format specified: " + f);
        savePng |= f == "png";
        saveExr |= f == "exr";  // PyramidLevel.h:515-516; pfm is always written, like the reference (:494)
      }
      if (F.s("output_formats").empty()) {
        LOG_WARNING("No explicit output formats specified. Forcing PFM...");
      }
    }
    // ---- rig (DerpCLI.cpp:185-192)
    rigSrc = load_rig(F.s("rig"));
    CHECK_MSG(!rigSrc.empty(), "no source cameras!");
    rigDst = filter_destinations(rigSrc, F.s("cameras"));
    CHECK_MSG(!rigDst.empty(), "no destination cameras!");
    if (filterOnly) {
      rigSrc = rigDst;  // the filter reads each destination's own colour (and mask) only
    }
    S = (int)rigSrc.size();
    D = (int)rigDst.size();
    // ---- pyramid geometry (DerpCLI.cpp:194-215)
    dispLevels = filterOnly && F.defs.count("disparity") && !F.s("disparity").empty() ? fs::path(F.s("disparity"))
                                                                                : fs::path(outputRoot) / "disparity_levels";
    pyramid_level_sizes(sizes, F.s("color"));
    pyramid_level_sizes(sizes, dispLevels);
    CHECK_MSG(!sizes.empty(), "No pyramid levels found in " + F.s("color"));
    numLevels = F.i("num_levels") == -1 ? sizes.rbegin()->first + 1 : F.i("num_levels");
    levelStart = F.i("level_start") >= 0 ? F.i("level_start") : numLevels - 1;
    levelEnd = 0;
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
    firstFrame = std::stoi(F.s("first"));
    numFrames = std::stoi(F.s("last")) - firstFrame + 1;
    // verifyInputImagePaths (DerpCLI.cpp:137-156)
    verify_image_paths(levelDir(F.s("color"), levelStart), rigSrc, F.s("first"), F.s("last"));
    if (useFg) {
      if (!filterOnly) {
        verify_image_paths(levelDir(F.s("background_disp"), levelStart), rigDst, F.s("background_frame"),
                           F.s("background_frame"));
      }
      verify_image_paths(levelDir(F.s("foreground_masks"), levelStart), rigDst, F.s("first"), F.s("last"));
    }
    if (filterOnly) {
      verify_image_paths(levelDir(dispLevels, levelStart), rigDst, F.s("first"), F.s("last"));
    } else if (levelStart < numLevels - 1) {
      verify_image_paths(levelDir(dispLevels, levelStart + 1), rigDst, F.s("first"), F.s("last"));
    }
    fs::create_directories(outputRoot);
    widthFull = (int)rigDst[0].resolution[0];
    heightFull = (int)rigDst[0].resolution[1];
    // levels outside [levelEnd, min(levelStart + 1, numLevels - 1)] are declared absent (no HBM spent on them)
    W.assign(numLevels, 0);
    H.assign(numLevels, 0);
    const int topLevel = filterOnly ? levelStart : std::min(levelStart + 1, numLevels - 1);
    for (int l = levelEnd; l <= topLevel; ++l) {
      CHECK_MSG(sizes.count(l), fmt("no images found for level %d", l));
      W[l] = sizes[l].first;
      H[l] = sizes[l].second;
    }
  }
  void setup_device(int device = -1) {
    // ---- context
    if (derp_create(&ctx, device >= 0 ? device : F.i("device"), rigSrc.data(), S, rigDst.data(), D) != 0) {
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
    DERP_OK(ctx, derp_set_pyramid(ctx, numLevels, W.data(), H.data(), widthFull, heightFull));
  }

  // createLevelOutputDirs (DerpUtil.cpp:311-330); `extra` = further per-level image types (time-filtered levels)
  void create_output_dirs(const std::vector<std::string>& extra = {}) const {
    for (int level = levelStart; level >= levelEnd; --level) {
      for (const auto& cam : rigDst) {
        fs::create_directories(fs::path(outputRoot) / "disparity" / cam.id);
        fs::create_directories(levelDir(dispLevels, level) / cam.id);
        for (const auto& t : extra) {
          fs::create_directories(levelDir(fs::path(outputRoot) / t, level) / cam.id);
        }
        if (F.b("save_debug_images")) {
          for (const char* t : {"cost", "confidence", "mismatches"}) {
            fs::create_directories(levelDir(fs::path(outputRoot) / t, level) / cam.id);
          }
        }
      }
    }
  }
};

// host memory the uploads / downloads go through: page-locked when the runtime grants it
struct Arena {
  void* p = nullptr;
  bool pinned = false;
  size_t bytes = 0;
  void ensure(size_t n) {
    if (n <= bytes) {
      return;
    }
    release();
    p = derp_host_alloc(n);
    pinned = p != nullptr;
    if (!p) {
      p = malloc(n);
    }
    CHECK_MSG(p != nullptr, "out of host memory");
    bytes = n;
  }
  void release() {
    if (p) {
      if (pinned) {
        derp_host_free(p);
      } else {
        free(p);
      }
    }
    p = nullptr;
    bytes = 0;
  }
};

// One frame's inputs (loadLevelImages, ImageUtil.h:79-94; DerpCLI.cpp:235-248,276-303): decoded by the pool into
// one of two staging arenas, then uploaded into the context's SELECTED frame slot.
struct FrameStager {
  const DerpJob& J;
  IoPool& pool;
  int inTop;
  std::map<int, size_t> offColor, offMask, offBg;
  size_t inBytes = 0, offPrev = 0;
  Arena arena[2];
  IoBatch batch[2];
  double waited = 0, uploading = 0;

  FrameStager(const DerpJob& job, IoPool& p) : J(job), pool(p) {
    // coarse masks feed the masked upsample (DerpCLI.cpp:280-285)
    inTop = J.useFg && J.levelStart < J.numLevels - 1 ? J.levelStart + 1 : J.levelStart;
    for (int level = inTop; level >= J.levelEnd; --level) {
      if (level <= J.levelStart) {
        offColor[level] = inBytes;
        inBytes += J.npx(level) * 6 * J.S;
      }
      if (J.useFg) {
        offMask[level] = inBytes;
        inBytes += (J.npx(level) * J.S + 15) / 16 * 16;
        if (level <= J.levelStart) {
          offBg[level] = inBytes;
          inBytes += J.npx(level) * 4 * J.D;
        }
      }
    }
    if (J.levelStart < J.numLevels - 1) {
      offPrev = inBytes;
      inBytes += J.npx(J.levelStart + 1) * 4 * J.D;
    }
  }
  ~FrameStager() {
    batch[0].wait();
    batch[1].wait();
    arena[0].release();
    arena[1].release();
  }

  void start_decode(int frameNumber, int parity) {
    const std::string frameName = zero_pad(frameNumber);
    Arena& A = arena[parity];
    A.ensure(inBytes);
    char* base = static_cast<char*>(A.p);
    IoBatch& B = batch[parity];
    const DerpJob* j = &J;
    const Flags& F = J.F;
    for (int level = inTop; level >= J.levelEnd; --level) {
      const int w = J.W[level], h = J.H[level];
      for (int s = 0; s < J.S; ++s) {
        if (level <= J.levelStart) {
          uint16_t* dst = reinterpret_cast<uint16_t*>(base + offColor[level]) + J.npx(level) * 3 * s;
          const fs::path path = image_path(DerpJob::levelDir(F.s("color"), level), J.rigSrc[s].id, frameName);
          B.add(pool, [=] { load_color_bgr16_into(path, dst, w, h); });
        }
        if (J.useFg) {
          uint8_t* dst = reinterpret_cast<uint8_t*>(base + offMask[level]) + J.npx(level) * s;
          const fs::path path = image_path(DerpJob::levelDir(F.s("foreground_masks"), level), J.rigSrc[s].id, frameName);
          B.add(pool, [=] {
            int mw, mh;
            const std::vector<uint8_t> m = load_mask(path, mw, mh);
            CHECK_MSG(mw == w && mh == h, "mask size mismatch: " + path.string());
            memcpy(dst, m.data(), m.size());
          });
        }
      }
      if (J.useFg && level <= J.levelStart) {
        for (int d = 0; d < J.D; ++d) {
          float* dst = reinterpret_cast<float*>(base + offBg[level]) + J.npx(level) * d;
          const fs::path path = image_path(DerpJob::levelDir(F.s("background_disp"), level), J.rigDst[d].id,
                                           F.s("background_frame"));
          B.add(pool, [=] {
            int bw, bh;
            const std::vector<float> bg = load_float(path, bw, bh);
            CHECK_MSG(bw == w && bh == h, "background disparity size mismatch: " + path.string());
            memcpy(dst, bg.data(), bg.size() * 4);
          });
        }
      }
    }
    if (J.levelStart < J.numLevels - 1) {  // resume: previous level from disk (DerpCLI.cpp:287-288)
      const int level = J.levelStart + 1, w = J.W[level], h = J.H[level];
      for (int d = 0; d < J.D; ++d) {
        float* dst = reinterpret_cast<float*>(base + offPrev) + J.npx(level) * d;
        const fs::path path = image_path(DerpJob::levelDir(j->dispLevels, level), J.rigDst[d].id, frameName, ".pfm");
        B.add(pool, [=] {
          int pw, ph;
          const std::vector<float> prev = load_float(path, pw, ph);
          CHECK_MSG(pw == w && ph == h, "previous-level disparity size mismatch: " + path.string());
          memcpy(dst, prev.data(), prev.size() * 4);
        });
      }
    }
  }

  void wait(int parity) {
    Timer t;
    batch[parity].wait();
    waited += t.s();
  }

  // into the frame slot currently selected in the context
  void upload(int parity) {
    Timer t;
    derp_ctx* ctx = J.ctx;
    const char* base = static_cast<const char*>(arena[parity].p);
    for (int level = inTop; level >= J.levelEnd; --level) {
      for (int s = 0; s < J.S; ++s) {
        if (level <= J.levelStart) {
          DERP_OK(ctx, derp_upload_color(ctx, level, s,
                                         reinterpret_cast<const uint16_t*>(base + offColor.at(level)) + J.npx(level) * 3 * s));
        }
        if (J.useFg) {
          DERP_OK(ctx, derp_upload_foreground_mask(
                           ctx, level, s, reinterpret_cast<const uint8_t*>(base + offMask.at(level)) + J.npx(level) * s));
        }
      }
      if (J.useFg && level <= J.levelStart) {
        for (int d = 0; d < J.D; ++d) {
          DERP_OK(ctx, derp_upload_background_disparity(
                           ctx, level, d, reinterpret_cast<const float*>(base + offBg.at(level)) + J.npx(level) * d));
        }
      }
    }
    if (J.levelStart < J.numLevels - 1) {
      for (int d = 0; d < J.D; ++d) {
        DERP_OK(ctx, derp_upload_disparity(ctx, J.levelStart + 1, d,
                                           reinterpret_cast<const float*>(base + offPrev) + J.npx(J.levelStart + 1) * d));
      }
    }
    uploading += t.s();
  }
};

// The inputs of a rank's frames in host memory, decoded level by level (DerpSequence): one buffer per (frame,
// level), one batch of pool jobs per (frame, level) that the level loop waits for individually, handed to the
// sequence driver with derp_seq_host_inputs (resident mode: uploaded then; out of core: streamed by the library
// whenever the frame's level is needed — the buffers stay alive until the store is destroyed).
struct FrameStore {
  const DerpJob& J;
  IoPool& pool;
  std::vector<int> frames;
  int inTop;
  // uninitialised host memory (a value-initialising container would write 4 GB of zeros before the first decode)
  template <typename T>
  struct Raw {
    T* p = nullptr;
    size_t n = 0;
    Raw() = default;
    Raw(const Raw&) = delete;
    Raw& operator=(const Raw&) = delete;
    Raw(Raw&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
    ~Raw() { free(p); }
    void resize(size_t count) {
      free(p);
      p = static_cast<T*>(malloc(count * sizeof(T)));
      CHECK_MSG(p != nullptr || count == 0, "out of host memory");
      n = count;
    }
    void release() {
      free(p);
      p = nullptr;
      n = 0;
    }
    T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
  };
  struct Level {
    Raw<uint16_t> color;  // [S][n][3]
    Raw<uint8_t> mask;    // [S][n]
    Raw<float> bg, prev;  // [D][n]
    std::unique_ptr<IoBatch> batch;
  };
  std::vector<std::vector<Level>> data;  // [frame index][level]
  double waited = 0;
  // Decodes are started in the order the level loop consumes them (schedule / pump), and only while the host
  // memory of the started-but-not-yet-handed-over (frame, level) buffers stays under `budget`: with the frames
  // resident in HBM a buffer is freed as soon as its level is handed over, so a chunk of any length needs a
  // bounded amount of host memory (round 3 decoded every level of every frame up front and kept all of it).
  // Out of core the library streams from these buffers whenever it needs a level: they stay, nothing is throttled.
  std::deque<std::pair<int, int>> pending;  // (frame index, level), consumption order
  std::vector<std::vector<char>> started;   // [frame index][level]
  size_t inFlight = 0, budget = (size_t)8 << 30;
  bool throttle = true;

  FrameStore(const DerpJob& job, IoPool& p, const std::vector<int>& owned) : J(job), pool(p), frames(owned) {
    inTop = J.levelStart < J.numLevels - 1 && !J.filterOnly ? J.levelStart + 1 : J.levelStart;
    data.resize(frames.size());
    for (auto& f : data) {
      f.resize(J.numLevels);
      for (auto& l : f) {
        l.batch.reset(new IoBatch);
      }
    }
    started.assign(frames.size(), std::vector<char>(J.numLevels, 0));
    if (const char* e = getenv("DERP_DECODE_BUDGET_GB")) {
      budget = (size_t)(atof(e) * (double)((size_t)1 << 30));
    }
  }
  size_t level_bytes(int level) const {
    const size_t n = J.npx(level);
    const bool compute = level <= J.levelStart;
    if (J.filterOnly) {
      return n * 6 * J.S + (J.useFg ? n * J.S : 0) + n * 4 * J.D;
    }
    return (compute ? n * 6 * J.S : 0) + (J.useFg ? n * J.S + (compute ? n * 4 * J.D : 0) : 0) + (compute ? 0 : n * 4 * J.D);
  }
  void schedule(int k, int level) { pending.emplace_back(k, level); }
  // start queued decodes while they fit the budget (one is always allowed: progress)
  void pump() {
    while (!pending.empty()) {
      const auto kl = pending.front();
      const size_t b = level_bytes(kl.second);
      if (throttle && inFlight > 0 && inFlight + b > budget) {
        break;
      }
      pending.pop_front();
      inFlight += b;
      started[kl.first][kl.second] = 1;
      start_decode(kl.first, kl.second);
    }
  }
  ~FrameStore() {
    for (auto& f : data) {
      for (auto& l : f) {
        std::unique_lock<std::mutex> lk(l.batch->mu);
        l.batch->cv.wait(lk, [&] { return l.batch->pending == 0; });
      }
    }
    for (auto& b : bounce) {
      b.release();
    }
  }

  void start_decode(int k, int level) {
    const std::string frameName = zero_pad(frames[k]);
    Level& L = data[k][level];
    IoBatch& B = *L.batch;
    const Flags& F = J.F;
    const int w = J.W[level], h = J.H[level];
    const size_t n = J.npx(level);
    const bool compute = level <= J.levelStart;  // level inTop > levelStart only feeds the upsample
    if (compute) {
      L.color.resize(n * 3 * J.S);
    }
    if (J.useFg) {
      L.mask.resize(n * J.S);
      if (compute && !J.filterOnly) {
        L.bg.resize(n * J.D);
      }
    }
    for (int s = 0; s < J.S; ++s) {
      if (compute) {
        uint16_t* dst = L.color.data() + n * 3 * s;
        const fs::path path = image_path(DerpJob::levelDir(F.s("color"), level), J.rigSrc[s].id, frameName);
        B.add(pool, [=] { load_color_bgr16_into(path, dst, w, h); });
      }
      if (J.useFg) {
        uint8_t* dst = L.mask.data() + n * s;
        const fs::path path = image_path(DerpJob::levelDir(F.s("foreground_masks"), level), J.rigSrc[s].id, frameName);
        B.add(pool, [=] {
          int mw, mh;
          const std::vector<uint8_t> m = load_mask(path, mw, mh);
          CHECK_MSG(mw == w && mh == h, "mask size mismatch: " + path.string());
          memcpy(dst, m.data(), m.size());
        });
      }
    }
    if (J.useFg && compute && !J.filterOnly) {
      for (int d = 0; d < J.D; ++d) {
        float* dst = L.bg.data() + n * d;
        const fs::path path = image_path(DerpJob::levelDir(F.s("background_disp"), level), J.rigDst[d].id,
                                         F.s("background_frame"));
        B.add(pool, [=] {
          int bw, bh;
          const std::vector<float> bg = load_float(path, bw, bh);
          CHECK_MSG(bw == w && bh == h, "background disparity size mismatch: " + path.string());
          memcpy(dst, bg.data(), bg.size() * 4);
        });
      }
    }
    if (!compute || J.filterOnly) {  // resume: previous level from disk (DerpCLI.cpp:287-288); or the level to filter
      L.prev.resize(n * J.D);
      for (int d = 0; d < J.D; ++d) {
        float* dst = L.prev.data() + n * d;
        // (the filter's input may be whatever cv::imread reads, first extension of the directory: ImageUtil.h:48-56)
        const fs::path path = J.filterOnly ? image_path(DerpJob::levelDir(J.dispLevels, level), J.rigDst[d].id, frameName)
                                           : image_path(DerpJob::levelDir(J.dispLevels, level), J.rigDst[d].id, frameName, ".pfm");
        B.add(pool, [=] {
          int pw, ph;
          const std::vector<float> prev = load_float(path, pw, ph);
          CHECK_MSG(pw == w && ph == h, "previous-level disparity size mismatch: " + path.string());
          memcpy(dst, prev.data(), prev.size() * 4);
        });
      }
    }
  }

  void wait(int k, int level) {
    Timer t;
    while (!started[k][level]) {  // not started for lack of budget: the consumer is here, so it goes now
      CHECK_MSG(!pending.empty(), "frame level was never scheduled for decoding");
      const auto kl = pending.front();
      pending.pop_front();
      inFlight += level_bytes(kl.second);
      started[kl.first][kl.second] = 1;
      start_decode(kl.first, kl.second);
    }
    data[k][level].batch->wait();
    waited += t.s();
  }

  // The decoded level becomes the sequence driver's input for (frame, level). The images were decoded into
  // pageable memory (decoding starts before the HIP runtime exists, and a long sequence must not pin the host's
  // RAM), from which uploads run at ~3 GB/s. With the frame resident in HBM the large levels therefore go through
  // a small ring of page-locked bounce planes: pool workers copy plane s + 1.. while plane s moves at the PCIe rate,
  // on the library's copy stream, i.e. behind the compute of the frame before.
  // Out of core the library streams from the (pageable) buffers itself, whenever the frame's level is needed.
  static constexpr int kBounce = 4;
  Arena bounce[kBounce];
  // page-lock the bounce planes at the finest level's size once (on whichever thread calls this), not level by level
  void reserve_bounce(int level) {
    const size_t bytes = J.npx(level) * 3 * sizeof(uint16_t);
    for (auto& b : bounce) {
      b.ensure(bytes);
    }
  }
  IoBatch bounceReady[kBounce];
  void hand_over(derp_seq* seq, int k, int level, bool resident) {
    derp_ctx* ctx = J.ctx;
    Level& L = data[k][level];
    const size_t plane = J.npx(level) * 3;  // u16 elements of one camera's image
    // Every level but the thumbnails goes through the page-locked bounce planes and the copy stream (the other path
    // copies on the COMPUTE stream and waits for it). Planes under 4 MB are copied into the bounce plane right here: a
    // free pool thread can be a whole PNG inflation away. (Handing the runtime the pageable buffer directly — one
    // call for all planes — took 25 ms per frame at the 256-px level while the inflation had every CPU busy.)
    const bool ring = resident && !L.color.empty() && plane * sizeof(uint16_t) >= (64u << 10);
    if (ring) {
      const bool inlineCopy = plane * sizeof(uint16_t) < (4u << 20);
      auto stage = [&](int s) {
        Arena& A = bounce[s % kBounce];
        A.ensure(plane * sizeof(uint16_t));
        void* dst = A.p;
        const uint16_t* src = L.color.data() + plane * s;
        if (inlineCopy) {
          memcpy(dst, src, plane * sizeof(uint16_t));
        } else {
          bounceReady[s % kBounce].add(pool, [=] { memcpy(dst, src, plane * sizeof(uint16_t)); }, 0);
        }
      };
      for (int s = 0; s < std::min(kBounce, J.S); ++s) {
        stage(s);
      }
      for (int s = 0; s < J.S; ++s) {
        bounceReady[s % kBounce].wait();
        DERP_OK(ctx, derp_seq_upload_color_plane(seq, frames[k], level, s, static_cast<const uint16_t*>(bounce[s % kBounce].p)));
        if (s + kBounce < J.S) {
          stage(s + kBounce);
        }
      }
    }
    if ((!ring && !L.color.empty()) || !L.mask.empty()) {
      DERP_OK(ctx, derp_seq_host_inputs(seq, frames[k], level, ring || L.color.empty() ? nullptr : L.color.data(),
                                        L.mask.empty() ? nullptr : L.mask.data(), L.bg.empty() ? nullptr : L.bg.data()));
    }
    if (!L.prev.empty()) {
      const size_t dn = J.npx(level);
      if (ring && J.filterOnly) {
        // the level to filter is as large as the colour: through the same page-locked planes (a float plane fits a
        // colour plane's 6 bytes per pixel), pool workers copying plane d + 1.. while plane d moves at the PCIe rate
        const bool inlineCopy = dn * sizeof(float) < (4u << 20);
        auto stage = [&](int d) {
          void* dst = bounce[d % kBounce].p;
          const float* src = L.prev.data() + dn * d;
          if (inlineCopy) {
            memcpy(dst, src, dn * sizeof(float));
          } else {
            bounceReady[d % kBounce].add(pool, [=] { memcpy(dst, src, dn * sizeof(float)); }, 0);
          }
        };
        for (int d = 0; d < std::min(kBounce, J.D); ++d) {
          stage(d);
        }
        for (int d = 0; d < J.D; ++d) {
          bounceReady[d % kBounce].wait();
          DERP_OK(ctx, derp_seq_upload_disparity(seq, frames[k], level, d, static_cast<const float*>(bounce[d % kBounce].p)));
          if (d + kBounce < J.D) {
            stage(d + kBounce);
          }
        }
      } else {
        for (int d = 0; d < J.D; ++d) {
          DERP_OK(ctx, derp_seq_upload_disparity(seq, frames[k], level, d, L.prev.data() + dn * d));
        }
      }
    }
    if (resident) {  // every byte is in HBM (the calls above return after their copies): give the host memory back
      L.color.release();
      L.mask.release();
      L.bg.release();
      L.prev.release();
      inFlight -= std::min(inFlight, level_bytes(level));
      pump();
    }
  }
};

// saveResults (PyramidLevel.h:487-529): the selected frame's level is downloaded into page-locked memory and
// written by the pool (PFM always, PNG on request) into every directory of `dirs`.
struct LevelWriter {
  const DerpJob& J;
  IoPool& pool;
  Arena arena[2];
  IoBatch batch[2];
  double waited = 0, downloading = 0;
  LevelWriter(const DerpJob& job, IoPool& p) : J(job), pool(p) {}
  ~LevelWriter() {
    finish();
    arena[0].release();
    arena[1].release();
    for (auto& r : ring) {
      r.mem.release();
    }
  }
  // make arena[parity] (>= bytes) available: the files written from it earlier are on disk
  void begin(int parity, size_t bytes) {
    Timer t;
    batch[parity].wait();
    waited += t.s();
    arena[parity].ensure(bytes);
  }
  void save(int parity, size_t offset, int level, const std::string& frameName, const std::vector<fs::path>& dirs) {
    derp_ctx* ctx = J.ctx;
    const int w = J.W[level], h = J.H[level];
    const bool png = J.savePng, exr = J.saveExr;
    for (int d = 0; d < J.D; ++d) {
      float* disp = reinterpret_cast<float*>(static_cast<char*>(arena[parity].p) + offset) + J.npx(level) * d;
      {
        Timer t;
        DERP_OK(ctx, derp_download_disparity(ctx, level, d, disp));
        downloading += t.s();
      }
      std::vector<fs::path> bases;
      for (const auto& dir : dirs) {
        bases.push_back(DerpJob::levelDir(dir, level) / J.rigDst[d].id);
      }
      batch[parity].add(pool, [=] {
        for (const auto& base : bases) {
          write_pfm(base / (frameName + ".pfm"), disp, w, h);
          if (png) {
            write_disparity_png(base / (frameName + ".png"), disp, w, h);
          }
          if (exr) {
            write_exr_f32(base / (frameName + ".exr"), disp, w, h);
          }
        }
      }, 1);
    }
  }
  // DerpSequence: a frame's level leaves through one of a few per-frame buffers — ONE download for all D planes (a
  // copy call costs ~0.5 ms whatever its size, and there are 16 planes x 8 frames x 10 levels), D write jobs that
  // share the buffer, and the buffer is free again when the last of them is done. The buffers are plain heap memory
  // unless DERP_PIN_DOWNLOADS is set: page-locking what the two finest levels of an 8-frame chunk need (2.6 GB) costs
  // 0.4 s and stalls every other HIP call of the process meanwhile, whichever thread does it, while the copy into
  // pageable memory fits the slack the host thread has behind each frame's compute.
  struct FrameBuf {
    Arena mem;
    int writers = 0;  // write jobs still reading it; 0 = free
  };
  std::vector<FrameBuf> ring;
  std::mutex ringMu;
  std::condition_variable ringCv;
  IoBatch ringBatch;
  bool pinRing = getenv("DERP_PIN_DOWNLOADS") != nullptr;
  void reserve_ring(int slots) { ring.resize(slots); }
  int ring_acquire(size_t bytes, int writers) {
    Timer t;
    int got = -1;
    {
      std::unique_lock<std::mutex> lk(ringMu);
      ringCv.wait(lk, [&] {
        for (size_t i = 0; i < ring.size(); ++i) {
          if (ring[i].writers == 0) {
            got = (int)i;
            return true;
          }
        }
        return false;
      });
      ring[got].writers = writers;
    }
    ringBatch.raise_if_failed();  // a write job of an earlier frame failed: stop here, not after the whole level
    waited += t.s();
    Arena& A = ring[got].mem;
    if (A.bytes < bytes) {
      if (pinRing) {
        A.ensure(bytes);
      } else {
        A.release();
        A.p = malloc(bytes);
        CHECK_MSG(A.p != nullptr, "out of host memory");
        A.pinned = false;
        A.bytes = bytes;
      }
    }
    return got;
  }
  void ring_release(int i) {
    bool freed;
    {
      std::lock_guard<std::mutex> lk(ringMu);
      freed = --ring[i].writers == 0;
    }
    if (freed) {
      ringCv.notify_one();
    }
  }
  // a sequence frame's level (resident slot or out-of-core host store) instead of the selected frame's.
  // fromScratch: the frame was filtered ahead of the level's Transfer (derp_seq_level_filter_frame); its filtered
  // level comes from the filter's scratch over the copy stream while the following frames compute
  void save_seq(derp_seq* seq, int frame, int level, const std::string& frameName, const std::vector<fs::path>& dirs,
                bool pngToo, bool fromScratch = false) {
    derp_ctx* ctx = J.ctx;
    const int w = J.W[level], h = J.H[level];
    const bool png = J.savePng && pngToo, exr = J.saveExr && pngToo;
    const size_t n = J.npx(level);
    const int slot = ring_acquire(n * sizeof(float) * J.D, J.D);
    float* all = static_cast<float*>(ring[slot].mem.p);
    {
      Timer t;
      if (fromScratch) {
        DERP_OK(ctx, derp_seq_download_filtered(seq, frame, level, -1, all));
      } else {
        DERP_OK(ctx, derp_seq_download_disparity(seq, frame, level, -1, all));
      }
      downloading += t.s();
    }
    for (int d = 0; d < J.D; ++d) {
      const float* disp = all + n * d;
      std::vector<fs::path> bases;
      for (const auto& dir : dirs) {
        bases.push_back(DerpJob::levelDir(dir, level) / J.rigDst[d].id);
      }
      ringBatch.add(pool, [=] {
        // the slot goes back on EVERY exit path: a write that fails (disk full, unwritable output) throws, the batch
        // keeps the message, and ring_acquire / finish() on the host thread turn it into the fatal exit — a slot
        // that stayed taken would park the host thread in ring_acquire for ever instead
        struct Release {
          LevelWriter* w;
          int slot;
          ~Release() { w->ring_release(slot); }
        } release{this, slot};
        for (const auto& base : bases) {
          write_pfm(base / (frameName + ".pfm"), disp, w, h);
          if (png) {
            write_disparity_png(base / (frameName + ".png"), disp, w, h);
          }
          if (exr) {
            write_exr_f32(base / (frameName + ".exr"), disp, w, h);
          }
        }
      }, 1);
    }
  }
  void finish() {
    Timer t;
    batch[0].wait();
    batch[1].wait();
    ringBatch.wait();
    waited += t.s();
  }
};

// debug images of the level processed last (PyramidLevel.h:418-485), written inline
inline void save_debug_images(const DerpJob& J, int level, const std::string& frameName) {
  derp_ctx* ctx = J.ctx;
  const size_t n = J.npx(level);
  for (int d = 0; d < J.D; ++d) {
    std::vector<float> cost(n), conf(n);
    DERP_OK(ctx, derp_download_cost(ctx, d, cost.data(), conf.data()));
    for (auto& v : cost) {
      v *= 255.0f / 100.0f / 65535.0f * 257.0f;  // kScaleCostPlot, 8-bit range in a 16-bit file
    }
    write_disparity_png(DerpJob::levelDir(fs::path(J.outputRoot) / "cost", level) / J.rigDst[d].id / (frameName + ".png"),
                        cost.data(), J.W[level], J.H[level]);
    for (auto& v : conf) {
      v *= 255.0f * 100.0f / 65535.0f * 257.0f;  // kScaleConfidencePlot
    }
    write_disparity_png(
        DerpJob::levelDir(fs::path(J.outputRoot) / "confidence", level) / J.rigDst[d].id / (frameName + ".png"),
        conf.data(), J.W[level], J.H[level]);
    std::vector<uint8_t> mm(n);
    DERP_OK(ctx, derp_download_mismatch_mask(ctx, d, mm.data()));
    std::vector<uint16_t> mpx(n);
    for (size_t i = 0; i < n; ++i) {
      mpx[i] = mm[i] ? 255 : 0;
    }
    write_png(DerpJob::levelDir(fs::path(J.outputRoot) / "mismatches", level) / J.rigDst[d].id / (frameName + ".png"),
              mpx.data(), J.W[level], J.H[level], 1, 8);
  }
}

}  // namespace cli
