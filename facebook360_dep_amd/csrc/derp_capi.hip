// C-ABI implementation of include/derp_hip.h: context, HBM-resident pyramid, level driver.
// Host side mirrors the reference's DerpCLI level loop (DerpCLI.cpp:220-323) and processLevel
// (Derp.cpp:1005-1034). No CPU compute path: every stage is a kernel in derp_kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/derp_hip.h"
#include "derp_kernels.h"

using namespace derp;

namespace {

enum Stage {
  ST_FOV = 0,
  ST_VARIANCE,
  ST_OWN_BIAS,
  ST_UPSAMPLE,
  ST_PROJ_WARP,
  ST_REPROJECT,
  ST_PROJ_BIAS,
  ST_BRUTE,
  ST_RANDOM,
  ST_PINGPONG,
  ST_MISMATCH,
  ST_BILATERAL,
  ST_MEDIAN,
  ST_MASKFOV,
  ST_TEMPORAL,
  ST_LANES,  // wall of a level whose frames ran on overlapping work lanes (their per-stage spans overlap in time)
  ST_COUNT
};
const char* kStageNames[ST_COUNT] = {"fov_mask",  "variance",    "own_bias",         "upsample",  "proj_warp",
                                     "reproject", "proj_bias",   "brute_force",      "random_proposals",
                                     "ping_pong", "mismatches",  "bilateral",        "median",    "mask_fov",
                                     "temporal",  "lanes_wall"};
constexpr int kMaxLevels = 24;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t n) {
    if (n <= bytes) {
      return 0;
    }
    if (p) {
      (void)hipFree(p);
      p = nullptr;
      bytes = 0;
    }
    if (hipMalloc(&p, n) != hipSuccess) {
      p = nullptr;
      return 1;
    }
    bytes = n;
    return 0;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
    }
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct TimedSpan {
  int stage, level;
  hipEvent_t a, b;
};

struct LanczosTab {
  DevBuf ofs, coef;
};

struct AreaTabDev {  // computeResizeAreaTab of one axis, resident in HBM
  DevBuf start, si, alpha;
  int iscale = 0;
};

}  // namespace

struct derp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copyStream = nullptr;  // input uploads of a frame that is not being computed (sequence driver), with
  DevBuf copyStaging;                // their own staging buffer: they overlap the compute of the frame before
  DevBuf cnVert, cnRgba, cnZ, cnAcc, cnOut, cnBig, cnNBig;  // derp_canopy_cubemap's buffers, kept between calls
  std::string err;
  derp_options opt;
  int S = 0, D = 0;
  std::vector<Cam> camsSrcH, camsDstH;
  std::vector<int> dst2srcH;
  DevBuf camsSrc, camsDst, dst2src;

  int numLevels = 0, widthFull = 0, heightFull = 0;
  std::vector<int> LW, LH;
  // HBM-resident pyramid of the SELECTED frame slot; the other slots' pyramids are parked in `parked`
  // (derp_set_frame_slots / derp_select_frame: several frames of one sequence resident on this GPU)
  std::vector<DevBuf> pyrColor, pyrFg, pyrBg, pyrDisp;
  std::vector<char> haveBg, haveDisp;
  struct FrameSlot {
    std::vector<DevBuf> pyrColor, pyrFg, pyrBg, pyrDisp;
    std::vector<char> haveBg, haveDisp;
  };
  std::vector<FrameSlot> parked;  // parked[curSlot] is empty while that slot is selected
  int curSlot = 0;
  int xcdRotate = 1;

  // working level
  int cur = -1;
  int DB = 0;  // dst batch that fits the table budget
  DevBuf srcVar, ownBias, fovMask, maskAnd, disparity, cost, confidence, dispRes, costRes, changed, tmpF, rank, mismatchMask, pairCount;
  DevBuf temporalCarry;  // accumulators of a temporal window longer than one launch holds
  DevBuf tileSeen;       // k_reproject_bias: per (table, tile) whether any map position is valid
  int colorTablesCleanLevel = -1;  // level whose colour / bias tables were written in full since its warps were built
  DevBuf projColorT;  // projColor again in 4x4-texel tiles: the random-proposal kernel's copy (DERP_RANDOM_TILED)
  DevBuf projWarp, projColor, projBias, projWarpInv, bruteCost, bruteConf, lanczosTmp, staging, stagingB;
  DevBuf rayDir, behind;  // per destination pixel: ray direction [3][D][n] f64, sources facing away [D][n] (k_pixel_rays)
  int warpCachedLevel = -1;
  bool randomRanThisLevel = false;  // cost / confidence hold random-proposal results for this level
  bool tablesValid = false;
  DevBuf counters;  // [ST_COUNT][kMaxLevels][4] u64
  std::map<std::pair<int, int>, LanczosTab*> lanczos;
  std::map<std::pair<int, int>, AreaTabDev*> areaTabs;
  DevBuf fullFrame;
  DevBuf devMask;  // derp_dev_mask result (not a working buffer)
  DevBuf rephotoColor, rephotoDisp;  // derp_rephotograph_upload: S planes of BGR u16 / f32 disparity
  int rephotoW = 0, rephotoH = 0;
  DevBuf spiral;
  int spiralN = 0, spiralRadius = -1;

  // Work lanes (round 6): a second, third ... working set + stream for processLevel of ANOTHER frame of a sequence at the
  // same coarse level (derp_seq_level_compute). The frames of a level are independent, and at the coarse levels one
  // frame's kernels fill a fraction of the chip (level 6 of the 16-camera rig: 784 waves for 4096 wave slots) and are
  // bound by their own serial latency — on lanes the frames' kernels overlap. A lane holds everything processLevel writes
  // per frame; the rig-only tables of the level (projWarp, projWarpInv, rayDir, behind, resampling tables) stay shared.
  struct WorkLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    DevBuf srcVar, ownBias, fovMask, maskAnd, disparity, cost, confidence, dispRes, costRes, changed, tmpF, rank, mismatchMask,
        pairCount, tileSeen, projColor, projBias, projColorT, bruteCost, bruteConf, lanczosTmp, staging, stagingB;
    int colorTablesCleanLevel = -1;
  };
  std::vector<WorkLane*> lanes;
  hipEvent_t laneReady = nullptr;  // recorded on the main stream behind what the lanes' frames depend on
  int activeLane = -1;             // the lane whose members are swapped in (-1: the context's own)

  bool profiling = false;
  bool noMemo = false;  // DERP_NO_MEMO (developer switch), read once in derp_create
  // waves per SIMD of the random-proposal / ping-pong kernels (0 = what their registers and LDS allow: four up to 16
  // cameras): a launch can ask for fewer by reserving more LDS per (one-wave) block — DERP_RANDOM_WAVES / DERP_PP_WAVES,
  // developer A/B switches
  int randomWaves = 0, ppWaves = 0;
  size_t ldsPerCu = 160 * 1024;  // hipDeviceProp.maxSharedMemoryPerMultiProcessor (derp_create)
  bool noTemporalTile = false;  // DERP_NO_TEMPORAL_TILE (developer A/B: the direct form of the temporal filter)
  bool noBlankSkip = false;     // DERP_NO_BLANK_SKIP (developer A/B: every frame rewrites the blank tiles of the colour tables)
  std::vector<TimedSpan> spans;
  double accMs[ST_COUNT][kMaxLevels];
  int accLaunch[ST_COUNT][kMaxLevels];
};

namespace {

int fail(derp_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) {
    c->err = buf;
  }
  return 1;
}

#define HIPCHK(c, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      return fail(c, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
    }                                                                                     \
  } while (0)
#define ALLOC(c, buf, n)                                                         \
  do {                                                                           \
    if ((buf).ensure(n)) {                                                       \
      return fail(c, "out of device memory allocating %zu bytes (%s)", (size_t)(n), #buf); \
    }                                                                            \
  } while (0)
#define KCHECK(c) HIPCHK(c, hipGetLastError())
#define TRY(expr)       \
  do {                  \
    int r_ = (expr);    \
    if (r_) {           \
      return r_;        \
    }                   \
  } while (0)

struct Span {
  derp_ctx* c;
  int stage, level;
  hipEvent_t a = nullptr, b = nullptr;
  Span(derp_ctx* ctx, int st, int lv) : c(ctx), stage(st), level(lv) {
    if (c->profiling) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, c->stream);
    }
  }
  ~Span() {
    if (c->profiling && a) {
      (void)hipEventRecord(b, c->stream);
      c->spans.push_back({stage, level, a, b});
    }
  }
};

void drain_spans(derp_ctx* c) {
  for (auto& s : c->spans) {
    (void)hipEventSynchronize(s.b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.a, s.b);
    if (s.level >= 0 && s.level < kMaxLevels) {
      c->accMs[s.stage][s.level] += ms;
      c->accLaunch[s.stage][s.level] += 1;
    }
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  c->spans.clear();
}

size_t npx(const derp_ctx* c, int level) {
  return (size_t)c->LW[level] * c->LH[level];
}

unsigned long long* counter_slot(derp_ctx* c, int stage, int level) {
  return c->counters.as<unsigned long long>() + ((size_t)stage * kMaxLevels + level) * 4;
}

LevelView make_view(derp_ctx* c, int stage, int dst0, int nd) {
  LevelView V;
  const int L = c->cur;
  V.W = c->LW[L];
  V.H = c->LH[L];
  V.Wd = (double)V.W;
  V.Hd = (double)V.H;
  V.S = c->S;
  V.D = nd;
  V.level = L;
  V.numLevels = c->numLevels;
  V.dst0 = dst0;
  V.hasFg = c->opt.use_foreground_masks;
  // PyramidLevel.h:232-236 — scale = float(width) / heightFullSize (reference quirk kept)
  const float scale = float(V.W) / c->heightFull;
  const float scaleVar = scale * scale;
  V.varNoiseFloor = std::max(c->opt.var_noise_floor * scaleVar, kMinVar);
  V.varHighThresh = c->opt.var_high_thresh;
  V.minDepthM = c->opt.min_depth_m;
  V.maxDepthM = c->opt.max_depth_m;
  V.randomProposals = c->opt.random_proposals;
  V.partialCoverage = c->opt.partial_coverage;
  V.xcdRotate = c->xcdRotate;
  V.camsSrc = c->camsSrc.as<Cam>();
  V.camsDst = c->camsDst.as<Cam>();
  V.dst2src = c->dst2src.as<int>();
  V.srcColor = c->pyrColor[L].as<ushort4>();
  V.ownBias = c->ownBias.as<ushort4>();
  V.srcVar = c->srcVar.as<float>();
  V.srcFg = c->pyrFg[L].as<uint8_t>();
  V.rayDir = c->rayDir.as<double>();
  V.behind = c->behind.as<unsigned>();
  V.rayStride = (size_t)c->D * V.W * V.H;
  V.projWarp = c->projWarp.as<float2>();
  V.projColor = c->projColor.as<ushort4>();
  V.projBias = c->projBias.as<ushort4>();
  V.projColorT = c->projColorT.as<ushort4>();
  V.disparity = c->disparity.as<float>();
  V.cost = c->cost.as<float>();
  V.confidence = c->confidence.as<float>();
  V.bgDisp = c->pyrBg[L].as<float>();
  V.fovMask = c->fovMask.as<uint8_t>();
  V.pairCount = c->pairCount.as<uint8_t>();
  V.counters = counter_slot(c, stage, L);
  return V;
}

dim3 grid2d(int w, int h, int z, dim3 b) {
  return dim3((w + b.x - 1) / b.x, (h + b.y - 1) / b.y, z);
}
const dim3 kBlk2d(32, 8, 1);

// k_blur3_u16: 64 x 4 threads, each a column strip of kBlurRows rows
const dim3 kBlurBlk(64, 4, 1);
dim3 blur_grid(int ow, int oh, int planes) {
  return dim3((ow + 63) / 64, (oh + 4 * kBlurRows - 1) / (4 * kBlurRows), planes);
}

int flat_grid(size_t n) {
  return (int)std::min<size_t>((n + 255) / 256, 2048 * 4);
}

// ---- Lanczos4 tables: resize.cpp interpolateLanczos4 + offset computation (fp64 libm on host) ----
void lanczos_coeffs(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < 1.1920928955078125e-07f) {
    for (int i = 0; i < 8; i++) {
      coeffs[i] = 0;
    }
    coeffs[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * M_PI * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
  for (int i = 0; i < 8; i++) {
    const double y = -(x + 3 - i) * M_PI * 0.25;
    coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += coeffs[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; i++) {
    coeffs[i] *= sum;
  }
}

int get_lanczos(derp_ctx* c, int ssize, int dsize, LanczosTab** out) {
  auto key = std::make_pair(ssize, dsize);
  auto it = c->lanczos.find(key);
  if (it != c->lanczos.end()) {
    *out = it->second;
    return 0;
  }
  std::vector<int> ofs(dsize);
  std::vector<float> coef((size_t)dsize * 8);
  const double scale = (double)ssize / dsize;
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)f;
    s -= (s > f);  // cvFloor
    f -= s;
    ofs[d] = s;
    lanczos_coeffs(f, &coef[(size_t)d * 8]);
  }
  LanczosTab* t = new LanczosTab;
  ALLOC(c, t->ofs, ofs.size() * sizeof(int));
  ALLOC(c, t->coef, coef.size() * sizeof(float));
  HIPCHK(c, hipMemcpyAsync(t->ofs.p, ofs.data(), ofs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(t->coef.p, coef.data(), coef.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host vectors go out of scope
  c->lanczos[key] = t;
  *out = t;
  return 0;
}

// cv::resize INTER_AREA, one axis: resize.cpp computeResizeAreaTab (fractional scales) or the integer
// scale factor of the "area fast" paths (|scale - round(scale)| < DBL_EPSILON)
int get_area_tab(derp_ctx* c, int ssize, int dsize, AreaTabDev** out, bool forceTable = false) {
  // cv::resize takes the integer-factor paths only when BOTH axes have integer factors; otherwise both
  // axes go through computeResizeAreaTab — forceTable builds the table of an integer-factor axis
  auto key = std::make_pair(forceTable ? -ssize : ssize, dsize);
  auto it = c->areaTabs.find(key);
  if (it != c->areaTabs.end()) {
    *out = it->second;
    return 0;
  }
  AreaTabDev* t = new AreaTabDev;
  const double scale = (double)ssize / dsize;
  const int iscale = (int)std::nearbyint(scale);
  std::vector<int> start(dsize + 1, 0), si;
  std::vector<float> alpha;
  if (!forceTable && iscale >= 1 && std::abs(scale - iscale) < 2.220446049250313e-16) {
    t->iscale = iscale;
  } else {
    for (int dx = 0; dx < dsize; ++dx) {
      start[dx] = (int)si.size();
      const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
      const double cellWidth = std::min(scale, ssize - fsx1);
      int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
      sx2 = std::min(sx2, ssize - 1);
      sx1 = std::min(sx1, sx2);
      if (sx1 - fsx1 > 1e-3) {
        si.push_back(sx1 - 1);
        alpha.push_back((float)((sx1 - fsx1) / cellWidth));
      }
      for (int sx = sx1; sx < sx2; ++sx) {
        si.push_back(sx);
        alpha.push_back(float(1.0 / cellWidth));
      }
      if (fsx2 - sx2 > 1e-3) {
        si.push_back(sx2);
        alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth));
      }
    }
    start[dsize] = (int)si.size();
  }
  if (si.empty()) {
    si.push_back(0);
    alpha.push_back(0.f);
  }
  ALLOC(c, t->start, start.size() * sizeof(int));
  ALLOC(c, t->si, si.size() * sizeof(int));
  ALLOC(c, t->alpha, alpha.size() * sizeof(float));
  HIPCHK(c, hipMemcpy(t->start.p, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(t->si.p, si.data(), si.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(t->alpha.p, alpha.data(), alpha.size() * sizeof(float), hipMemcpyHostToDevice));
  c->areaTabs[key] = t;
  *out = t;
  return 0;
}

// cv2.resize(src, (dw, dh), INTER_AREA): kind 0 BGR u16 -> BGRX, 1 u8 (-> {0,1} when threshold >= 0), 2 f32
int resize_area_dev(derp_ctx* c, int kind, const void* src, int sw, int sh, void* dst, int dw, int dh, int threshold) {
  if (dw > sw || dh > sh) {
    // enlarging along an axis: cv::resize(INTER_AREA) turns into its bilinear emulation (float images only here)
    if (kind < 2) {
      return fail(c, "pyramid levels must not be larger than the full-size frame (%dx%d -> %dx%d)", sw, sh, dw, dh);
    }
    const dim3 g = grid2d(dw, dh, 1, kBlk2d);
    if (kind == 3) {
      hipLaunchKernelGGL(k_resize_linear_area_f32<3>, g, kBlk2d, 0, c->stream, (const float*)src, sw, sh, (float*)dst, dw, dh);
    } else {
      hipLaunchKernelGGL(k_resize_linear_area_f32<1>, g, kBlk2d, 0, c->stream, (const float*)src, sw, sh, (float*)dst, dw, dh);
    }
    KCHECK(c);
    return 0;
  }
  AreaTabDev *tx, *ty;
  TRY(get_area_tab(c, sw, dw, &tx));
  TRY(get_area_tab(c, sh, dh, &ty));
  if ((tx->iscale > 0) != (ty->iscale > 0)) {
    if (tx->iscale > 0) {
      TRY(get_area_tab(c, sw, dw, &tx, true));
    } else {
      TRY(get_area_tab(c, sh, dh, &ty, true));
    }
  }
  AreaAxis ax{tx->start.as<int>(), tx->si.as<int>(), tx->alpha.as<float>(), tx->iscale};
  AreaAxis ay{ty->start.as<int>(), ty->si.as<int>(), ty->alpha.as<float>(), ty->iscale};
  const dim3 g = grid2d(dw, dh, 1, kBlk2d);
  if (kind == 0) {
    hipLaunchKernelGGL(k_resize_area<0>, g, kBlk2d, 0, c->stream, src, sw, sh, dst, dw, dh, ax, ay, threshold);
  } else if (kind == 1) {
    hipLaunchKernelGGL(k_resize_area<1>, g, kBlk2d, 0, c->stream, src, sw, sh, dst, dw, dh, ax, ay, threshold);
  } else if (kind == 3) {
    hipLaunchKernelGGL(k_resize_area<3>, g, kBlk2d, 0, c->stream, src, sw, sh, dst, dw, dh, ax, ay, threshold);
  } else {
    hipLaunchKernelGGL(k_resize_area<2>, g, kBlk2d, 0, c->stream, src, sw, sh, dst, dw, dh, ax, ay, threshold);
  }
  KCHECK(c);
  return 0;
}

// UpsampleDisparityLib.cpp:27-54 — clockwise outward spiral of diameter w
std::vector<int2> make_spiral(int w) {
  int x = 0, y = 0, dx = 0, dy = -1, t = w;
  const int samples = t * t;
  std::vector<int2> locs;
  for (int i = 0; i < samples; ++i) {
    const bool vx = (-w / 2 <= x) && (x <= w / 2), vy = (-w / 2 <= y) && (y <= w / 2);
    if (vx && vy) {
      locs.push_back(make_int2(x, y));
    }
    if (x == y || ((x < 0) && (x == -y)) || ((x > 0) && (x == 1 - y))) {
      t = dx;
      dx = -dy;
      dy = t;
    }
    x += dx;
    y += dy;
  }
  return locs;
}

int ensure_spiral(derp_ctx* c, int radius) {
  if (c->spiralRadius == radius) {
    return 0;
  }
  const std::vector<int2> s = make_spiral(radius * 2 + 1);
  ALLOC(c, c->spiral, s.size() * sizeof(int2));
  HIPCHK(c, hipMemcpy(c->spiral.p, s.data(), s.size() * sizeof(int2), hipMemcpyHostToDevice));
  c->spiralN = (int)s.size();
  c->spiralRadius = radius;
  return 0;
}

// upsampleDisparityInPlace for `nd` planes living on the device (Lanczos path) or one plane (mask path)
int upsample_lanczos_dev(derp_ctx* c, const float* in, int sw, int sh, float* out, int dw, int dh, int planes,
                         size_t inStride, size_t outStride) {
  if (sw == dw && sh == dh) {  // cv::resize to the same size is a copy (after NaN -> 1e-4)
    return fail(c, "upsample to identical size not supported");
  }
  LanczosTab *tx, *ty;
  TRY(get_lanczos(c, sw, dw, &tx));
  TRY(get_lanczos(c, sh, dh, &ty));
  const size_t tmpStride = (size_t)dw * sh;
  ALLOC(c, c->lanczosTmp, tmpStride * planes * sizeof(float));
  hipLaunchKernelGGL(k_lanczos_h, grid2d(dw, sh, planes, kBlk2d), kBlk2d, 0, c->stream, in, sw, sh, dw,
                     tx->ofs.as<int>(), tx->coef.as<float>(), c->lanczosTmp.as<float>(), inStride, tmpStride);
  KCHECK(c);
  hipLaunchKernelGGL(k_lanczos_v, grid2d(dw, dh, planes, kBlk2d), kBlk2d, 0, c->stream, c->lanczosTmp.as<float>(), sh,
                     dw, dh, ty->ofs.as<int>(), ty->coef.as<float>(), out, tmpStride, outStride);
  KCHECK(c);
  return 0;
}

// `planes` images (destination cameras) in one launch: every pointer addresses `planes` contiguous planes
int upsample_masked_dev(derp_ctx* c, const float* in, const uint8_t* mask, int sw, int sh, const uint8_t* maskUp,
                        const float* bgUp, float* out, int dw, int dh, int planes = 1) {
  // getRadius (UpsampleDisparityLib.cpp:93-96): int(scale*scale + 1), float arithmetic
  const float scale = float(dw) / float(sw);
  const int radius = (int)(scale * scale + 1);
  TRY(ensure_spiral(c, radius));
  ALLOC(c, c->lanczosTmp, (size_t)dw * dh * planes * sizeof(float));
  hipLaunchKernelGGL(k_upsample_nearest_masked, grid2d(dw, dh, planes, kBlk2d), kBlk2d, 0, c->stream, in, mask, sw, sh,
                     maskUp, dw, dh, c->lanczosTmp.as<float>());
  KCHECK(c);
  hipLaunchKernelGGL(k_spiral_fill, grid2d(dw, dh, planes, kBlk2d), kBlk2d, 0, c->stream, c->lanczosTmp.as<float>(), bgUp,
                     maskUp, dw, dh, c->spiral.as<int2>(), c->spiralN, out);
  KCHECK(c);
  return 0;
}

// Stored inverse warps (projWarpInv) are needed only for sources that are not the own source of a destination of the
// same batch: everywhere else projWarpInv(d, s) is projWarp(ds, own(d)) (derp_kernels.h, batch_dst_of_source).
bool batch_needs_inverse_warps(const derp_ctx* c, int dst0, int nd) {
  if (DERP_NO_WARP_IDENTITY) {
    return true;
  }
  std::vector<char> covered(c->S, 0);
  for (int d = dst0; d < dst0 + nd; ++d) {
    covered[c->dst2srcH[d]] = 1;
  }
  for (int s = 0; s < c->S; ++s) {
    if (!covered[s]) {
      return true;
    }
  }
  return false;
}

size_t table_bytes_per_dst(const derp_ctx* c, int W, int H, bool withInverse) {
  const size_t wp = (size_t)(W + 2 * kPadW) * (H + 2 * kPadW), cp = (size_t)(W + 2 * kPadC) * (H + 2 * kPadC);
  return (size_t)(c->S - 1) * (wp * sizeof(float2) + 2 * cp * sizeof(ushort4) + (withInverse ? (size_t)W * H * sizeof(float2) : 0) +
                               (DERP_RANDOM_TILED ? tiled_plane(W, H) * sizeof(ushort4) : 0));
}

int compute_fov_and_masks(derp_ctx* c, int level) {
  const int W = c->LW[level], H = c->LH[level];
  const size_t n = (size_t)W * H;
  {
    Span sp(c, ST_FOV, level);
    hipLaunchKernelGGL(k_fov_mask, grid2d(W, H, c->D, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>(), W, H,
                       c->fovMask.as<uint8_t>());
    KCHECK(c);
    hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(n), c->D), dim3(256), 0, c->stream, c->fovMask.as<uint8_t>(),
                       c->pyrFg[level].as<uint8_t>(), c->dst2src.as<int>(), 0, n, c->maskAnd.as<uint8_t>());
    KCHECK(c);
  }
  return 0;
}

int build_warp(derp_ctx* c, int dst0, int nd) {
  const int L = c->cur;
  Span sp(c, ST_PROJ_WARP, L);
  {
    const size_t n = (size_t)c->LW[L] * c->LH[L];
    ALLOC(c, c->rayDir, 3 * n * c->D * sizeof(double));
    ALLOC(c, c->behind, n * c->D * sizeof(unsigned));
  }
  LevelView V = make_view(c, ST_PROJ_WARP, dst0, nd);
  hipLaunchKernelGGL(k_proj_warp, grid2d(V.W + 2 * kPadW, V.H + 2 * kPadW, c->S, kBlk2d), kBlk2d, 0, c->stream, V,
                     c->projWarp.as<float2>());
  KCHECK(c);
  // the destination pixels' ray directions and behind-the-camera source masks: rig + level size only, like the warps
  hipLaunchKernelGGL(k_pixel_rays, grid2d(V.W, V.H, nd, kBlk2d), kBlk2d, 0, c->stream, V, c->rayDir.as<double>(),
                     c->behind.as<unsigned>());
  KCHECK(c);
  c->colorTablesCleanLevel = -1;  // new warps (another level, batch or rig state): the colour tables must be rewritten in full
  // ... and the inverse warps reprojectColors reads (projWarpInv, PyramidLevel.h:46-51) — those that are not a
  // projWarp table of this batch already (all of them are when every source is a destination of the batch)
  if (batch_needs_inverse_warps(c, dst0, nd)) {
    hipLaunchKernelGGL(k_proj_warp_inv, grid2d(V.W, V.H, nd, kBlk2d), kBlk2d, 0, c->stream, V, c->projWarpInv.as<float2>());
    KCHECK(c);
  }
  return 0;
}

int build_color_tables(derp_ctx* c, int dst0, int nd) {
  const int L = c->cur;
  LevelView V = make_view(c, ST_REPROJECT, dst0, nd);
  // colours and their 3x3 biases in one pass (the bias stage's time is inside ST_REPROJECT now)
  Span sp(c, ST_REPROJECT, L);
  const dim3 grid((V.W + kRbTile - 1) / kRbTile, (V.H + kRbTile - 1) / kRbTile, nd * (c->S - 1));
  ALLOC(c, c->tileSeen, (size_t)grid.x * grid.y * grid.z);
  // a frame that finds the tables of this level as an earlier frame left them (same warps: a sequence running the
  // level frame after frame) skips the tiles no source pixel maps into — they still hold their zeros
  const int skipBlank = c->colorTablesCleanLevel == L && nd == c->D && !c->noBlankSkip;
  hipLaunchKernelGGL(k_reproject_bias, grid, dim3(256), 0, c->stream, V, c->projWarpInv.as<float2>(),
                     c->projColor.as<ushort4>(), c->projBias.as<ushort4>(), c->projColorT.as<ushort4>(),
                     c->tileSeen.as<uint8_t>(), skipBlank);
  KCHECK(c);
  c->colorTablesCleanLevel = nd == c->D ? L : -1;
  return 0;
}

// number of cost-kernel blocks covering a W x H image (16x16 super-tiles of four 8x8 wave tiles)
int tiles_of(int W, int H, int& tilesX) {
  constexpr int B = DERP_TILE_BLOCK > 1 ? DERP_TILE_BLOCK : 1;  // tile grid padded to whole B x B squares
  tilesX = ((W + 15) / 16 + B - 1) / B * B;
  const int tilesY = ((H + 15) / 16 + B - 1) / B * B;
  return tilesX * tilesY * (256 / DERP_COST_BLOCK);
}
constexpr size_t kCostLdsPerSrc = (size_t)DERP_COST_BLOCK * sizeof(SsdPair);
int round8(int n) {
  return (n + 7) / 8 * 8;
}
// dynamic LDS of a one-wave block such that at most `waves` blocks per SIMD (4 x waves per CU) fit a CU's LDS
// (`ldsPerCu`: hipDeviceProp.maxSharedMemoryPerMultiProcessor, 160 KB on gfx950); `fixed` = the kernel's static LDS
size_t lds_for_waves(size_t needed, int waves, size_t ldsPerCu, size_t fixed) {
  if (waves <= 0 || waves >= 4) {
    return needed;
  }
  const size_t perBlock = ldsPerCu / (size_t)(4 * waves + 1) + 256;  // 4 * waves blocks fit, 4 * waves + 1 do not
  return std::max(needed, perBlock > fixed ? perBlock - fixed : needed);
}
constexpr size_t kCostLdsStatic = sizeof(PatchWin) * (DERP_COST_BLOCK / 64) + kAtanLutDoubles * sizeof(double);
// Which register budget of the two cost kernels to launch (k_ping_pong / k_random_proposals vs their _w3 twins): four waves
// per SIMD need sixteen one-wave blocks per CU, i.e. LDS for sixteen — true up to 16 cameras (9.3 KB each of 160 KB; measured
// 3.9 resident waves), not beyond (24 cameras: 13.3 KB, twelve blocks). DERP_COST_WAVES=3 / 4 forces one (developer A/B).
bool cost_four_waves(const derp_ctx* c) {
  if (const char* e = getenv("DERP_COST_WAVES")) {
    return atoi(e) >= 4;
  }
  return 16 * (kCostLdsPerSrc * (size_t)c->S + kCostLdsStatic) <= c->ldsPerCu && c->S <= 16;
}

int run_brute_force(derp_ctx* c, int dst0, int nd) {
  const int L = c->cur;
  if (L != c->numLevels - 1) {
    return 0;
  }
  Span sp(c, ST_BRUTE, L);
  LevelView V = make_view(c, ST_BRUTE, dst0, nd);
  const size_t n = (size_t)V.W * V.H;
  ALLOC(c, c->bruteCost, (size_t)nd * kNumDepths * n * sizeof(float));
  ALLOC(c, c->bruteConf, (size_t)nd * kNumDepths * n * sizeof(float));
  // 8 x 8 pixel strips over the interior (W - 2) x (H - 2) pixels, one wave each
  const int tilesX = std::max(1, (V.W - 2 + 7) / 8), tilesY = std::max(1, (V.H - 2 + 7) / 8);
  const int tiles = tilesX * tilesY;
  const size_t lds = kCostLdsPerSrc * (size_t)(c->S);
  hipLaunchKernelGGL(k_brute_costs, dim3(tiles, kNumDepths, nd), dim3(DERP_COST_BLOCK), lds, c->stream, V,
                     c->bruteCost.as<float>(), c->bruteConf.as<float>(), tilesX, tiles);
  KCHECK(c);
  hipLaunchKernelGGL(k_brute_select, grid2d(V.W, V.H, nd, kBlk2d), kBlk2d, 0, c->stream, V, c->bruteCost.as<float>(),
                     c->bruteConf.as<float>());
  KCHECK(c);
  hipLaunchKernelGGL(k_brute_margin, grid2d(V.W, V.H, nd, kBlk2d), kBlk2d, 0, c->stream, V);
  KCHECK(c);
  return 0;
}

int run_random_proposals(derp_ctx* c, int dst0, int nd) {
  const int L = c->cur;
  if (c->opt.random_proposals <= 0 || L == c->numLevels - 1) {
    return 0;
  }
  Span sp(c, ST_RANDOM, L);
  c->randomRanThisLevel = true;
  LevelView V = make_view(c, ST_RANDOM, dst0, nd);
  if (V.H > 2 && V.W > 2) {
    hipLaunchKernelGGL(k_row_rank, dim3(V.H - 2, nd), dim3(256), 0, c->stream, V, c->rank.as<int>());
    KCHECK(c);
  }
  int tilesX;
  const int tiles = tiles_of(V.W, V.H, tilesX);
  const size_t lds = lds_for_waves(kCostLdsPerSrc * (size_t)(c->S), c->randomWaves, c->ldsPerCu, kCostLdsStatic);
  hipLaunchKernelGGL(cost_four_waves(c) ? k_random_proposals : k_random_proposals_w3, dim3(round8(tiles), nd),
                     dim3(DERP_COST_BLOCK), lds, c->stream, V, c->rank.as<int>(), tilesX, tiles);
  KCHECK(c);
  return 0;
}

int run_ping_pong(derp_ctx* c, int dst0, int nd) {
  const int L = c->cur;
  if (L == c->numLevels - 1) {
    return 0;
  }
  Span sp(c, ST_PINGPONG, L);
  LevelView V = make_view(c, ST_PINGPONG, dst0, nd);
  const size_t n = (size_t)V.W * V.H;
  HIPCHK(c, hipMemsetAsync(c->changed.as<uint8_t>() + (size_t)dst0 * n, 1, n * nd, c->stream));
  int tilesX;
  const int tiles = tiles_of(V.W, V.H, tilesX);
  const size_t lds = lds_for_waves(kCostLdsPerSrc * (size_t)(c->S), c->ppWaves, c->ldsPerCu, kCostLdsStatic);
  for (int it = 1; it <= c->opt.ping_pong_iterations; ++it) {
    hipLaunchKernelGGL(cost_four_waves(c) ? k_ping_pong : k_ping_pong_w3, dim3(round8(tiles), nd), dim3(DERP_COST_BLOCK), lds,
                       c->stream, V, c->changed.as<uint8_t>(), c->dispRes.as<float>(), c->costRes.as<float>(), tilesX,
                       (int)(it == 1 && c->randomRanThisLevel && !c->noMemo));
    KCHECK(c);
    hipLaunchKernelGGL(k_ping_pong_commit, dim3(flat_grid(n * nd)), dim3(256), 0, c->stream,
                       c->disparity.as<float>() + (size_t)dst0 * n, c->cost.as<float>() + (size_t)dst0 * n,
                       c->dispRes.as<float>() + (size_t)dst0 * n, c->costRes.as<float>() + (size_t)dst0 * n,
                       c->changed.as<uint8_t>() + (size_t)dst0 * n, n * nd);
    KCHECK(c);
  }
  // cost / confidence now belong to ping-pong's result (+inf where every candidate was rejected): the
  // memoised candidate must not be served from them by a later derp_stage_ping_pong call
  if (dst0 + nd >= c->D) {
    c->randomRanThisLevel = false;
  }
  return 0;
}

// handleDisparityMismatches (Derp.cpp:722-748): after every destination finished ping-pong
int run_mismatches(derp_ctx* c) {
  const int L = c->cur;
  if (L > c->opt.mismatches_start_level || L == c->numLevels - 1) {
    return 0;
  }
  if (c->D != c->S) {
    return fail(c, "Check failed: rigDst.size() == rigSrc.size()  Mismatches only valid when considering all cameras");
  }
  for (int d = 0; d < c->D; ++d) {
    if (c->dst2srcH[d] != d) {
      return fail(c, "mismatch handling needs destinations in rig order (dst %d maps to src %d)", d, c->dst2srcH[d]);
    }
  }
  Span sp(c, ST_MISMATCH, L);
  c->randomRanThisLevel = false;  // the working disparity changes: cost[] no longer matches it
  LevelView V = make_view(c, ST_MISMATCH, 0, c->D);
  const size_t n = (size_t)V.W * V.H;
  hipLaunchKernelGGL(k_mismatch, dim3((V.W + 15) / 16, (V.H + 15) / 16, c->D), dim3(256), 256 * sizeof(float) * c->S,
                     c->stream, V, c->dispRes.as<float>(), c->mismatchMask.as<uint8_t>());
  KCHECK(c);
  HIPCHK(c, hipMemcpyAsync(c->disparity.p, c->dispRes.p, n * c->D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int bilateral_radius(int level) {  // Derp.cpp:876-878
  const float scale = std::pow(0.9f, level);
  return (int)std::max(std::ceil(5 * scale), float(1));
}

size_t bilateral_lds_bytes(int radius) {
  // rows of the tile are padded to a multiple of 16 texels (k_joint_bilateral: bank-conflict-free float4 reads)
  const size_t t = (size_t)((16 + 2 * radius + 15) & ~15) * (16 + 2 * radius);
  return t * 4 * sizeof(float) + ((t + 3) & ~(size_t)3);
}

int run_bilateral(derp_ctx* c) {
  const int L = c->cur;
  Span sp(c, ST_BILATERAL, L);
  c->randomRanThisLevel = false;  // the working disparity changes: cost[] no longer matches it
  const int W = c->LW[L], H = c->LH[L];
  const size_t n = (size_t)W * H;
  // weights passed (B, G, R) = (0.5, 1, 1) — Derp.cpp:893-896, Derp.h:44-48; sigma 0.005
  const int radius = bilateral_radius(L);
  hipLaunchKernelGGL(k_joint_bilateral<true>, dim3((W + 15) / 16, (H + 15) / 16, c->D), dim3(256),
                     bilateral_lds_bytes(radius), c->stream, c->disparity.as<float>(),
                     (const void*)c->pyrColor[L].as<ushort4>(), c->maskAnd.as<uint8_t>(), W, H, radius, 0.005f, 0.5f,
                     1.0f, 1.0f, c->tmpF.as<float>(), n, n, c->dst2src.as<int>());
  KCHECK(c);
  HIPCHK(c, hipMemcpyAsync(c->disparity.p, c->tmpF.p, n * c->D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int run_median(derp_ctx* c, bool fuseMaskFov) {
  const int L = c->cur;
  Span sp(c, ST_MEDIAN, L);
  c->randomRanThisLevel = false;  // the working disparity changes: cost[] no longer matches it
  const int W = c->LW[L], H = c->LH[L];
  const size_t n = (size_t)W * H;
  hipLaunchKernelGGL(k_masked_median, grid2d(W, H, c->D, kBlk2d), kBlk2d, 0, c->stream, c->disparity.as<float>(),
                     c->opt.use_foreground_masks ? c->pyrBg[L].as<float>() : (const float*)nullptr,
                     c->maskAnd.as<uint8_t>(), W, H, 1, c->tmpF.as<float>(), n,
                     fuseMaskFov ? c->fovMask.as<uint8_t>() : (const uint8_t*)nullptr);
  KCHECK(c);
  HIPCHK(c, hipMemcpyAsync(c->disparity.p, c->tmpF.p, n * c->D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int run_mask_fov(derp_ctx* c) {
  const int L = c->cur;
  Span sp(c, ST_MASKFOV, L);
  const size_t n = npx(c, L) * c->D;
  hipLaunchKernelGGL(k_mask_fov, dim3(flat_grid(n)), dim3(256), 0, c->stream, c->disparity.as<float>(),
                     c->fovMask.as<uint8_t>(), n);
  KCHECK(c);
  return 0;
}

int check_level(derp_ctx* c, int level) {
  if (!c || c->numLevels == 0) {
    return fail(c, "derp_set_pyramid has not been called");
  }
  if (level < 0 || level >= c->numLevels) {
    return fail(c, "level %d out of range [0, %d)", level, c->numLevels);
  }
  if (c->LW[level] <= 0 || c->LH[level] <= 0) {
    return fail(c, "level %d was declared absent in derp_set_pyramid", level);
  }
  return 0;
}

// Level set-up: what DerpCLI does before processLevel (DerpCLI.cpp:221-303)
int level_begin(derp_ctx* c, int level, bool buildAllTables) {
  TRY(check_level(c, level));
  if (c->S - 1 > kMaxSrc) {
    return fail(c, "too many source cameras (%d > %d)", c->S, kMaxSrc + 1);
  }
  c->cur = level;
  const int W = c->LW[level], H = c->LH[level];
  const size_t n = (size_t)W * H;
  if (W < 3 || H < 3) {
    return fail(c, "level %d is too small (%dx%d)", level, W, H);
  }
  TRY(compute_fov_and_masks(c, level));
  {
    Span sp(c, ST_VARIANCE, level);
    hipLaunchKernelGGL(k_variance, grid2d(W, H, c->S, kBlk2d), kBlk2d, 0, c->stream, c->pyrColor[level].as<ushort4>(),
                       W, H, c->srcVar.as<float>());
    KCHECK(c);
  }
  {
    Span sp(c, ST_OWN_BIAS, level);
    hipLaunchKernelGGL(k_blur3_u16, blur_grid(W, H, c->S), kBlurBlk, 0, c->stream,
                       c->pyrColor[level].as<ushort4>(), 0, c->ownBias.as<ushort4>(), 0, W, H, n, n);
    KCHECK(c);
  }
  // fresh PyramidLevel: disparity / cost / confidence start at 0 (PyramidLevel.h:209-221)
  HIPCHK(c, hipMemsetAsync(c->cost.p, 0, n * c->D * sizeof(float), c->stream));
  HIPCHK(c, hipMemsetAsync(c->confidence.p, 0, n * c->D * sizeof(float), c->stream));
  HIPCHK(c, hipMemsetAsync(c->mismatchMask.p, 0, n * c->D, c->stream));
  if (level < c->numLevels - 1 && !c->haveDisp[level + 1] && !buildAllTables) {
    return fail(c, "Missing disparity of level %d needed to start level %d", level + 1, level);
  }
  if (level < c->numLevels - 1 && c->haveDisp[level + 1]) {
    Span sp(c, ST_UPSAMPLE, level);
    const int sw = c->LW[level + 1], sh = c->LH[level + 1];
    if (!c->opt.use_foreground_masks) {
      TRY(upsample_lanczos_dev(c, c->pyrDisp[level + 1].as<float>(), sw, sh, c->disparity.as<float>(), W, H, c->D,
                               (size_t)sw * sh, n));
    } else {
      // masks = fov & fg at both sizes (UpsampleDisparityLib.cpp:163-179); coarse fov&fg recomputed into tmpF bytes
      ALLOC(c, c->staging, (size_t)sw * sh * c->D);
      hipLaunchKernelGGL(k_fov_mask, grid2d(sw, sh, c->D, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>(), sw, sh,
                         c->staging.as<uint8_t>());
      KCHECK(c);
      hipLaunchKernelGGL(k_and_masks, dim3(flat_grid((size_t)sw * sh), c->D), dim3(256), 0, c->stream,
                         c->staging.as<uint8_t>(), c->pyrFg[level + 1].as<uint8_t>(), c->dst2src.as<int>(), 0,
                         (size_t)sw * sh, c->staging.as<uint8_t>());
      KCHECK(c);
      TRY(upsample_masked_dev(c, c->pyrDisp[level + 1].as<float>(), c->staging.as<uint8_t>(), sw, sh,
                              c->maskAnd.as<uint8_t>(), c->pyrBg[level].as<float>(), c->disparity.as<float>(), W, H, c->D));
    }
  } else {
    HIPCHK(c, hipMemsetAsync(c->disparity.p, 0, n * c->D * sizeof(float), c->stream));
  }
  // table budget -> dst batch. When the buffers already hold every destination's tables of this level (the
  // steady state of a sequence: same levels frame after frame) there is nothing to ask the runtime.
  // (the inverse-warp table only exists when a batch holds a source that is not one of its destinations)
  const bool invAll = batch_needs_inverse_warps(c, 0, c->D);
  size_t per = table_bytes_per_dst(c, W, H, invAll);
  int DB = c->D;
  bool needInv = invAll;
  {
    const size_t wpAll = (size_t)(W + 2 * kPadW) * (H + 2 * kPadW) * (c->S - 1) * c->D * sizeof(float2);
    const size_t cpAll = (size_t)(W + 2 * kPadC) * (H + 2 * kPadC) * (c->S - 1) * c->D * sizeof(ushort4);
    const size_t ipAll = (size_t)W * H * (c->S - 1) * c->D * sizeof(float2);
    const bool resident = c->projWarp.bytes >= wpAll && c->projColor.bytes >= cpAll && c->projBias.bytes >= cpAll &&
        (!invAll || c->projWarpInv.bytes >= ipAll) && !getenv("DERP_TABLE_BUDGET_GB") &&
        (!DERP_RANDOM_TILED || c->projColorT.bytes >= tiled_plane(W, H) * (c->S - 1) * c->D * sizeof(ushort4));
    if (!resident) {
      size_t freeB = 0, totalB = 0;
      HIPCHK(c, hipMemGetInfo(&freeB, &totalB));
      size_t budget = freeB + c->projWarp.bytes + c->projColor.bytes + c->projBias.bytes + c->projWarpInv.bytes + c->projColorT.bytes;
      if (const char* e = getenv("DERP_TABLE_BUDGET_GB")) {
        budget = std::min<size_t>(budget, (size_t)(atof(e) * (1ull << 30)));
      } else {
        budget = (size_t)(budget * 0.85);
      }
      DB = (int)std::min<size_t>((size_t)c->D, budget / std::max<size_t>(per, 1));
      if (DB < c->D && !needInv) {  // batches: the sources outside a batch need stored inverse warps
        needInv = true;
        per = table_bytes_per_dst(c, W, H, true);
        DB = (int)std::min<size_t>((size_t)c->D, budget / std::max<size_t>(per, 1));
      }
      if (DB < 1) {
        return fail(c, "projection tables for one destination (%zu bytes) exceed the table budget (%zu bytes)", per, budget);
      }
      // equal-sized batches: 24 destinations under a 23-destination budget run as 12 + 12, not 23 + 1
      const int batches = (c->D + DB - 1) / DB;
      DB = (c->D + batches - 1) / batches;
    }
  }
  c->DB = DB;
  const size_t wp = (size_t)(W + 2 * kPadW) * (H + 2 * kPadW), cp = (size_t)(W + 2 * kPadC) * (H + 2 * kPadC);
  ALLOC(c, c->projWarp, (size_t)DB * (c->S - 1) * wp * sizeof(float2));
  ALLOC(c, c->projColor, (size_t)DB * (c->S - 1) * cp * sizeof(ushort4));
  ALLOC(c, c->projBias, (size_t)DB * (c->S - 1) * cp * sizeof(ushort4));
  if (needInv) {
    ALLOC(c, c->projWarpInv, (size_t)DB * (c->S - 1) * n * sizeof(float2));
  }
  if (DERP_RANDOM_TILED) {
    ALLOC(c, c->projColorT, (size_t)DB * (c->S - 1) * tiled_plane(W, H) * sizeof(ushort4));
  }
  c->tablesValid = false;
  c->randomRanThisLevel = false;
  if (buildAllTables) {
    if (DB < c->D) {
      return fail(c, "stage-level API needs all destinations' tables resident (batch %d < %d)", DB, c->D);
    }
    if (c->opt.rebuild_warp_tables || c->warpCachedLevel != level) {
      TRY(build_warp(c, 0, c->D));
      c->warpCachedLevel = level;
    }
  }
  return 0;
}

int level_end(derp_ctx* c) {
  const int L = c->cur;
  HIPCHK(c, hipMemcpyAsync(c->pyrDisp[L].p, c->disparity.p, npx(c, L) * c->D * sizeof(float),
                           hipMemcpyDeviceToDevice, c->stream));
  c->haveDisp[L] = 1;
  return 0;
}

int process_level(derp_ctx* c, int level) {
  TRY(level_begin(c, level, false));
  const int L = level;
  for (int d0 = 0; d0 < c->D; d0 += c->DB) {
    const int nd = std::min(c->DB, c->D - d0);
    const bool single = (c->DB == c->D);
    if (!single || c->opt.rebuild_warp_tables || c->warpCachedLevel != L) {
      TRY(build_warp(c, d0, nd));
      c->warpCachedLevel = single ? L : -1;
    }
    TRY(build_color_tables(c, d0, nd));
    TRY(run_brute_force(c, d0, nd));
    TRY(run_random_proposals(c, d0, nd));
    TRY(run_ping_pong(c, d0, nd));
  }
  TRY(run_mismatches(c));
  if (c->opt.do_bilateral_filter) {
    TRY(run_bilateral(c));
  }
  if (c->opt.do_median_filter) {
    TRY(run_median(c, true));
  } else {
    TRY(run_mask_fov(c));
  }
  TRY(level_end(c));
  return 0;
}

int need_current(derp_ctx* c, bool tables) {
  if (!c || c->cur < 0) {
    return fail(c, "derp_level_begin has not been called");
  }
  if (tables && !c->tablesValid) {
    return fail(c, "derp_stage_reproject_colors must run before this stage");
  }
  return 0;
}

template <typename T>
int upload_tmp(derp_ctx* c, DevBuf& buf, const T* host, size_t count) {
  ALLOC(c, buf, count * sizeof(T));
  HIPCHK(c, hipMemcpyAsync(buf.p, host, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return 0;
}


// HBM-resident pyramid of one frame (colour, fg masks, background disparity, result per level)
int alloc_pyramid(derp_ctx* c, std::vector<DevBuf>& color, std::vector<DevBuf>& fg, std::vector<DevBuf>& bg,
                  std::vector<DevBuf>& disp, std::vector<char>& haveBg, std::vector<char>& haveDisp) {
  const int num_levels = c->numLevels;
  color.resize(num_levels);
  fg.resize(num_levels);
  bg.resize(num_levels);
  disp.resize(num_levels);
  haveBg.assign(num_levels, 0);
  haveDisp.assign(num_levels, 0);
  for (int l = 0; l < num_levels; ++l) {
    const size_t n = npx(c, l);
    if (n == 0) {
      continue;  // level not present / not needed by this run
    }
    ALLOC(c, color[l], n * c->S * sizeof(ushort4));
    ALLOC(c, fg[l], n * c->S);
    ALLOC(c, bg[l], n * c->D * sizeof(float));
    ALLOC(c, disp[l], n * c->D * sizeof(float));
    HIPCHK(c, hipMemsetAsync(fg[l].p, 1, n * c->S, c->stream));  // generateAllPassMasks
    HIPCHK(c, hipMemsetAsync(bg[l].p, 0, n * c->D * sizeof(float), c->stream));
  }
  return 0;
}

// make `slot` the frame the pyramid members refer to
int select_frame(derp_ctx* c, int slot) {
  if (slot < 0 || slot >= (int)c->parked.size()) {
    return fail(c, "frame slot %d out of range [0, %d)", slot, (int)c->parked.size());
  }
  if (slot == c->curSlot) {
    return 0;
  }
  auto swap_with = [&](derp_ctx::FrameSlot& fs) {
    std::swap(fs.pyrColor, c->pyrColor);
    std::swap(fs.pyrFg, c->pyrFg);
    std::swap(fs.pyrBg, c->pyrBg);
    std::swap(fs.pyrDisp, c->pyrDisp);
    std::swap(fs.haveBg, c->haveBg);
    std::swap(fs.haveDisp, c->haveDisp);
  };
  swap_with(c->parked[c->curSlot]);  // park the active frame
  swap_with(c->parked[slot]);        // activate the requested one
  c->curSlot = slot;
  c->cur = -1;  // working buffers belong to the previously selected frame
  return 0;
}

// exchange the context's per-frame working set (and stream) with lane `i`'s: applied twice it is the identity
void lane_swap(derp_ctx* c, int i) {
  derp_ctx::WorkLane& w = *c->lanes[i];
  std::swap(c->stream, w.stream);
  std::swap(c->colorTablesCleanLevel, w.colorTablesCleanLevel);
  DevBuf* mine[] = {&c->srcVar, &c->ownBias, &c->fovMask, &c->maskAnd, &c->disparity, &c->cost, &c->confidence, &c->dispRes,
                    &c->costRes, &c->changed, &c->tmpF, &c->rank, &c->mismatchMask, &c->pairCount, &c->tileSeen, &c->projColor,
                    &c->projBias, &c->projColorT, &c->bruteCost, &c->bruteConf, &c->lanczosTmp, &c->staging, &c->stagingB};
  DevBuf* theirs[] = {&w.srcVar, &w.ownBias, &w.fovMask, &w.maskAnd, &w.disparity, &w.cost, &w.confidence, &w.dispRes,
                      &w.costRes, &w.changed, &w.tmpF, &w.rank, &w.mismatchMask, &w.pairCount, &w.tileSeen, &w.projColor,
                      &w.projBias, &w.projColorT, &w.bruteCost, &w.bruteConf, &w.lanczosTmp, &w.staging, &w.stagingB};
  for (size_t k = 0; k < sizeof(mine) / sizeof(mine[0]); ++k) {
    std::swap(*mine[k], *theirs[k]);
  }
}

// lanes 0 .. count - 1 exist and hold working buffers for a level of `n` pixels
int lanes_prepare(derp_ctx* c, int count, size_t n) {
  while ((int)c->lanes.size() < count) {
    auto* w = new derp_ctx::WorkLane();
    c->lanes.push_back(w);
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess) {
      return fail(c, "work lane: hipStreamCreate / hipEventCreate failed");
    }
  }
  if (!c->laneReady) {
    HIPCHK(c, hipEventCreateWithFlags(&c->laneReady, hipEventDisableTiming));
  }
  for (int i = 0; i < count; ++i) {
    derp_ctx::WorkLane& w = *c->lanes[i];
    ALLOC(c, w.srcVar, n * c->S * sizeof(float));
    ALLOC(c, w.ownBias, n * c->S * sizeof(ushort4));
    ALLOC(c, w.fovMask, n * c->D);
    ALLOC(c, w.maskAnd, n * c->D);
    for (DevBuf* b : {&w.disparity, &w.cost, &w.confidence, &w.dispRes, &w.costRes, &w.tmpF, &w.rank}) {
      ALLOC(c, *b, n * c->D * sizeof(float));
    }
    ALLOC(c, w.changed, n * c->D);
    ALLOC(c, w.mismatchMask, n * c->D);
    ALLOC(c, w.pairCount, n * c->D);
  }
  return 0;
}

// processLevel of the frame in slot `slot` on lane `i` (i < 0: on the context's own working set and stream). The lane's
// stream first waits for c->laneReady.
int process_level_on_lane(derp_ctx* c, int i, int slot, int level) {
  if (i < 0) {
    TRY(select_frame(c, slot));
    return process_level(c, level);
  }
  HIPCHK(c, hipStreamWaitEvent(c->lanes[i]->stream, c->laneReady, 0));
  lane_swap(c, i);
  c->activeLane = i;
  int rc = select_frame(c, slot);
  if (!rc) {
    rc = process_level(c, level);
  }
  if (!rc && hipEventRecord(c->lanes[i]->done, c->stream) != hipSuccess) {
    rc = fail(c, "work lane: hipEventRecord failed");
  }
  lane_swap(c, i);
  c->activeLane = -1;
  c->cur = -1;  // the context's own working buffers do not hold that frame's level
  return rc;
}

// temporalJointBilateralFilter (TemporalBilateralFilter.h:126-215) of `planes` planes over a window of n frames,
// frame `centre` being the one filtered: one launch per kMaxTemporalFrames frames, the sums carried between them
int temporal_launch(derp_ctx* c, const void* const* guides, const float* const* images, const uint8_t* const* masks, int n,
                    int centre, int W, int H, int planes, float sigma, int radius, float w0, float w1, float w2,
                    float* out, const int* dst2src) {
  if (n < 1 || centre < 0 || centre >= n) {
    return fail(c, "temporal window must hold at least the centre frame");
  }
  if (n > kMaxTemporalFrames) {
    ALLOC(c, c->temporalCarry, (size_t)planes * W * H * sizeof(float2));
  }
  for (int t0 = 0; t0 < n; t0 += kMaxTemporalFrames) {
    TemporalFrames F;
    F.n = std::min(kMaxTemporalFrames, n - t0);
    for (int t = 0; t < F.n; ++t) {
      F.guides[t] = reinterpret_cast<const ushort4*>(guides[t0 + t]);
      F.images[t] = images[t0 + t];
      F.masks[t] = masks[t0 + t];
    }
    F.refGuide = reinterpret_cast<const ushort4*>(guides[centre]);
    F.refImage = images[centre];
    F.refMask = masks[centre];
    F.carry = n > kMaxTemporalFrames ? c->temporalCarry.as<float2>() : nullptr;
    F.first = t0 == 0;
    F.last = t0 + F.n >= n;
    // taps staged through LDS unless the halo makes the tile too big (two buffers of (32 + 2r) x (8 + 2r) x 9 bytes)
    const size_t lds = 2 * ((((size_t)(32 + 2 * radius) * (8 + 2 * radius) * 9) + 15) & ~(size_t)15);
    if (radius >= 0 && lds <= 48 * 1024 && !c->noTemporalTile) {
      hipLaunchKernelGGL(k_temporal_tiled, grid2d(W, H, planes, kBlk2d), kBlk2d, lds, c->stream, F, W, H, sigma, radius, w0,
                         w1, w2, out, dst2src);
    } else {
      hipLaunchKernelGGL(k_temporal, grid2d(W, H, planes, kBlk2d), kBlk2d, 0, c->stream, F, W, H, sigma, radius, w0, w1, w2,
                         out, dst2src);
    }
    KCHECK(c);
  }
  return 0;
}

}  // namespace

// =========================================================================================
extern "C" {

void derp_options_default(derp_options* o) {
  o->min_depth_m = 0.5f;
  o->max_depth_m = 1e4f;
  o->var_noise_floor = 4e-5f;
  o->var_high_thresh = 1e-3f;
  o->random_proposals = 2;
  o->ping_pong_iterations = 1;
  o->mismatches_start_level = -1;
  o->do_bilateral_filter = 1;
  o->do_median_filter = 1;
  o->use_foreground_masks = 0;
  o->partial_coverage = 0;
  o->rebuild_warp_tables = 1;
}

static thread_local std::string g_create_error;

int derp_create(derp_ctx** out, int device, const derp_camera_desc* src, int n_src, const derp_camera_desc* dst,
                int n_dst) {
  if (!out) {
    return 1;
  }
  *out = nullptr;
  derp_ctx* c = new derp_ctx;
  derp_options_default(&c->opt);
  memset(c->accMs, 0, sizeof c->accMs);
  memset(c->accLaunch, 0, sizeof c->accLaunch);
  auto bail = [&](const std::string& m) {
    g_create_error = m;
    for (DevBuf* b : {&c->camsSrc, &c->camsDst, &c->dst2src, &c->counters}) {
      b->release();
    }
    if (c->stream) {
      (void)hipStreamDestroy(c->stream);
    }
    if (c->copyStream) {
      (void)hipStreamDestroy(c->copyStream);
    }
    delete c;
    return 1;
  };
  if (n_src <= 0 || n_dst <= 0) {
    return bail("no source / destination cameras!");
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    return bail("no HIP device present: the depth path has no CPU fallback");
  }
  if (device < 0 || device >= count) {
    return bail("HIP device index out of range");
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    return bail("hipGetDeviceProperties failed");
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("DERP_ALLOW_ANY_ARCH")) {
    return bail(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  }
  if (hipSetDevice(device) != hipSuccess) {
    return bail("hipSetDevice failed");
  }
  c->device = device;
  if (prop.maxSharedMemoryPerMultiProcessor > 0) {
    c->ldsPerCu = prop.maxSharedMemoryPerMultiProcessor;
  }
  if (const char* e = getenv("DERP_XCD_ROTATE")) {
    c->xcdRotate = atoi(e);
  }
  c->noMemo = getenv("DERP_NO_MEMO") != nullptr;
  if (const char* e = getenv("DERP_RANDOM_WAVES")) {
    c->randomWaves = atoi(e);
  }
  if (const char* e = getenv("DERP_PP_WAVES")) {
    c->ppWaves = atoi(e);
  }
  c->noTemporalTile = getenv("DERP_NO_TEMPORAL_TILE") != nullptr;
  c->noBlankSkip = getenv("DERP_NO_BLANK_SKIP") != nullptr;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking) != hipSuccess) {
    return bail("hipStreamCreate failed");
  }
  c->S = n_src;
  c->D = n_dst;
  c->parked.assign(1, derp_ctx::FrameSlot());
  c->camsSrcH.resize(n_src);
  c->camsDstH.resize(n_dst);
  for (int i = 0; i < n_src; ++i) {
    if (const char* m = host_prepare_camera(src[i], c->camsSrcH[i])) {
      return bail(std::string("camera ") + src[i].id + ": " + m);
    }
  }
  c->dst2srcH.assign(n_dst, 0);
  for (int i = 0; i < n_dst; ++i) {
    if (const char* m = host_prepare_camera(dst[i], c->camsDstH[i])) {
      return bail(std::string("camera ") + dst[i].id + ": " + m);
    }
    bool found = false;
    for (int s = 0; s < n_src; ++s) {  // mapSrcToDstIndexes, DerpUtil.cpp:75-89
      if (strncmp(dst[i].id, src[s].id, sizeof dst[i].id) == 0) {
        c->dst2srcH[i] = s;
        found = true;
        break;
      }
    }
    if (!found) {
      return bail(std::string("destination camera ") + dst[i].id + " is not a source camera");
    }
    // The reference's destinations ARE rig cameras (filterDestinations, Derp.cpp:42-70, keeps a subset of the rig), and
    // k_reproject_bias relies on it: projWarpInv(d, s) is read from projWarp(ds, own) when s is destination ds. A
    // descriptor that shares an id with a source but not its intrinsics / pose would silently warp with the wrong camera.
    if (memcmp(&c->camsDstH[i], &c->camsSrcH[c->dst2srcH[i]], sizeof(Cam)) != 0) {
      return bail(std::string("destination camera ") + dst[i].id + " differs from the source camera of the same id "
                  "(destinations must be cameras of the source rig, as filterDestinations makes them)");
    }
  }
  if (c->camsSrc.ensure(sizeof(Cam) * n_src) || c->camsDst.ensure(sizeof(Cam) * n_dst) ||
      c->dst2src.ensure(sizeof(int) * n_dst) || c->counters.ensure(sizeof(unsigned long long) * ST_COUNT * kMaxLevels * 4)) {
    return bail("out of device memory");
  }
  (void)hipMemcpy(c->camsSrc.p, c->camsSrcH.data(), sizeof(Cam) * n_src, hipMemcpyHostToDevice);
  (void)hipMemcpy(c->camsDst.p, c->camsDstH.data(), sizeof(Cam) * n_dst, hipMemcpyHostToDevice);
  (void)hipMemcpy(c->dst2src.p, c->dst2srcH.data(), sizeof(int) * n_dst, hipMemcpyHostToDevice);
  (void)hipMemset(c->counters.p, 0, c->counters.bytes);
  *out = c;
  return 0;
}

void derp_destroy(derp_ctx* c) {
  if (!c) {
    return;
  }
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  drain_spans(c);
  for (auto* v : {&c->pyrColor, &c->pyrFg, &c->pyrBg, &c->pyrDisp}) {
    for (auto& b : *v) {
      b.release();
    }
  }
  for (auto& fs : c->parked) {
    for (auto* v : {&fs.pyrColor, &fs.pyrFg, &fs.pyrBg, &fs.pyrDisp}) {
      for (auto& b : *v) {
        b.release();
      }
    }
  }
  c->devMask.release();
  for (DevBuf* b : {&c->camsSrc, &c->camsDst, &c->dst2src, &c->srcVar, &c->ownBias, &c->fovMask, &c->maskAnd,
                    &c->disparity, &c->cost, &c->confidence, &c->dispRes, &c->costRes, &c->changed, &c->tmpF, &c->rank, &c->mismatchMask, &c->pairCount,
                    &c->projWarp, &c->projColor, &c->projBias, &c->projColorT, &c->projWarpInv, &c->temporalCarry, &c->tileSeen, &c->rayDir, &c->behind, &c->bruteCost, &c->bruteConf, &c->lanczosTmp,
                    &c->staging, &c->stagingB, &c->counters, &c->spiral}) {
    b->release();
  }
  for (auto& kv : c->lanczos) {
    kv.second->ofs.release();
    kv.second->coef.release();
    delete kv.second;
  }
  for (auto& kv : c->areaTabs) {
    kv.second->start.release();
    kv.second->si.release();
    kv.second->alpha.release();
    delete kv.second;
  }
  c->fullFrame.release();
  c->rephotoColor.release();
  c->rephotoDisp.release();
  c->copyStaging.release();
  for (DevBuf* b : {&c->cnVert, &c->cnRgba, &c->cnZ, &c->cnAcc, &c->cnOut, &c->cnBig, &c->cnNBig}) {
    b->release();
  }
  for (derp_ctx::WorkLane* w : c->lanes) {
    (void)hipStreamSynchronize(w->stream);
    for (DevBuf* b : {&w->srcVar, &w->ownBias, &w->fovMask, &w->maskAnd, &w->disparity, &w->cost, &w->confidence, &w->dispRes,
                      &w->costRes, &w->changed, &w->tmpF, &w->rank, &w->mismatchMask, &w->pairCount, &w->tileSeen, &w->projColor,
                      &w->projBias, &w->projColorT, &w->bruteCost, &w->bruteConf, &w->lanczosTmp, &w->staging, &w->stagingB}) {
      b->release();
    }
    (void)hipStreamDestroy(w->stream);
    (void)hipEventDestroy(w->done);
    delete w;
  }
  if (c->laneReady) {
    (void)hipEventDestroy(c->laneReady);
  }
  (void)hipStreamDestroy(c->stream);
  (void)hipStreamDestroy(c->copyStream);
  delete c;
}

const char* derp_last_error(const derp_ctx* c) {
  return c ? c->err.c_str() : g_create_error.c_str();
}

int derp_set_options(derp_ctx* c, const derp_options* o) {
  if (!c || !o) {
    return 1;
  }
  if (o->random_proposals < 0) {
    return fail(c, "Check failed: random_proposals >= 0");
  }
  c->opt = *o;
  return 0;
}

int derp_set_pyramid(derp_ctx* c, int num_levels, const int* widths, const int* heights, int width_full,
                     int height_full) {
  if (!c) {
    return 1;
  }
  if (num_levels <= 0 || num_levels > kMaxLevels) {
    return fail(c, "num_levels %d out of range (1..%d)", num_levels, kMaxLevels);
  }
  HIPCHK(c, hipSetDevice(c->device));
  c->numLevels = num_levels;
  c->widthFull = width_full;
  c->heightFull = height_full;
  c->LW.assign(widths, widths + num_levels);
  c->LH.assign(heights, heights + num_levels);
  // a new geometry drops every frame slot but the selected one (their buffers have the old sizes)
  for (auto& fs : c->parked) {
    for (auto* v : {&fs.pyrColor, &fs.pyrFg, &fs.pyrBg, &fs.pyrDisp}) {
      for (auto& b : *v) {
        b.release();
      }
    }
  }
  c->parked.assign(1, derp_ctx::FrameSlot());
  c->curSlot = 0;
  TRY(alloc_pyramid(c, c->pyrColor, c->pyrFg, c->pyrBg, c->pyrDisp, c->haveBg, c->haveDisp));
  size_t nmax = 0;
  for (int l = 0; l < num_levels; ++l) {
    nmax = std::max(nmax, npx(c, l));
  }
  ALLOC(c, c->srcVar, nmax * c->S * sizeof(float));
  ALLOC(c, c->ownBias, nmax * c->S * sizeof(ushort4));
  ALLOC(c, c->fovMask, nmax * c->D);
  ALLOC(c, c->maskAnd, nmax * c->D);
  for (DevBuf* b : {&c->disparity, &c->cost, &c->confidence, &c->dispRes, &c->costRes, &c->tmpF, &c->rank}) {
    ALLOC(c, *b, nmax * c->D * sizeof(float));
  }
  ALLOC(c, c->changed, nmax * c->D);
  ALLOC(c, c->mismatchMask, nmax * c->D);
  ALLOC(c, c->pairCount, nmax * c->D);
  c->cur = -1;
  c->warpCachedLevel = -1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int derp_set_frame_slots(derp_ctx* c, int n_slots) {
  if (!c || c->numLevels == 0) {
    return fail(c, "derp_set_pyramid has not been called");
  }
  if (n_slots < 1 || n_slots > 4096) {
    return fail(c, "n_slots %d out of range (1..4096)", n_slots);
  }
  HIPCHK(c, hipSetDevice(c->device));
  TRY(select_frame(c, 0));
  for (int k = (int)c->parked.size() - 1; k >= n_slots; --k) {
    for (auto* v : {&c->parked[k].pyrColor, &c->parked[k].pyrFg, &c->parked[k].pyrBg, &c->parked[k].pyrDisp}) {
      for (auto& b : *v) {
        b.release();
      }
    }
  }
  const int had = (int)c->parked.size();
  c->parked.resize(n_slots);
  for (int k = had; k < n_slots; ++k) {
    derp_ctx::FrameSlot& fs = c->parked[k];
    TRY(alloc_pyramid(c, fs.pyrColor, fs.pyrFg, fs.pyrBg, fs.pyrDisp, fs.haveBg, fs.haveDisp));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int derp_select_frame(derp_ctx* c, int slot) {
  if (!c || c->numLevels == 0) {
    return fail(c, "derp_set_pyramid has not been called");
  }
  return select_frame(c, slot);
}

int derp_frame_slots(const derp_ctx* c, int* n_slots, int* selected) {
  if (!c) {
    return 1;
  }
  if (n_slots) {
    *n_slots = (int)c->parked.size();
  }
  if (selected) {
    *selected = c->curSlot;
  }
  return 0;
}

int derp_bind_thread(derp_ctx* c) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}
void* derp_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void derp_host_free(void* p) {
  if (p) {
    (void)hipHostFree(p);
  }
}
int derp_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0 || hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return 1;
  }
  return 0;
}
void derp_host_unregister(void* p) {
  if (p && hipHostUnregister(p) != hipSuccess) {
    (void)hipGetLastError();
  }
}

int derp_upload_color(derp_ctx* c, int level, int s, const uint16_t* bgr) {
  TRY(check_level(c, level));
  if (s < 0 || s >= c->S || !bgr) {
    return fail(c, "bad source index / null image");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  TRY(upload_tmp(c, c->staging, bgr, n * 3));
  hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, c->staging.as<uint16_t>(),
                     c->pyrColor[level].as<ushort4>() + (size_t)s * n, n);
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));  // staging buffer is reused by the next upload
  if (c->warpCachedLevel == level) {
    // colour does not affect the warp tables; nothing to invalidate
  }
  return 0;
}

int derp_upload_foreground_mask(derp_ctx* c, int level, int s, const uint8_t* mask) {
  TRY(check_level(c, level));
  if (s < 0 || s >= c->S || !mask) {
    return fail(c, "bad source index / null mask");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  HIPCHK(c, hipMemcpy(c->pyrFg[level].as<uint8_t>() + (size_t)s * n, mask, n, hipMemcpyHostToDevice));
  return 0;
}

int derp_upload_background_disparity(derp_ctx* c, int level, int d, const float* disp) {
  TRY(check_level(c, level));
  if (d < 0 || d >= c->D || !disp) {
    return fail(c, "bad destination index / null image");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  HIPCHK(c, hipMemcpy(c->pyrBg[level].as<float>() + (size_t)d * n, disp, n * sizeof(float), hipMemcpyHostToDevice));
  c->haveBg[level] = 1;
  return 0;
}

int derp_upload_disparity(derp_ctx* c, int level, int d, const float* disp) {
  TRY(check_level(c, level));
  if (d < 0 || d >= c->D || !disp) {
    return fail(c, "bad destination index / null image");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  HIPCHK(c, hipMemcpy(c->pyrDisp[level].as<float>() + (size_t)d * n, disp, n * sizeof(float), hipMemcpyHostToDevice));
  c->haveDisp[level] = 1;
  return 0;
}

int derp_process_level(derp_ctx* c, int level) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipSetDevice(c->device));
  return process_level(c, level);
}

int derp_process_pyramid(derp_ctx* c, int level_start, int level_end_) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (level_start < level_end_) {
    return fail(c, "Check failed: level_start >= level_end (%d vs %d)", level_start, level_end_);
  }
  for (int level = level_start; level >= level_end_; --level) {
    TRY(process_level(c, level));
  }
  return 0;
}

int derp_synchronize(derp_ctx* c) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // coverage CHECK of computeBruteForceDisparity (Derp.cpp:334-349)
  if (c->numLevels > 0 && !c->opt.partial_coverage && !c->opt.use_foreground_masks) {
    unsigned long long v[4];
    HIPCHK(c, hipMemcpy(v, counter_slot(c, ST_BRUTE, c->numLevels - 1), sizeof v, hipMemcpyDeviceToHost));
    if (v[2] != 0) {
      return fail(c, "Check failed: partialCoverage || useForegroundMasks  Insufficient coverage at %llu pixels", v[2]);
    }
  }
  return 0;
}

int derp_download_disparity(derp_ctx* c, int level, int d, float* disparity) {
  TRY(check_level(c, level));
  if (d < 0 || d >= c->D || !disparity) {
    return fail(c, "bad destination index / null output");
  }
  if (!c->haveDisp[level]) {
    return fail(c, "level %d has not been processed", level);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = npx(c, level);
  HIPCHK(c, hipMemcpy(disparity, c->pyrDisp[level].as<float>() + (size_t)d * n, n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int derp_download_cost(derp_ctx* c, int d, float* cost, float* confidence) {
  TRY(need_current(c, false));
  if (d < 0 || d >= c->D) {
    return fail(c, "bad destination index");
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = npx(c, c->cur);
  if (cost) {
    HIPCHK(c, hipMemcpy(cost, c->cost.as<float>() + (size_t)d * n, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (confidence) {
    HIPCHK(c, hipMemcpy(confidence, c->confidence.as<float>() + (size_t)d * n, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  return 0;
}

// ---- stage-level API ----
int derp_level_begin(derp_ctx* c, int level) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipSetDevice(c->device));
  return level_begin(c, level, true);
}
int derp_stage_reproject_colors(derp_ctx* c) {
  TRY(need_current(c, false));
  TRY(build_color_tables(c, 0, c->D));
  c->tablesValid = true;
  return 0;
}
int derp_stage_brute_force(derp_ctx* c) {
  TRY(need_current(c, true));
  return run_brute_force(c, 0, c->D);
}
int derp_stage_random_proposals(derp_ctx* c) {
  TRY(need_current(c, true));
  return run_random_proposals(c, 0, c->D);
}
int derp_stage_ping_pong(derp_ctx* c) {
  TRY(need_current(c, true));
  return run_ping_pong(c, 0, c->D);
}
int derp_stage_mismatches(derp_ctx* c) {
  TRY(need_current(c, false));
  return run_mismatches(c);
}
int derp_stage_bilateral_filter(derp_ctx* c) {
  TRY(need_current(c, false));
  return run_bilateral(c);
}
int derp_stage_median_filter(derp_ctx* c) {
  TRY(need_current(c, false));
  return run_median(c, false);
}
int derp_stage_mask_fov(derp_ctx* c) {
  TRY(need_current(c, false));
  return run_mask_fov(c);
}
int derp_level_end(derp_ctx* c) {
  TRY(need_current(c, false));
  return level_end(c);
}
int derp_set_level_disparity(derp_ctx* c, int d, const float* disp) {
  TRY(need_current(c, false));
  if (d < 0 || d >= c->D || !disp) {
    return fail(c, "bad destination index / null image");
  }
  const size_t n = npx(c, c->cur);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->disparity.as<float>() + (size_t)d * n, disp, n * sizeof(float), hipMemcpyHostToDevice));
  c->randomRanThisLevel = false;  // cost[] no longer belongs to the working disparity
  return 0;
}
int derp_get_level_disparity(derp_ctx* c, int d, float* disp) {
  TRY(need_current(c, false));
  if (d < 0 || d >= c->D || !disp) {
    return fail(c, "bad destination index / null output");
  }
  const size_t n = npx(c, c->cur);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(disp, c->disparity.as<float>() + (size_t)d * n, n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int derp_cost_map(derp_ctx* c, int d, const float* disp, float* cost, float* confidence) {
  TRY(need_current(c, true));
  if (d < 0 || d >= c->D || !disp || !cost || !confidence) {
    return fail(c, "bad arguments");
  }
  const int L = c->cur;
  const size_t n = npx(c, L);
  TRY(upload_tmp(c, c->staging, disp, n));
  ALLOC(c, c->stagingB, 2 * n * sizeof(float));
  HIPCHK(c, hipMemsetAsync(c->stagingB.p, 0xff, 2 * n * sizeof(float), c->stream));  // NaN fill
  LevelView V = make_view(c, ST_PINGPONG, 0, c->D);
  int tilesX;
  const int tiles = tiles_of(V.W, V.H, tilesX);
  const size_t lds = kCostLdsPerSrc * (size_t)(c->S);
  hipLaunchKernelGGL(k_cost_map, dim3(tiles), dim3(DERP_COST_BLOCK), lds, c->stream, V, d, c->staging.as<float>(),
                     c->stagingB.as<float>(), c->stagingB.as<float>() + n, tilesX);
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(cost, c->stagingB.p, n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(confidence, c->stagingB.as<float>() + n, n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int derp_debug_atan2_ypos(derp_ctx* c, const double* y, const double* x, double* out, size_t n) {
  if (!c || !y || !x || !out) {
    return fail(c, "bad arguments");
  }
  ALLOC(c, c->staging, 3 * n * sizeof(double));
  double* d = c->staging.as<double>();
  HIPCHK(c, hipMemcpyAsync(d, y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + n, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_debug_atan2_ypos, dim3(flat_grid(n)), dim3(256), 0, c->stream, d, d + n, d + 2 * n, n);
  KCHECK(c);
  HIPCHK(c, hipMemcpyAsync(out, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int derp_debug_download(derp_ctx* c, int d, int s, int which, void* out) {
  TRY(need_current(c, false));
  const int L = c->cur;
  const int W = c->LW[L], H = c->LH[L];
  const size_t n = (size_t)W * H;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (which == 4) {
    HIPCHK(c, hipMemcpy(out, c->srcVar.as<float>() + (size_t)s * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  if (which == 5) {
    HIPCHK(c, hipMemcpy(out, c->fovMask.as<uint8_t>() + (size_t)d * n, n, hipMemcpyDeviceToHost));
    return 0;
  }
  if (d < 0 || d >= c->D || s < 0 || s >= c->S || s == c->dst2srcH[d]) {
    return fail(c, "bad (dst, src) pair");
  }
  const size_t tab = (size_t)d * (c->S - 1) + (s < c->dst2srcH[d] ? s : s - 1);
  if (which == 0) {
    const int PW = W + 2 * kPadW, PH = H + 2 * kPadW;
    std::vector<float2> tmp((size_t)PW * PH);
    HIPCHK(c, hipMemcpy(tmp.data(), c->projWarp.as<float2>() + tab * tmp.size(), tmp.size() * sizeof(float2),
                        hipMemcpyDeviceToHost));
    float2* o = reinterpret_cast<float2*>(out);
    for (int y = 0; y < H; ++y) {
      memcpy(o + (size_t)y * W, &tmp[(size_t)(y + kPadW) * PW + kPadW], (size_t)W * sizeof(float2));
    }
    return 0;
  }
  if (which == 2 || which == 3) {
    const int PW = W + 2 * kPadC, PH = H + 2 * kPadC;
    std::vector<ushort4> tmp((size_t)PW * PH);
    const ushort4* base = (which == 2 ? c->projColor.as<ushort4>() : c->projBias.as<ushort4>()) + tab * tmp.size();
    HIPCHK(c, hipMemcpy(tmp.data(), base, tmp.size() * sizeof(ushort4), hipMemcpyDeviceToHost));
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
    for (int y = 0; y < H; ++y) {
      for (int x = 0; x < W; ++x) {
        const ushort4 q = tmp[(size_t)(y + kPadC) * PW + x + kPadC];
        o[((size_t)y * W + x) * 3 + 0] = q.x;
        o[((size_t)y * W + x) * 3 + 1] = q.y;
        o[((size_t)y * W + x) * 3 + 2] = q.z;
      }
    }
    return 0;
  }
  return fail(c, "unknown table id %d", which);
}

// ---- pyramid builder (scripts/render/resize.py:51-85) ----
static int build_pyramid(derp_ctx* c, int kind, int index, int count, const void* host, size_t elem, int w, int h,
                         int threshold) {
  if (!c || c->numLevels == 0) {
    return fail(c, "derp_set_pyramid has not been called");
  }
  if (index < 0 || index >= count || !host || w <= 0 || h <= 0) {
    return fail(c, "bad camera index / null image");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  ALLOC(c, c->fullFrame, n * elem);
  HIPCHK(c, hipMemcpyAsync(c->fullFrame.p, host, n * elem, hipMemcpyHostToDevice, c->stream));
  for (int l = 0; l < c->numLevels; ++l) {
    const size_t nl = npx(c, l);
    if (nl == 0) {
      continue;
    }
    void* dst = kind == 0 ? (void*)(c->pyrColor[l].as<ushort4>() + (size_t)index * nl)
        : kind == 1       ? (void*)(c->pyrFg[l].as<uint8_t>() + (size_t)index * nl)
                          : (void*)(c->pyrBg[l].as<float>() + (size_t)index * nl);
    TRY(resize_area_dev(c, kind, c->fullFrame.p, w, h, dst, c->LW[l], c->LH[l], threshold));
    if (kind == 2) {
      c->haveBg[l] = 1;
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));  // fullFrame is reused by the next call
  return 0;
}
int derp_build_pyramid_color(derp_ctx* c, int src, const uint16_t* bgr, int w, int h) {
  return build_pyramid(c, 0, src, c ? c->S : 0, bgr, 6, w, h, -1);
}
int derp_build_pyramid_foreground_mask(derp_ctx* c, int src, const uint8_t* mask, int w, int h, int threshold) {
  return build_pyramid(c, 1, src, c ? c->S : 0, mask, 1, w, h, threshold);
}
int derp_build_pyramid_background_disparity(derp_ctx* c, int dst, const float* disp, int w, int h) {
  return build_pyramid(c, 2, dst, c ? c->D : 0, disp, 4, w, h, -1);
}
int derp_download_level_color(derp_ctx* c, int level, int src, uint16_t* bgr) {
  TRY(check_level(c, level));
  if (src < 0 || src >= c->S || !bgr) {
    return fail(c, "bad source index / null output");
  }
  const size_t n = npx(c, level);
  ALLOC(c, c->staging, n * 6);
  hipLaunchKernelGGL(k_bgrx_to_bgr, dim3(flat_grid(n)), dim3(256), 0, c->stream,
                     c->pyrColor[level].as<ushort4>() + (size_t)src * n, c->staging.as<uint16_t>(), n);
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(bgr, c->staging.p, n * 6, hipMemcpyDeviceToHost));
  return 0;
}
int derp_download_level_mask(derp_ctx* c, int level, int src, uint8_t* mask) {
  TRY(check_level(c, level));
  if (src < 0 || src >= c->S || !mask) {
    return fail(c, "bad source index / null output");
  }
  const size_t n = npx(c, level);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(mask, c->pyrFg[level].as<uint8_t>() + (size_t)src * n, n, hipMemcpyDeviceToHost));
  return 0;
}
int derp_download_level_background(derp_ctx* c, int level, int dst, float* disp) {
  TRY(check_level(c, level));
  if (dst < 0 || dst >= c->D || !disp) {
    return fail(c, "bad destination index / null output");
  }
  const size_t n = npx(c, level);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(disp, c->pyrBg[level].as<float>() + (size_t)dst * n, n * 4, hipMemcpyDeviceToHost));
  return 0;
}
// one image: kind 0 = BGR u16 x3, 1 = u8, 2 = f32, 3 = BGR f32 x3 (host in / host out)
int derp_resize_area(derp_ctx* c, int kind, const void* src, int w, int h, void* dst, int dw, int dh) {
  if (!c || !src || !dst || kind < 0 || kind > 3 || w <= 0 || h <= 0 || dw <= 0 || dh <= 0) {
    return fail(c, "bad arguments");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t elem = kind == 0 ? 6 : kind == 1 ? 1 : kind == 3 ? 12 : 4, n = (size_t)w * h, nd = (size_t)dw * dh;
  DevBuf in, out, out3;
  int rc = 0;
  if (in.ensure(n * elem) || out.ensure(nd * (kind == 0 ? 8 : elem)) || (kind == 0 && out3.ensure(nd * 6))) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(in.p, src, n * elem, hipMemcpyHostToDevice);
    rc = resize_area_dev(c, kind, in.p, w, h, out.p, dw, dh, -1);
    if (!rc && kind == 0) {
      hipLaunchKernelGGL(k_bgrx_to_bgr, dim3(flat_grid(nd)), dim3(256), 0, c->stream, out.as<ushort4>(),
                         out3.as<uint16_t>(), nd);
    }
    if (!rc && (hipStreamSynchronize(c->stream) != hipSuccess ||
                hipMemcpy(dst, kind == 0 ? out3.p : out.p, nd * elem, hipMemcpyDeviceToHost) != hipSuccess)) {
      rc = fail(c, "HIP error in derp_resize_area: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* b : {&in, &out, &out3}) {
    b->release();
  }
  return rc;
}

// ---- GenerateForegroundMasks (source/render/BackgroundSubtractionUtil.h:20-60) ----
int derp_generate_foreground_mask(derp_ctx* c, const uint16_t* template_bgr, const uint16_t* frame_bgr, int w, int h,
                                  int blur_radius, float threshold, int morph_closing_size, uint8_t* mask01) {
  if (!c || !template_bgr || !frame_bgr || !mask01 || w <= 0 || h <= 0 || blur_radius < 0 || blur_radius > 3 ||
      morph_closing_size < 0 || !(threshold >= 0)) {
    return fail(c, "bad arguments (blur_radius must be 0..3)");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  DevBuf raw, t4, f4, tb, fb, m0, m1;
  int rc = 0;
  if (raw.ensure(n * 6) || t4.ensure(n * 8) || f4.ensure(n * 8) || tb.ensure(n * 8) || fb.ensure(n * 8) || m0.ensure(n) ||
      m1.ensure(n)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(raw.p, template_bgr, n * 6, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, raw.as<uint16_t>(), t4.as<ushort4>(), n);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(raw.p, frame_bgr, n * 6, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, raw.as<uint16_t>(), f4.as<ushort4>(), n);
    const ushort4 *tp = t4.as<ushort4>(), *fp = f4.as<ushort4>();
    if (blur_radius > 0) {
      hipLaunchKernelGGL(k_gauss_u16, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, t4.as<ushort4>(), tb.as<ushort4>(), w, h, blur_radius);
      hipLaunchKernelGGL(k_gauss_u16, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, f4.as<ushort4>(), fb.as<ushort4>(), w, h, blur_radius);
      tp = tb.as<ushort4>();
      fp = fb.as<ushort4>();
    }
    hipLaunchKernelGGL(k_fg_threshold, dim3(flat_grid(n)), dim3(256), 0, c->stream, tp, fp, n, threshold, m0.as<uint8_t>());
    const uint8_t* res = m0.as<uint8_t>();
    if (morph_closing_size > 0) {
      hipLaunchKernelGGL(k_morph_rect, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, m0.as<uint8_t>(), m1.as<uint8_t>(), w, h, morph_closing_size, 1);
      hipLaunchKernelGGL(k_morph_rect, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, m1.as<uint8_t>(), m0.as<uint8_t>(), w, h, morph_closing_size, 0);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(mask01, res, n, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_generate_foreground_mask: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* b : {&raw, &t4, &f4, &tb, &fb, &m0, &m1}) {
    b->release();
  }
  return rc;
}

// ---- sibling binaries' kernels, host-pointer convenience forms ----
int derp_layer_disparities(derp_ctx* c, const float* foreground, const float* background, size_t n, uint8_t* out) {
  if (!c || !foreground || !background || !out) {
    return fail(c, "bad arguments");
  }
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf f, b, o;
  int rc = 0;
  if (f.ensure(n * 4) || b.ensure(n * 4) || o.ensure(n)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(f.p, foreground, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b.p, background, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layer_disparities, dim3(flat_grid(n)), dim3(256), 0, c->stream, f.as<float>(), b.as<float>(), n,
                       o.as<uint8_t>());
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, o.p, n, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_layer_disparities: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* x : {&f, &b, &o}) {
    x->release();
  }
  return rc;
}
// ---- rephotography score (RephotographyUtil.h:38-116, ComputeRephotographyErrors.cpp:69-189) ----
int derp_ssim(derp_ctx* c, const float* x_bgr, const float* y_bgr, int w, int h, int blur_radius, float alpha,
              float beta, float gamma, float* score_bgr) {
  auto is01 = [](float v) { return v == 0.0f || v == 1.0f; };
  if (!c || !x_bgr || !y_bgr || !score_bgr || w <= 0 || h <= 0 || blur_radius < 1 || blur_radius > 15) {
    return fail(c, "bad arguments (blur_radius must be 1..15)");
  }
  if (!is01(alpha) || !is01(beta) || !is01(gamma)) {
    return fail(c, "exponents other than 0 and 1 are not supported (computeScoreMap uses MSSIM = 1,1,1 / NCC = 0,0,1)");
  }
  HIPCHK(c, hipSetDevice(c->device));
  // getGaussianKernel(2r + 1, 1.5, CV_32F): OpenCV 4's order of operations, in double, rounded to float
  GaussCoef coef{};
  {
    const int n = 2 * blur_radius + 1;
    const double sigma = 1.5f, scale2X = -0.125 / (sigma * sigma);
    double t[16], sum = 0;
    for (int i = 0, x = 1 - n; i < blur_radius; ++i, x += 2) {
      t[i] = std::exp((double)(x * x) * scale2X);
      sum += t[i];
    }
    sum *= 2;
    sum += 1;
    const double mul = 1.0 / sum;
    coef.k[0] = (float)mul;
    for (int i = 0; i < blur_radius; ++i) {
      coef.k[blur_radius - i] = (float)(t[i] * mul);
    }
  }
  const size_t n3 = (size_t)w * h * 3, bytes = n3 * 4;
  DevBuf x, y, muX, muY, a, b, cc, tmp, s2x, s2y, sxy;
  int rc = 0;
  bool oom = false;
  for (DevBuf* buf : {&x, &y, &muX, &muY, &a, &b, &cc, &tmp, &s2x, &s2y, &sxy}) {
    oom = oom || buf->ensure(bytes);
  }
  if (oom) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(x.p, x_bgr, bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(y.p, y_bgr, bytes, hipMemcpyHostToDevice);
    const dim3 grid = grid2d(w * 3, h, 1, kBlk2d);
    auto blur = [&](const DevBuf& in, DevBuf& out) {
      hipLaunchKernelGGL(k_gauss_f32c3, grid, kBlk2d, 0, c->stream, in.as<float>(), tmp.as<float>(), w, h, blur_radius, coef, 0);
      hipLaunchKernelGGL(k_gauss_f32c3, grid, kBlk2d, 0, c->stream, tmp.as<float>(), out.as<float>(), w, h, blur_radius, coef, 1);
    };
    blur(x, muX);
    blur(y, muY);
    hipLaunchKernelGGL(k_ssim_moments, dim3(flat_grid(n3)), dim3(256), 0, c->stream, x.as<float>(), y.as<float>(),
                       muX.as<float>(), muY.as<float>(), a.as<float>(), b.as<float>(), cc.as<float>(), n3);
    blur(a, s2x);
    blur(b, s2y);
    blur(cc, sxy);
    // the score overwrites `a`
    hipLaunchKernelGGL(k_ssim_score, dim3(flat_grid(n3)), dim3(256), 0, c->stream, muX.as<float>(), muY.as<float>(),
                       s2x.as<float>(), s2y.as<float>(), sxy.as<float>(), alpha != 0.0f, beta != 0.0f, gamma != 0.0f,
                       a.as<float>(), n3);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(score_bgr, a.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_ssim: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* buf : {&x, &y, &muX, &muY, &a, &b, &cc, &tmp, &s2x, &s2y, &sxy}) {
    buf->release();
  }
  return rc;
}

int derp_average_score(const float* score_bgr, const uint8_t* mask, int w, int h, double* avg_bgr3) {
  if (!score_bgr || !mask || !avg_bgr3 || w <= 0 || h <= 0) {
    return 1;
  }
  const size_t n = (size_t)w * h;
  for (int ch = 0; ch < 3; ++ch) {
    double sum = 0;
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
      const float v = score_bgr[i * 3 + ch];
      if (mask[i] && !std::isnan(v)) {
        sum += v;
        ++cnt;
      }
    }
    avg_bgr3[ch] = cnt ? sum / (double)cnt : 0.0;
  }
  return 0;
}

int derp_rephotograph_upload(derp_ctx* c, const uint16_t* const* colors, const float* const* disparities, int w, int h) {
  if (!c || !colors || !disparities || w <= 0 || h <= 0) {
    return fail(c, "bad arguments");
  }
  if ((size_t)w * h > (1u << 24) || c->S > 256) {
    return fail(c, "rephotography keys hold 24 bits of pixel index and 8 bits of camera index");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  c->rephotoW = c->rephotoH = 0;
  ALLOC(c, c->rephotoColor, (size_t)c->S * n * 6);
  ALLOC(c, c->rephotoDisp, (size_t)c->S * n * 4);
  for (int s = 0; s < c->S; ++s) {
    if (!colors[s] || !disparities[s]) {
      return fail(c, "null colour / disparity for source %d", s);
    }
    HIPCHK(c, hipMemcpy((char*)c->rephotoColor.p + (size_t)s * n * 6, colors[s], n * 6, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy((char*)c->rephotoDisp.p + (size_t)s * n * 4, disparities[s], n * 4, hipMemcpyHostToDevice));
  }
  c->rephotoW = w;
  c->rephotoH = h;
  return 0;
}

int derp_rephotograph_render(derp_ctx* c, int target, float* out_bgra) {
  if (!c || !out_bgra || target < 0 || target >= c->S) {
    return fail(c, "bad arguments");
  }
  if (c->rephotoW <= 0) {
    return fail(c, "derp_rephotograph_upload has not been called");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const int w = c->rephotoW, h = c->rephotoH;
  const size_t n = (size_t)w * h;
  DevBuf key, out;
  int rc = 0;
  if (key.ensure(n * 8) || out.ensure(n * 16)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemsetAsync(key.p, 0xff, n * 8, c->stream);
    hipLaunchKernelGGL(k_rephoto_splat, grid2d(w, h, c->S, kBlk2d), kBlk2d, 0, c->stream, c->camsSrc.as<Cam>(), target,
                       c->rephotoDisp.as<float>(), w, h, key.as<unsigned long long>());
    hipLaunchKernelGGL(k_rephoto_resolve, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, c->camsSrc.as<Cam>(), target,
                       c->rephotoColor.as<uint16_t>(), key.as<unsigned long long>(), w, h, out.as<float4>());
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(out_bgra, out.p, n * 16, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_rephotograph_render: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  key.release();
  out.release();
  return rc;
}

int derp_rephotograph(derp_ctx* c, int target, const uint16_t* const* colors, const float* const* disparities, int w,
                      int h, float* out_bgra) {
  if (!c || target < 0 || target >= c->S) {
    return fail(c, "bad arguments");
  }
  TRY(derp_rephotograph_upload(c, colors, disparities, w, h));
  return derp_rephotograph_render(c, target, out_bgra);
}

// CanopyScene::cubemap for the cameras `include[s] != 0` of the last derp_rephotograph_upload, seen from
// `centre` (rig space): BGRA float [6 * edge][edge] (ComputeRephotographyErrors.cpp:77-95 generateCubemaps)
int derp_canopy_cubemap(derp_ctx* c, const uint8_t* include, const double* centre, int edge, float* out_bgra) {
  if (!c || !include || !centre || !out_bgra || edge < 1 || edge > 8192) {
    return fail(c, "bad arguments");
  }
  if (c->rephotoW <= 0) {
    return fail(c, "derp_rephotograph_upload has not been called");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const int w = c->rephotoW, h = c->rephotoH, E = edge;
  const size_t n = (size_t)w * h, nf = (size_t)E * E;
  // mip chain geometry (glGenerateMipmap): level sizes halve, rounding down, never below 1
  CanopyMips M;
  M.n = 0;
  size_t texels = 0;
  for (int lw = w, lh = h;; lw = std::max(1, lw >> 1), lh = std::max(1, lh >> 1)) {
    if (M.n >= kCanopyMaxLevels) {
      return fail(c, "image too large for the mip chain");
    }
    M.w[M.n] = lw;
    M.h[M.n] = lh;
    M.off[M.n] = (unsigned)texels;
    texels += (size_t)lw * lh;
    ++M.n;
    if (lw == 1 && lh == 1) {
      break;
    }
  }
  int nInc = 0;
  for (int s = 0; s < c->S; ++s) {
    nInc += include[s] != 0;
  }
  // per included camera: mesh vertices + the colour mip chain, built once and reused by the six faces
  DevBuf &vert = c->cnVert, &rgba = c->cnRgba, &zbuf = c->cnZ, &acc = c->cnAcc, &out = c->cnOut, &big = c->cnBig,
         &nBig = c->cnNBig;
  int rc = 0;
  if (vert.ensure((size_t)std::max(nInc, 1) * n * 16) || rgba.ensure((size_t)std::max(nInc, 1) * texels * 16) ||
      zbuf.ensure(nf * 8) || acc.ensure(nf * 16) || out.ensure(nf * 6 * 16) || big.ensure(n * 2 * sizeof(unsigned)) ||
      nBig.ensure(sizeof(unsigned))) {
    rc = fail(c, "out of device memory");
  } else {
    const float cx = (float)centre[0], cy = (float)centre[1], cz = (float)centre[2];  // position.cast<float>()
    std::vector<int> slotOf(c->S, -1);
    for (int s = 0, k = 0; s < c->S; ++s) {
      if (!include[s]) {
        continue;
      }
      slotOf[s] = k;
      float4* v = vert.as<float4>() + (size_t)k * n;
      float4* tex = rgba.as<float4>() + (size_t)k * texels;
      hipLaunchKernelGGL(k_canopy_mesh, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, c->camsSrc.as<Cam>(), s,
                         c->rephotoColor.as<uint16_t>() + (size_t)s * n * 3, c->rephotoDisp.as<float>() + (size_t)s * n, w,
                         h, v, tex);
      for (int l = 1; l < M.n; ++l) {
        hipLaunchKernelGGL(k_canopy_mip, grid2d(M.w[l], M.h[l], 1, kBlk2d), kBlk2d, 0, c->stream, tex + M.off[l - 1],
                           M.w[l - 1], M.h[l - 1], tex + M.off[l], M.w[l], M.h[l]);
      }
      ++k;
    }
    // the reference's order: face-outer, camera-inner (the accumulation order of the cameras is part of the result)
    for (int face = 0; face < 6 && !rc; ++face) {
      (void)hipMemsetAsync(acc.p, 0, nf * 16, c->stream);
      for (int s = 0; s < c->S; ++s) {
        if (!include[s]) {
          continue;
        }
        const float4* v = vert.as<float4>() + (size_t)slotOf[s] * n;
        const float4* tex = rgba.as<float4>() + (size_t)slotOf[s] * texels;
        (void)hipMemsetAsync(zbuf.p, 0, nf * 8, c->stream);
        (void)hipMemsetAsync(nBig.p, 0, sizeof(unsigned), c->stream);
        hipLaunchKernelGGL(k_canopy_raster, grid2d(w - 1, h - 1, 2, kBlk2d), kBlk2d, 0, c->stream, v, tex, M, w, h, cx, cy, cz,
                           face, E, zbuf.as<unsigned long long>(), big.as<unsigned>(), nBig.as<unsigned>());
        hipLaunchKernelGGL(k_canopy_raster_big, dim3(4096), dim3(256), 0, c->stream, v, tex, M, w, h, cx, cy, cz, face, E,
                           zbuf.as<unsigned long long>(), big.as<unsigned>(), nBig.as<unsigned>());
        hipLaunchKernelGGL(k_canopy_resolve, grid2d(E, E, 1, kBlk2d), kBlk2d, 0, c->stream, v, tex, M, w, h, cx, cy, cz, face, E,
                           zbuf.as<unsigned long long>(), acc.as<float4>());
      }
      hipLaunchKernelGGL(k_canopy_finish, grid2d(E, E, 1, kBlk2d), kBlk2d, 0, c->stream, acc.as<float4>(), face, E,
                         out.as<float4>());
      if (hipGetLastError() != hipSuccess) {
        rc = fail(c, "HIP error launching the canopy kernels");
      }
    }
    if (!rc && (hipStreamSynchronize(c->stream) != hipSuccess ||
                hipMemcpy(out_bgra, out.p, nf * 6 * 16, hipMemcpyDeviceToHost) != hipSuccess)) {
      rc = fail(c, "HIP error in derp_canopy_cubemap: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  return rc;
}

int derp_download_mismatch_mask(derp_ctx* c, int d, uint8_t* out) {
  TRY(need_current(c, false));
  if (d < 0 || d >= c->D || !out) {
    return fail(c, "bad destination index / null output");
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t n = npx(c, c->cur);
  HIPCHK(c, hipMemcpy(out, c->mismatchMask.as<uint8_t>() + (size_t)d * n, n, hipMemcpyDeviceToHost));
  return 0;
}
int derp_fov_mask(derp_ctx* c, int d, int w, int h, uint8_t* out) {
  if (!c || !out || d < 0 || d >= c->D || w <= 0 || h <= 0) {
    return fail(c, "bad arguments");
  }
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf m;
  if (m.ensure((size_t)w * h)) {
    return fail(c, "out of device memory");
  }
  hipLaunchKernelGGL(k_fov_mask, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>() + d, w, h,
                     m.as<uint8_t>());
  int rc = 0;
  if (hipStreamSynchronize(c->stream) != hipSuccess ||
      hipMemcpy(out, m.p, (size_t)w * h, hipMemcpyDeviceToHost) != hipSuccess) {
    rc = fail(c, "HIP error in derp_fov_mask: %s", hipGetErrorString(hipGetLastError()));
  }
  m.release();
  return rc;
}

int derp_upsample_disparity(derp_ctx* c, int d, const float* disp, int w, int h, const float* bg_disp_up,
                            const uint8_t* fg_mask, const uint8_t* fg_mask_up, int w_up, int h_up, int use_fg,
                            float* out) {
  if (!c || !disp || !out || d < 0 || d >= c->D) {
    return fail(c, "bad arguments");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h, nu = (size_t)w_up * h_up;
  DevBuf in, res, m, mu, bg, fov, fovu;
  int rc = 0;
  auto cleanup = [&]() {
    for (DevBuf* b : {&in, &res, &m, &mu, &bg, &fov, &fovu}) {
      b->release();
    }
  };
  do {
    if (in.ensure(n * 4) || res.ensure(nu * 4)) {
      rc = fail(c, "out of device memory");
      break;
    }
    (void)hipMemcpy(in.p, disp, n * 4, hipMemcpyHostToDevice);
    if (!use_fg) {
      rc = upsample_lanczos_dev(c, in.as<float>(), w, h, res.as<float>(), w_up, h_up, 1, n, nu);
    } else {
      if (!bg_disp_up || !fg_mask || !fg_mask_up) {
        rc = fail(c, "foreground-mask upsample needs bg_disp_up, fg_mask and fg_mask_up");
        break;
      }
      if (m.ensure(n) || mu.ensure(nu) || bg.ensure(nu * 4) || fov.ensure(n) || fovu.ensure(nu)) {
        rc = fail(c, "out of device memory");
        break;
      }
      (void)hipMemcpy(m.p, fg_mask, n, hipMemcpyHostToDevice);
      (void)hipMemcpy(mu.p, fg_mask_up, nu, hipMemcpyHostToDevice);
      (void)hipMemcpy(bg.p, bg_disp_up, nu * 4, hipMemcpyHostToDevice);
      const int zero = 0;
      DevBuf idx;
      if (idx.ensure(sizeof(int))) {
        rc = fail(c, "out of device memory");
        break;
      }
      (void)hipMemcpy(idx.p, &zero, sizeof(int), hipMemcpyHostToDevice);
      // fov masks of camera d at both sizes, AND-ed with the fg masks (UpsampleDisparityLib.cpp:163-179)
      hipLaunchKernelGGL(k_fov_mask, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>() + d, w, h,
                         fov.as<uint8_t>());
      hipLaunchKernelGGL(k_fov_mask, grid2d(w_up, h_up, 1, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>() + d,
                         w_up, h_up, fovu.as<uint8_t>());
      hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(n), 1), dim3(256), 0, c->stream, fov.as<uint8_t>(),
                         m.as<uint8_t>(), idx.as<int>(), 0, n, fov.as<uint8_t>());
      hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(nu), 1), dim3(256), 0, c->stream, fovu.as<uint8_t>(),
                         mu.as<uint8_t>(), idx.as<int>(), 0, nu, fovu.as<uint8_t>());
      rc = upsample_masked_dev(c, in.as<float>(), fov.as<uint8_t>(), w, h, fovu.as<uint8_t>(), bg.as<float>(),
                               res.as<float>(), w_up, h_up);
      (void)hipStreamSynchronize(c->stream);
      idx.release();
    }
    if (rc) {
      break;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(out, res.p, nu * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_upsample_disparity");
    }
  } while (0);
  cleanup();
  return rc;
}

int derp_joint_bilateral_u16(derp_ctx* c, const float* image, const uint16_t* guide, const uint8_t* mask, int w, int h,
                             int radius, float sigma, float w0, float w1, float w2, float* out) {
  if (!c || !image || !guide || !mask || !out || radius < 0 || bilateral_lds_bytes(radius) > 64 * 1024) {
    return fail(c, "bad arguments (radius must be in [0, 47])");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  DevBuf im, g3, g4, m, res;
  int rc = 0;
  if (im.ensure(n * 4) || g3.ensure(n * 6) || g4.ensure(n * 8) || m.ensure(n) || res.ensure(n * 4)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(im.p, image, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(g3.p, guide, n * 6, hipMemcpyHostToDevice);
    (void)hipMemcpy(m.p, mask, n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, g3.as<uint16_t>(), g4.as<ushort4>(), n);
    hipLaunchKernelGGL(k_joint_bilateral<true>, dim3((w + 15) / 16, (h + 15) / 16, 1), dim3(256),
                       bilateral_lds_bytes(radius), c->stream, im.as<float>(), (const void*)g4.as<ushort4>(),
                       m.as<uint8_t>(), w, h, radius, sigma, w0, w1, w2, res.as<float>(), n, n, (const int*)nullptr);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, res.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_joint_bilateral_u16: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* b : {&im, &g3, &g4, &m, &res}) {
    b->release();
  }
  return rc;
}

int derp_joint_bilateral_f32(derp_ctx* c, const float* image, const float* guide, const uint8_t* mask, int w, int h,
                             int radius, float sigma, float w0, float w1, float w2, float* out) {
  if (!c || !image || !guide || !mask || !out || radius < 0 || bilateral_lds_bytes(radius) > 64 * 1024) {
    return fail(c, "bad arguments (radius must be in [0, 47])");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  DevBuf im, g, m, res;
  int rc = 0;
  if (im.ensure(n * 4) || g.ensure(n * 12) || m.ensure(n) || res.ensure(n * 4)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(im.p, image, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(g.p, guide, n * 12, hipMemcpyHostToDevice);
    (void)hipMemcpy(m.p, mask, n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_joint_bilateral<false>, dim3((w + 15) / 16, (h + 15) / 16, 1), dim3(256),
                       bilateral_lds_bytes(radius), c->stream, im.as<float>(), (const void*)g.as<float>(),
                       m.as<uint8_t>(), w, h, radius, sigma, w0, w1, w2, res.as<float>(), n, n, (const int*)nullptr);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, res.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_joint_bilateral_f32: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* b : {&im, &g, &m, &res}) {
    b->release();
  }
  return rc;
}

int derp_masked_median(derp_ctx* c, const float* image, const float* background, const uint8_t* mask, int w, int h,
                       int radius, float* out) {
  if (!c || !image || !mask || !out || radius < 1 || radius > 2) {
    return fail(c, "bad arguments (radius must be 1 or 2)");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  DevBuf im, bg, m, res;
  int rc = 0;
  if (im.ensure(n * 4) || (background && bg.ensure(n * 4)) || m.ensure(n) || res.ensure(n * 4)) {
    rc = fail(c, "out of device memory");
  } else {
    (void)hipMemcpy(im.p, image, n * 4, hipMemcpyHostToDevice);
    if (background) {
      (void)hipMemcpy(bg.p, background, n * 4, hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(m.p, mask, n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_masked_median, grid2d(w, h, 1, kBlk2d), kBlk2d, 0, c->stream, im.as<float>(),
                       background ? bg.as<float>() : (const float*)nullptr, m.as<uint8_t>(), w, h, radius,
                       res.as<float>(), n, (const uint8_t*)nullptr);
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, res.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_masked_median: %s", hipGetErrorString(hipGetLastError()));
    }
  }
  for (DevBuf* b : {&im, &bg, &m, &res}) {
    b->release();
  }
  return rc;
}

int derp_temporal_filter_dev(derp_ctx* c, const void* const* guides, const float* const* disps,
                             const uint8_t* const* masks, int n_frames, int w, int h, int frame_offset, float sigma,
                             int space_radius, float w0, float w1, float w2, float* out_dev) {
  if (!c || n_frames < 1 || frame_offset < 0 || frame_offset >= n_frames) {
    return fail(c, "temporal window must hold at least one frame and contain the centre frame");
  }
  HIPCHK(c, hipSetDevice(c->device));
  return temporal_launch(c, guides, disps, masks, n_frames, frame_offset, w, h, 1, sigma, space_radius, w0, w1, w2, out_dev,
                         nullptr);
}

int derp_temporal_filter(derp_ctx* c, const uint16_t* const* guides, const float* const* disps,
                         const uint8_t* const* masks, int n_frames, int w, int h, int frame_offset, float sigma,
                         int space_radius, float w0, float w1, float w2, float* out) {
  if (!c || n_frames < 1) {
    return fail(c, "temporal window must hold at least one frame");
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)w * h;
  std::vector<DevBuf> g4(n_frames), im(n_frames), m(n_frames);
  DevBuf g3, res;
  int rc = 0;
  std::vector<const void*> gpv(n_frames);
  std::vector<const float*> ipv(n_frames);
  std::vector<const uint8_t*> mpv(n_frames);
  const void** gp = gpv.data();
  const float** ip = ipv.data();
  const uint8_t** mp = mpv.data();
  do {
    if (g3.ensure(n * 6) || res.ensure(n * 4)) {
      rc = fail(c, "out of device memory");
      break;
    }
    for (int t = 0; t < n_frames && !rc; ++t) {
      if (g4[t].ensure(n * 8) || im[t].ensure(n * 4) || m[t].ensure(n)) {
        rc = fail(c, "out of device memory");
        break;
      }
      (void)hipMemcpy(g3.p, guides[t], n * 6, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, g3.as<uint16_t>(),
                         g4[t].as<ushort4>(), n);
      (void)hipStreamSynchronize(c->stream);
      (void)hipMemcpy(im[t].p, disps[t], n * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(m[t].p, masks[t], n, hipMemcpyHostToDevice);
      gp[t] = g4[t].p;
      ip[t] = im[t].as<float>();
      mp[t] = m[t].as<uint8_t>();
    }
    if (rc) {
      break;
    }
    rc = derp_temporal_filter_dev(c, gp, ip, mp, n_frames, w, h, frame_offset, sigma, space_radius, w0, w1, w2,
                                  res.as<float>());
    if (rc) {
      break;
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(out, res.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error in derp_temporal_filter: %s", hipGetErrorString(hipGetLastError()));
    }
  } while (0);
  for (auto* v : {&g4, &im, &m}) {
    for (auto& b : *v) {
      b.release();
    }
  }
  g3.release();
  res.release();
  return rc;
}

int derp_dev_disparity(derp_ctx* c, int level, int d, float** ptr, size_t* bytes) {
  TRY(check_level(c, level));
  if (d < 0 || d >= c->D || !ptr || !bytes) {
    return fail(c, "bad destination index / null output");
  }
  const size_t n = npx(c, level);
  *ptr = c->pyrDisp[level].as<float>() + (size_t)d * n;
  *bytes = n * sizeof(float);
  return 0;
}
int derp_dev_color(derp_ctx* c, int level, int s, void** ptr, size_t* bytes) {
  TRY(check_level(c, level));
  if (s < 0 || s >= c->S || !ptr || !bytes) {
    return fail(c, "bad source index / null output");
  }
  const size_t n = npx(c, level);
  *ptr = c->pyrColor[level].as<ushort4>() + (size_t)s * n;
  *bytes = n * sizeof(ushort4);
  return 0;
}
int derp_dev_mask(derp_ctx* c, int level, int d, uint8_t** ptr, size_t* bytes) {
  TRY(check_level(c, level));
  if (d < 0 || d >= c->D || !ptr || !bytes) {
    return fail(c, "bad destination index / null output");
  }
  HIPCHK(c, hipSetDevice(c->device));
  // fov & fg of `level` (TemporalBilateralFilter.cpp:150-160) for every destination, into a buffer of its
  // own (never a working buffer of the level loop), complete when this call returns
  const int W = c->LW[level], H = c->LH[level];
  const size_t n = (size_t)W * H;
  ALLOC(c, c->devMask, n * c->D);
  hipLaunchKernelGGL(k_fov_mask, grid2d(W, H, c->D, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>(), W, H,
                     c->devMask.as<uint8_t>());
  KCHECK(c);
  hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(n), c->D), dim3(256), 0, c->stream, c->devMask.as<uint8_t>(),
                     c->pyrFg[level].as<uint8_t>(), c->dst2src.as<int>(), 0, n, c->devMask.as<uint8_t>());
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *ptr = c->devMask.as<uint8_t>() + (size_t)d * n;
  *bytes = n;
  return 0;
}

int derp_get_counters(derp_ctx* c, uint64_t* n_cost, uint64_t* n_pair, uint64_t* insufficient) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<unsigned long long> h((size_t)ST_COUNT * kMaxLevels * 4);
  HIPCHK(c, hipMemcpy(h.data(), c->counters.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  uint64_t a = 0, b = 0, i = 0;
  for (size_t k = 0; k < h.size(); k += 4) {
    a += h[k];
    b += h[k + 1];
    i += h[k + 2];
  }
  if (n_cost) {
    *n_cost = a;
  }
  if (n_pair) {
    *n_pair = b;
  }
  if (insufficient) {
    *insufficient = i;
  }
  return 0;
}
int derp_reset_counters(derp_ctx* c) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipMemsetAsync(c->counters.p, 0, c->counters.bytes, c->stream));
  return 0;
}
int derp_profile_enable(derp_ctx* c, int on) {
  if (!c) {
    return 1;
  }
  c->profiling = on != 0;
  return 0;
}
int derp_profile_reset(derp_ctx* c) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  memset(c->accMs, 0, sizeof c->accMs);
  memset(c->accLaunch, 0, sizeof c->accLaunch);
  return derp_reset_counters(c);
}
int derp_profile_query(derp_ctx* c, const char* stage, int level, double* ms, int* launches, uint64_t* n_cost,
                       uint64_t* n_pair) {
  if (!c || !stage) {
    return 1;
  }
  int st = -1;
  for (int i = 0; i < ST_COUNT; ++i) {
    if (strcmp(stage, kStageNames[i]) == 0) {
      st = i;
    }
  }
  if (st < 0) {
    return fail(c, "unknown stage '%s'", stage);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  drain_spans(c);
  std::vector<unsigned long long> h((size_t)kMaxLevels * 4);
  HIPCHK(c, hipMemcpy(h.data(), counter_slot(c, st, 0), h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double m = 0;
  int l = 0;
  uint64_t a = 0, b = 0;
  for (int lv = 0; lv < kMaxLevels; ++lv) {
    if (level >= 0 && lv != level) {
      continue;
    }
    m += c->accMs[st][lv];
    l += c->accLaunch[st][lv];
    a += h[(size_t)lv * 4];
    b += h[(size_t)lv * 4 + 1];
  }
  if (ms) {
    *ms = m;
  }
  if (launches) {
    *launches = l;
  }
  if (n_cost) {
    *n_cost = a;
  }
  if (n_pair) {
    *n_pair = b;
  }
  return 0;
}
int derp_profile_memoised(derp_ctx* c, const char* stage, int level, uint64_t* n_memoised) {
  if (!c || !stage || !n_memoised) {
    return 1;
  }
  int st = -1;
  for (int i = 0; i < ST_COUNT; ++i) {
    if (strcmp(stage, kStageNames[i]) == 0) {
      st = i;
    }
  }
  if (st < 0) {
    return fail(c, "unknown stage '%s'", stage);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<unsigned long long> h((size_t)kMaxLevels * 4);
  HIPCHK(c, hipMemcpy(h.data(), counter_slot(c, st, 0), h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  uint64_t m = 0;
  for (int lv = 0; lv < kMaxLevels; ++lv) {
    if (level < 0 || lv == level) {
      m += h[(size_t)lv * 4 + 3];
    }
  }
  *n_memoised = m;
  return 0;
}
int derp_device_memory(derp_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!c) {
    return 1;
  }
  HIPCHK(c, hipSetDevice(c->device));
  size_t f = 0, t = 0;
  HIPCHK(c, hipMemGetInfo(&f, &t));
  if (free_bytes) {
    *free_bytes = f;
  }
  if (total_bytes) {
    *total_bytes = t;
  }
  return 0;
}

int derp_device_name(derp_ctx* c, char* buf, int n) {
  if (!c || !buf || n <= 0) {
    return 1;
  }
  hipDeviceProp_t prop;
  HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
  snprintf(buf, n, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

// ---- host-only self checks ----
namespace {
struct HostPairs {
  SsdPair* p;
  SsdPair get(int i) const {
    return p[i];
  }
  void set(int i, const SsdPair& v) {
    p[i] = v;
  }
};
}  // namespace
int derp_host_nth_element_pairs(float* pairs, int n, int nth) {
  HostPairs acc{reinterpret_cast<SsdPair*>(pairs)};
  GccSelect<HostPairs> sel(acc);
  sel.nth_element(nth, n);
  return 0;
}
float derp_host_minstd_uniform(int seed, uint64_t draw_index, float a, float b) {
  uint32_t state = minstd_jump(minstd_seed(seed), draw_index);
  return minstd_uniform(state, a, b);
}

}  // extern "C"

#include "derp_sequence.h"
