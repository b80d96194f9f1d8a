// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// CPU restatement of facebook360_dep's depth_estimation hot path,
// plain C++17, no third-party dependencies. Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.
//
// Each function cites the reference file:line it follows (paths relative to
// /root/reference/). Pinning status:
//   * Camera maths          — pinned by the reference's own known-answer tests
//                             (source/test/util/{FTheta,Rectilinear,Orthographic}Test.cpp)
//                             and by vectors generated from scripts/util/camera.py.
//   * cost / propagation / filters / upsample — PARITY UNPINNED by the reference
//     (its only checks need S3 datasets); OpenCV-defined arithmetic restated in
//     oracle_cv.h is likewise unpinned. The restatement + committed goldens are the pin.
#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "oracle_camera.h"
#include "oracle_cv.h"
#include "oracle_canopy.h"

namespace oracle {

// ---- constants: Derp.h:26-48, DerpUtil.h:22-43 ----
static const int kSearchWindowRadius = 1;
static const int kMinOverlappingCams = 2;
static const int kNumDepths = 150;
static const float kRandomPropMaxCost = 5.0;
static const float kRandomPropHighVarDeviation = 0.1;
static const int kMedianFilterRadius = 1;
static const int kBilateralSpaceRadiusMin = 1;
static const int kBilateralSpaceRadiusMax = 5;
static const float kBilateralSigma = 0.005;
static const float kBilateralWeightR = 1.0;
static const float kBilateralWeightG = 1.0;
static const float kBilateralWeightB = 0.5;
static const float kLevelScale = 0.9f;
static const float kRgbWeights[3] = {0.3333f, 0.3334f, 0.3333f};
static const float kMinVar = 1.0f / 12.0f / 65025.0f;
static const int kCandidateTemplate[9][2] =
    {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-2, -2}, {2, -2}, {-2, 2}, {2, 2}};

template <typename T>
static inline T clampT(const T& x, const T& a, const T& b) { // MathUtil.h:37-39
  return x < a ? a : x > b ? b : x;
}

// ThreadPool.h:23-57 runs one task per row in batches; results do not depend on
// the batching, so a plain strided parallel-for is equivalent.
static void parallelFor(int begin, int end, int threads, const std::function<void(int)>& fn) {
  int n = threads < 0 ? std::max(1u, std::thread::hardware_concurrency()) : threads;
  if (n <= 1 || end - begin <= 1) {
    for (int i = begin; i < end; ++i) {
      fn(i);
    }
    return;
  }
  std::atomic<int> next(begin);
  std::vector<std::thread> pool;
  for (int t = 0; t < n; ++t) {
    pool.emplace_back([&] {
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= end) {
          return;
        }
        fn(i);
      }
    });
  }
  for (auto& th : pool) {
    th.join();
  }
}

// ---- CvUtil.h:78-120: clampToEdge / bilerp / getPixelBilinear ----
// NOTE (SURVEY §7 "Truncating bilinear"): for Vec<ushort,3> the per-channel call
// resolves to the scalar template with T=ushort, whose return type truncates the
// float expression to an integer. Vec2f / float stay float.
template <typename T>
static inline const T& clampToEdge(const Img<T>& src, int x, int y) {
  return src.at(clampT(y, 0, src.h - 1), clampT(x, 0, src.w - 1));
}
static inline uint16_t
bilerpU16(uint16_t p00, uint16_t p01, uint16_t p10, uint16_t p11, float xw, float yw) {
  return (uint16_t)(
      (1 - xw) * (1 - yw) * p00 + xw * (1 - yw) * p01 + (1 - xw) * yw * p10 + xw * yw * p11);
}
static inline float bilerpF(float p00, float p01, float p10, float p11, float xw, float yw) {
  return (1 - xw) * (1 - yw) * p00 + xw * (1 - yw) * p01 + (1 - xw) * yw * p10 + xw * yw * p11;
}
static inline Px3w getPixelBilinear(const Img<Px3w>& src, const float x, const float y) {
  const float xf = std::round(x);
  const float yf = std::round(y);
  const int xi = xf;
  const int yi = yf;
  const Px3w& p00 = clampToEdge(src, xi - 1, yi - 1);
  const Px3w& p01 = clampToEdge(src, xi, yi - 1);
  const Px3w& p10 = clampToEdge(src, xi - 1, yi);
  const Px3w& p11 = clampToEdge(src, xi, yi);
  const float xw = x - xf + 0.5f, yw = y - yf + 0.5f;
  Px3w r;
  for (int i = 0; i < 3; ++i) {
    r.c[i] = bilerpU16(p00.c[i], p01.c[i], p10.c[i], p11.c[i], xw, yw);
  }
  return r;
}
static inline Px2f getPixelBilinear(const Img<Px2f>& src, const float x, const float y) {
  const float xf = std::round(x);
  const float yf = std::round(y);
  const int xi = xf;
  const int yi = yf;
  const Px2f& p00 = clampToEdge(src, xi - 1, yi - 1);
  const Px2f& p01 = clampToEdge(src, xi, yi - 1);
  const Px2f& p10 = clampToEdge(src, xi - 1, yi);
  const Px2f& p11 = clampToEdge(src, xi, yi);
  const float xw = x - xf + 0.5f, yw = y - yf + 0.5f;
  Px2f r;
  for (int i = 0; i < 2; ++i) {
    r.c[i] = bilerpF(p00.c[i], p01.c[i], p10.c[i], p11.c[i], xw, yw);
  }
  return r;
}
static inline float getPixelBilinear(const Img<float>& src, const float x, const float y) {
  const float xf = std::round(x);
  const float yf = std::round(y);
  const int xi = xf;
  const int yi = yf;
  return bilerpF(
      clampToEdge(src, xi - 1, yi - 1),
      clampToEdge(src, xi, yi - 1),
      clampToEdge(src, xi - 1, yi),
      clampToEdge(src, xi, yi),
      x - xf + 0.5f,
      y - yf + 0.5f);
}

// ---- DerpUtil.cpp:38-73 ----
static inline V3 dstToWorldPoint(
    const Camera& camDst, const int x, const int y, const float disparity, const int dstW, const int dstH) {
  V2 p = {(x + 0.5) / dstW, (y + 0.5) / dstH};
  if (!camDst.isNormalized()) {
    p = {p.x * camDst.resolution.x, p.y * camDst.resolution.y};
  }
  return camDst.rig(p, 1.0f / disparity);
}
static inline bool
worldToSrcPoint(V2& pSrc, const V3& pWorld, const Camera& camSrc, const int srcW, const int srcH) {
  if (!camSrc.sees(pWorld, pSrc)) {
    return false;
  }
  if (camSrc.isNormalized()) {
    pSrc.x *= srcW;
    pSrc.y *= srcH;
  }
  return true;
}

// ---- DerpUtil.cpp:126-162 ----
static inline std::pair<float, float> computeSSD(
    const Img<Px3w>& dstColor,
    const int x,
    const int y,
    const Px3w& dstBias,
    const Img<Px3w>& dstSrcColor,
    const float xDstSrc,
    const float yDstSrc,
    const Px3w& dstSrcBias,
    const int radius) {
  float bias[3];
  for (int c = 0; c < 3; ++c) {
    bias[c] = float(dstBias.c[c]) - float(dstSrcBias.c[c]);
  }
  std::pair<float, float> ssd = {0.0f, 0.0f};
  for (int dx = -radius; dx <= radius; ++dx) {
    for (int dy = -radius; dy <= radius; ++dy) {
      const Px3w& cDstW = dstColor.at(y + dy, x + dx);
      const Px3w cSrcW = getPixelBilinear(dstSrcColor, xDstSrc + dx, yDstSrc + dy);
      float diffBias[3], diffNoBias[3];
      for (int c = 0; c < 3; ++c) {
        diffBias[c] = float(cDstW.c[c]) - float(cSrcW.c[c]);
        diffNoBias[c] = diffBias[c] - bias[c];
      }
      // cv::Vec::dot accumulates left to right from 0
      float d0 = 0, d1 = 0;
      for (int c = 0; c < 3; ++c) {
        d0 += diffBias[c] * diffBias[c];
        d1 += diffNoBias[c] * diffNoBias[c];
      }
      ssd.first += d0;
      ssd.second += d1;
    }
  }
  const float maxDepth = 65535.0f;
  const float scaleFactor = 1.0f / (maxDepth * maxDepth);
  ssd.first *= scaleFactor;
  ssd.second *= scaleFactor;
  return ssd;
}

// ---- per-level state: PyramidLevel.h:24-131 ----
struct Params {
  int32_t level, numLevels;
  int32_t width, height;
  int32_t widthFull, heightFull;
  float minDepthM, maxDepthM;
  float varNoiseFloorFull, varHighThresh;
  int32_t randomProposals, pingPongIterations, mismatchesStartLevel;
  int32_t doBilateral, doMedian, useFgMasks, partialCoverage;
  int32_t threads;
};

struct Counters {
  uint64_t nCost = 0, nPair = 0;
};

struct Level {
  Params p;
  Rig rigSrc, rigDst; // normalised
  std::vector<int> dst2src;
  int S, D, W, H;
  float varNoiseFloor, varHighThresh;
  bool hasFg;

  std::vector<Img<Px3w>> srcColor;
  std::vector<Img<float>> srcVariance;
  std::vector<Img<uint8_t>> srcFg;

  std::vector<Img<float>> disparity, cost, confidence, bgDisp;
  std::vector<Img<uint8_t>> fovMask, mismatchMask;

  // proj[d*S+s]
  std::vector<Img<Px2f>> projWarp, projWarpInv;
  std::vector<Img<Px3w>> projColor, projColorBias;

  std::atomic<uint64_t> nCost{0}, nPair{0};
  std::atomic<int> insufficientCoverage{0};
  std::atomic<int> coverageCheckFailed{0};

  int idx(int d, int s) const {
    return d * S + s;
  }
  const Img<Px3w>& dstProjColor(int d) const {
    return projColor[idx(d, dst2src[d])];
  }
  const Img<Px3w>& dstProjColorBias(int d) const {
    return projColorBias[idx(d, dst2src[d])];
  }
  const Img<float>& dstVariance(int d) const {
    return srcVariance[dst2src[d]];
  }
  const Img<uint8_t>& dstFg(int d) const {
    return srcFg[dst2src[d]];
  }
  void flush(const Counters& c) {
    nCost += c.nCost;
    nPair += c.nPair;
  }
};

// ---- DerpUtil.cpp:214-237 + PyramidLevel.h:232-247 ----
static Img<float> computeImageVariance(const Img<Px3w>& image) {
  const int w = image.w, h = image.h;
  Img<Px3f> imageF(w, h), sq(w, h), mean, meanSq;
  const float scale = 1.0f / 65535.0f; // CvUtil.h:196-207 convertTo(CV_32F, 1/65535)
  for (size_t i = 0; i < image.d.size(); ++i) {
    for (int c = 0; c < 3; ++c) {
      const float v = image.d[i].c[c] * scale;
      imageF.d[i].c[c] = v;
      sq.d[i].c[c] = v * v;
    }
  }
  blur3x3F32C3(imageF, mean);
  blur3x3F32C3(sq, meanSq);
  Img<float> var(w, h);
  for (size_t i = 0; i < var.d.size(); ++i) {
    float v[3];
    for (int c = 0; c < 3; ++c) {
      v[c] = meanSq.d[i].c[c] - mean.d[i].c[c] * mean.d[i].c[c];
    }
    // varChannels[0]*w[2] + varChannels[1]*w[1] + varChannels[2]*w[0]   (BGR order)
    const float t = v[0] * kRgbWeights[2] + v[1] * kRgbWeights[1];
    var.d[i] = t * 1.0f + v[2] * kRgbWeights[0];
  }
  return var;
}

// ---- DerpUtil.cpp:239-276 ----
static Img<uint8_t> generateFovMask(const Camera& cam, int w, int h) {
  Img<uint8_t> m(w, h);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      V2 p = {x + 0.5, y + 0.5};
      if (cam.isNormalized()) {
        p = {p.x / w, p.y / h};
      }
      m.at(y, x) = !cam.isOutsideImageCircle(p);
    }
  }
  return m;
}

// ---- ImageUtil.cpp:142-167 ----
static Img<Px2f> computeWarpDstToSrc(const Camera& dst, const Camera& src) {
  const int dw = (int)dst.resolution.x, dh = (int)dst.resolution.y;
  const float nan = std::numeric_limits<float>::quiet_NaN();
  Img<Px2f> warp(dw, dh, Px2f{{nan, nan}});
  if (dst.id == src.id) {
    return warp;
  }
  for (int y = 0; y < dh; ++y) {
    for (int x = 0; x < dw; ++x) {
      const V2 dstPixel = {x + 0.5, y + 0.5};
      if (dst.isOutsideImageCircle(dstPixel)) {
        continue;
      }
      const V3 rig = dst.rigNearInfinity(dstPixel);
      V2 srcPixel;
      if (!src.sees(rig, srcPixel)) {
        continue;
      }
      warp.at(y, x) = Px2f{{float(srcPixel.x - 0.5f), float(srcPixel.y - 0.5f)}};
    }
  }
  return warp;
}

// ---- Derp.cpp:955-976 ----
static void precomputeProjections(Level& L) {
  const int n = L.D * L.S;
  L.projWarp.assign(n, Img<Px2f>());
  L.projWarpInv.assign(n, Img<Px2f>());
  parallelFor(0, n, L.p.threads, [&](int i) {
    const int d = i / L.S, s = i % L.S;
    const Camera camDst = L.rigDst[d].rescale({double(L.W), double(L.H)});
    const Camera camSrc = L.rigSrc[s].rescale({double(L.W), double(L.H)});
    L.projWarp[i] = computeWarpDstToSrc(camSrc, camDst);
    L.projWarpInv[i] = computeWarpDstToSrc(camDst, camSrc);
  });
}

// ---- Derp.cpp:978-1003 ----
static void reprojectColors(Level& L) {
  const int n = L.D * L.S;
  L.projColor.assign(n, Img<Px3w>());
  L.projColorBias.assign(n, Img<Px3w>());
  parallelFor(0, n, L.p.threads, [&](int i) {
    const int d = i / L.S, s = i % L.S;
    if (s == L.dst2src[d]) {
      L.projColor[i] = L.srcColor[s];
    } else {
      remapCubicU16C3(L.srcColor[s], L.projWarpInv[i], L.projColor[i]);
    }
    blur3x3U16C3(L.projColor[i], L.projColorBias[i]);
  });
}

// ---- Derp.cpp:104-226 ----
static inline std::pair<float, float>
computeCost(const Level& L, const int dstIdx, const float disparity, const int x, const int y, Counters& cnt) {
  ++cnt.nCost;
  const Img<Px3w>& dstColor = L.dstProjColor(dstIdx);
  const Camera& camDst = L.rigDst[dstIdx];
  const V3 pWorld = dstToWorldPoint(camDst, x, y, disparity, dstColor.w, dstColor.h);

  using SSDPair = std::pair<float, float>;
  SSDPair SSDs[64];
  int ssdCount = 0;
  const Img<Px3w>& dstColorBias = L.dstProjColorBias(dstIdx);
  for (int srcIdx = 0; srcIdx < L.S; ++srcIdx) {
    if (srcIdx == L.dst2src[dstIdx]) {
      continue;
    }
    const Camera& camSrc = L.rigSrc[srcIdx];
    V2 pSrc;
    if (!worldToSrcPoint(pSrc, pWorld, camSrc, L.W, L.H)) {
      continue;
    }
    const Img<Px2f>& dstProjWarp = L.projWarp[L.idx(dstIdx, srcIdx)];
    const Px2f pDstSrc = getPixelBilinear(dstProjWarp, float(pSrc.x), float(pSrc.y));
    const float xDstSrc = pDstSrc.c[0] + 0.5;
    const float yDstSrc = pDstSrc.c[1] + 0.5;
    if (std::isnan(xDstSrc) || std::isnan(yDstSrc)) {
      continue;
    }
    ++cnt.nPair;
    const Img<Px3w>& dstSrcColorBias = L.projColorBias[L.idx(dstIdx, srcIdx)];
    const Px3w dstSrcBias = getPixelBilinear(dstSrcColorBias, xDstSrc, yDstSrc);
    const Px3w& dstBias = dstColorBias.at(y, x);
    const Img<Px3w>& dstSrcColor = L.projColor[L.idx(dstIdx, srcIdx)];
    SSDs[ssdCount] = computeSSD(
        dstColor, x, y, dstBias, dstSrcColor, xDstSrc, yDstSrc, dstSrcBias, kSearchWindowRadius);
    ++ssdCount;
  }

  int keep = kMinOverlappingCams - 1;
  if (ssdCount < keep) {
    return {FLT_MAX, 0.0f};
  }
  keep = std::max<int>(keep, ssdCount - 2);
  std::nth_element(SSDs, SSDs + keep, SSDs + ssdCount);
  float cost = 0;
  for (int i = 0; i < keep; ++i) {
    cost += SSDs[i].second;
  }
  cost /= keep;
  const float trustCoef = 1.0f / keep;
  const float dstVariance = L.dstVariance(dstIdx).at(y, x);
  const float confidence = std::max(dstVariance, kMinVar);
  const float costFinal = cost * trustCoef / confidence;
  return {costFinal, confidence};
}

// ImageUtil.cpp:100-107
static inline double
probeDisparity(const int probe, const int probeCount, const double minD, const double maxD) {
  const double fraction = double(probe) / double(probeCount - 1);
  return fraction * minD + (1 - fraction) * maxD;
}

// ---- Derp.cpp:230-382 ----
static void computeBruteForceDisparity(Level& L, const int dstIdx) {
  Img<float>& dstDisparity = L.disparity[dstIdx];
  Img<float>& dstCosts = L.cost[dstIdx];
  Img<float>& dstConfidences = L.confidence[dstIdx];
  const float nan = std::numeric_limits<float>::quiet_NaN();

  std::vector<float> disparities(kNumDepths);
  const float minDisparity = 1.0f / L.p.maxDepthM;
  const float maxDisparity = 1.0f / L.p.minDepthM;
  for (int i = 0; i < kNumDepths; ++i) {
    disparities[i] = probeDisparity(i, kNumDepths, minDisparity, maxDisparity);
  }
  std::vector<Img<float>> costs(kNumDepths), confidences(kNumDepths);
  const Img<uint8_t>& fov = L.fovMask[dstIdx];
  const Img<uint8_t>& fg = L.dstFg(dstIdx);
  parallelFor(0, kNumDepths, L.p.threads, [&](int i) {
    Counters cnt;
    costs[i] = Img<float>(L.W, L.H, nan);
    confidences[i] = Img<float>(L.W, L.H, nan);
    const float disparity = disparities[i];
    const int radius = kSearchWindowRadius;
    for (int y = radius; y < L.H - radius; ++y) {
      for (int x = radius; x < L.W - radius; ++x) {
        const bool closer = L.hasFg ? (L.bgDisp[dstIdx].at(y, x) < disparity) : true;
        const bool ignore = !fov.at(y, x) || !fg.at(y, x) || !closer;
        if (ignore) {
          costs[i].at(y, x) = nan;
          confidences[i].at(y, x) = nan;
        } else {
          std::tie(costs[i].at(y, x), confidences[i].at(y, x)) =
              computeCost(L, dstIdx, disparity, x, y, cnt);
        }
      }
    }
    L.flush(cnt);
  });

  const int margin = kSearchWindowRadius;
  for (int y = margin; y < L.H - margin; ++y) {
    for (int x = margin; x < L.W - margin; ++x) {
      if (!fov.at(y, x)) {
        dstDisparity.at(y, x) = nan;
        continue;
      }
      if (!fg.at(y, x)) {
        dstDisparity.at(y, x) = L.bgDisp[dstIdx].at(y, x);
        continue;
      }
      float minCost = FLT_MAX;
      float minCostConfidence = 0;
      int best = -1;
      for (int i = 0; i < kNumDepths; ++i) {
        const float c = costs[i].at(y, x);
        if (c < minCost) {
          minCost = c;
          minCostConfidence = confidences[i].at(y, x);
          best = i;
        }
      }
      if (best == -1) {
        // reference: CHECK(partialCoverage || useForegroundMasks) then LOG(WARNING)
        if (!(L.p.partialCoverage || L.p.useFgMasks)) {
          L.coverageCheckFailed++;
        }
        L.insufficientCoverage++;
        dstDisparity.at(y, x) = minDisparity;
      } else {
        dstDisparity.at(y, x) = disparities[best];
      }
      dstCosts.at(y, x) = minCost;
      dstConfidences.at(y, x) = minCostConfidence;
    }
  }
  if (margin > 0) {
    for (int y = 0; y < L.H; ++y) {
      for (int x = 0; x < L.W; ++x) {
        if (x < margin || x >= L.W - margin || y < margin || y >= L.H - margin) {
          if (!fg.at(y, x)) {
            dstDisparity.at(y, x) = L.bgDisp[dstIdx].at(y, x);
            continue;
          }
          const int yy = clampT(y, margin, L.H - margin - 1);
          const int xx = clampT(x, margin, L.W - margin - 1);
          dstDisparity.at(y, x) = dstDisparity.at(yy, xx);
          dstCosts.at(y, x) = dstCosts.at(yy, xx);
          dstConfidences.at(y, x) = dstConfidences.at(yy, xx);
        }
      }
    }
  }
}

// ---- Derp.cpp:750-824 ----
static void randomProposalRow(Level& L, const int dstIdx, const int y, Counters& cnt) {
  std::default_random_engine engine;
  engine.seed(y * L.p.level);
  Img<float>& dstDisparity = L.disparity[dstIdx];
  Img<float>& dstCosts = L.cost[dstIdx];
  Img<float>& dstConfidence = L.confidence[dstIdx];
  const Img<float>& variance = L.dstVariance(dstIdx);
  const int numProposals = L.p.randomProposals;
  for (int x = kSearchWindowRadius; x < L.W - kSearchWindowRadius; ++x) {
    if (!L.fovMask[dstIdx].at(y, x)) {
      continue;
    }
    float currDisp = dstDisparity.at(y, x);
    if (!L.dstFg(dstIdx).at(y, x)) {
      dstDisparity.at(y, x) = L.bgDisp[dstIdx].at(y, x);
      continue;
    }
    const float varHighDev = kRandomPropHighVarDeviation * L.varHighThresh;
    const float varHighThresh = std::max(varHighDev, L.varNoiseFloor);
    if (variance.at(y, x) < varHighThresh) {
      continue;
    }
    float currCost, currConfidence;
    std::tie(currCost, currConfidence) = computeCost(L, dstIdx, currDisp, x, y, cnt);
    const float costThresh = std::fmin(0.5f * currCost, kRandomPropMaxCost);
    const float minDisp = L.hasFg ? L.bgDisp[dstIdx].at(y, x) : (1.0f / L.p.maxDepthM);
    const float maxDisp = 1.0f / L.p.minDepthM;
    float amplitude = (maxDisp - minDisp) / 2.0f;
    for (int i = 0; i < numProposals; ++i) {
      float propDisp = std::uniform_real_distribution<float>(
          std::max(float(minDisp), currDisp - amplitude),
          std::min(float(maxDisp), currDisp + amplitude))(engine);
      float propCost, propConfidence;
      std::tie(propCost, propConfidence) = computeCost(L, dstIdx, propDisp, x, y, cnt);
      if (propCost < currCost && propCost < costThresh) {
        currCost = propCost;
        currDisp = propDisp;
        currConfidence = propConfidence;
        amplitude /= 2.0f;
      }
    }
    dstDisparity.at(y, x) = currDisp;
    dstCosts.at(y, x) = currCost;
    dstConfidence.at(y, x) = currConfidence;
  }
}

// Derp.cpp:844-873
static void randomProposals(Level& L) {
  if (L.p.randomProposals <= 0 || L.p.level == L.p.numLevels - 1) {
    return;
  }
  for (int d = 0; d < L.D; ++d) {
    parallelFor(kSearchWindowRadius, L.H - kSearchWindowRadius, L.p.threads, [&](int y) {
      Counters cnt;
      randomProposalRow(L, d, y, cnt);
      L.flush(cnt);
    });
  }
}

// ---- Derp.cpp:403-551 ----
static void pingPong(Level& L, std::vector<float>* changedPctOut) {
  if (L.p.level == L.p.numLevels - 1) {
    return;
  }
  for (int dstIdx = 0; dstIdx < L.D; ++dstIdx) {
    Img<float>& disp = L.disparity[dstIdx];
    Img<float>& costs = L.cost[dstIdx];
    Img<float> dispRes = disp;
    Img<float> costsRes(L.W, L.H, INFINITY);
    Img<float> confidencesRes(L.W, L.H, 0.f);
    Img<uint8_t> changed(L.W, L.H, 1);
    const Img<uint8_t>& maskFov = L.fovMask[dstIdx];
    const Img<float>& variance = L.dstVariance(dstIdx);
    const Img<float>& confidences = L.confidence[dstIdx];
    for (int it = 1; it <= L.p.pingPongIterations; ++it) {
      const int radius = kSearchWindowRadius;
      parallelFor(radius, L.H - radius, L.p.threads, [&](int y) {
        Counters cnt;
        for (int x = radius; x < L.W - radius; ++x) {
          if (!maskFov.at(y, x)) {
            continue;
          }
          if (!L.dstFg(dstIdx).at(y, x)) {
            dispRes.at(y, x) = L.bgDisp[dstIdx].at(y, x);
            continue;
          }
          if (variance.at(y, x) < L.varNoiseFloor) {
            continue;
          }
          float bestCost = INFINITY;
          float bestDisparity = disp.at(y, x);
          float bestConfidence = confidences.at(y, x);
          const float backgroundDisparity = L.hasFg ? L.bgDisp[dstIdx].at(y, x) : 0;
          for (int k = 0; k < 9; ++k) {
            const int xx = clampT(x + kCandidateTemplate[k][0], 0, L.W - 1);
            const int yy = clampT(y + kCandidateTemplate[k][1], 0, L.H - 1);
            if (maskFov.at(yy, xx)) {
              const float d = disp.at(yy, xx);
              if (d >= backgroundDisparity && changed.at(yy, xx)) {
                const auto cv = computeCost(L, dstIdx, d, x, y, cnt);
                if (cv.first < bestCost) {
                  bestCost = cv.first;
                  bestDisparity = d;
                  bestConfidence = cv.second;
                }
              }
            }
          }
          dispRes.at(y, x) = bestDisparity;
          costsRes.at(y, x) = bestCost;
          confidencesRes.at(y, x) = bestConfidence;
        }
        L.flush(cnt);
      });
      // changed = disp != dispRes (NaN != NaN is true, as cv::compare CMP_NE)
      int count = 0, countFov = 0;
      for (size_t i = 0; i < disp.d.size(); ++i) {
        changed.d[i] = disp.d[i] != dispRes.d[i];
        count += changed.d[i];
        countFov += maskFov.d[i] != 0;
      }
      disp = dispRes;
      costs = costsRes;
      if (changedPctOut) {
        changedPctOut->push_back(100.0f * count / countFov);
      }
    }
  }
}

// ---- Derp.cpp:553-748: handleDisparityMismatches (off unless --mismatches_start_level >= level) ----
static void handleDisparityMismatches(Level& L) {
  if (L.p.level > L.p.mismatchesStartLevel || L.p.level == L.p.numLevels - 1) {
    return;
  }
  // CHECK_EQ(rigDst.size(), rigSrc.size()): the reference indexes dstDisparity(srcIdx)
  const float nan = std::numeric_limits<float>::quiet_NaN();
  std::vector<Img<float>> newDisp(L.D);
  parallelFor(0, L.D, L.p.threads, [&](int dstIdx) {
    const Img<float>& dstDisp = L.disparity[dstIdx];
    Img<uint8_t>& dstMask = L.mismatchMask[dstIdx];
    Img<float> dstDispNew(L.W, L.H, nan);
    const Img<float>& dstVar = L.dstVariance(dstIdx);
    const Camera& camDst = L.rigDst[dstIdx];
    for (int y = 0; y < L.H; ++y) {
      for (int x = 0; x < L.W; ++x) {
        if (!L.fovMask[dstIdx].at(y, x)) {
          continue;
        }
        std::vector<float> dispMatches, dispMismatches;
        // getSrcMismatches (Derp.cpp:553-603)
        if (L.dstFg(dstIdx).at(y, x)) {
          const V3 ptWorld = dstToWorldPoint(camDst, x, y, dstDisp.at(y, x), L.W, L.H);
          for (int srcIdx = 0; srcIdx < L.S; ++srcIdx) {
            if (srcIdx == L.dst2src[dstIdx]) {
              continue;
            }
            V2 ptSrc;
            if (!worldToSrcPoint(ptSrc, ptWorld, L.rigSrc[srcIdx], L.W, L.H)) {
              continue;
            }
            const float dSrc = getPixelBilinear(L.disparity[srcIdx], float(ptSrc.x), float(ptSrc.y));
            static const float kFractionChange = 0.1f;
            const float dDstMin = (1.0f - kFractionChange) * dstDisp.at(y, x);
            const float dDstMax = (1.0f + kFractionChange) * dstDisp.at(y, x);
            if (dDstMin <= dSrc && dSrc <= dDstMax) {
              dispMatches.push_back(dSrc);
            } else {
              dispMismatches.push_back(dSrc);
            }
          }
        }
        // updateDstDisparityAndMismatchMask (Derp.cpp:605-652)
        const float dispCurr = dstDisp.at(y, x);
        bool mask;
        float dispNew;
        if (dispMatches.size() + dispMismatches.size() == 0) {
          mask = false;
          dispNew = dispCurr;
        } else if (
            int(dispMatches.size()) >= kMinOverlappingCams - 1 || L.varHighThresh < dstVar.at(y, x) ||
            dstVar.at(y, x) < L.varNoiseFloor) {
          mask = false;
          dispNew = dispCurr;
        } else {
          mask = true;
          std::sort(dispMismatches.begin(), dispMismatches.end());
          int closer;
          for (closer = 0; closer < int(dispMismatches.size()); ++closer) {
            if (dispMismatches[closer] >= dispCurr) {
              break;
            }
          }
          const int median = closer / 2;
          dispNew = std::min(dispCurr, dispMismatches[median]);
        }
        dstMask.at(y, x) = mask;
        dstDispNew.at(y, x) = dispNew;
      }
    }
    newDisp[dstIdx] = dstDispNew;
  });
  for (int d = 0; d < L.D; ++d) {
    L.disparity[d] = newDisp[d];
  }
}

// ---- TemporalBilateralFilter.h:39-124, TGuide = Vec3w ----
static Img<float> generalizedJointBilateralFilterU16(
    const Img<float>& image,
    const Img<Px3w>& guide,
    const Img<Px3w>& neighborGuide,
    const Img<uint8_t>& mask,
    const int radius,
    const float sigma,
    const float weight0,
    const float weight1,
    const float weight2,
    const int threads) {
  Img<float> dest(image.w, image.h);
  parallelFor(0, image.h, threads, [&](int y) {
    for (int x = 0; x < image.w; ++x) {
      if (!mask.at(y, x)) {
        dest.at(y, x) = image.at(y, x);
        continue;
      }
      const Px3w guideColor = guide.at(y, x);
      float sumWeight = 0.0f;
      float weightedAvg = 0.0f;
      const float guideFactor = 1 / 65535.0f;
      const float neighborFactor = 1 / 65535.0f;
      for (int v = -radius; v <= radius; ++v) {
        for (int u = -radius; u <= radius; ++u) {
          const int sampleX = clampT(x + u, 0, image.w - 1);
          const int sampleY = clampT(y + v, 0, image.h - 1);
          if (!mask.at(sampleY, sampleX)) {
            continue;
          }
          const Px3w& nb = neighborGuide.at(sampleY, sampleX);
          auto sq = [](float a) { return a * a; };
          const float colorDiffSq =
              weight0 * sq((guideColor.c[0] * guideFactor) - (nb.c[0] * neighborFactor)) +
              weight1 * sq((guideColor.c[1] * guideFactor) - (nb.c[1] * neighborFactor)) +
              weight2 * sq((guideColor.c[2] * guideFactor) - (nb.c[2] * neighborFactor));
          const float weight = expf((-colorDiffSq / 3.0f) / (2.0f * (sigma * sigma)));
          sumWeight += weight;
          weightedAvg += weight * image.at(sampleY, sampleX);
        }
      }
      if (sumWeight != 0.0f) {
        weightedAvg /= sumWeight;
        dest.at(y, x) = weightedAvg;
      } else {
        dest.at(y, x) = image.at(y, x);
      }
    }
  });
  return dest;
}

// TGuide = Vec3f (UpsampleDisparity.cpp:109-128): maxPixelValue(CV_32F) = 1
static Img<float> generalizedJointBilateralFilterF32(
    const Img<float>& image,
    const Img<Px3f>& guide,
    const Img<uint8_t>& mask,
    const int radius,
    const float sigma,
    const float weight0,
    const float weight1,
    const float weight2,
    const int threads) {
  Img<float> dest(image.w, image.h);
  parallelFor(0, image.h, threads, [&](int y) {
    for (int x = 0; x < image.w; ++x) {
      if (!mask.at(y, x)) {
        dest.at(y, x) = image.at(y, x);
        continue;
      }
      const Px3f guideColor = guide.at(y, x);
      float sumWeight = 0.0f;
      float weightedAvg = 0.0f;
      const float factor = 1 / 1.0f;
      for (int v = -radius; v <= radius; ++v) {
        for (int u = -radius; u <= radius; ++u) {
          const int sampleX = clampT(x + u, 0, image.w - 1);
          const int sampleY = clampT(y + v, 0, image.h - 1);
          if (!mask.at(sampleY, sampleX)) {
            continue;
          }
          const Px3f& nb = guide.at(sampleY, sampleX);
          auto sq = [](float a) { return a * a; };
          const float colorDiffSq = weight0 * sq((guideColor.c[0] * factor) - (nb.c[0] * factor)) +
              weight1 * sq((guideColor.c[1] * factor) - (nb.c[1] * factor)) +
              weight2 * sq((guideColor.c[2] * factor) - (nb.c[2] * factor));
          const float weight = expf((-colorDiffSq / 3.0f) / (2.0f * (sigma * sigma)));
          sumWeight += weight;
          weightedAvg += weight * image.at(sampleY, sampleX);
        }
      }
      if (sumWeight != 0.0f) {
        weightedAvg /= sumWeight;
        dest.at(y, x) = weightedAvg;
      } else {
        dest.at(y, x) = image.at(y, x);
      }
    }
  });
  return dest;
}

static int bilateralRadius(int level) { // Derp.cpp:876-878
  const float scale = std::pow(kLevelScale, level);
  return std::max(std::ceil(kBilateralSpaceRadiusMax * scale), float(kBilateralSpaceRadiusMin));
}

// ---- Derp.cpp:875-902 ----
static void bilateralFilter(Level& L) {
  const int spaceRadius = bilateralRadius(L.p.level);
  for (int d = 0; d < L.D; ++d) {
    Img<float>& disparity = L.disparity[d];
    const Img<Px3w>& color = L.srcColor[L.dst2src[d]];
    Img<uint8_t> mask(L.W, L.H);
    for (size_t i = 0; i < mask.d.size(); ++i) {
      mask.d[i] = L.fovMask[d].d[i] & L.dstFg(d).d[i];
    }
    const Img<float> filtered = generalizedJointBilateralFilterU16(
        disparity,
        color,
        color,
        mask,
        spaceRadius,
        kBilateralSigma,
        kBilateralWeightB,
        kBilateralWeightG,
        kBilateralWeightR,
        L.p.threads);
    for (size_t i = 0; i < mask.d.size(); ++i) {
      if (L.dstFg(d).d[i]) {
        disparity.d[i] = filtered.d[i];
      }
    }
  }
}

// ---- CvUtil.h:336-385 ----
static Img<float> maskedMedianBlur(
    const Img<float>& mat, const Img<float>& background, const Img<uint8_t>& mask, const int radius) {
  Img<float> blurred(mat.w, mat.h, 0.0f);
  for (int y = 0; y < mat.h; ++y) {
    for (int x = 0; x < mat.w; ++x) {
      std::vector<float> values;
      if (!mask.at(y, x)) {
        if (!background.empty()) {
          blurred.at(y, x) = background.at(y, x);
        }
        continue;
      }
      for (int yy = y - radius; yy <= y + radius; ++yy) {
        for (int xx = x - radius; xx <= x + radius; ++xx) {
          if (0 > yy || yy >= mat.h || 0 > xx || xx >= mat.w) {
            continue;
          }
          if (!mask.at(yy, xx)) {
            continue;
          }
          if (std::isnan(mat.at(yy, xx)) || mat.at(yy, xx) == 0) {
            continue;
          }
          values.push_back(mat.at(yy, xx));
        }
      }
      if (!values.empty()) {
        const size_t n = values.size() / 2;
        std::partial_sort(values.begin(), values.begin() + n + 1, values.end());
        if (values.size() % 2 == 1) {
          blurred.at(y, x) = values[n];
        } else {
          blurred.at(y, x) = (values[n - 1] + values[n]) / 2.0;
        }
      }
    }
  }
  return blurred;
}

// ---- Derp.cpp:904-920 ----
static void medianFilter(Level& L) {
  parallelFor(0, L.D, L.p.threads, [&](int d) {
    Img<uint8_t> mask(L.W, L.H);
    for (size_t i = 0; i < mask.d.size(); ++i) {
      mask.d[i] = L.fovMask[d].d[i] & L.dstFg(d).d[i];
    }
    L.disparity[d] = maskedMedianBlur(L.disparity[d], L.bgDisp[d], mask, kMedianFilterRadius);
  });
}

// ---- Derp.cpp:940-951 ----
static void maskFov(Level& L) {
  const float nan = std::numeric_limits<float>::quiet_NaN();
  for (int d = 0; d < L.D; ++d) {
    for (size_t i = 0; i < L.disparity[d].d.size(); ++i) {
      if (!L.fovMask[d].d[i]) {
        L.disparity[d].d[i] = nan;
      }
    }
  }
}

// ---- UpsampleDisparityLib.cpp:27-147 ----
static std::vector<std::pair<int, int>> spiral(const int w) {
  int x = 0, y = 0, dx = 0, dy = -1, t = w;
  const int samples = t * t;
  std::vector<std::pair<int, int>> locs;
  for (int i = 0; i < samples; ++i) {
    const bool isValidX = (-w / 2 <= x) && (x <= w / 2);
    const bool isValidY = (-w / 2 <= y) && (y <= w / 2);
    if (isValidX && isValidY) {
      locs.emplace_back(x, y);
    }
    const bool isCorner = x == y;
    const bool isEdgeLeftX = (x < 0) && (x == -y);
    const bool isEdgeRightX = (x > 0) && (x == 1 - y);
    if (isCorner || isEdgeLeftX || isEdgeRightX) {
      t = dx;
      dx = -dy;
      dy = t;
    }
    x += dx;
    y += dy;
  }
  return locs;
}

static Img<float> replaceNans(
    const Img<float>& dispUp, const Img<float>& bgDispUp, const Img<uint8_t>& maskUp, const int radius) {
  Img<float> dispOut = dispUp;
  const auto spiralLocs = spiral(radius * 2 + 1);
  for (int py = 0; py < dispUp.h; ++py) {
    for (int px = 0; px < dispUp.w; ++px) {
      // maskNan = maskUp with (dispUp > 0) cleared: true = NaN (or <= 0) inside mask
      if (!maskUp.at(py, px) || dispUp.at(py, px) > 0) {
        continue;
      }
      for (const auto& loc : spiralLocs) {
        const int xx = clampT(px + loc.first, 0, dispUp.w - 1);
        const int yy = clampT(py + loc.second, 0, dispUp.h - 1);
        const float d = dispUp.at(yy, xx);
        if (d > 0) {
          dispOut.at(py, px) = d;
          break;
        }
      }
    }
  }
  for (size_t i = 0; i < dispOut.d.size(); ++i) {
    if (std::isnan(dispOut.d[i]) || dispOut.d[i] == 0) {
      dispOut.d[i] = bgDispUp.d[i];
    }
  }
  return dispOut;
}

static int getUpsampleRadius(int w, int wUp) { // UpsampleDisparityLib.cpp:93-96
  const float scale = float(wUp) / float(w);
  return scale * scale + 1;
}

static Img<float> upsampleDisparity(
    const Img<float>& disp,
    const Img<float>& bgDispUp,
    const Img<uint8_t>& mask, // fov & fg at coarse size
    const Img<uint8_t>& maskUp, // fov & fg at up size
    const int wUp,
    const int hUp,
    const bool useForegroundMasks) {
  const float nan = std::numeric_limits<float>::quiet_NaN();
  Img<float> dispUp;
  if (useForegroundMasks) {
    const int radius = getUpsampleRadius(mask.w, wUp);
    Img<float> dispSmallMasked = disp;
    for (size_t i = 0; i < disp.d.size(); ++i) {
      if (mask.d[i] == 0) {
        dispSmallMasked.d[i] = nan;
      }
    }
    Img<float> dispUpMasked;
    resizeNearest(dispSmallMasked, wUp, hUp, dispUpMasked);
    for (size_t i = 0; i < dispUpMasked.d.size(); ++i) {
      if (maskUp.d[i] == 0) {
        dispUpMasked.d[i] = nan;
      }
    }
    dispUp = replaceNans(dispUpMasked, bgDispUp, maskUp, radius);
  } else {
    const float minDisp = 1e-4;
    Img<float> dispSmallMasked = disp;
    for (size_t i = 0; i < disp.d.size(); ++i) {
      if (disp.d[i] != disp.d[i]) {
        dispSmallMasked.d[i] = minDisp;
      }
    }
    resizeLanczos4F32(dispSmallMasked, wUp, hUp, dispUp);
  }
  return dispUp;
}

// ---- TemporalBilateralFilter.h:126-172 (guide = Vec3w) ----
static Img<float> temporalJointBilateralFilter(
    const std::vector<const Px3w*>& guides,
    const std::vector<const float*>& images,
    const std::vector<const uint8_t*>& masks,
    const int w,
    const int h,
    const int frameOffset,
    const float sigma,
    const int spatialRadius,
    const float weight0,
    const float weight1,
    const float weight2,
    const int threads) {
  Img<float> result(w, h);
  const float maxImageValue = 65535.0f;
  const int nT = (int)guides.size();
  parallelFor(0, h, threads, [&](int y) {
    for (int x = 0; x < w; ++x) {
      const size_t c = size_t(y) * w + x;
      if (!masks[frameOffset][c]) {
        result.at(y, x) = images[frameOffset][c];
        continue;
      }
      float weightedSumPix = 0.0f;
      float sumWeight = 0.0f;
      const Px3w referenceColor = guides[frameOffset][c];
      for (int t = 0; t < nT; ++t) {
        for (int u = -spatialRadius; u <= spatialRadius; ++u) {
          for (int v = -spatialRadius; v <= spatialRadius; ++v) {
            const int sampleX = clampT(x + u, 0, w - 1);
            const int sampleY = clampT(y + v, 0, h - 1);
            const size_t si = size_t(sampleY) * w + sampleX;
            if (!masks[t][si]) {
              continue;
            }
            const Px3w sampleColor = guides[t][si];
            auto sq = [](float a) { return a * a; };
            // ushort - ushort promotes to int (exact, signed), then / float
            const float weightedDiff =
                weight0 * sq((referenceColor.c[0] - sampleColor.c[0]) / maxImageValue) +
                weight1 * sq((referenceColor.c[1] - sampleColor.c[1]) / maxImageValue) +
                weight2 * sq((referenceColor.c[2] - sampleColor.c[2]) / maxImageValue);
            const float weight = expf(-weightedDiff / (sigma * sigma));
            weightedSumPix += images[t][c] * weight; // centre pixel of frame t (TemporalBilateralFilter.h:165)
            sumWeight += weight;
          }
        }
      }
      result.at(y, x) = (weightedSumPix / sumWeight);
    }
  });
  return result;
}

// ---- source/render/BackgroundSubtractionUtil.h:20-60: generateForegroundMask<Vec3w, Vec3f> ----
// (the producer of the masks DerpCLI consumes.) OpenCV-defined steps, restated:
//  * cv::GaussianBlur(ksize 2r+1, sigma 0) on CV_16UC3: sigma <= 0 and ksize <= 7 selects the fixed
//    small kernels {1,2,1}/4, {1,4,6,4,1}/16, {2,7,14,18,14,7,2}/64; OpenCV 4's 16-bit path is fixed
//    point (ufixedpoint32), exact products, rounded once half-up; BORDER_REFLECT_101.
//  * convertTo(CV_32F, 1/65535), cv::absdiff, cv::norm(Vec3f) accumulates squares in double.
//  * cv::morphologyEx(MORPH_CLOSE, rect k x k, anchor (k/2, k/2)): dilate then erode with the SAME
//    (unreflected) element, out-of-image taps ignored (morphologyDefaultBorderValue).
static Img<uint8_t> generateForegroundMask(
    const Img<Px3w>& templ, const Img<Px3w>& frame, int blurRadius, float threshold, int morphSize) {
  const int w = templ.w, h = templ.h;
  auto blur = [&](const Img<Px3w>& in) {
    if (blurRadius <= 0) {
      return in;
    }
    static const int k3[] = {1, 2, 1}, k5[] = {1, 4, 6, 4, 1}, k7[] = {2, 7, 14, 18, 14, 7, 2};
    const int* k = blurRadius == 1 ? k3 : blurRadius == 2 ? k5 : k7;
    const int shift1 = blurRadius == 1 ? 2 : blurRadius == 2 ? 4 : 6;
    Img<Px3w> out(w, h);
    for (int y = 0; y < h; ++y) {
      for (int x = 0; x < w; ++x) {
        for (int c = 0; c < 3; ++c) {
          uint64_t acc = 0;
          for (int j = -blurRadius; j <= blurRadius; ++j) {
            const int yy = reflect101(y + j, h);
            uint64_t row = 0;
            for (int i = -blurRadius; i <= blurRadius; ++i) {
              row += (uint64_t)k[i + blurRadius] * in.at(yy, reflect101(x + i, w)).c[c];
            }
            acc += (uint64_t)k[j + blurRadius] * row;
          }
          const int sh = 2 * shift1;
          out.at(y, x).c[c] = (uint16_t)std::min<uint64_t>(65535, (acc + (1ull << (sh - 1))) >> sh);
        }
      }
    }
    return out;
  };
  const Img<Px3w> tb = blur(templ), fb = blur(frame);
  Img<uint8_t> mask(w, h, 0);
  const float s = 1.0f / 65535.0f;
  for (size_t i = 0; i < mask.d.size(); ++i) {
    double acc = 0;
    for (int c = 0; c < 3; ++c) {
      const float d = std::abs(tb.d[i].c[c] * s - fb.d[i].c[c] * s);
      acc += (double)d * d;
    }
    mask.d[i] = std::sqrt(acc) > threshold;
  }
  if (morphSize > 0) {
    const int a = morphSize / 2;
    auto pass = [&](const Img<uint8_t>& in, bool dilate) {
      Img<uint8_t> out(w, h);
      for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
          uint8_t v = dilate ? 0 : 1;
          for (int j = 0; j < morphSize; ++j) {
            for (int i = 0; i < morphSize; ++i) {
              const int yy = y + j - a, xx = x + i - a;
              if (yy < 0 || yy >= h || xx < 0 || xx >= w) {
                continue;
              }
              v = dilate ? std::max(v, in.at(yy, xx)) : std::min(v, in.at(yy, xx));
            }
          }
          out.at(y, x) = v;
        }
      }
      return out;
    };
    mask = pass(pass(mask, true), false);
  }
  return mask;
}

// ---- source/render/RephotographyUtil.h:38-116 + ComputeRephotographyErrors.cpp:69-189 ----
// (the reference's quality gate for DerpCLI.) The score arithmetic (computeSSIM,
// averageScore, formatResults) is restated operation for operation. OpenCV-defined steps:
//  * cv::GaussianBlur(ksize 2r+1, sigma 1.5) on CV_32FC3: getGaussianKernel(n, sigma, CV_32F) of
//    OpenCV 4 = t_i = exp(-0.125 / sigma^2 * (2i - n + 1)^2) summed from the outside in, doubled,
//    plus the centre 1, kernel = t_i * (1 / sum) rounded to float; separable row then column pass
//    in float, symmetric form k0 * x0 + k1 * (x-1 + x1) + ..., BORDER_REFLECT_101. Whether OpenCV's
//    SIMD build contracts those into FMAs is build dependent: this restatement does not (parity
//    unpinned, like every OpenCV call site — SURVEY §8c).
//  * MatExpr arithmetic in float: (2 * A + c) rounds once; 1.0f / M is an IEEE division; cv::pow with
//    exponent 1 copies and with exponent 0 yields 1; cv::sqrt is IEEE.
static std::vector<float> gaussianKernel32f(int radius, double sigma) {
  const int n = 2 * radius + 1;
  std::vector<double> t(radius);
  const double scale2X = -0.125 / (sigma * sigma);
  double sum = 0;
  for (int i = 0, x = 1 - n; i < radius; ++i, x += 2) {
    t[i] = std::exp((double)(x * x) * scale2X);
    sum += t[i];
  }
  sum *= 2;
  sum += 1;
  const double mul = 1.0 / sum;
  std::vector<float> k(n);
  for (int i = 0; i < radius; ++i) {
    k[i] = k[n - 1 - i] = (float)(t[i] * mul);
  }
  k[radius] = (float)mul;
  return k;
}

// 3-channel interleaved float image blur (rephoto_util::blur)
static std::vector<float> gaussianBlur32f(const std::vector<float>& in, int w, int h, int radius) {
  const std::vector<float> k = gaussianKernel32f(radius, 1.5f);
  std::vector<float> tmp(in.size()), out(in.size());
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      for (int c = 0; c < 3; ++c) {
        float s = in[((size_t)y * w + x) * 3 + c] * k[radius];
        for (int i = 1; i <= radius; ++i) {
          const float a = in[((size_t)y * w + reflect101(x - i, w)) * 3 + c];
          const float b = in[((size_t)y * w + reflect101(x + i, w)) * 3 + c];
          s += (a + b) * k[radius + i];
        }
        tmp[((size_t)y * w + x) * 3 + c] = s;
      }
    }
  }
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      for (int c = 0; c < 3; ++c) {
        float s = tmp[((size_t)y * w + x) * 3 + c] * k[radius];
        for (int i = 1; i <= radius; ++i) {
          const float a = tmp[((size_t)reflect101(y - i, h) * w + x) * 3 + c];
          const float b = tmp[((size_t)reflect101(y + i, h) * w + x) * 3 + c];
          s += (a + b) * k[radius + i];
        }
        out[((size_t)y * w + x) * 3 + c] = s;
      }
    }
  }
  return out;
}

// RephotographyUtil.h:38-86 (alpha, beta, gamma in {0, 1}: MSSIM = 1,1,1; NCC = 0,0,1)
static std::vector<float> computeSSIM(
    const std::vector<float>& x, const std::vector<float>& y, int w, int h, int blurRadius, float alpha, float beta,
    float gamma) {
  const size_t n = x.size();
  const std::vector<float> muX = gaussianBlur32f(x, w, h, blurRadius), muY = gaussianBlur32f(y, w, h, blurRadius);
  std::vector<float> a(n), b(n), c(n);
  for (size_t i = 0; i < n; ++i) {
    const float dx = x[i] - muX[i], dy = y[i] - muY[i];
    a[i] = dx * dx;
    b[i] = dy * dy;
    c[i] = dx * dy;
  }
  const std::vector<float> sig2X = gaussianBlur32f(a, w, h, blurRadius), sig2Y = gaussianBlur32f(b, w, h, blurRadius),
                           sigXY = gaussianBlur32f(c, w, h, blurRadius);
  const float c1 = 0.0001f, c2 = 0.0009f, c3 = (float)((double)0.0009f / 2.0);
  auto ipow = [](float v, float e) { return e == 0.0f ? 1.0f : e == 1.0f ? v : std::pow(v, e); };
  std::vector<float> out(n);
  for (size_t i = 0; i < n; ++i) {
    const float mu2X = muX[i] * muX[i], mu2Y = muY[i] * muY[i], muXY = muX[i] * muY[i];
    const float sigX = std::sqrt(sig2X[i]), sigY = std::sqrt(sig2Y[i]);
    const float sxy = sigX * sigY;
    const float luminance = ipow((2 * muXY + c1) * (1.0f / (mu2X + mu2Y + c1)), alpha);
    const float contrast = ipow((2 * sxy + c2) * (1.0f / (sig2X[i] + sig2Y[i] + c2)), beta);
    const float structure = ipow((sigXY[i] + c3) * (1.0f / (sxy + c3)), gamma);
    out[i] = contrast * luminance * structure;
  }
  return out;
}

// RephotographyUtil.h:88-108: per channel mean over mask != 0 and finite-or-inf (NaN excluded), in double
static void averageScore(const std::vector<float>& score, const uint8_t* mask, size_t npx, double out[3]) {
  for (int c = 0; c < 3; ++c) {
    double sum = 0;
    size_t cnt = 0;
    for (size_t i = 0; i < npx; ++i) {
      const float v = score[i * 3 + c];
      if (mask[i] && !std::isnan(v)) {
        sum += v;
        ++cnt;
      }
    }
    out[c] = cnt ? sum / (double)cnt : 0.0;
  }
}

// Camera-space rephotography: what the other cameras' colour + disparity say camera `target` sees.
// The reference renders both sides as OpenGL cubemaps of disparity meshes centred on the target
// camera (CanopyScene; out of scope). Here, pass 1: every valid pixel of every other camera becomes a
// point at 1/disparity along its ray (dstToWorldPoint), is projected with Camera::sees into the
// target image and its distance to the target position (float) is written to the 2x2 pixels whose
// centres surround it; the smallest (distance, camera, pixel) key wins. Pass 2: each covered target
// pixel's own ray is walked to that distance, the point is projected into the winning camera and its
// colour fetched with getPixelBilinear. Output BGRA float in [0, 1], alpha = covered.
static void rephotograph(
    const Rig& rig, int target, const uint16_t* const* colors, const float* const* disps, int w, int h, float* outBgra) {
  const size_t n = (size_t)w * h;
  std::vector<uint64_t> key(n, ~0ull);
  const Camera& camT = rig[target];
  for (int j = 0; j < (int)rig.size(); ++j) {
    if (j == target) {
      continue;
    }
    for (int y = 0; y < h; ++y) {
      for (int x = 0; x < w; ++x) {
        const float d = disps[j][(size_t)y * w + x];
        if (!(d > 0) || std::isinf(d)) {
          continue;
        }
        const V3 p = dstToWorldPoint(rig[j], x, y, d, w, h);
        V2 pix;
        if (!worldToSrcPoint(pix, p, camT, w, h)) {
          continue;
        }
        const double dx = p.x - camT.position.x, dy = p.y - camT.position.y, dz = p.z - camT.position.z;
        const float dist = (float)std::sqrt(dx * dx + (dy * dy + dz * dz));
        uint32_t bits;
        std::memcpy(&bits, &dist, 4);
        const uint64_t k = ((uint64_t)bits << 32) | ((uint64_t)j << 24) | (uint64_t)((size_t)y * w + x);
        const int x0 = (int)std::floor(pix.x - 0.5), y0 = (int)std::floor(pix.y - 0.5);
        for (int yy = y0; yy <= y0 + 1; ++yy) {
          for (int xx = x0; xx <= x0 + 1; ++xx) {
            if (xx >= 0 && yy >= 0 && xx < w && yy < h) {
              uint64_t& slot = key[(size_t)yy * w + xx];
              slot = std::min(slot, k);
            }
          }
        }
      }
    }
  }
  const float s = 1.0f / 65535.0f;
  std::vector<Img<Px3w>> imgs;
  for (int j = 0; j < (int)rig.size(); ++j) {
    Img<Px3w> im(w, h);
    std::memcpy((void*)im.d.data(), colors[j], n * 6);
    imgs.push_back(std::move(im));
  }
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const size_t i = (size_t)y * w + x;
      float* o = outBgra + i * 4;
      o[0] = o[1] = o[2] = o[3] = 0.0f;
      if (key[i] == ~0ull) {
        continue;
      }
      const int j = (int)((key[i] >> 24) & 0xff);
      const uint32_t bits = (uint32_t)(key[i] >> 32);
      float dist;
      std::memcpy(&dist, &bits, 4);
      V2 p = {(x + 0.5) / w, (y + 0.5) / h};
      if (!camT.isNormalized()) {
        p = {p.x * camT.resolution.x, p.y * camT.resolution.y};
      }
      const V3 pWorld = camT.rig(p, (double)dist);
      V2 ps;
      if (!worldToSrcPoint(ps, pWorld, rig[j], w, h)) {
        continue;
      }
      const Px3w c = getPixelBilinear(imgs[j], float(ps.x), float(ps.y));
      for (int k = 0; k < 3; ++k) {
        o[k] = c.c[k] * s;
      }
      o[3] = 1.0f;
    }
  }
}

} // namespace oracle

// =====================================================================
// C interface (ctypes). All images row-major; colour = interleaved BGR u16.
// =====================================================================
using namespace oracle;

extern "C" {

struct OracleRig {
  Rig cams;
};

OracleRig* oracle_rig_create(const CameraJson* cams, int n) {
  OracleRig* r = new OracleRig;
  for (int i = 0; i < n; ++i) {
    r->cams.emplace_back(cams[i]);
  }
  return r;
}
void oracle_rig_destroy(OracleRig* r) {
  delete r;
}
int oracle_rig_size(const OracleRig* r) {
  return (int)r->cams.size();
}
int oracle_cam_valid(const OracleRig* r, int i) {
  return r->cams[i].valid;
}
void oracle_rig_normalize(OracleRig* r) { // Camera.cpp:236-242
  for (Camera& c : r->cams) {
    if (!c.isNormalized()) {
      c.normalize();
    }
  }
}
void oracle_cam_rescale(OracleRig* r, int i, double w, double h) {
  r->cams[i] = r->cams[i].rescale({w, h});
}
// state dump: position[3], R[9], resolution[2], principal[2], focal[2], dist[3], distMax, cosFov
void oracle_cam_get(const OracleRig* r, int i, double* out23) {
  const Camera& c = r->cams[i];
  double* o = out23;
  *o++ = c.position.x;
  *o++ = c.position.y;
  *o++ = c.position.z;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) {
      *o++ = c.R[a][b];
    }
  }
  *o++ = c.resolution.x;
  *o++ = c.resolution.y;
  *o++ = c.principal.x;
  *o++ = c.principal.y;
  *o++ = c.focal.x;
  *o++ = c.focal.y;
  *o++ = c.dist[0];
  *o++ = c.dist[1];
  *o++ = c.dist[2];
  *o++ = c.distMax;
  *o++ = c.cosFov;
}
void oracle_cam_set_fov(OracleRig* r, int i, double fov, int setDefault) {
  if (setDefault) {
    r->cams[i].setDefaultFov();
  } else {
    r->cams[i].setFov(fov);
  }
}
double oracle_cam_get_fov(const OracleRig* r, int i) {
  return r->cams[i].getFov();
}
void oracle_cam_set_distortion(OracleRig* r, int i, const double* d3, int setDefault) {
  if (setDefault) {
    r->cams[i].setDefaultDistortion();
  } else {
    r->cams[i].setDistortion(d3);
  }
}
void oracle_cam_pixel(const OracleRig* r, int i, const double* xyz, int n, double* out) {
  for (int k = 0; k < n; ++k) {
    const V2 p = r->cams[i].pixel({xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]});
    out[2 * k] = p.x;
    out[2 * k + 1] = p.y;
  }
}
void oracle_cam_rig(const OracleRig* r, int i, const double* pix, const double* depth, int n, double* out) {
  for (int k = 0; k < n; ++k) {
    const V3 p = r->cams[i].rig({pix[2 * k], pix[2 * k + 1]}, depth[k]);
    out[3 * k] = p.x;
    out[3 * k + 1] = p.y;
    out[3 * k + 2] = p.z;
  }
}
void oracle_cam_sees(const OracleRig* r, int i, const double* xyz, int n, uint8_t* sees, double* pix) {
  for (int k = 0; k < n; ++k) {
    V2 p = {NAN, NAN};
    sees[k] = r->cams[i].sees({xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]}, p);
    pix[2 * k] = p.x;
    pix[2 * k + 1] = p.y;
  }
}
int oracle_cam_is_outside_image_circle(const OracleRig* r, int i, double px, double py) {
  return r->cams[i].isOutsideImageCircle({px, py});
}
int oracle_cam_is_outside_sensor(const OracleRig* r, int i, double px, double py) {
  return r->cams[i].isOutsideSensor({px, py});
}
int oracle_cam_is_behind(const OracleRig* r, int i, double x, double y, double z) {
  return r->cams[i].isBehind({x, y, z});
}
double oracle_cam_distort(const OracleRig* r, int i, double v) {
  return r->cams[i].distort(v);
}
double oracle_cam_undistort(const OracleRig* r, int i, double v) {
  return r->cams[i].undistort(v);
}

// ---- level ----
Level* oracle_level_create(
    const OracleRig* rigSrc, const OracleRig* rigDst, const int* dst2src, const Params* p) {
  Level* L = new Level;
  L->p = *p;
  L->rigSrc = rigSrc->cams;
  L->rigDst = rigDst->cams;
  L->S = (int)L->rigSrc.size();
  L->D = (int)L->rigDst.size();
  L->dst2src.assign(dst2src, dst2src + L->D);
  L->W = p->width;
  L->H = p->height;
  L->hasFg = p->useFgMasks != 0;
  // PyramidLevel.h:232-236 (note: width / heightFullSize — reference quirk kept)
  const float scale = float(L->W) / p->heightFull;
  const float scaleVar = scale * scale;
  L->varNoiseFloor = std::max(p->varNoiseFloorFull * scaleVar, kMinVar);
  L->varHighThresh = p->varHighThresh;
  L->srcColor.resize(L->S);
  L->srcVariance.resize(L->S);
  L->srcFg.assign(L->S, Img<uint8_t>(L->W, L->H, 1));
  L->disparity.assign(L->D, Img<float>(L->W, L->H, 0.f));
  L->cost.assign(L->D, Img<float>(L->W, L->H, 0.f));
  L->confidence.assign(L->D, Img<float>(L->W, L->H, 0.f));
  L->mismatchMask.assign(L->D, Img<uint8_t>(L->W, L->H, 0));
  L->bgDisp.assign(L->D, Img<float>());
  L->fovMask.resize(L->D);
  parallelFor(0, L->D, p->threads, [&](int d) {
    L->fovMask[d] = generateFovMask(L->rigDst[d], L->W, L->H);
  });
  return L;
}
void oracle_level_destroy(Level* L) {
  delete L;
}
void oracle_level_set_src(Level* L, int s, const uint16_t* bgr, const uint8_t* fg) {
  L->srcColor[s] = Img<Px3w>(L->W, L->H);
  memcpy(L->srcColor[s].d.data(), bgr, size_t(L->W) * L->H * 6);
  if (fg) {
    memcpy(L->srcFg[s].d.data(), fg, size_t(L->W) * L->H);
  }
  L->srcVariance[s] = computeImageVariance(L->srcColor[s]);
}
void oracle_level_set_dst(Level* L, int d, const float* disparity, const float* bg) {
  if (disparity) {
    memcpy(L->disparity[d].d.data(), disparity, size_t(L->W) * L->H * 4);
  }
  if (bg) {
    L->bgDisp[d] = Img<float>(L->W, L->H);
    memcpy(L->bgDisp[d].d.data(), bg, size_t(L->W) * L->H * 4);
  }
}
void oracle_level_precompute_projections(Level* L) {
  precomputeProjections(*L);
}
void oracle_level_reproject_colors(Level* L) {
  reprojectColors(*L);
}
void oracle_level_brute_force(Level* L) { // Derp.cpp:826-842, 384-401
  if (L->p.level == L->p.numLevels - 1) {
    for (int d = 0; d < L->D; ++d) {
      computeBruteForceDisparity(*L, d);
    }
  }
}
void oracle_level_random_proposals(Level* L) {
  randomProposals(*L);
}
void oracle_level_ping_pong(Level* L) {
  pingPong(*L, nullptr);
}
void oracle_level_bilateral(Level* L) {
  bilateralFilter(*L);
}
void oracle_level_median(Level* L) {
  medianFilter(*L);
}
void oracle_level_mask_fov(Level* L) {
  maskFov(*L);
}
void oracle_level_mismatches(Level* L) {
  handleDisparityMismatches(*L);
}
void oracle_level_get_mismatch_mask(Level* L, int d, uint8_t* out) {
  memcpy(out, L->mismatchMask[d].d.data(), size_t(L->W) * L->H);
}
// LayerDisparities.cpp:45-55: mask = fg > 0; layer = fg*mask + bg*(1-mask); imwrite(layer*255) -> 8-bit
void oracle_layer_disparities(const float* fg, const float* bg, size_t n, uint8_t* out) {
  for (size_t i = 0; i < n; ++i) {
    const float mask = fg[i] > 0.0f ? 1.0f : 0.0f;
    const float layer = fg[i] * mask + bg[i] * (1 - mask);
    const int r = cvRoundF(layer * 255.0f);
    out[i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
  }
}
// Derp.cpp:1005-1034
void oracle_level_process(Level* L) {
  reprojectColors(*L);
  oracle_level_brute_force(L);
  randomProposals(*L);
  pingPong(*L, nullptr);
  handleDisparityMismatches(*L);
  if (L->p.doBilateral) {
    bilateralFilter(*L);
  }
  if (L->p.doMedian) {
    medianFilter(*L);
  }
  maskFov(*L);
}
// cost of a caller-supplied disparity map at every interior pixel (test hook over computeCost)
void oracle_level_cost_map(Level* L, int d, const float* disp, float* cost, float* conf) {
  parallelFor(1, L->H - 1, L->p.threads, [&](int y) {
    Counters cnt;
    for (int x = 1; x < L->W - 1; ++x) {
      const auto r = computeCost(*L, d, disp[size_t(y) * L->W + x], x, y, cnt);
      cost[size_t(y) * L->W + x] = r.first;
      conf[size_t(y) * L->W + x] = r.second;
    }
    L->flush(cnt);
  });
}
void oracle_level_get_dst(Level* L, int d, float* disparity, float* cost, float* confidence) {
  const size_t n = size_t(L->W) * L->H * 4;
  if (disparity) {
    memcpy(disparity, L->disparity[d].d.data(), n);
  }
  if (cost) {
    memcpy(cost, L->cost[d].d.data(), n);
  }
  if (confidence) {
    memcpy(confidence, L->confidence[d].d.data(), n);
  }
}
void oracle_level_get_fov_mask(Level* L, int d, uint8_t* out) {
  memcpy(out, L->fovMask[d].d.data(), size_t(L->W) * L->H);
}
void oracle_level_get_variance(Level* L, int s, float* out) {
  memcpy(out, L->srcVariance[s].d.data(), size_t(L->W) * L->H * 4);
}
// which: 0 projWarp (float2), 1 projWarpInv (float2), 2 projColor (u16x3), 3 projColorBias (u16x3)
void oracle_level_get_proj(Level* L, int d, int s, int which, void* out) {
  const int i = L->idx(d, s);
  const size_t px = size_t(L->W) * L->H;
  switch (which) {
    case 0:
      memcpy(out, L->projWarp[i].d.data(), px * 8);
      break;
    case 1:
      memcpy(out, L->projWarpInv[i].d.data(), px * 8);
      break;
    case 2:
      memcpy(out, L->projColor[i].d.data(), px * 6);
      break;
    case 3:
      memcpy(out, L->projColorBias[i].d.data(), px * 6);
      break;
  }
}
void oracle_level_get_counters(Level* L, uint64_t* nCost, uint64_t* nPair, int* insufficient, int* checkFailed) {
  *nCost = L->nCost;
  *nPair = L->nPair;
  *insufficient = L->insufficientCoverage;
  *checkFailed = L->coverageCheckFailed;
}
float oracle_level_var_noise_floor(Level* L) {
  return L->varNoiseFloor;
}

// ---- upsample (UpsampleDisparityLib.cpp:149-182): rig must be normalised ----
void oracle_upsample_disparity(
    const OracleRig* rigDst,
    int d,
    const float* disp,
    int w,
    int h,
    const float* bgDispUp,
    const uint8_t* fgMask,
    const uint8_t* fgMaskUp,
    int wUp,
    int hUp,
    int useFg,
    float* out) {
  Img<float> in(w, h);
  memcpy(in.d.data(), disp, size_t(w) * h * 4);
  Img<float> bg;
  Img<uint8_t> m, mUp;
  if (useFg) {
    bg = Img<float>(wUp, hUp);
    memcpy(bg.d.data(), bgDispUp, size_t(wUp) * hUp * 4);
    m = generateFovMask(rigDst->cams[d], w, h);
    mUp = generateFovMask(rigDst->cams[d], wUp, hUp);
    for (size_t i = 0; i < m.d.size(); ++i) {
      m.d[i] &= fgMask[i];
    }
    for (size_t i = 0; i < mUp.d.size(); ++i) {
      mUp.d[i] &= fgMaskUp[i];
    }
  }
  const Img<float> up = upsampleDisparity(in, bg, m, mUp, wUp, hUp, useFg != 0);
  memcpy(out, up.d.data(), size_t(wUp) * hUp * 4);
}

// ---- standalone filters ----
void oracle_joint_bilateral_u16(
    const float* image, const uint16_t* guide, const uint8_t* mask, int w, int h, int radius, float sigma,
    float w0, float w1, float w2, int threads, float* out) {
  Img<float> im(w, h);
  memcpy(im.d.data(), image, size_t(w) * h * 4);
  Img<Px3w> g(w, h);
  memcpy(g.d.data(), guide, size_t(w) * h * 6);
  Img<uint8_t> m(w, h);
  memcpy(m.d.data(), mask, size_t(w) * h);
  const Img<float> r = generalizedJointBilateralFilterU16(im, g, g, m, radius, sigma, w0, w1, w2, threads);
  memcpy(out, r.d.data(), size_t(w) * h * 4);
}
void oracle_joint_bilateral_f32(
    const float* image, const float* guide, const uint8_t* mask, int w, int h, int radius, float sigma,
    float w0, float w1, float w2, int threads, float* out) {
  Img<float> im(w, h);
  memcpy(im.d.data(), image, size_t(w) * h * 4);
  Img<Px3f> g(w, h);
  memcpy(g.d.data(), guide, size_t(w) * h * 12);
  Img<uint8_t> m(w, h);
  memcpy(m.d.data(), mask, size_t(w) * h);
  const Img<float> r = generalizedJointBilateralFilterF32(im, g, m, radius, sigma, w0, w1, w2, threads);
  memcpy(out, r.d.data(), size_t(w) * h * 4);
}
void oracle_masked_median(
    const float* image, const float* background, const uint8_t* mask, int w, int h, int radius, float* out) {
  Img<float> im(w, h), bg;
  memcpy(im.d.data(), image, size_t(w) * h * 4);
  if (background) {
    bg = Img<float>(w, h);
    memcpy(bg.d.data(), background, size_t(w) * h * 4);
  }
  Img<uint8_t> m(w, h);
  memcpy(m.d.data(), mask, size_t(w) * h);
  const Img<float> r = maskedMedianBlur(im, bg, m, radius);
  memcpy(out, r.d.data(), size_t(w) * h * 4);
}
// TemporalBilateralFilter.cpp:121-184 core: n frames of (guide, disparity, mask=fg&fov)
void oracle_temporal_filter(
    const uint16_t* const* guides, const float* const* images, const uint8_t* const* masks, int n, int w, int h,
    int frameOffset, float sigma, int spatialRadius, float w0, float w1, float w2, int threads, float* out) {
  std::vector<const Px3w*> g(n);
  std::vector<const float*> im(n);
  std::vector<const uint8_t*> m(n);
  for (int i = 0; i < n; ++i) {
    g[i] = reinterpret_cast<const Px3w*>(guides[i]);
    im[i] = images[i];
    m[i] = masks[i];
  }
  const Img<float> r =
      temporalJointBilateralFilter(g, im, m, w, h, frameOffset, sigma, spatialRadius, w0, w1, w2, threads);
  memcpy(out, r.d.data(), size_t(w) * h * 4);
}
int oracle_temporal_space_radius(int level) { // TemporalBilateralFilter.cpp:165-168
  const float scale = std::pow(kLevelScale, level);
  return std::max(std::ceil(1 * scale), float(1));
}
int oracle_bilateral_radius(int level) {
  return bilateralRadius(level);
}
int oracle_upsample_radius(int w, int wUp) {
  return getUpsampleRadius(w, wUp);
}

// ---- cv primitives exposed for unit tests ----
void oracle_cv_remap_cubic_u16c3(const uint16_t* src, int sw, int sh, const float* map, int dw, int dh, uint16_t* out) {
  Img<Px3w> s(sw, sh);
  memcpy(s.d.data(), src, size_t(sw) * sh * 6);
  Img<Px2f> m(dw, dh);
  memcpy(m.d.data(), map, size_t(dw) * dh * 8);
  Img<Px3w> d;
  remapCubicU16C3(s, m, d);
  memcpy(out, d.d.data(), size_t(dw) * dh * 6);
}
void oracle_cv_blur3_u16c3(const uint16_t* src, int w, int h, uint16_t* out) {
  Img<Px3w> s(w, h), d;
  memcpy(s.d.data(), src, size_t(w) * h * 6);
  blur3x3U16C3(s, d);
  memcpy(out, d.d.data(), size_t(w) * h * 6);
}
void oracle_cv_blur3_f32c3(const float* src, int w, int h, float* out) {
  Img<Px3f> s(w, h), d;
  memcpy(s.d.data(), src, size_t(w) * h * 12);
  blur3x3F32C3(s, d);
  memcpy(out, d.d.data(), size_t(w) * h * 12);
}
void oracle_cv_resize_lanczos4(const float* src, int sw, int sh, int dw, int dh, float* out) {
  Img<float> s(sw, sh), d;
  memcpy(s.d.data(), src, size_t(sw) * sh * 4);
  resizeLanczos4F32(s, dw, dh, d);
  memcpy(out, d.d.data(), size_t(dw) * dh * 4);
}
void oracle_cv_resize_nearest_f32(const float* src, int sw, int sh, int dw, int dh, float* out) {
  Img<float> s(sw, sh), d;
  memcpy(s.d.data(), src, size_t(sw) * sh * 4);
  resizeNearest(s, dw, dh, d);
  memcpy(out, d.d.data(), size_t(dw) * dh * 4);
}
void oracle_cv_variance(const uint16_t* src, int w, int h, float* out) {
  Img<Px3w> s(w, h);
  memcpy(s.d.data(), src, size_t(w) * h * 6);
  const Img<float> v = computeImageVariance(s);
  memcpy(out, v.d.data(), size_t(w) * h * 4);
}
void oracle_generate_foreground_mask(
    const uint16_t* templ, const uint16_t* frame, int w, int h, int blurRadius, float threshold, int morphSize, uint8_t* out) {
  Img<Px3w> t(w, h), f(w, h);
  memcpy(t.d.data(), templ, size_t(w) * h * 6);
  memcpy(f.d.data(), frame, size_t(w) * h * 6);
  const Img<uint8_t> m = generateForegroundMask(t, f, blurRadius, threshold, morphSize);
  memcpy(out, m.d.data(), size_t(w) * h);
}
// pyramid builder: scripts/render/resize.py:51-85 — cv2.resize(full frame, (w, h), INTER_AREA) per level
void oracle_cv_resize_area_u16c3(const uint16_t* src, int sw, int sh, int dw, int dh, uint16_t* out) {
  resizeAreaCv<uint16_t, 3>(src, sw, sh, out, dw, dh);
}
void oracle_cv_resize_area_u8(const uint8_t* src, int sw, int sh, int dw, int dh, uint8_t* out) {
  resizeAreaCv<uint8_t, 1>(src, sw, sh, out, dw, dh);
}
void oracle_cv_resize_area_f32(const float* src, int sw, int sh, int dw, int dh, float* out) {
  resizeAreaCv<float, 1>(src, sw, sh, out, dw, dh);
}
// cv_util::resizeImage<cv::Vec3f> (CvUtil.h:139-147): the colour guide of UpsampleDisparity.cpp:117
void oracle_cv_resize_area_f32c3(const float* src, int sw, int sh, int dw, int dh, float* out) {
  resizeAreaCv<float, 3>(src, sw, sh, out, dw, dh);
}
// libstdc++ behaviours the random-proposal stage depends on (Derp.cpp:757-758,806-808)
void oracle_minstd_uniform(int seed, int n, float a, float b, float* out) {
  std::default_random_engine engine;
  engine.seed(seed);
  for (int i = 0; i < n; ++i) {
    out[i] = std::uniform_real_distribution<float>(a, b)(engine);
  }
}
// std::nth_element on pair<float,float> (Derp.cpp:210): permuted array back to the caller
void oracle_nth_element_pairs(float* pairs, int n, int nth) {
  std::pair<float, float>* p = reinterpret_cast<std::pair<float, float>*>(pairs);
  std::nth_element(p, p + nth, p + n);
}

// ---- rephotography score (RephotographyUtil.h) ----
void oracle_gaussian_blur_f32c3(const float* src, int w, int h, int radius, float* out) {
  const std::vector<float> in(src, src + (size_t)w * h * 3);
  const std::vector<float> r = gaussianBlur32f(in, w, h, radius);
  std::memcpy(out, r.data(), r.size() * 4);
}
void oracle_compute_ssim(
    const float* x, const float* y, int w, int h, int blurRadius, float alpha, float beta, float gamma, float* out) {
  const size_t n = (size_t)w * h * 3;
  const std::vector<float> r =
      computeSSIM(std::vector<float>(x, x + n), std::vector<float>(y, y + n), w, h, blurRadius, alpha, beta, gamma);
  std::memcpy(out, r.data(), n * 4);
}
void oracle_average_score(const float* score, const uint8_t* mask, int w, int h, double* out3) {
  const size_t n = (size_t)w * h;
  averageScore(std::vector<float>(score, score + n * 3), mask, n, out3);
}
// CanopyScene::cubemap (CanopyScene.cpp:198-374) of the cameras include[s] != 0 seen from `centre`
void oracle_canopy_cubemap(const OracleRig* r, const uint16_t* const* colors, const float* const* disps, int w, int h,
                           const uint8_t* include, const double* centre, int edge, float* outBgra) {
  canopyCubemap(r->cams, colors, disps, w, h, include, centre, edge, outBgra);
}
void oracle_rephotograph(
    const OracleRig* r, int target, const uint16_t* const* colors, const float* const* disps, int w, int h, float* outBgra) {
  rephotograph(r->cams, target, colors, disps, w, h, outBgra);
}

} // extern "C"
