// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restatement of the OpenCV primitives the depth path calls. OpenCV is NOT in
// /root/reference (find_package(OpenCV 4), CMakeLists.txt:67; the Dockerfile
// builds 3.4.3, Dockerfile:120) — so these follow OpenCV's published generic
// (non-IPP, non-AVX-dispatched, scalar-order) algorithms. PARITY UNPINNED at
// these call sites: no reference test in the tree pins their bit-level output.
//
//   remapCubicU16C3   <- cv::remap(INTER_CUBIC, BORDER_CONSTANT)   DerpUtil.cpp:199-205
//   blur3x3U16C3      <- cv::blur 3x3 on CV_16UC3                  DerpUtil.cpp:208-210 -> CvUtil.h:314-323
//   blur3x3F32C3      <- cv::blur 3x3 on CV_32FC3                  DerpUtil.cpp:214-224
//   resizeLanczos4F32 <- cv::resize(INTER_LANCZOS4) on CV_32FC1    UpsampleDisparityLib.cpp:145
//   resizeNearest     <- cv::resize(INTER_NEAREST)                 UpsampleDisparityLib.cpp:125
//   resizeAreaU16C3 / resizeAreaF32 <- cv::resize(INTER_AREA)      scripts/render/resize.py:79, CvUtil.h:139-147
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

namespace oracle {

template <typename T>
struct Img {
  int w = 0, h = 0;
  std::vector<T> d;
  Img() {}
  Img(int w_, int h_) : w(w_), h(h_), d(size_t(w_) * h_) {}
  Img(int w_, int h_, const T& v) : w(w_), h(h_), d(size_t(w_) * h_, v) {}
  T& at(int y, int x) {
    return d[size_t(y) * w + x];
  }
  const T& at(int y, int x) const {
    return d[size_t(y) * w + x];
  }
  bool empty() const {
    return d.empty();
  }
};

struct Px3w {
  uint16_t c[3];
};
struct Px3f {
  float c[3];
};
struct Px2f {
  float c[2];
};

// cvRound(float): SSE cvtss2si, round-half-even, "integer indefinite" on NaN/overflow
static inline int cvRoundF(float v) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) {
    return std::numeric_limits<int>::min();
  }
  return (int)std::nearbyintf(v);
}
static inline int cvRoundD(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) {
    return std::numeric_limits<int>::min();
  }
  return (int)std::nearbyint(v);
}
static inline int cvFloorF(float v) {
  const int i = (int)v;
  return i - (i > v);
}
static inline int cvFloorD(double v) {
  const int i = (int)v;
  return i - (i > v);
}
static inline uint16_t satU16(int v) {
  return (uint16_t)(v < 0 ? 0 : v > 65535 ? 65535 : v);
}
static inline int16_t satS16(int v) {
  return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
}
static inline int reflect101(int p, int len) {
  if (len == 1) {
    return 0;
  }
  while (p < 0 || p >= len) {
    if (p < 0) {
      p = -p;
    } else {
      p = 2 * len - 2 - p;
    }
  }
  return p;
}

// ---- bicubic table: imgwarp.cpp interpolateCubic / initInterTab1D / initInterTab2D ----
static const int kInterBits = 5;
static const int kInterTabSize = 1 << kInterBits;

static inline void interpolateCubic(float x, float* coeffs) {
  const float A = -0.75f;
  coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

struct CubicTab {
  float tab1[kInterTabSize][4];
  CubicTab() {
    const float scale = 1.f / kInterTabSize;
    for (int i = 0; i < kInterTabSize; ++i) {
      interpolateCubic(i * scale, tab1[i]);
    }
  }
};
static inline const CubicTab& cubicTab() {
  static const CubicTab t;
  return t;
}

// cv::remap, CV_16UC3 source, CV_32FC2 map, INTER_CUBIC, BORDER_CONSTANT(0).
// imgwarp.cpp: RemapInvoker (float map -> fixed point, 1/32 px) + remapBicubic<Cast<float,ushort>,float,1>
static inline void
remapCubicU16C3(const Img<Px3w>& src, const Img<Px2f>& map, Img<Px3w>& dst) {
  const CubicTab& T = cubicTab();
  dst = Img<Px3w>(map.w, map.h);
  const int sw = src.w, sh = src.h;
  const unsigned width1 = std::max(sw - 3, 0), height1 = std::max(sh - 3, 0);
  for (int dy = 0; dy < map.h; ++dy) {
    for (int dx = 0; dx < map.w; ++dx) {
      const Px2f m = map.at(dy, dx);
      const int fsx = cvRoundF(m.c[0] * kInterTabSize);
      const int fsy = cvRoundF(m.c[1] * kInterTabSize);
      const int fx = fsx & (kInterTabSize - 1);
      const int fy = fsy & (kInterTabSize - 1);
      const int sx = satS16(fsx >> kInterBits) - 1;
      const int sy = satS16(fsy >> kInterBits) - 1;
      float w[16];
      for (int k1 = 0; k1 < 4; ++k1) {
        const float vy = T.tab1[fy][k1];
        for (int k2 = 0; k2 < 4; ++k2) {
          w[k1 * 4 + k2] = vy * T.tab1[fx][k2];
        }
      }
      Px3w out;
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        for (int k = 0; k < 3; ++k) {
          const uint16_t* S0 = &src.at(sy, sx).c[k];
          const uint16_t* S1 = &src.at(sy + 1, sx).c[k];
          const uint16_t* S2 = &src.at(sy + 2, sx).c[k];
          const uint16_t* S3 = &src.at(sy + 3, sx).c[k];
          float sum = S0[0] * w[0] + S0[3] * w[1] + S0[6] * w[2] + S0[9] * w[3];
          sum += S1[0] * w[4] + S1[3] * w[5] + S1[6] * w[6] + S1[9] * w[7];
          sum += S2[0] * w[8] + S2[3] * w[9] + S2[6] * w[10] + S2[9] * w[11];
          sum += S3[0] * w[12] + S3[3] * w[13] + S3[6] * w[14] + S3[9] * w[15];
          out.c[k] = satU16(cvRoundF(sum));
        }
      } else if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
        out.c[0] = out.c[1] = out.c[2] = 0;
      } else {
        int xs[4], ys[4];
        for (int i = 0; i < 4; ++i) {
          xs[i] = ((unsigned)(sx + i) < (unsigned)sw) ? sx + i : -1;
          ys[i] = ((unsigned)(sy + i) < (unsigned)sh) ? sy + i : -1;
        }
        for (int k = 0; k < 3; ++k) {
          const float cv = 0.f;
          float sum = cv * 1;
          for (int i = 0; i < 4; ++i) {
            if (ys[i] < 0) {
              continue;
            }
            for (int j = 0; j < 4; ++j) {
              if (xs[j] >= 0) {
                sum += (src.at(ys[i], xs[j]).c[k] - cv) * w[i * 4 + j];
              }
            }
          }
          out.c[k] = satU16(cvRoundF(sum));
        }
      }
      dst.at(dy, dx) = out;
    }
  }
}

// cv::blur 3x3 CV_16UC3, BORDER_REFLECT_101. box_filter: RowSum<ushort,int>,
// ColumnSum<int,ushort> with scale 1/9 -> round-to-nearest of sum/9 (never a tie).
static inline void blur3x3U16C3(const Img<Px3w>& src, Img<Px3w>& dst) {
  dst = Img<Px3w>(src.w, src.h);
  for (int y = 0; y < src.h; ++y) {
    const int ys[3] = {reflect101(y - 1, src.h), y, reflect101(y + 1, src.h)};
    for (int x = 0; x < src.w; ++x) {
      const int xs[3] = {reflect101(x - 1, src.w), x, reflect101(x + 1, src.w)};
      for (int k = 0; k < 3; ++k) {
        int s = 0;
        for (int j = 0; j < 3; ++j) {
          for (int i = 0; i < 3; ++i) {
            s += src.at(ys[j], xs[i]).c[k];
          }
        }
        dst.at(y, x).c[k] = satU16(cvRoundD(s * (1. / 9)));
      }
    }
  }
}

// cv::blur 3x3 CV_32FC3, BORDER_REFLECT_101. RowSum<float,double> (ksize==3 special
// case: S[i]+S[i+cn]+S[i+2cn] left to right), ColumnSum<double,float>: running SUM
// over rows (add newest, emit (float)(s0 * 1/9), subtract oldest).
static inline void blur3x3F32C3(const Img<Px3f>& src, Img<Px3f>& dst) {
  dst = Img<Px3f>(src.w, src.h);
  const int w = src.w, h = src.h;
  auto rowSum = [&](int yy, std::vector<double>& out) {
    const int y = reflect101(yy, h);
    for (int x = 0; x < w; ++x) {
      const int x0 = reflect101(x - 1, w), x2 = reflect101(x + 1, w);
      for (int k = 0; k < 3; ++k) {
        out[x * 3 + k] =
            (double)src.at(y, x0).c[k] + (double)src.at(y, x).c[k] + (double)src.at(y, x2).c[k];
      }
    }
  };
  std::vector<double> rows[3];
  for (auto& r : rows) {
    r.resize(size_t(w) * 3);
  }
  std::vector<double> SUM(size_t(w) * 3, 0.0);
  // prime with extended rows -1 and 0
  rowSum(-1, rows[0]);
  rowSum(0, rows[1]);
  for (int i = 0; i < w * 3; ++i) {
    SUM[i] += rows[0][i];
  }
  for (int i = 0; i < w * 3; ++i) {
    SUM[i] += rows[1][i];
  }
  const double scale = 1. / 9;
  for (int y = 0; y < h; ++y) {
    // ring: rows[(y)%3] = extended row y-1 (oldest), rows[(y+1)%3] = row y, rows[(y+2)%3] = row y+1 (newest)
    std::vector<double>& Sp = rows[(y + 2) % 3];
    const std::vector<double>& Sm = rows[y % 3];
    rowSum(y + 1, Sp);
    for (int x = 0; x < w; ++x) {
      for (int k = 0; k < 3; ++k) {
        const int i = x * 3 + k;
        const double s0 = SUM[i] + Sp[i];
        dst.at(y, x).c[k] = (float)(s0 * scale);
        SUM[i] = s0 - Sm[i];
      }
    }
  }
}

// ---- cv::resize INTER_LANCZOS4, CV_32FC1 (resize.cpp: resizeGeneric_, HResizeLanczos4,
// VResizeLanczos4 scalar order, replicate border) ----
static inline void interpolateLanczos4(float x, float* coeffs) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[][2] = {
      {1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < std::numeric_limits<float>::epsilon()) {
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

static inline void lanczosAxis(int ssize, int dsize, std::vector<int>& ofs, std::vector<float>& coef) {
  const double scale = (double)ssize / dsize;
  ofs.resize(dsize);
  coef.resize(size_t(dsize) * 8);
  for (int d = 0; d < dsize; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    const int s = cvFloorF(f);
    f -= s;
    ofs[d] = s;
    interpolateLanczos4(f, &coef[size_t(d) * 8]);
  }
}

static inline void resizeLanczos4F32(const Img<float>& src, int dw, int dh, Img<float>& dst) {
  if (src.w == dw && src.h == dh) {
    dst = src;
    return;
  }
  std::vector<int> xofs, yofs;
  std::vector<float> alpha, beta;
  lanczosAxis(src.w, dw, xofs, alpha);
  lanczosAxis(src.h, dh, yofs, beta);
  // horizontal pass on every source row
  Img<float> hbuf(dw, src.h);
  for (int y = 0; y < src.h; ++y) {
    const float* S = &src.d[size_t(y) * src.w];
    for (int dx = 0; dx < dw; ++dx) {
      const float* a = &alpha[size_t(dx) * 8];
      const int sx = xofs[dx] - 3;
      float v = 0;
      for (int j = 0; j < 8; ++j) {
        int sxj = sx + j;
        sxj = sxj < 0 ? 0 : sxj >= src.w ? src.w - 1 : sxj;
        v += S[sxj] * a[j];
      }
      hbuf.at(y, dx) = v;
    }
  }
  dst = Img<float>(dw, dh);
  for (int dy = 0; dy < dh; ++dy) {
    const float* b = &beta[size_t(dy) * 8];
    const float* rows[8];
    for (int k = 0; k < 8; ++k) {
      int sy = yofs[dy] - 3 + k;
      sy = sy < 0 ? 0 : sy >= src.h ? src.h - 1 : sy;
      rows[k] = &hbuf.d[size_t(sy) * dw];
    }
    for (int x = 0; x < dw; ++x) {
      dst.at(dy, x) = rows[0][x] * b[0] + rows[1][x] * b[1] + rows[2][x] * b[2] + rows[3][x] * b[3] +
          rows[4][x] * b[4] + rows[5][x] * b[5] + rows[6][x] * b[6] + rows[7][x] * b[7];
    }
  }
}

// cv::resize INTER_NEAREST (resizeNN): sx = min(floor(x * ssize/dsize), ssize-1)
template <typename T>
static inline void resizeNearest(const Img<T>& src, int dw, int dh, Img<T>& dst) {
  dst = Img<T>(dw, dh);
  const double ifx = (double)src.w / dw, ify = (double)src.h / dh;
  for (int y = 0; y < dh; ++y) {
    const int sy = std::min(cvFloorD(y * ify), src.h - 1);
    for (int x = 0; x < dw; ++x) {
      const int sx = std::min(cvFloorD(x * ifx), src.w - 1);
      dst.at(y, x) = src.at(sy, sx);
    }
  }
}

// cv::resize INTER_AREA (resize.cpp computeResizeAreaTab + ResizeArea_Invoker), float
// accumulation. Used for the pyramid builder (integer and fractional down-scales).
struct AreaTabEntry {
  int di, si;
  float alpha;
};
static inline void areaTab(int ssize, int dsize, std::vector<AreaTabEntry>& tab) {
  const double scale = (double)ssize / dsize;
  tab.clear();
  for (int dx = 0; dx < dsize; dx++) {
    const double fsx1 = dx * scale;
    const double fsx2 = fsx1 + scale;
    const double cellWidth = std::min(scale, ssize - fsx1);
    int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) {
      tab.push_back({dx, sx1 - 1, (float)((sx1 - fsx1) / cellWidth)});
    }
    for (int sx = sx1; sx < sx2; sx++) {
      tab.push_back({dx, sx, float(1.0 / cellWidth)});
    }
    if (fsx2 - sx2 > 1e-3) {
      tab.push_back(
          {dx, sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
    }
  }
}

// generic area resize over `cn` interleaved float-convertible channels; T -> float -> T
template <typename T, int CN, typename Cast>
static inline void
resizeAreaGeneric(const T* src, int sw, int sh, int dw, int dh, T* dst, Cast cast) {
  std::vector<AreaTabEntry> xtab, ytab;
  areaTab(sw, dw, xtab);
  areaTab(sh, dh, ytab);
  std::vector<float> buf(size_t(dw) * CN), sum(size_t(dw) * CN, 0.f);
  int prev_dy = ytab.empty() ? -1 : ytab[0].di;
  auto flush = [&](int dy) {
    for (int i = 0; i < dw * CN; ++i) {
      dst[size_t(dy) * dw * CN + i] = cast(sum[i]);
    }
  };
  for (size_t j = 0; j < ytab.size(); ++j) {
    const float beta = ytab[j].alpha;
    const int dy = ytab[j].di, sy = ytab[j].si;
    const T* S = src + size_t(sy) * sw * CN;
    std::fill(buf.begin(), buf.end(), 0.f);
    for (const AreaTabEntry& e : xtab) {
      for (int c = 0; c < CN; ++c) {
        buf[e.di * CN + c] += S[e.si * CN + c] * e.alpha;
      }
    }
    if (dy != prev_dy) {
      flush(prev_dy);
      for (int i = 0; i < dw * CN; ++i) {
        sum[i] = beta * buf[i];
      }
      prev_dy = dy;
    } else {
      for (int i = 0; i < dw * CN; ++i) {
        sum[i] += beta * buf[i];
      }
    }
  }
  if (prev_dy >= 0) {
    flush(prev_dy);
  }
}

// cv::resize(..., INTER_AREA) when the image is ENLARGED along at least one axis (resize.cpp: "true area interpolation
// is only implemented for the case scale_x >= 1 && scale_y >= 1; in other cases it is emulated using some variant of
// bilinear"): the INTER_LINEAR machinery (ksize 2) with the area-mode tap positions and weights
//   sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) * inv_scale, fx = fx <= 0 ? 0 : fx - floor(fx)
// on BOTH axes, float weights (1 - fx, fx), taps past the last column read as the last column times 1, rows clamped;
// horizontal pass first, then the vertical one (HResizeLinear / VResizeLinear<float>, scalar order, no FMA).
// Needed by cv_util::resizeImage<Vec3f> (CvUtil.h:139-147) when UpsampleDisparity's colour guide is smaller than the
// output (UpsampleDisparity.cpp:117). Float images only. Unpinned like every OpenCV primitive here.
struct LinearAreaAxis {
  std::vector<int> ofs;
  std::vector<float> w1;  // weight of tap ofs + 1; weight of tap ofs is 1 - w1
  std::vector<char> one;  // the second tap lies beyond the image: the first tap alone, times 1
};
static inline void linearAreaAxis(int ssize, int dsize, LinearAreaAxis& a) {
  const double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
  a.ofs.resize(dsize);
  a.w1.resize(dsize);
  a.one.resize(dsize);
  for (int dx = 0; dx < dsize; ++dx) {
    int sx = cvFloorD(dx * scale);
    float fx = (float)((dx + 1) - (sx + 1) * inv_scale);
    fx = fx <= 0 ? 0.f : fx - (float)cvFloorD(fx);
    if (sx < 0) {
      fx = 0, sx = 0;
    }
    a.one[dx] = sx + 1 >= ssize;
    if (sx >= ssize - 1) {
      fx = 0, sx = ssize - 1;
    }
    a.ofs[dx] = sx;
    a.w1[dx] = fx;
  }
}
template <int CN>
static inline void resizeLinearAreaF32(const float* src, int sw, int sh, float* dst, int dw, int dh) {
  LinearAreaAxis ax, ay;
  linearAreaAxis(sw, dw, ax);
  linearAreaAxis(sh, dh, ay);
  std::vector<float> r0((size_t)dw * CN), r1((size_t)dw * CN);
  auto hrow = [&](int sy, std::vector<float>& out) {
    const float* S = src + (size_t)sy * sw * CN;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = ax.ofs[dx];
      for (int c = 0; c < CN; ++c) {
        out[(size_t)dx * CN + c] = ax.one[dx] ? S[sx * CN + c] * 1.f
                                              : S[sx * CN + c] * (1.f - ax.w1[dx]) + S[(sx + 1) * CN + c] * ax.w1[dx];
      }
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    const int sy = ay.ofs[dy];
    hrow(std::min(std::max(sy, 0), sh - 1), r0);
    hrow(std::min(std::max(sy + 1, 0), sh - 1), r1);
    const float b0 = 1.f - ay.w1[dy], b1 = ay.w1[dy];
    for (int i = 0; i < dw * CN; ++i) {
      dst[(size_t)dy * dw * CN + i] = r0[i] * b0 + r1[i] * b1;
    }
  }
}

// cv::resize(..., INTER_AREA) as resize.cpp dispatches it when shrinking:
//   * both scale factors integer ("is_area_fast"):
//       - scale 2x2 on 8U/16U: ResizeAreaFastVec_SIMD_*: (a + b + c + d + 2) >> 2
//       - otherwise ResizeAreaFast_<T, float>: float sum over the block in row-major order, * (1.f / area),
//         saturate_cast<T> (round-half-even for integers)
//   * fractional scale: computeResizeAreaTab + ResizeArea_Invoker (resizeAreaGeneric above)
// CN interleaved channels. castI = integer rounding cast (u8 / u16); floats pass through.
template <typename T, int CN>
static inline void resizeAreaCv(const T* src, int sw, int sh, T* dst, int dw, int dh) {
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const int iscale_x = (int)std::nearbyint(scale_x) < 1 ? 1 : cvRoundD(scale_x);
  const int iscale_y = cvRoundD(scale_y) < 1 ? 1 : cvRoundD(scale_y);
  const bool fast = std::abs(scale_x - iscale_x) < 2.220446049250313e-16 && std::abs(scale_y - iscale_y) < 2.220446049250313e-16;
  auto cast = [](float v) -> T {
    if (std::is_same<T, float>::value) {
      return (T)v;
    }
    const int r = cvRoundF(v);
    const int hi = std::is_same<T, uint8_t>::value ? 255 : 65535;
    return (T)(r < 0 ? 0 : r > hi ? hi : r);
  };
  if (sw == dw && sh == dh) {
    memcpy(dst, src, sizeof(T) * (size_t)sw * sh * CN);
    return;
  }
  if (scale_x < 1 || scale_y < 1) {  // enlarging along an axis: the bilinear emulation (float images only here)
    if constexpr (std::is_same<T, float>::value) {
      resizeLinearAreaF32<CN>(src, sw, sh, dst, dw, dh);
    }
    return;
  }
  if (fast) {
    const int area = iscale_x * iscale_y;
    const float scale = 1.f / area;
    for (int dy = 0; dy < dh; ++dy) {
      for (int dx = 0; dx < dw; ++dx) {
        for (int c = 0; c < CN; ++c) {
          if (iscale_x == 2 && iscale_y == 2 && !std::is_same<T, float>::value) {
            const T* r0 = src + ((size_t)(2 * dy) * sw + 2 * dx) * CN + c;
            const T* r1 = r0 + (size_t)sw * CN;
            dst[((size_t)dy * dw + dx) * CN + c] = (T)(((int)r0[0] + (int)r0[CN] + (int)r1[0] + (int)r1[CN] + 2) >> 2);
          } else {
            float sum = 0;
            for (int sy = 0; sy < iscale_y; ++sy) {
              for (int sx = 0; sx < iscale_x; ++sx) {
                sum += (float)src[((size_t)(dy * iscale_y + sy) * sw + dx * iscale_x + sx) * CN + c];
              }
            }
            dst[((size_t)dy * dw + dx) * CN + c] = cast(sum * scale);
          }
        }
      }
    }
    return;
  }
  resizeAreaGeneric<T, CN>(src, sw, sh, dw, dh, dst, cast);
}

} // namespace oracle
