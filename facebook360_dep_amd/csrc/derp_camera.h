// Camera model of the depth path, fp64, usable from host and device.
// Follows the reference's source/util/Camera.h:121-378 and Camera.cpp:77-242 operation for
// operation (Camera::Real = double, Camera.h:33). 3-vector reductions use Eigen 3.3's
// unrolled association a0 + (a1 + a2). Compile with -ffp-contract=off: the reference's
// x86-64 release build has no FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/derp_hip.h"

#define DERP_HD __host__ __device__ __forceinline__

namespace derp {

struct Cam {  // normalised camera (resolution 1x1) unless rescaled by the caller
  double pos[3];
  double R[9];  // row-major; rows = right, up, backward
  double principal[2];
  double focal[2];
  double dist[3];
  double dist_max;
  double cos_fov;
  double edge_sq;       // |cameraToSensor((0, sinFov, -cosFov))|^2, for isOutsideImageCircle
  int32_t type;
  int32_t default_fov;  // isDefaultFov()
  int32_t dist_zero;    // getDistortion().isZero()
  int32_t pad;
};

struct D2 {
  double x, y;
};
struct D3 {
  double x, y, z;
};

DERP_HD double sum3(double a, double b, double c) {
  return a + (b + c);
}

// Camera.h:238-253
DERP_HD double distort_factor(const Cam& c, double r2) {
  double result = c.dist[2];
  result = c.dist[1] + r2 * result;
  result = c.dist[0] + r2 * result;
  return 1.0 + r2 * result;
}
DERP_HD double distort(const Cam& c, double r) {
  r = (c.dist_max < r) ? c.dist_max : r;  // std::min(r, distortionMax_)
  return distort_factor(c, r * r) * r;
}

// Camera.h:255-284 (Newton, <= 10 steps, smidgen = 1/kNearInfinity)
DERP_HD double undistort(const Cam& c, const double y) {
  if (c.dist_zero) {
    return y;
  }
  if (y >= distort(c, c.dist_max)) {
    return c.dist_max;
  }
  const double smidgen = 1.0 / 1e4;
  double x0 = 0, y0 = 0, dy0 = 1;
  for (int step = 0; step < 10; ++step) {
    const double x1 = (y - y0) / dy0 + x0;
    const double y1 = distort(c, x1);
    if (fabs(y1 - y) < smidgen) {
      return x1;
    }
    const double dy1 = (distort(c, x1 + smidgen) - y1) / smidgen;
    x0 = x1;
    y0 = y1;
    dy0 = dy1;
  }
  return x0;
}

static constexpr int kAtanLutDoubles = 20;  // atan2_ypos_lut's table: 5 rows of {atanhi, atanlo, c, pad}
#if defined(__HIP_DEVICE_COMPILE__)
// IEEE fp64 division for operands whose quotient neither overflows nor underflows: the reciprocal-refinement
// sequence the compiler emits for `/` (two Newton steps, quotient, one correction) without its exponent
// pre-scaling and special-case fix-up — the same roundings, so the same correctly rounded quotient.
__device__ __forceinline__ double div_plain(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// atan2(y, x) for y >= 0 (the FTHETA projection: y = |xy|, x = -z), result in [0, pi]. fdlibm's atan
// (s_atan.c: breakpoints 7/16, 11/16, 19/16, 39/16, the 11-term polynomial aT[], atanhi / atanlo) with the
// argument reduction written on the pair (y, x) — (y - c x) / (x + c y) instead of (t - c) / (1 + c t) — so
// that it takes ONE division; <= 1 ulp from glibc's atan2 (test_lean_atan2_against_libm compares 4 x 10^6
// arguments through derp_debug_atan2_ypos), like the device library's routine it replaces in the cost kernels,
// at less than half its instruction count.
// An fp64 literal for the hot chain, delivered in a scalar register pair at the point of use. gfx950 has no 64-bit
// literal operands, and left alone the compiler parks every coefficient in a vector register pair hoisted out of
// the candidate loop — two dozen registers that the allocator then spills to scratch INSIDE the dependent chain as
// soon as anything else needs room (measured: +30 % on level-0 ping-pong). The opaque asm makes the value a
// scalar one: if it is kept across the loop it costs scalar registers, whose spills are lane moves, not memory.
// (Not volatile: a volatile asm is a scheduling barrier and serialises the two Horner chains, +25 %.)
__device__ __forceinline__ double sk(double v) {
  asm("" : "+s"(v));
  return v;
}
// sqrt(x) exactly as the compiler expands llvm.sqrt.f64 on this target (scale tiny arguments by 2^256, v_rsq_f64, one
// Goldschmidt step on (g, h), two residual corrections, scale back by 2^-128, pass +-0 / +inf through) — operation for
// operation, so the same bits — but without its three 32-bit literals: the expansion selects 256 / -128 with v_cndmask,
// which (vcc + one scalar = the constant-bus limit) forces them, and the class mask, into vector registers that then
// ride through every loop of the cost kernels. Here the exponents come from the comparison bit by shifts and the mask
// sits in a scalar register: three more 32-bit instructions per projection, three VGPRs fewer everywhere.
__device__ __forceinline__ double sqrt_lean(double x) {
  const int scaled = (int)(x < sk(0x1.0p-767));
  const double xs = __builtin_ldexp(x, scaled << 8);
  const double y = __builtin_amdgcn_rsq(xs);
  double g = xs * y;
  double h = y * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  double d = __builtin_fma(-g, g, xs);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, xs);
  g = __builtin_fma(d, h, g);
  const double res = __builtin_ldexp(g, -(scaled << 7));
  int mask = 0x260;  // +-0 and +inf
  asm("" : "+s"(mask));
  return __builtin_amdgcn_class(xs, mask) ? xs : res;
}
__device__ __forceinline__ double atan2_ypos(double y, double x) {
  const double ax = fabs(x);
  const bool c0 = y < 0.4375 * ax, c1 = y < 0.6875 * ax, c2 = y < 1.1875 * ax, c3 = y < 2.4375 * ax;
  const double c = c1 ? 0.5 : c2 ? 1.0 : 1.5;
  double num = __builtin_fma(-c, ax, y), den = __builtin_fma(c, y, ax);
  num = c0 ? y : c3 ? num : -ax;
  den = c0 ? ax : c3 ? den : y;
  const double hi = c0 ? 0.0 : c1 ? sk(4.63647609000806093515e-01) : c2 ? sk(7.85398163397448278999e-01)
                     : c3 ? sk(9.82793723247329054082e-01) : sk(1.57079632679489655800e+00);
  const double lo = c0 ? 0.0 : c1 ? sk(2.26987774529616870924e-17) : c2 ? sk(3.06161699786838301793e-17)
                     : c3 ? sk(1.39033110312309984516e-17) : sk(6.12323399573676603587e-17);
  const double t = div_plain(num, den);
  const double z = t * t, w = z * z;
  double s1 = __builtin_fma(w, sk(1.62858201153657823623e-02), sk(4.97687799461593236017e-02));
  s1 = __builtin_fma(w, s1, sk(6.66107313738753120669e-02));
  s1 = __builtin_fma(w, s1, sk(9.09088713343650656196e-02));
  s1 = __builtin_fma(w, s1, sk(1.42857142725034663711e-01));
  s1 = __builtin_fma(w, s1, sk(3.33333333333329318027e-01));
  s1 = z * s1;
  double s2 = __builtin_fma(w, sk(-3.65315727442169155270e-02), sk(-5.83357013379057348645e-02));
  s2 = __builtin_fma(w, s2, sk(-7.69187620504482999495e-02));
  s2 = __builtin_fma(w, s2, sk(-1.11111104054623557880e-01));
  s2 = __builtin_fma(w, s2, sk(-1.99999999998764832476e-01));
  s2 = w * s2;
  double r = hi - ((t * (s1 + s2) - lo) - t);
  if (x < 0) {
    r = sk(3.1415926535897931160E+00) - (r - sk(1.2246467991473531772E-16));
  }
  return r;
}

// The same routine with the five-way choice of (atanhi, atanlo, c) read from a table in LDS by a per-lane index
// instead of selected from scalar literals. With the literals in scalar registers (sk() above) the compiler can only
// choose per lane by branching on the execution mask: 8 nested s_and_saveexec / s_cbranch diamonds — about a hundred
// scalar instructions and a dozen v_readlane / v_writelane scalar-spill moves inside every projection. Here the
// index is the number of breakpoints the lane's argument lies above (four compares feeding v_addc), the constants
// arrive with two LDS reads, and c = 0 serves the first interval: fma(-0, ax, y) = y and fma(0, y, ax) = ax exactly.
// Bit-identical to atan2_ypos for finite arguments (test_lean_atan2_against_libm runs both).
// lut: kAtanLutRows rows of {atanhi, atanlo, c, pad}, 32 bytes each (atan_lut_fill).
__device__ __forceinline__ void atan_lut_fill(double* lut) {
  const int i = (int)threadIdx.x;
  if (i < kAtanLutDoubles) {
    const int row = i >> 2, col = i & 3;
    const double hi = row == 1 ? 4.63647609000806093515e-01 : row == 2 ? 7.85398163397448278999e-01
                    : row == 3 ? 9.82793723247329054082e-01 : row == 4 ? 1.57079632679489655800e+00 : 0.0;
    const double lo = row == 1 ? 2.26987774529616870924e-17 : row == 2 ? 3.06161699786838301793e-17
                    : row == 3 ? 1.39033110312309984516e-17 : row == 4 ? 6.12323399573676603587e-17 : 0.0;
    const double c = row == 1 ? 0.5 : row == 2 ? 1.0 : row == 3 ? 1.5 : 0.0;
    lut[i] = col == 0 ? hi : col == 1 ? lo : col == 2 ? c : 0.0;
  }
}
__device__ __forceinline__ double atan2_ypos_lut(double y, double x, const double* lut) {
  const double ax = fabs(x);
  const int idx = (int)!(y < 0.4375 * ax) + (int)!(y < 0.6875 * ax) + (int)!(y < 1.1875 * ax) + (int)!(y < 2.4375 * ax);
  const double* row = lut + idx * 4;
  const double hi = row[0], lo = row[1], c = row[2];
  double num = __builtin_fma(-c, ax, y), den = __builtin_fma(c, y, ax);
  const bool top = idx == 4;
  num = top ? -ax : num;
  den = top ? y : den;
  const double t = div_plain(num, den);
  const double z = t * t, w = z * z;
  double s1 = __builtin_fma(w, sk(1.62858201153657823623e-02), sk(4.97687799461593236017e-02));
  s1 = __builtin_fma(w, s1, sk(6.66107313738753120669e-02));
  s1 = __builtin_fma(w, s1, sk(9.09088713343650656196e-02));
  s1 = __builtin_fma(w, s1, sk(1.42857142725034663711e-01));
  s1 = __builtin_fma(w, s1, sk(3.33333333333329318027e-01));
  s1 = z * s1;
  double s2 = __builtin_fma(w, sk(-3.65315727442169155270e-02), sk(-5.83357013379057348645e-02));
  s2 = __builtin_fma(w, s2, sk(-7.69187620504482999495e-02));
  s2 = __builtin_fma(w, s2, sk(-1.11111104054623557880e-01));
  s2 = __builtin_fma(w, s2, sk(-1.99999999998764832476e-01));
  s2 = w * s2;
  double r = hi - ((t * (s1 + s2) - lo) - t);
  if (x < 0) {
    r = sk(3.1415926535897931160E+00) - (r - sk(1.2246467991473531772E-16));
  }
  return r;
}
#endif

// Camera.h:301-341. LEAN (device, cost kernels only): fp64 atan2 / division through the routines above
// (1 = interval constants from scalar literals, 2 = from the table `atanLut` in LDS).
// LEAN: bits 0-1 = the short atan2 / division (1: constants from scalar literals, 2: from the LDS table); bit 2 = square
// roots through sqrt_lean in every camera type (the same bits as sqrt(), none of its vector-register literals)
#if defined(__HIP_DEVICE_COMPILE__)
#define DERP_SQRT(x) ((LEAN & 4) ? sqrt_lean(x) : sqrt(x))
#else
#define DERP_SQRT(x) sqrt(x)
#endif
template <int LEAN = 0>
DERP_HD D2 camera_to_sensor(const Cam& c, const D3& p, const double* atanLut = nullptr) {
#ifdef DERP_MIX_HOT_ONLY
  if (DERP_MIX_HOT_ONLY || c.type == DERP_FTHETA) {
#else
  if (c.type == DERP_FTHETA) {
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const double xy = DERP_SQRT(p.x * p.x + p.y * p.y);
    if (LEAN & 3) {
      const double r = (LEAN & 3) == 2 ? atan2_ypos_lut(xy, -p.z, atanLut) : atan2_ypos(xy, -p.z);
      const double s = div_plain(distort(c, r), xy);
      return {s * p.x, s * p.y};
    }
#else
    const double xy = sqrt(p.x * p.x + p.y * p.y);
#endif
    const double r = atan2(xy, -p.z);
    const double s = distort(c, r) / xy;
    return {s * p.x, s * p.y};
  } else if (c.type == DERP_RECTILINEAR) {
    const double xy = DERP_SQRT(p.x * p.x + p.y * p.y);
    double r;
    if (-p.z <= 0) {
#if defined(__HIP_DEVICE_COMPILE__)
      r = sk(tan(M_PI / 2));  // (a scalar: left alone the folded literal rides in a vector register pair through every kernel)
#else
      r = tan(M_PI / 2);
#endif
    } else {
      r = xy / -p.z;
    }
    const double s = distort(c, r) / xy;
    return {s * p.x, s * p.y};
  } else if (c.type == DERP_EQUISOLID) {
    const double xy = DERP_SQRT(p.x * p.x + p.y * p.y);
    const double n = DERP_SQRT(sum3(p.x * p.x, p.y * p.y, p.z * p.z));
    const double r = 2 * DERP_SQRT((1 + p.z / n) / 2);
    const double s = distort(c, r) / xy;
    return {s * p.x, s * p.y};
  } else {
    double px, py;
    if (p.z < 0) {
      const double n = DERP_SQRT(sum3(p.x * p.x, p.y * p.y, p.z * p.z));
      px = p.x / n;
      py = p.y / n;
    } else {
      const double n = DERP_SQRT(p.x * p.x + p.y * p.y);
      px = p.x / n;
      py = p.y / n;
    }
    const double f = distort_factor(c, px * px + py * py);
    return {f * px, f * py};
  }
}

#undef DERP_SQRT
// Camera.h:344-378
DERP_HD D3 sensor_to_camera(const Cam& c, const D2& s) {
  const double sq = s.x * s.x + s.y * s.y;
  if (sq == 0) {
    return {0, 0, -1};
  }
  const double norm = sqrt(sq);
  const double r = undistort(c, norm);
  double theta;
  if (c.type == DERP_FTHETA) {
    theta = r;
  } else if (c.type == DERP_RECTILINEAR) {
    theta = atan(r);
  } else if (c.type == DERP_EQUISOLID) {
    theta = r <= 2 ? 2 * asin(r / 2) : M_PI;
  } else {
    theta = r <= 1 ? asin(r) : M_PI / 2;
  }
  const double k = sin(theta) / norm;
  return {k * s.x, k * s.y, -cos(theta)};
}

// Camera.h:131-138: direction of Ray(position, rotation^T * unit); pix in the units of
// (principal, focal) passed in (normalised or level pixels).
DERP_HD D3 rig_direction(const Cam& c, double px, double py, double prx, double pry, double fx, double fy) {
  const D2 sensor = {(px - prx) / fx, (py - pry) / fy};
  const D3 u = sensor_to_camera(c, sensor);
  return {
      sum3(c.R[0] * u.x, c.R[3] * u.y, c.R[6] * u.z),
      sum3(c.R[1] * u.x, c.R[4] * u.y, c.R[7] * u.z),
      sum3(c.R[2] * u.x, c.R[5] * u.y, c.R[8] * u.z)};
}

// Camera.h:166-178, pix already in sensor units: (pix - principal) / focal
DERP_HD bool outside_image_circle(const Cam& c, double px, double py, double prx, double pry, double fx, double fy) {
  if (c.default_fov) {
    return false;
  }
  const double sx = (px - prx) / fx, sy = (py - pry) / fy;
  return sx * sx + sy * sy >= c.edge_sq;
}

// Camera.h:184-190 (+154-164, 121-128, 180-182). Returns false if the point is outside the
// FOV cone or projects off the sensor; pix in units of (principal, focal, res).
template <int LEAN = 0>
DERP_HD bool sees(const Cam& c, const D3& rig, double prx, double pry, double fx, double fy,
                  double resx, double resy, D2& pix, const double* atanLut = nullptr) {
  const D3 v = {rig.x - c.pos[0], rig.y - c.pos[1], rig.z - c.pos[2]};
  // backward().dot(v): also the camera-space z used below. forward().dot(v) is its exact negation
  // (negating every product and the sums commutes with round-to-nearest), so it is not recomputed.
  const double back = sum3(c.R[6] * v.x, c.R[7] * v.y, c.R[8] * v.z);
  if (c.cos_fov != -1) {
    if (c.cos_fov == 0) {
      // isBehind: backward().dot(v) >= 0
      if (back >= 0) {
        return false;
      }
    } else {
      const double dot = -back;
      const double sq = sum3(v.x * v.x, v.y * v.y, v.z * v.z);
      if (dot * fabs(dot) <= c.cos_fov * fabs(c.cos_fov) * sq) {
        return false;
      }
    }
  }
  const D3 cam = {
      sum3(c.R[0] * v.x, c.R[1] * v.y, c.R[2] * v.z),
      sum3(c.R[3] * v.x, c.R[4] * v.y, c.R[5] * v.z),
      back};
  const D2 s = camera_to_sensor<LEAN>(c, cam, atanLut);
  pix.x = fx * s.x + prx;
  pix.y = fy * s.y + pry;
  return !(0 > pix.x || pix.x >= resx || 0 > pix.y || pix.y >= resy);
}

// ---------------------------------------------------------------------------------------
// Host-side construction: Camera::Camera(json) + normalize(). Returns NULL on success or a
// message describing the CHECK that would have fired in the reference.
// ---------------------------------------------------------------------------------------
inline double host_poly(const double* k, int deg, double x) {
  double acc = k[deg];
  for (int i = deg; i-- > 0;) {
    acc = acc * x + k[i];
  }
  return acc;
}

// smallest root > 0 of k[0] + k[1] y + ... + k[deg] y^deg (k[0] = 1), deg <= 3; +inf if none.
// Reference: Eigen::PolynomialSolver real roots (Camera.cpp:139-150).
inline double host_smallest_positive_root(const double* k, int deg) {
  double stops[4];
  int n_stops = 0;
  if (deg == 3) {
    const double a = 3 * k[3], b = 2 * k[2], c = k[1];
    const double disc = b * b - 4 * a * c;
    if (disc >= 0) {
      const double q = -0.5 * (b + copysign(sqrt(disc), b));
      double r0 = q / a, r1 = (q != 0) ? c / q : r0;
      if (r0 > r1) {
        const double t = r0;
        r0 = r1;
        r1 = t;
      }
      if (r0 > 0) {
        stops[n_stops++] = r0;
      }
      if (r1 > 0 && r1 != r0) {
        stops[n_stops++] = r1;
      }
    }
  } else if (deg == 2) {
    const double r = -k[1] / (2 * k[2]);
    if (r > 0) {
      stops[n_stops++] = r;
    }
  }
  double left = 0, f_left = k[0];
  for (int seg = 0; seg <= n_stops; ++seg) {
    double right;
    if (seg < n_stops) {
      right = stops[seg];
    } else {
      right = (left > 0 ? left : 1.0) * 2;
      int tries = 0;
      while (host_poly(k, deg, right) * f_left > 0 && tries++ < 2000) {
        right *= 2;
      }
      if (tries >= 2000 || !isfinite(right)) {
        return INFINITY;
      }
    }
    const double f_right = host_poly(k, deg, right);
    if (f_right == 0) {
      return right;
    }
    if ((f_right > 0) != (f_left > 0)) {
      double lo = left, hi = right;
      const bool lo_pos = f_left > 0;
      for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) {
          break;
        }
        const double fm = host_poly(k, deg, mid);
        if (fm == 0) {
          return mid;
        }
        if ((fm > 0) == lo_pos) {
          lo = mid;
        } else {
          hi = mid;
        }
      }
      return 0.5 * (lo + hi);
    }
    left = right;
    f_left = f_right;
  }
  return INFINITY;
}

inline const char* host_prepare_camera(const derp_camera_desc& j, Cam& c) {
  c.type = j.type;
  if (j.type < 0 || j.type > 3) {
    return "unknown camera type";
  }
  for (int i = 0; i < 3; ++i) {
    c.pos[i] = j.origin[i];
  }
  // --- setRotation(forward, up, right): Camera.cpp:77-87
  const double* f = j.forward;
  const double* u = j.up;
  const double* r = j.right;
  const double cx = r[1] * u[2] - r[2] * u[1], cy = r[2] * u[0] - r[0] * u[2], cz = r[0] * u[1] - r[1] * u[0];
  if (!(sum3(cx * f[0], cy * f[1], cz * f[2]) < 0)) {
    return "rotation must be right-handed";
  }
  const double m[3][3] = {{r[0], r[1], r[2]}, {u[0], u[1], u[2]}, {-f[0], -f[1], -f[2]}};
  for (int a = 0; a < 3; ++a) {
    const double n2 = sum3(m[0][a] * m[0][a], m[1][a] * m[1][a], m[2][a] * m[2][a]);
    if (fabs(n2 - 1.0) > 1e-3) {
      return "rotation is not close to unitary";
    }
    for (int b = 0; b < a; ++b) {
      if (fabs(sum3(m[0][a] * m[0][b], m[1][a] * m[1][b], m[2][a] * m[2][b])) > 1e-3) {
        return "rotation is not close to unitary";
      }
    }
  }
  // Eigen: AngleAxis(matrix) = AngleAxis(Quaternion(matrix)); then toRotationMatrix()
  double qx, qy, qz, qw;
  {
    double q[4];
    double t = sum3(m[0][0], m[1][1], m[2][2]);
    if (t > 0) {
      t = sqrt(t + 1.0);
      q[3] = 0.5 * t;
      t = 0.5 / t;
      q[0] = (m[2][1] - m[1][2]) * t;
      q[1] = (m[0][2] - m[2][0]) * t;
      q[2] = (m[1][0] - m[0][1]) * t;
    } else {
      int i = 0;
      if (m[1][1] > m[0][0]) {
        i = 1;
      }
      if (m[2][2] > m[i][i]) {
        i = 2;
      }
      const int jj = (i + 1) % 3, kk = (jj + 1) % 3;
      t = sqrt(m[i][i] - m[jj][jj] - m[kk][kk] + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[kk][jj] - m[jj][kk]) * t;
      q[jj] = (m[jj][i] + m[i][jj]) * t;
      q[kk] = (m[kk][i] + m[i][kk]) * t;
    }
    qx = q[0];
    qy = q[1];
    qz = q[2];
    qw = q[3];
  }
  double n = sqrt(sum3(qx * qx, qy * qy, qz * qz));
  double angle, ax, ay, az;
  if (n != 0) {
    angle = 2.0 * atan2(n, fabs(qw));
    if (qw < 0) {
      n = -n;
    }
    ax = qx / n;
    ay = qy / n;
    az = qz / n;
  } else {
    angle = 0;
    ax = 1;
    ay = 0;
    az = 0;
  }
  {
    const double s = sin(angle), cs = cos(angle);
    const double sx = s * ax, sy = s * ay, sz = s * az;
    const double c1x = (1.0 - cs) * ax, c1y = (1.0 - cs) * ay, c1z = (1.0 - cs) * az;
    double t;
    t = c1x * ay;
    c.R[1] = t - sz;
    c.R[3] = t + sz;
    t = c1x * az;
    c.R[2] = t + sy;
    c.R[6] = t - sy;
    t = c1y * az;
    c.R[5] = t - sx;
    c.R[7] = t + sx;
    c.R[0] = c1x * ax + cs;
    c.R[4] = c1y * ay + cs;
    c.R[8] = c1z * az + cs;
  }
  // --- principal / distortion / fov / focal: Camera.cpp:44-70
  double res[2] = {j.resolution[0], j.resolution[1]};
  double pr[2] = {j.has_principal ? j.principal[0] : res[0] / 2, j.has_principal ? j.principal[1] : res[1] / 2};
  c.dist[0] = c.dist[1] = c.dist[2] = 0;
  c.dist_max = INFINITY;
  if (j.has_distortion) {  // setDistortion, Camera.cpp:119-154
    int count = 3;
    while (count > 0 && j.distortion[count - 1] == 0) {
      --count;
    }
    if (count > 0) {
      double k[4] = {1, 0, 0, 0};
      for (int i = 0; i < count; ++i) {
        k[i + 1] = j.distortion[i] * (2 * i + 3);
      }
      const double y = host_smallest_positive_root(k, count);
      for (int i = 0; i < 3; ++i) {
        c.dist[i] = j.distortion[i];
      }
      c.dist_max = sqrt(y);
    }
  }
  c.dist_zero = (c.dist[0] == 0 && c.dist[1] == 0 && c.dist[2] == 0);
  const double def_cos = (j.type == DERP_RECTILINEAR || j.type == DERP_ORTHOGRAPHIC) ? 0.0 : -1.0;
  if (j.has_fov) {
    c.cos_fov = cos(j.fov);
    if (!(c.cos_fov >= def_cos)) {
      return "fov exceeds the camera type's default";
    }
  } else {
    c.cos_fov = def_cos;
  }
  c.default_fov = (c.cos_fov == def_cos);
  // --- normalize(): Camera.cpp:225-229
  c.principal[0] = pr[0] / res[0];
  c.principal[1] = pr[1] / res[1];
  c.focal[0] = j.focal[0] / res[0];
  c.focal[1] = j.focal[1] / res[1];
  // edge point of the FOV cone in sensor space (Camera.h:171-173)
  c.edge_sq = 0;
  if (!c.default_fov) {
    const double sin_fov = sqrt(1 - c.cos_fov * c.cos_fov);
    const D2 e = camera_to_sensor(c, D3{0, sin_fov, -c.cos_fov});
    c.edge_sq = e.x * e.x + e.y * e.y;
  }
  c.pad = 0;
  return nullptr;
}

}  // namespace derp
