// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// CPU restatement of the reference's camera model (fp64), dependency-free.
// Follows /root/reference/source/util/Camera.h:32-419 and Camera.cpp:21-242.
// Eigen-defined arithmetic (3-vector reductions, AngleAxis re-unitarisation,
// PolynomialSolver) is restated from Eigen 3.3's published algorithms; see
// DESIGN.md "Oracle" for which parts are pinned by the reference's own tests
// (FThetaTest/RectilinearTest/OrthographicTest known answers) and which are not.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace oracle {

struct V2 {
  double x, y;
};
struct V3 {
  double x, y, z;
};

// Eigen's unrolled, non-vectorised reduction of a fixed-size 3-vector splits
// the range in halves: a0 + (a1 + a2)  (Eigen/src/Core/Redux.h, redux_novec_unroller).
static inline double sum3(double a0, double a1, double a2) {
  return a0 + (a1 + a2);
}
static inline double dot3(const V3& a, const V3& b) {
  return sum3(a.x * b.x, a.y * b.y, a.z * b.z);
}
static inline double sqnorm3(const V3& a) {
  return dot3(a, a);
}
static inline V3 sub3(const V3& a, const V3& b) {
  return {a.x - b.x, a.y - b.y, a.z - b.z};
}

enum CameraType { FTHETA = 0, RECTILINEAR = 1, EQUISOLID = 2, ORTHOGRAPHIC = 3 };

// Raw JSON fields of one camera (Camera.cpp:30-75). Shared POD layout with the
// Python side (tests/oracle_lib.py) — this is *input data*, not shared logic.
struct CameraJson {
  int32_t type;
  int32_t has_principal;
  int32_t has_distortion;
  int32_t has_fov;
  double origin[3];
  double forward[3];
  double up[3];
  double right[3];
  double resolution[2];
  double focal[2];
  double principal[2];
  double distortion[3];
  double fov;
  char id[64];
};

struct Camera {
  static constexpr double kNearInfinity = 1e4; // Camera.cpp:19

  int type = FTHETA;
  V3 position{0, 0, 0};
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; // rows: right, up, backward
  V2 resolution{1, 1};
  V2 principal{0.5, 0.5};
  double dist[3] = {0, 0, 0};
  double distMax = std::numeric_limits<double>::infinity();
  V2 focal{1, -1};
  double cosFov = -1;
  std::string id;
  bool valid = true; // false if a CHECK in the reference would have fired
  std::string error;

  // ---- construction (Camera.cpp:30-154) ----
  explicit Camera(const CameraJson& j) {
    id = j.id;
    type = j.type;
    position = {j.origin[0], j.origin[1], j.origin[2]};
    setRotation(
        {j.forward[0], j.forward[1], j.forward[2]},
        {j.up[0], j.up[1], j.up[2]},
        {j.right[0], j.right[1], j.right[2]});
    resolution = {j.resolution[0], j.resolution[1]};
    if (j.has_principal) {
      principal = {j.principal[0], j.principal[1]};
    } else {
      principal = {resolution.x / 2, resolution.y / 2};
    }
    if (j.has_distortion) {
      setDistortion(j.distortion);
    } else {
      setDefaultDistortion();
    }
    if (j.has_fov) {
      setFov(j.fov);
    } else {
      setDefaultFov();
    }
    focal = {j.focal[0], j.focal[1]};
  }
  Camera() {}

  static double defaultCosFov(int t) { // Camera.cpp:190-198
    return (t == RECTILINEAR || t == ORTHOGRAPHIC) ? 0.0 : -1.0;
  }
  void setDefaultFov() {
    cosFov = defaultCosFov(type);
  }
  void setFov(double fov) { // Camera.cpp:204-207
    cosFov = std::cos(fov);
    if (!(cosFov >= defaultCosFov(type))) {
      valid = false;
      error = "fov larger than the type's default";
    }
  }
  double getFov() const {
    return std::acos(cosFov);
  }
  bool isDefaultFov() const {
    return cosFov == defaultCosFov(type);
  }

  // Camera.cpp:77-87 — rows = right/up/-forward, then re-unitarise through
  // Eigen::AngleAxis (matrix -> quaternion -> angle/axis -> matrix).
  void setRotation(const V3& forward, const V3& up, const V3& right) {
    // right.cross(up).dot(forward) < 0
    const V3 c = {
        right.y * up.z - right.z * up.y, right.z * up.x - right.x * up.z,
        right.x * up.y - right.y * up.x};
    if (!(dot3(c, forward) < 0)) {
      valid = false;
      error = "rotation must be right-handed";
    }
    double m[3][3] = {
        {right.x, right.y, right.z}, {up.x, up.y, up.z}, {-forward.x, -forward.y, -forward.z}};
    // isUnitary(tol=1e-3): columns unit length and mutually orthogonal (Eigen Fuzzy.h)
    for (int i = 0; i < 3 && valid; ++i) {
      const V3 ci = {m[0][i], m[1][i], m[2][i]};
      if (std::abs(sqnorm3(ci) - 1.0) > 1e-3) { // isApprox(|c|^2, 1, prec)
        valid = false;
        error = "rotation is not close to unitary";
      }
      for (int jj = 0; jj < i; ++jj) {
        const V3 cj = {m[0][jj], m[1][jj], m[2][jj]};
        if (std::abs(dot3(ci, cj)) > 1e-3) {
          valid = false;
          error = "rotation is not close to unitary";
        }
      }
    }
    // --- Quaternion from matrix (Eigen Quaternion.h, quaternionbase_assign_impl<.,3,3>)
    double q[4]; // x y z w
    double t = sum3(m[0][0], m[1][1], m[2][2]);
    if (t > 0) {
      t = std::sqrt(t + 1.0);
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
      const int j = (i + 1) % 3;
      const int k = (j + 1) % 3;
      t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
      q[i] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[k][j] - m[j][k]) * t;
      q[j] = (m[j][i] + m[i][j]) * t;
      q[k] = (m[k][i] + m[i][k]) * t;
    }
    // --- AngleAxis from quaternion (Eigen AngleAxis.h operator=(QuaternionBase))
    double n = std::sqrt(sum3(q[0] * q[0], q[1] * q[1], q[2] * q[2]));
    double angle;
    V3 axis;
    if (n != 0.0) {
      angle = 2.0 * std::atan2(n, std::abs(q[3]));
      if (q[3] < 0) {
        n = -n;
      }
      axis = {q[0] / n, q[1] / n, q[2] / n};
    } else {
      angle = 0;
      axis = {1, 0, 0};
    }
    // --- AngleAxis::toRotationMatrix
    const double s = std::sin(angle);
    const double cc = std::cos(angle);
    const V3 sin_axis = {s * axis.x, s * axis.y, s * axis.z};
    const V3 cos1_axis = {(1.0 - cc) * axis.x, (1.0 - cc) * axis.y, (1.0 - cc) * axis.z};
    double tmp;
    tmp = cos1_axis.x * axis.y;
    R[0][1] = tmp - sin_axis.z;
    R[1][0] = tmp + sin_axis.z;
    tmp = cos1_axis.x * axis.z;
    R[0][2] = tmp + sin_axis.y;
    R[2][0] = tmp - sin_axis.y;
    tmp = cos1_axis.y * axis.z;
    R[1][2] = tmp - sin_axis.x;
    R[2][1] = tmp + sin_axis.x;
    R[0][0] = cos1_axis.x * axis.x + cc;
    R[1][1] = cos1_axis.y * axis.y + cc;
    R[2][2] = cos1_axis.z * axis.z + cc;
  }

  void setDefaultDistortion() { // Camera.cpp:114-117
    dist[0] = dist[1] = dist[2] = 0;
    distMax = std::numeric_limits<double>::infinity();
  }

  // Camera.cpp:119-154. distortionMax = sqrt(smallest positive real root of the
  // derivative polynomial in y=x^2: 1 + 3 d0 y + 5 d1 y^2 + 7 d2 y^3).
  // The reference uses Eigen::PolynomialSolver (companion-matrix eigenvalues,
  // real roots = |imag| < 1e-12); here: monotone-interval bisection to 1 ulp.
  void setDistortion(const double d[3]) {
    int count = 3;
    while (d[count - 1] == 0) {
      if (--count == 0) {
        setDefaultDistortion();
        return;
      }
    }
    double c[4] = {1, 0, 0, 0};
    for (int i = 0; i < count; ++i) {
      c[i + 1] = d[i] * (2 * i + 3);
    }
    const double y = smallestPositiveRoot(c, count);
    dist[0] = d[0];
    dist[1] = d[1];
    dist[2] = d[2];
    distMax = std::sqrt(y);
  }

  static double polyval(const double* c, int deg, double x) {
    double r = c[deg];
    for (int i = deg - 1; i >= 0; --i) {
      r = r * x + c[i];
    }
    return r;
  }

  static double smallestPositiveRoot(const double* c, int deg) {
    const double inf = std::numeric_limits<double>::infinity();
    // split (0, inf) at the positive critical points of p
    std::vector<double> cuts;
    if (deg == 3) { // p' = c1 + 2 c2 x + 3 c3 x^2
      const double a = 3 * c[3], b = 2 * c[2], cc = c[1];
      const double disc = b * b - 4 * a * cc;
      if (disc >= 0) {
        const double sq = std::sqrt(disc);
        const double qq = -0.5 * (b + (b >= 0 ? sq : -sq));
        double r1 = qq / a;
        double r2 = (qq != 0) ? cc / qq : r1;
        if (r1 > r2) {
          std::swap(r1, r2);
        }
        if (r1 > 0) {
          cuts.push_back(r1);
        }
        if (r2 > 0 && r2 != r1) {
          cuts.push_back(r2);
        }
      }
    } else if (deg == 2) {
      const double r = -c[1] / (2 * c[2]);
      if (r > 0) {
        cuts.push_back(r);
      }
    }
    double lo = 0;
    double flo = c[0]; // = 1 > 0
    for (size_t seg = 0; seg <= cuts.size(); ++seg) {
      double hi;
      if (seg < cuts.size()) {
        hi = cuts[seg];
      } else {
        // expand until sign change or give up
        hi = (lo > 0 ? lo : 1.0) * 2;
        int guard = 0;
        while (polyval(c, deg, hi) * flo > 0 && guard++ < 2000) {
          hi *= 2;
        }
        if (guard >= 2000 || !std::isfinite(hi)) {
          return inf;
        }
      }
      const double fhi = polyval(c, deg, hi);
      if (fhi == 0) {
        return hi;
      }
      if ((fhi > 0) != (flo > 0)) {
        double a = lo, b = hi;
        bool apos = flo > 0;
        for (int it = 0; it < 200; ++it) {
          const double m = 0.5 * (a + b);
          if (m <= a || m >= b) {
            break;
          }
          const double fm = polyval(c, deg, m);
          if (fm == 0) {
            return m;
          }
          if ((fm > 0) == apos) {
            a = m;
          } else {
            b = m;
          }
        }
        return 0.5 * (a + b);
      }
      lo = hi;
      flo = fhi;
    }
    return inf;
  }

  // ---- rescale / normalize (Camera.cpp:217-242) ----
  Camera rescale(const V2& newRes) const {
    Camera r = *this;
    r.principal.x *= newRes.x / r.resolution.x;
    r.principal.y *= newRes.y / r.resolution.y;
    r.focal.x *= newRes.x / r.resolution.x;
    r.focal.y *= newRes.y / r.resolution.y;
    r.resolution = newRes;
    return r;
  }
  void normalize() {
    principal = {principal.x / resolution.x, principal.y / resolution.y};
    focal = {focal.x / resolution.x, focal.y / resolution.y};
    resolution = {1, 1};
  }
  bool isNormalized() const {
    return resolution.x == 1 && resolution.y == 1;
  }

  V3 forwardV() const {
    return {-R[2][0], -R[2][1], -R[2][2]};
  }
  V3 backwardV() const {
    return {R[2][0], R[2][1], R[2][2]};
  }

  // ---- distortion (Camera.h:238-284) ----
  double distortFactor(double rSquared) const {
    double result = dist[2];
    result = dist[1] + rSquared * result;
    result = dist[0] + rSquared * result;
    return 1 + rSquared * result;
  }
  double distort(double r) const {
    r = std::min(r, distMax);
    return distortFactor(r * r) * r;
  }
  bool distortionIsZero() const {
    return dist[0] == 0 && dist[1] == 0 && dist[2] == 0;
  }
  double undistort(const double y) const {
    if (distortionIsZero()) {
      return y;
    }
    if (y >= distort(distMax)) {
      return distMax;
    }
    const double smidgen = 1.0 / kNearInfinity;
    const int kMaxSteps = 10;
    double x0 = 0;
    double y0 = 0;
    double dy0 = 1;
    for (int step = 0; step < kMaxSteps; ++step) {
      const double x1 = (y - y0) / dy0 + x0;
      const double y1 = distort(x1);
      if (std::abs(y1 - y) < smidgen) {
        return x1;
      }
      const double dy1 = (distort(x1 + smidgen) - y1) / smidgen;
      x0 = x1;
      y0 = y1;
      dy0 = dy1;
    }
    return x0;
  }

  // ---- projection (Camera.h:301-341) ----
  V2 cameraToSensor(const V3& c) const {
    if (type == FTHETA) {
      const double xy = std::sqrt(c.x * c.x + c.y * c.y);
      const double r = std::atan2(xy, -c.z);
      const double s = distort(r) / xy;
      return {s * c.x, s * c.y};
    } else if (type == RECTILINEAR) {
      const double xy = std::sqrt(c.x * c.x + c.y * c.y);
      double r;
      if (-c.z <= 0) {
        r = std::tan(M_PI / 2);
      } else {
        r = xy / -c.z;
      }
      const double s = distort(r) / xy;
      return {s * c.x, s * c.y};
    } else if (type == EQUISOLID) {
      const double xy = std::sqrt(c.x * c.x + c.y * c.y);
      const double r = 2 * std::sqrt((1 + c.z / std::sqrt(sqnorm3(c))) / 2);
      const double s = distort(r) / xy;
      return {s * c.x, s * c.y};
    } else {
      V2 pre;
      if (c.z < 0) {
        const double n = std::sqrt(sqnorm3(c));
        pre = {c.x / n, c.y / n};
      } else {
        const double n = std::sqrt(c.x * c.x + c.y * c.y);
        pre = {c.x / n, c.y / n};
      }
      const double f = distortFactor(pre.x * pre.x + pre.y * pre.y);
      return {f * pre.x, f * pre.y};
    }
  }

  // Camera.h:344-378
  V3 sensorToCamera(const V2& sensor) const {
    const double squaredNorm = sensor.x * sensor.x + sensor.y * sensor.y;
    if (squaredNorm == 0) {
      return {0, 0, -1};
    }
    const double norm = std::sqrt(squaredNorm);
    const double r = undistort(norm);
    double theta;
    if (type == FTHETA) {
      theta = r;
    } else if (type == RECTILINEAR) {
      theta = std::atan(r);
    } else if (type == EQUISOLID) {
      theta = r <= 2 ? 2 * std::asin(r / 2) : M_PI;
    } else {
      theta = r <= 1 ? std::asin(r) : M_PI / 2;
    }
    const double s = std::sin(theta) / norm;
    return {s * sensor.x, s * sensor.y, -std::cos(theta)};
  }

  // Camera.h:121-128 — rotation * (rig - position): 3x3 * 3x1 is Eigen's
  // coefficient-based lazy product, each coefficient a 3-term redux.
  V2 pixel(const V3& rig) const {
    const V3 v = sub3(rig, position);
    const V3 cam = {
        sum3(R[0][0] * v.x, R[0][1] * v.y, R[0][2] * v.z),
        sum3(R[1][0] * v.x, R[1][1] * v.y, R[1][2] * v.z),
        sum3(R[2][0] * v.x, R[2][1] * v.y, R[2][2] * v.z)};
    const V2 sensor = cameraToSensor(cam);
    return {focal.x * sensor.x + principal.x, focal.y * sensor.y + principal.y};
  }

  // Camera.h:131-143 — Ray(position, rotation^T * unit).pointAt(depth)
  V3 rigDirection(const V2& pix) const {
    const V2 sensor = {(pix.x - principal.x) / focal.x, (pix.y - principal.y) / focal.y};
    const V3 u = sensorToCamera(sensor);
    return {
        sum3(R[0][0] * u.x, R[1][0] * u.y, R[2][0] * u.z),
        sum3(R[0][1] * u.x, R[1][1] * u.y, R[2][1] * u.z),
        sum3(R[0][2] * u.x, R[1][2] * u.y, R[2][2] * u.z)};
  }
  V3 rig(const V2& pix, double depth) const {
    const V3 d = rigDirection(pix);
    return {position.x + d.x * depth, position.y + d.y * depth, position.z + d.z * depth};
  }
  V3 rigNearInfinity(const V2& pix) const {
    return rig(pix, kNearInfinity);
  }

  bool isBehind(const V3& rig) const { // Camera.h:150-152
    return dot3(backwardV(), sub3(rig, position)) >= 0;
  }
  bool isOutsideFov(const V3& rig) const { // Camera.h:154-164
    if (cosFov == -1) {
      return false;
    }
    if (cosFov == 0) {
      return isBehind(rig);
    }
    const V3 v = sub3(rig, position);
    const double dot = dot3(forwardV(), v);
    return dot * std::abs(dot) <= cosFov * std::abs(cosFov) * sqnorm3(v);
  }
  bool isOutsideImageCircle(const V2& pix) const { // Camera.h:166-178
    if (isDefaultFov()) {
      return false;
    }
    const double sinFov = std::sqrt(1 - cosFov * cosFov);
    const V2 edge = cameraToSensor({0, sinFov, -cosFov});
    const V2 sensor = {(pix.x - principal.x) / focal.x, (pix.y - principal.y) / focal.y};
    return sensor.x * sensor.x + sensor.y * sensor.y >= edge.x * edge.x + edge.y * edge.y;
  }
  bool isOutsideSensor(const V2& pix) const { // Camera.h:180-182
    return 0 > pix.x || pix.x >= resolution.x || 0 > pix.y || pix.y >= resolution.y;
  }
  bool sees(const V3& rig, V2& pix) const { // Camera.h:184-190
    if (isOutsideFov(rig)) {
      return false;
    }
    pix = pixel(rig);
    return !isOutsideSensor(pix);
  }
};

using Rig = std::vector<Camera>;

} // namespace oracle
