// CPU restatement of the reference's rephotography renderer: CanopyScene::cubemap
// (source/render/CanopyScene.cpp:36-69 Canopy::render, :72-196 shaders, :198-283 accumulate / CanopyScene::render,
// :285-374 createCubemapTexture / cubemap, :447-476 disparityMesh / alphaFov) as ComputeRephotographyErrors.cpp:77-95
// drives it. The reference renders with OpenGL; this is a software rasteriser with the same pipeline:
//   per camera ("canopy"): a vertex per disparity pixel at camera.rig(pixel centre, 1 / disparity), the triangle
//   strip of stripify() (two triangles per pixel quad), depth test GL_LEQUAL, fragment = bilinear sample of the
//   camera's colour (alpha = inside the image circle), discarded when alpha == 0, alpha *= minor axis of the screen
//   -> texture Jacobian (dFdx / dFdy of texVar) * cone(texVar);
//   accumulate: weight = exp(30 alpha) - 1, premultiplied sum over the cameras; un-premultiply; NaN -> 0;
//   six 90-degree faces (+X -X +Y -Y +Z -Z, EXT_texture_cube_map axes), stacked top to bottom.
// What OpenGL leaves to the implementation is fixed here as follows: pixel centres at half
// integers with an inclusive edge test (no holes; double hits resolve through the depth test, later triangle
// wins ties like GL_LEQUAL); triangles with a vertex nearer than the near plane (0.1 m) or with a NaN vertex
// are dropped instead of clipped; derivatives are the fine 2x2-quad differences of the triangle's
// perspective-correct interpolant; the texture is filtered the way the reference asks OpenGL to
// (source/gpu/GlUtil.h:316-338: glGenerateMipmap, GL_LINEAR_MIPMAP_LINEAR, maximum anisotropy): a 2x2-box mip
// chain, and per fragment the EXT_texture_filter_anisotropic recipe — N = min(ceil(Pmax / Pmin), 16) trilinear taps
// along the major axis of the pixel footprint at LOD log2(Pmax / N) — with the LOD fraction taken linear in the
// footprint inside an octave (no log2: exact in fp32, so HIP and this restatement agree bit for bit). fp32
// throughout, like the shaders; the mesh vertices come from the fp64 camera model and are rounded to fp32
// (cv::Vec3f). Test infrastructure only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle_camera.h"

namespace oracle {

struct CanopyTri {
  float sx[3], sy[3], invd[3], tu[3], tv[3];
  float area;
};

// EXT_texture_cube_map table (CanopyScene.cpp:318-332): major axis, sc, tc
static const int kCubeAxes[6][3][2] = {  // {axis index 0..2, sign}
    {{0, +1}, {2, -1}, {1, -1}}, {{0, -1}, {2, +1}, {1, -1}}, {{1, +1}, {0, +1}, {2, +1}},
    {{1, -1}, {0, +1}, {2, -1}}, {{2, +1}, {0, +1}, {1, -1}}, {{2, -1}, {0, -1}, {1, -1}}};

static const int kCanopyMaxAniso = 16;  // GL_MAX_TEXTURE_MAX_ANISOTROPY of current hardware

struct CanopyMesh {
  int w = 0, h = 0;
  std::vector<float> v;        // [h][w][3] rig-space vertex, NaN when the disparity is unusable
  std::vector<float> rgba;     // mip chain, level 0 first: [h_k][w_k][4] B, G, R in [0, 1], A = inside the image circle
  std::vector<size_t> mipOff;  // float offset of each level in rgba
  std::vector<int> mipW, mipH;
};

// glGenerateMipmap: level k = 2x2 box of level k - 1 (sizes halve, rounding down, never below 1; an odd last
// row / column repeats its edge texel)
static inline void canopyBuildMips(CanopyMesh& m) {
  m.mipOff.assign(1, 0);
  m.mipW.assign(1, m.w);
  m.mipH.assign(1, m.h);
  while (m.mipW.back() > 1 || m.mipH.back() > 1) {
    const int sw = m.mipW.back(), sh = m.mipH.back();
    const int dw = std::max(1, sw >> 1), dh = std::max(1, sh >> 1);
    const size_t so = m.mipOff.back(), dofs = m.rgba.size();
    m.rgba.resize(dofs + (size_t)dw * dh * 4);
    for (int y = 0; y < dh; ++y) {
      for (int x = 0; x < dw; ++x) {
        const int x0 = std::min(2 * x, sw - 1), x1 = std::min(2 * x + 1, sw - 1);
        const int y0 = std::min(2 * y, sh - 1), y1 = std::min(2 * y + 1, sh - 1);
        for (int c = 0; c < 4; ++c) {
          const float a = m.rgba[so + ((size_t)y0 * sw + x0) * 4 + c], b = m.rgba[so + ((size_t)y0 * sw + x1) * 4 + c];
          const float e = m.rgba[so + ((size_t)y1 * sw + x0) * 4 + c], f = m.rgba[so + ((size_t)y1 * sw + x1) * 4 + c];
          m.rgba[dofs + ((size_t)y * dw + x) * 4 + c] = ((a + b) + (e + f)) * 0.25f;
        }
      }
    }
    m.mipOff.push_back(dofs);
    m.mipW.push_back(dw);
    m.mipH.push_back(dh);
  }
}

static inline CanopyMesh canopyMesh(const Camera& cam, const uint16_t* bgr, const float* disp, int w, int h) {
  CanopyMesh m;
  m.w = w;
  m.h = h;
  m.v.resize((size_t)w * h * 3);
  m.rgba.resize((size_t)w * h * 4);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const size_t i = (size_t)y * w + x;
      V2 p = {(x + 0.5) / w, (y + 0.5) / h};
      if (!cam.isNormalized()) {
        p = {p.x * cam.resolution.x, p.y * cam.resolution.y};
      }
      const float distance = 1.0f / disp[i];  // disparityMesh, CanopyScene.cpp:453
      const V3 rig = cam.rig(p, (double)distance);
      m.v[3 * i + 0] = (float)rig.x;
      m.v[3 * i + 1] = (float)rig.y;
      m.v[3 * i + 2] = (float)rig.z;
      for (int c = 0; c < 3; ++c) {
        m.rgba[4 * i + c] = (float)bgr[3 * i + c] / 65535.0f;  // GL_RGBA16 texel -> [0, 1]
      }
      m.rgba[4 * i + 3] = cam.isOutsideImageCircle(p) ? 0.0f : 1.0f;  // alphaFov, CanopyScene.cpp:465-476
    }
  }
  canopyBuildMips(m);
  return m;
}

// triangle `t` (0: A B C, 1: B C D of the quad at (qx, qy): A = (x, y), B = (x, y+1), C = (x+1, y), D = (x+1, y+1))
static inline bool canopySetup(const CanopyMesh& m, int qx, int qy, int t, const float centre[3], int face, int E,
                               CanopyTri& T) {
  static const int off[2][3][2] = {{{0, 0}, {0, 1}, {1, 0}}, {{0, 1}, {1, 0}, {1, 1}}};
  const float scaleX = (float)(1.0 / m.w), scaleY = (float)(1.0 / m.h);  // Canopy::scale
  for (int k = 0; k < 3; ++k) {
    const int vx = qx + off[t][k][0], vy = qy + off[t][k][1];
    const float* p = &m.v[((size_t)vy * m.w + vx) * 3];
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) {
      return false;
    }
    const float q[3] = {p[0] - centre[0], p[1] - centre[1], p[2] - centre[2]};
    const float d = kCubeAxes[face][0][1] * q[kCubeAxes[face][0][0]];
    const float cx = kCubeAxes[face][1][1] * q[kCubeAxes[face][1][0]];
    const float cy = kCubeAxes[face][2][1] * q[kCubeAxes[face][2][0]];
    if (!(d >= 0.1f)) {  // kNearZ
      return false;
    }
    T.sx[k] = (cx / d + 1.0f) * 0.5f * (float)E;
    T.sy[k] = (cy / d + 1.0f) * 0.5f * (float)E;
    T.invd[k] = 1.0f / d;
    T.tu[k] = scaleX * ((float)vx + 0.5f);
    T.tv[k] = scaleY * ((float)vy + 0.5f);
  }
  T.area = (T.sx[1] - T.sx[0]) * (T.sy[2] - T.sy[0]) - (T.sx[2] - T.sx[0]) * (T.sy[1] - T.sy[0]);
  return T.area != 0.0f && std::isfinite(T.area);
}

// barycentrics of (px, py); returns false when outside (only checked when `test`)
static inline bool canopyBary(const CanopyTri& T, float px, float py, float l[3], bool test) {
  const float e0 = (T.sx[2] - T.sx[1]) * (py - T.sy[1]) - (T.sy[2] - T.sy[1]) * (px - T.sx[1]);
  const float e1 = (T.sx[0] - T.sx[2]) * (py - T.sy[2]) - (T.sy[0] - T.sy[2]) * (px - T.sx[2]);
  const float e2 = (T.sx[1] - T.sx[0]) * (py - T.sy[0]) - (T.sy[1] - T.sy[0]) * (px - T.sx[0]);
  l[0] = e0 / T.area;
  l[1] = e1 / T.area;
  l[2] = e2 / T.area;
  return !test || (l[0] >= 0.0f && l[1] >= 0.0f && l[2] >= 0.0f);
}
static inline float canopyInvZ(const CanopyTri& T, const float l[3]) {
  return l[0] * T.invd[0] + l[1] * T.invd[1] + l[2] * T.invd[2];
}
static inline void canopyTex(const CanopyTri& T, float px, float py, float& u, float& v) {
  float l[3];
  canopyBary(T, px, py, l, false);
  const float iz = canopyInvZ(T, l);
  u = (l[0] * (T.tu[0] * T.invd[0]) + l[1] * (T.tu[1] * T.invd[1]) + l[2] * (T.tu[2] * T.invd[2])) / iz;
  v = (l[0] * (T.tv[0] * T.invd[0]) + l[1] * (T.tv[1] * T.invd[1]) + l[2] * (T.tv[2] * T.invd[2])) / iz;
}
// GL_LINEAR inside mip level `k`, clamp to edge; out = B, G, R, A
static inline void canopyBilinear(const CanopyMesh& m, int k, float u, float v, float out[4]) {
  const int w = m.mipW[k], h = m.mipH[k];
  const float* img = m.rgba.data() + m.mipOff[k];
  const float fx = u * (float)w - 0.5f, fy = v * (float)h - 0.5f;
  const float x0f = std::floor(fx), y0f = std::floor(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int xa = std::min(std::max(x0, 0), w - 1), xb = std::min(std::max(x0 + 1, 0), w - 1);
  const int ya = std::min(std::max(y0, 0), h - 1), yb = std::min(std::max(y0 + 1, 0), h - 1);
  for (int c = 0; c < 4; ++c) {
    const float c00 = img[((size_t)ya * w + xa) * 4 + c], c10 = img[((size_t)ya * w + xb) * 4 + c];
    const float c01 = img[((size_t)yb * w + xa) * 4 + c], c11 = img[((size_t)yb * w + xb) * 4 + c];
    const float top = c00 * (1.0f - ax) + c10 * ax, bot = c01 * (1.0f - ax) + c11 * ax;
    out[c] = top * (1.0f - ay) + bot * ay;
  }
}

// texture(sampler, texVar) under GL_LINEAR_MIPMAP_LINEAR + maximum anisotropy; (ax, ay) = dFdx(texVar),
// (bx, by) = dFdy(texVar), in normalised texture coordinates
static inline void canopySample(const CanopyMesh& m, float u, float v, float ax, float ay, float bx, float by, float out[4]) {
  const float axT = ax * (float)m.w, ayT = ay * (float)m.h, bxT = bx * (float)m.w, byT = by * (float)m.h;
  const float px2 = axT * axT + ayT * ayT, py2 = bxT * bxT + byT * byT;
  const bool xMajor = px2 >= py2;
  const float pMax = std::sqrt(xMajor ? px2 : py2), pMin = std::sqrt(xMajor ? py2 : px2);
  if (!(pMax > 0.0f) || !std::isfinite(pMax)) {
    canopyBilinear(m, 0, u, v, out);
    return;
  }
  int n = kCanopyMaxAniso;
  if (pMin > 0.0f) {
    const float r = std::ceil(pMax / pMin);
    n = r < (float)kCanopyMaxAniso ? (int)r : kCanopyMaxAniso;
  }
  const float rho = pMax / (float)n;
  const int top = (int)m.mipW.size() - 1;
  int level = 0;
  float frac = 0.0f;
  if (rho > 1.0f) {
    int e;
    const float mant = std::frexp(rho, &e);  // rho = mant * 2^e, mant in [0.5, 1)
    level = e - 1;
    frac = 2.0f * mant - 1.0f;
    if (level >= top) {
      level = top;
      frac = 0.0f;
    }
  }
  const float du = xMajor ? ax : bx, dv = xMajor ? ay : by;
  float sum[4] = {0, 0, 0, 0};
  for (int i = 1; i <= n; ++i) {
    const float t = (float)i / (float)(n + 1) - 0.5f;
    const float uu = u + du * t, vv = v + dv * t;
    float a[4], b[4];
    canopyBilinear(m, level, uu, vv, a);
    if (frac > 0.0f) {
      canopyBilinear(m, level + 1, uu, vv, b);
      for (int c = 0; c < 4; ++c) {
        a[c] = a[c] * (1.0f - frac) + b[c] * frac;
      }
    }
    for (int c = 0; c < 4; ++c) {
      sum[c] += a[c];
    }
  }
  for (int c = 0; c < 4; ++c) {
    out[c] = sum[c] / (float)n;
  }
}

// texVar and its fine 2x2-quad derivatives at pixel (i, j) of triangle T
static inline void canopyGrad(const CanopyTri& T, int i, int j, float& u, float& v, float& ax, float& ay, float& bx,
                              float& by) {
  canopyTex(T, i + 0.5f, j + 0.5f, u, v);
  const int ib = i & ~1, jb = j & ~1;
  float ua, va, ub, vb;
  canopyTex(T, ib + 0.5f, j + 0.5f, ua, va);
  canopyTex(T, ib + 1.5f, j + 0.5f, ub, vb);
  ax = ub - ua;  // dFdx(texVar)
  ay = vb - va;
  canopyTex(T, i + 0.5f, jb + 0.5f, ua, va);
  canopyTex(T, i + 0.5f, jb + 1.5f, ub, vb);
  bx = ub - ua;  // dFdy(texVar)
  by = vb - va;
}

// cameras `include[s] != 0` rendered from `centre` -> BGRA float [6 * E][E]
static inline void canopyCubemap(const Rig& rig, const uint16_t* const* colors, const float* const* disps, int w, int h,
                                 const uint8_t* include, const double centreD[3], int E, float* out) {
  const float centre[3] = {(float)centreD[0], (float)centreD[1], (float)centreD[2]};  // position.cast<float>()
  std::vector<CanopyMesh> meshes(rig.size());
  for (size_t s = 0; s < rig.size(); ++s) {
    if (include[s]) {
      meshes[s] = canopyMesh(rig[s], colors[s], disps[s], w, h);
    }
  }
  const size_t nFace = (size_t)E * E;
  std::vector<float> acc(nFace * 4);
  std::vector<uint64_t> zbuf(nFace);
  for (int face = 0; face < 6; ++face) {
    std::fill(acc.begin(), acc.end(), 0.0f);
    for (size_t s = 0; s < rig.size(); ++s) {
      if (!include[s]) {
        continue;
      }
      const CanopyMesh& m = meshes[s];
      std::fill(zbuf.begin(), zbuf.end(), 0ull);
      for (int qy = 0; qy + 1 < h; ++qy) {
        for (int qx = 0; qx + 1 < w; ++qx) {
          for (int t = 0; t < 2; ++t) {
            CanopyTri T;
            if (!canopySetup(m, qx, qy, t, centre, face, E, T)) {
              continue;
            }
            const float minx = std::min(T.sx[0], std::min(T.sx[1], T.sx[2])), maxx = std::max(T.sx[0], std::max(T.sx[1], T.sx[2]));
            const float miny = std::min(T.sy[0], std::min(T.sy[1], T.sy[2])), maxy = std::max(T.sy[0], std::max(T.sy[1], T.sy[2]));
            if (!(maxx >= 0.0f && maxy >= 0.0f && minx <= (float)E && miny <= (float)E)) {
              continue;
            }
            const int i0 = std::max(0, (int)std::ceil(minx - 0.5f)), i1 = std::min(E - 1, (int)std::floor(maxx - 0.5f));
            const int j0 = std::max(0, (int)std::ceil(miny - 0.5f)), j1 = std::min(E - 1, (int)std::floor(maxy - 0.5f));
            const uint32_t triId = (uint32_t)(((size_t)qy * w + qx) * 2 + t);
            for (int j = j0; j <= j1; ++j) {
              for (int i = i0; i <= i1; ++i) {
                float l[3];
                if (!canopyBary(T, i + 0.5f, j + 0.5f, l, true)) {
                  continue;
                }
                const float iz = canopyInvZ(T, l);
                if (!(iz > 0.0f)) {
                  continue;
                }
                float u, v, ax, ay, bx, by, c[4];
                canopyGrad(T, i, j, u, v, ax, ay, bx, by);
                canopySample(m, u, v, ax, ay, bx, by, c);
                if (c[3] == 0.0f) {
                  continue;  // discard: no colour, no depth
                }
                uint32_t bits;
                std::memcpy(&bits, &iz, 4);
                const uint64_t key = ((uint64_t)bits << 32) | triId;  // nearer = larger 1/z; ties: the later triangle
                uint64_t& slot = zbuf[(size_t)j * E + i];
                slot = std::max(slot, key);
              }
            }
          }
        }
      }
      // resolve the canopy and accumulate it
      for (int j = 0; j < E; ++j) {
        for (int i = 0; i < E; ++i) {
          const uint64_t key = zbuf[(size_t)j * E + i];
          if (!key) {
            continue;
          }
          const uint32_t triId = (uint32_t)key;
          const int t = triId & 1, q = triId >> 1, qx = q % w, qy = q / w;
          CanopyTri T;
          canopySetup(m, qx, qy, t, centre, face, E, T);
          float u, v, ax, ay, bx, by, c[4];
          canopyGrad(T, i, j, u, v, ax, ay, bx, by);
          canopySample(m, u, v, ax, ay, bx, by, c);
          const float aa = ax * ax + ay * ay, bb = bx * bx + by * by, ab = ax * bx + ay * by;
          const float hx = (aa - bb) / 2.0f;
          const float minor = (aa + bb) / 2.0f - std::sqrt(hx * hx + ab * ab);
          float alpha = c[3] * minor;
          const float du = u - 0.5f, dv = v - 0.5f;
          const float cone = std::max(1.0f / 255.0f, 1.0f - 2.0f * std::sqrt(du * du + dv * dv));
          alpha *= cone;
          const float weight = std::exp(30.0f * alpha) - 1.0f;  // accumulateFS, kLogK = 30
          float* a = &acc[((size_t)j * E + i) * 4];
          a[0] = weight * c[0] + a[0];  // GL_SRC_ALPHA, GL_ONE
          a[1] = weight * c[1] + a[1];
          a[2] = weight * c[2] + a[2];
          a[3] = weight + a[3];
        }
      }
    }
    // un-premultiply, NaN -> 0, GL row j (bottom-up) -> image row E - 1 - j of face `face`
    for (int j = 0; j < E; ++j) {
      for (int i = 0; i < E; ++i) {
        const float* a = &acc[((size_t)j * E + i) * 4];
        float* o = out + (((size_t)face * E + (E - 1 - j)) * E + i) * 4;
        for (int c = 0; c < 4; ++c) {
          const float v = a[c] / a[3];
          o[c] = v != v ? 0.0f : v;
        }
      }
    }
  }
}

}  // namespace oracle
