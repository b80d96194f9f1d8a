// HIP kernels of the depth_estimation hot path for gfx950 (MI355X / CDNA4).
//
// Arithmetic follows the reference operation for operation (see each kernel's citation);
// the translation unit is compiled with -ffp-contract=off (the reference's x86-64 build has
// no FMA) and with correctly rounded fp32 divide/sqrt. MFMA is not used: the path is
// gather / compare, not a dense contraction.
//
// HBM layout (per level):
//   srcColor   [S][H][W]            ushort4 (B,G,R,0)          8 B texels, one aligned load
//   ownBias    [S][H][W]            ushort4  3x3 box of srcColor (dstProjColorBias(dst))
//   projWarp   [DB][S-1][H+2][W+2]  float2, 1-texel replicated ring  (clamp-to-edge taps
//                                    of getPixelBilinear become plain loads)
//   projColor  [DB][S-1][H+4][W+4]  ushort4, 2-texel replicated ring
//   projBias   [DB][S-1][H+4][W+4]  ushort4, 2-texel replicated ring
//   disparity / cost / confidence / variance  float [.][H][W]; masks uint8
// A wave covers an 8x8 pixel tile (the cost kernels run one wave per block) so that the 64 lanes'
// gathers into one source table fall into a few neighbouring cache lines; blocks are
// remapped so that each XCD (own L2) walks a contiguous band of tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "derp_camera.h"
#include "gcc_algos.h"

namespace derp {

// waves per SIMD the register allocator must leave room for (__launch_bounds__). Round 6: FOUR for every cost kernel
// (<= 128 VGPRs) — the destination patch lives in LDS (PatchWin), the 4x4 block is streamed by columns (no parked sums),
// and the candidate loops carry a handful of values. Rounds 2-5 ran three (164-167 VGPRs), an odd count at which a plain
// fp32 op costs 2.8 instead of 2.08 cycles (tools/valu_ubench.hip). Measured on one box (profiles/r06_kernel_variants.txt):
// level-0 ping-pong 59.8 -> 55.0 ms, random proposals 42.3 -> 39.9 ms per frame; the same sources held to three waves:
// 61.8 / 42.0. What still spills at 128 is cold: a few per-pixel / per-candidate values around the source loops.
#ifndef DERP_COST_MIN_WAVES
#define DERP_COST_MIN_WAVES 4
#endif
#ifndef DERP_RANDOM_MIN_WAVES
#define DERP_RANDOM_MIN_WAVES 4
#endif
#ifndef DERP_COST_BLOCK
#define DERP_COST_BLOCK 64
#endif
#ifndef DERP_TILE_BLOCK
#define DERP_TILE_BLOCK 4
#endif
// skip, per wave, the sources that face away from the wave's pixels for every candidate depth (behind_sources)
#ifndef DERP_SOURCE_CULL
#define DERP_SOURCE_CULL 1
#endif
// fp64 atan2 / division of the cost kernels' projection through the short routines of derp_camera.h (0 = the
// device library's)
#ifndef DERP_LEAN_PROJ
#define DERP_LEAN_PROJ 1
#endif
// the short atan2's per-interval constants from a table in LDS (1) or from scalar literals chosen by branching (0)
#ifndef DERP_ATAN_LUT
#define DERP_ATAN_LUT 1
#endif
// tools/valu_model.py only (never a product build): compile the cost kernels without their cold paths — the
// tap-by-tap SSD fallback and the non-FTHETA camera types — so that the static instruction mix it weights the
// typed VALU counters with is the mix of the hot loops
#ifndef DERP_MIX_HOT_ONLY
#define DERP_MIX_HOT_ONLY 0
#endif
// computeSSD's 4x4-block arithmetic in plain fp32 (1) or packed fp32 (0), per kernel family
// random proposals: each XCD walks its own band of tiles (1, like the coherent kernels) or all XCDs sweep the image
// together (0: consecutive blocks, which the hardware deals round-robin to the XCDs, are neighbouring tiles)
#ifndef DERP_RANDOM_SWIZZLE
#define DERP_RANDOM_SWIZZLE 1
#endif
#ifndef DERP_RANDOM_SSD_SCALAR
#define DERP_RANDOM_SSD_SCALAR 1
#endif
#ifndef DERP_COST_SSD_SCALAR
#define DERP_COST_SSD_SCALAR 0
#endif
// random proposals, round 5: every lane of a wave gathers at its own place on its epipolar curve; the kernel's
// gathers miss L2 (66 % hits, 141 GB through the memory-side counters per level-0 launch at config 2 in round 4).
//  DERP_RANDOM_BLOCK_BIAS (on)  srcBias = bilinear sample of projBias = 2x2 taps of the 3x3 box of projColor — which are
//                        sums over exactly the 4x4 block already in registers: the two projBias loads (2.1 cache lines
//                        per pair, and a third of the kernel's table footprint) become ~110 VALU instructions. Exact:
//                        the box is an integer sum < 2^24 (exact in fp32) and trunc((s + 4) * fl(1/9)) == (s + 4) / 9
//                        for every s <= 9 * 65535 (checked exhaustively, tests/test_abi.py). Taps on the image's first /
//                        last row or column (BORDER_REFLECT_101 there, a replicated ring in the table) load projBias.
//                        Measured (profiles/r05_kernel_variants.txt): 142 -> 53 GB per launch, L2 hits 66 -> 82 %,
//                        config 4's random proposals 485 -> 367 ms.
//  DERP_RANDOM_TILED (off: measured, rejected)  the 4x4 block from a second copy of projColor stored in 4x4-texel tiles
//                        of one 128-byte line each (k_reproject_bias writes both): 3.06 lines per block instead of
//                        4.75, but sixteen 8-byte loads instead of eight 16-byte ones. 53 -> 48 GB, and slower: +2 %
//                        at config 2; at config 4 the copy costs a third destination batch (+25 % table bytes) and the
//                        tiled stores +25 ms of k_reproject_bias.
#ifndef DERP_RANDOM_TILED
#define DERP_RANDOM_TILED 0
#endif
#ifndef DERP_RANDOM_BLOCK_BIAS
#define DERP_RANDOM_BLOCK_BIAS 1
#endif
// ping-pong: read the pixel's ray direction from its table per candidate instead of keeping it (in scratch) across the loop
#ifndef DERP_PP_RELOAD_RAY
#define DERP_PP_RELOAD_RAY 1
#endif
// a wave whose lanes are all in computeSSD's exact case runs a copy of the 4x4-block arithmetic specialised for it
#ifndef DERP_UNIFORM_WEIGHTS
#define DERP_UNIFORM_WEIGHTS 1
#endif
// ... also of the plain-fp32 form (random proposals): + 7 VGPRs there, i.e. the third wave unless they are found elsewhere
#ifndef DERP_UNIFORM_WEIGHTS_SCALAR
#define DERP_UNIFORM_WEIGHTS_SCALAR 0
#endif
// random proposals: the same (round 6: six registers the fourth wave needs)
#ifndef DERP_RANDOM_RELOAD_RAY
#define DERP_RANDOM_RELOAD_RAY 1
#endif
#ifndef DERP_RANDOM_RECONVERT
#define DERP_RANDOM_RECONVERT 1
#endif
// developer A/B: 1 = compute and store every inverse warp (round 4) instead of reading projWarp(ds, own) where source s
// is a destination of the batch
#ifndef DERP_NO_WARP_IDENTITY
#define DERP_NO_WARP_IDENTITY 0
#endif
static constexpr int kPadW = 1;   // ring of projWarp
static constexpr int kPadC = 2;   // ring of projColor / projBias
static constexpr int kMaxSrc = 32;
static constexpr int kNumDepths = 150;                 // Derp.h:33
static constexpr float kMinVar = 1.0f / 12.0f / 65025.0f;  // DerpUtil.h:32

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned int u4a8 __attribute__((ext_vector_type(4), aligned(8)));

struct LevelView {
  int W, H, S, D;             // D = dst cameras in this batch
  double Wd, Hd;              // (double)W, (double)H: kernel arguments, i.e. scalar registers in the cost kernels' projection
  int level, numLevels;
  int dst0;                   // first dst of the batch (global dst index = dst0 + dl)
  int hasFg;
  float varNoiseFloor, varHighThresh;
  float minDepthM, maxDepthM;
  int randomProposals;
  int partialCoverage;
  int xcdRotate;              // rotate the XCD -> band map per destination (load balance)
  const Cam* camsSrc;         // [S] normalised
  const Cam* camsDst;         // [Dtotal] normalised
  const int* dst2src;         // [Dtotal]
  const ushort4* srcColor;    // [S][H*W]
  const ushort4* ownBias;     // [S][H*W]
  const float* srcVar;        // [S][H*W]
  const uint8_t* srcFg;       // [S][H*W]
  const double* rayDir;       // [3][Dtotal][H*W] dst ray direction per pixel, planar (k_pixel_rays)
  const unsigned* behind;     // [Dtotal][H*W] slots of the sources that face away from the pixel's ray
  size_t rayStride;           // Dtotal * H * W
  const float2* projWarp;     // [D][S-1] padded
  const ushort4* projColor;
  const ushort4* projBias;
  const ushort4* projColorT;  // projColor again in 4x4-texel tiles (tile_index); null = not kept
  // per dst (global index)
  float* disparity;           // [Dtotal][H*W]
  float* cost;
  float* confidence;
  const float* bgDisp;        // [Dtotal][H*W] or null
  const uint8_t* fovMask;     // [Dtotal][H*W]
  uint8_t* pairCount;            // [Dtotal][H*W] #pairs behind cost[] where random proposals evaluated the pixel
  unsigned long long* counters;  // [0] nCost [1] nPair [2] insufficient coverage [3] cost evaluations served from memo
};

__device__ __forceinline__ size_t warp_plane(const LevelView& V) {
  return (size_t)(V.W + 2 * kPadW) * (V.H + 2 * kPadW);
}
__device__ __forceinline__ size_t color_plane(const LevelView& V) {
  return (size_t)(V.W + 2 * kPadC) * (V.H + 2 * kPadC);
}
// projColorT: the padded (W + 4) x (H + 4) texel grid of a projColor plane in 4x4-texel tiles, one 128-byte line each
__host__ __device__ __forceinline__ int tiled_tiles_x(int W) {
  return (W + 2 * kPadC + 3) >> 2;
}
__host__ __device__ __forceinline__ size_t tiled_plane(int W, int H) {
  return (size_t)tiled_tiles_x(W) * (size_t)((H + 2 * kPadC + 3) >> 2) * 16;
}
// texel index of padded coordinates (ox, oy)
__device__ __forceinline__ size_t tiled_index(int tilesX, int ox, int oy) {
  return ((size_t)(oy >> 2) * tilesX + (ox >> 2)) * 16 + (size_t)((oy & 3) * 4 + (ox & 3));
}
__device__ __forceinline__ int slot(int s, int own) {
  return s < own ? s : s - 1;
}
// element `i` of a per-destination plane through a 32-bit BYTE offset from its (wave-uniform) base: the access keeps the
// scalar-base + 32-bit-offset form, and no 64-bit per-lane address (a register pair the allocator parks in scratch) exists.
// A plane is far below 4 GB.
template <typename T>
__device__ __forceinline__ T& at32(T* base, unsigned i) {
  return *reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + i * (unsigned)sizeof(T));
}

// XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give block b
// the tile (b % 8) * ceil(n/8) + b / 8 — each XCD's L2 then serves one contiguous band.
// `rot` rotates which band an XCD gets (callers pass the destination index): the top and bottom bands of
// every image hold the clipped FOV-circle corners, i.e. less work, and without the rotation the same two
// XCDs would draw them for every destination of the launch and idle at its end.
__device__ __forceinline__ int xcd_swizzle(int b, int n, int rot) {
  const int per = (n + 7) >> 3;
  const int t = (((b & 7) + rot) & 7) * per + (b >> 3);
  return t;
}

// Pixel handled by this thread. A wave owns an 8x8 pixel tile; four consecutive waves form a 16x16
// super-tile (raster order over super-tiles). `item` numbers the blocks of the launch; a block holds
// blockDim.x / 64 consecutive waves. Cost kernels run ONE wave per block (DERP_COST_BLOCK = 64): the
// waves of a pixel tile finish at very different times, and a single-wave block frees its slot at once.
__device__ __forceinline__ void tile_pixel(int item, int tilesX, int& x, int& y) {
  const int lane = threadIdx.x & 63;
  const int gw = item * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  const int super = gw >> 2, quad = gw & 3;
  // super-tiles are walked in DERP_TILE_BLOCK x DERP_TILE_BLOCK squares (128 x 128 px for 8): the tiles an
  // XCD works on at one time then cover a compact square instead of a 16-px-high strip, which shrinks
  // the union of the source-image footprints their gathers fall into (tilesX is padded to the block size)
  constexpr int B = DERP_TILE_BLOCK;
  const int blk = super / (B * B), in = super % (B * B);
  const int blocksX = tilesX / B;
  const int tx = (blk % blocksX) * B + (in % B), ty = (blk / blocksX) * B + (in / B);
  x = tx * 16 + (quad & 1) * 8 + (lane & 7);
  y = ty * 16 + (quad >> 1) * 8 + (lane >> 3);
}

// ----------------------------------------------------------------------------------------
// bilinear helpers — CvUtil.h:78-120. Weights: x - round(x) + 0.5; taps (xi-1, yi-1)..(xi, yi)
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float bilerp_f(float p00, float p01, float p10, float p11, float w00, float w01,
                                          float w10, float w11) {
  return w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11;
}
// scalar bilerp<T = ushort>: float expression truncated to ushort
__device__ __forceinline__ float bilerp_u16(float p00, float p01, float p10, float p11, float w00, float w01,
                                            float w10, float w11) {
  // v lies in [0, 65535 * (1 + 2^-22)], so the reference's (ushort) conversion is a plain truncation
  const float v = w00 * p00 + w01 * p01 + w10 * p10 + w11 * p11;
  return __builtin_truncf(v);
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float a) {
  return (v2f){a, a};
}
__device__ __forceinline__ v2f trunc2(v2f v) {
  return (v2f){__builtin_truncf(v.x), __builtin_truncf(v.y)};
}
// low / high ushort of a packed texel word as (B, G)
__device__ __forceinline__ v2f bg_of(unsigned u) {
  return (v2f){(float)(u & 0xffff), (float)(u >> 16)};
}

// The destination colours a wave's 8x8 pixel tile compares against — the 3x3 patches of its 64 pixels overlap, so the
// wave keeps ONE 10x10 window of its destination camera's colour (as floats) in LDS instead of 27 floats per lane in
// registers (round 6: those 27 + the 18 parked per-offset sums were what pinned the cost kernels at 164-167 VGPRs, an odd
// three waves per SIMD). (B, G) and R live in separate planes: a (B, G) pair arrives as one aligned register pair
// (ds_read_b64), and R of the offsets dy = -1 / +1 as one pair too (ds_read2_b32, two rows of the R plane).
static constexpr int kWinW = 10, kWinTexels = kWinW * kWinW;
struct PatchWin {
  v2f bg[kWinTexels];
  float r[kWinTexels];
};

struct PixCtx {
  D3 rayO, rayD;        // dst ray (Camera::rig(pixel) of the dst pixel centre)
  // dst colour 3x3: offset (dx, dy) = (ix - 1, iy - 1) is texel [iy * kWinW + ix] from the lane's corner of the window
  const v2f* winBG;
  const float* winR;
  v2f dstBiasBG;
  float dstBiasR;
  __device__ __forceinline__ v2f patchBG(int ix, int iy) const {
    return winBG[iy * kWinW + ix];
  }
  __device__ __forceinline__ float patchR(int ix, int iy) const {
    return winR[iy * kWinW + ix];
  }
  __device__ __forceinline__ float patch(int ix, int iy, int c) const {
    return c == 2 ? patchR(ix, iy) : c == 0 ? patchBG(ix, iy).x : patchBG(ix, iy).y;
  }
  __device__ __forceinline__ float dstBias(int c) const {
    return c == 0 ? dstBiasBG.x : c == 1 ? dstBiasBG.y : dstBiasR;
  }
  float confidence;     // max(variance, kMinVar)
};

// Fill the wave's window: texel (i, j) = pixel (x0 - 1 + i, y0 - 1 + j) of camera `own`, clamped to the image (a clamped
// texel is only ever part of the patch of a border pixel, and those never reach computeCost). Every lane of the wave
// takes part; the caller's barrier (one wave per block: a wait on the LDS queue) publishes it.
__device__ __forceinline__ void patch_window_fill(const LevelView& V, int own, int x0, int y0, PatchWin* win) {
  const ushort4* col = V.srcColor + (size_t)own * ((size_t)V.W * V.H);
  for (int t = (int)(threadIdx.x & 63); t < kWinTexels; t += 64) {
    const int j = t / kWinW, i = t - j * kWinW;
    const int xx = min(max(x0 - 1 + i, 0), V.W - 1), yy = min(max(y0 - 1 + j, 0), V.H - 1);
    const ushort4 q = col[(size_t)yy * V.W + xx];
    win->bg[t] = (v2f){(float)q.x, (float)q.y};
    win->r[t] = (float)q.z;
  }
}

// selection scratch in LDS: pairs[i][thread]
struct LdsPairs {
  SsdPair* base;
  int stride;
  const double* atanLut;  // atan_lut_fill's table, or null: select the constants from scalar literals
  __device__ __forceinline__ SsdPair get(int i) const {
    return base[i * stride];
  }
  __device__ __forceinline__ void set(int i, const SsdPair& v) {
    base[i * stride] = v;
  }
};

// The texels one computeSSD call reads, requested ahead of the arithmetic: the 4x4 block of projColor
// around round(x, y) (rows yi-2..yi+1, cols xi-2..xi+1) and the 2x2 bias taps, as ten 16-byte loads. The
// block lies inside the padded table for every x, y the tables can produce, and only the rare path at the
// bottom of ssd_arith (taps not block shaped) ignores it.
struct SsdTexels {
  u4a8 raw[4][2];
  u4a8 ba, bb;
};
// The random-proposal form: the block from the tiled copy `colT` (sixteen 8-byte loads: a texel row of the block can
// straddle two tiles), the bias taps only for lanes whose taps touch the image's first / last row or column
// (`border`; everywhere else ssd_arith sums them from the block).
template <bool TILED, bool BLOCK_BIAS>
__device__ __forceinline__ void ssd_issue_random(const LevelView& V, const ushort4* __restrict__ col, const ushort4* __restrict__ colT,
                                                 const ushort4* __restrict__ bia, float xDstSrc, float yDstSrc, SsdTexels& T,
                                                 bool& border) {
  const int pitch = V.W + 2 * kPadC;
  const int xi = (int)roundf(xDstSrc), yi = (int)roundf(yDstSrc);
  const unsigned off = ((unsigned)(yi - 2 + kPadC) * (unsigned)pitch + (unsigned)(xi - 2 + kPadC)) * 8u;
  if constexpr (TILED) {
    const unsigned xa = (unsigned)(xi - 2 + kPadC), ya = (unsigned)(yi - 2 + kPadC);
    const unsigned rowTiles = (unsigned)tiled_tiles_x(V.W) * 128u;  // bytes of one row of tiles
    const unsigned xr = xa & 3u, yr = ya & 3u;
    const unsigned base0 = (ya >> 2) * rowTiles + (xa >> 2) * 128u + yr * 32u + xr * 8u;
    const char* base = reinterpret_cast<const char*>(colT);
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      const unsigned ro = base0 + (unsigned)row * 32u + ((yr + (unsigned)row > 3u) ? rowTiles - 128u : 0u);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned o = ro + (unsigned)k * 8u + ((xr + (unsigned)k > 3u) ? 96u : 0u);
        const u2 t = *reinterpret_cast<const u2*>(base + o);
        if (k & 1) {
          T.raw[row][k >> 1].z = t.x;
          T.raw[row][k >> 1].w = t.y;
        } else {
          T.raw[row][k >> 1].x = t.x;
          T.raw[row][k >> 1].y = t.y;
        }
      }
    }
  } else {
    const char* base = reinterpret_cast<const char*>(col);
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      T.raw[row][0] = *reinterpret_cast<const u4a8*>(base + (off + (unsigned)row * (unsigned)pitch * 8u));
      T.raw[row][1] = *reinterpret_cast<const u4a8*>(base + (off + (unsigned)row * (unsigned)pitch * 8u + 16u));
    }
  }
  // bias taps (xi - 1 .. xi, yi - 1 .. yi): their 3x3 boxes lie inside the image — where the table's box (BORDER_REFLECT_101)
  // and the block's (replicated ring) are the same nine texels — iff 2 <= xi <= W - 2 and 2 <= yi <= H - 2
  border = !BLOCK_BIAS || xi < 2 || yi < 2 || xi > V.W - 2 || yi > V.H - 2;
  T.ba = T.bb = (u4a8){0u, 0u, 0u, 0u};
  if (border) {
    const unsigned boff = off + ((unsigned)pitch + 1u) * 8u;  // (yi - 1, xi - 1)
    const char* bbase = reinterpret_cast<const char*>(bia);
    T.ba = *reinterpret_cast<const u4a8*>(bbase + boff);
    T.bb = *reinterpret_cast<const u4a8*>(bbase + (boff + (unsigned)pitch * 8u));
  }
}
__device__ __forceinline__ void ssd_issue(const LevelView& V, const ushort4* __restrict__ col,
                                          const ushort4* __restrict__ bia, float xDstSrc, float yDstSrc, SsdTexels& T) {
  const int pitch = V.W + 2 * kPadC;
  const int xi = (int)roundf(xDstSrc), yi = (int)roundf(yDstSrc);
  // one table plane is far below 4 GB: 32-bit byte offsets from the (wave-uniform) plane base let the
  // loads use the scalar-base + 32-bit-offset addressing form instead of 64-bit pointer arithmetic
  const unsigned off = ((unsigned)(yi - 2 + kPadC) * (unsigned)pitch + (unsigned)(xi - 2 + kPadC)) * 8u;
  const char* base = reinterpret_cast<const char*>(col);
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    T.raw[row][0] = *reinterpret_cast<const u4a8*>(base + (off + (unsigned)row * (unsigned)pitch * 8u));
    T.raw[row][1] = *reinterpret_cast<const u4a8*>(base + (off + (unsigned)row * (unsigned)pitch * 8u + 16u));
  }
#ifdef DERP_ABLATE_NO_BIAS_LOAD  // developer ablation: same arithmetic, two loads fewer (results are wrong)
  T.ba = T.raw[1][0];
  T.bb = T.raw[2][0];
  return;
#endif
  const unsigned boff = off + ((unsigned)pitch + 1u) * 8u;  // (yi - 1, xi - 1)
  const char* bbase = reinterpret_cast<const char*>(bia);
  T.ba = *reinterpret_cast<const u4a8*>(bbase + boff);
  T.bb = *reinterpret_cast<const u4a8*>(bbase + (boff + (unsigned)pitch * 8u));
}

// computeSSD (DerpUtil.cpp:126-162) for one source whose projected tables are `col` / `bias`; `T` holds the
// texels ssd_issue requested for (xDstSrc, yDstSrc).
template <bool SCALAR, bool BLOCK_BIAS = false>
__device__ __forceinline__ SsdPair ssd_arith(const LevelView& V, const PixCtx& px, const ushort4* __restrict__ col,
                                             const SsdTexels& T, float xDstSrc, float yDstSrc, bool border = true) {
#ifdef DERP_ABLATE_NO_SSD
  return {xDstSrc * 1e-3f, yDstSrc * 1e-3f};
#endif
#ifdef DERP_ABLATE_SSD_LOADS_ONLY  // developer ablation: the ten loads stay, the arithmetic is a few xors
  {
    unsigned a = T.ba.x ^ T.bb.y, b = T.ba.z ^ T.bb.w;
    for (int r = 0; r < 4; ++r) {
      a ^= T.raw[r][0].x ^ T.raw[r][0].z ^ T.raw[r][1].x ^ T.raw[r][1].z;
      b ^= T.raw[r][0].y ^ T.raw[r][0].w ^ T.raw[r][1].y ^ T.raw[r][1].w;
    }
    return {(float)(a & 0xffff) * 1e-3f + xDstSrc * 1e-3f, (float)(b & 0xffff) * 1e-3f + yDstSrc * 1e-3f};
  }
#endif
  const int pitch = V.W + 2 * kPadC;
  const u4a8 (&raw)[4][2] = T.raw;
  // texel (row r, column k) of the 4x4 block: the (B | G << 16) word and the R word
  auto word_bg = [&](const u4a8 (&q)[4][2], int r, int k) { return (k & 1) ? q[r][k >> 1].z : q[r][k >> 1].x; };
  auto word_r = [&](const u4a8 (&q)[4][2], int r, int k) { return (k & 1) ? q[r][k >> 1].w : q[r][k >> 1].y; };
  // --- srcBias = getPixelBilinear(dstSrcColorBias, xDstSrc, yDstSrc)
  float bias[3];
  if constexpr (BLOCK_BIAS) {
    // The four taps are 3x3 boxes of projColor (colorBias, DerpUtil.cpp:208-210: cv::blur on CV_16UC3 = exact integer sum,
    // round(s / 9)) over block columns 0..2 / 1..3 and rows 0..2 / 1..3. Columns are visited 3, 2, 1, 0 so that the floats
    // of columns 1 and 0 are the ones the block arithmetic below starts with; columns 2 and 3 are converted again there
    // (their words pass through an empty asm below: kept live instead, the 24 floats cost the kernel a wave).
    const float xf = roundf(xDstSrc), yf = roundf(yDstSrc);
    const float xw = xDstSrc - xf + 0.5f, yw = yDstSrc - yf + 0.5f;
    const float w00 = (1 - xw) * (1 - yw), w01 = xw * (1 - yw), w10 = (1 - xw) * yw, w11 = xw * yw;
    if constexpr (!SCALAR) {
      // packed form: (B, G) of a texel as one register pair; R of (rows 0..2, rows 1..3) as one pair
      v2f topBG[2], botBG[2], tbR[2];  // [tap column]
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        const v2f m = bg_of(word_bg(raw, 1, k)) + bg_of(word_bg(raw, 2, k));
        const v2f vT = m + bg_of(word_bg(raw, 0, k)), vB = m + bg_of(word_bg(raw, 3, k));
        const float mR = (float)(word_r(raw, 1, k) & 0xffff) + (float)(word_r(raw, 2, k) & 0xffff);
        const v2f vR = splat2(mR) + (v2f){(float)(word_r(raw, 0, k) & 0xffff), (float)(word_r(raw, 3, k) & 0xffff)};
        const v2f four = splat2(4.0f);  // first column of a box: start it at + 4 (round(s / 9) = (s + 4) / 9)
        if (k == 3) {
          topBG[1] = vT + four, botBG[1] = vB + four, tbR[1] = vR + four;
        } else if (k == 2) {
          topBG[1] += vT, botBG[1] += vB, tbR[1] += vR;
          topBG[0] = vT + four, botBG[0] = vB + four, tbR[0] = vR + four;
        } else if (k == 1) {
          topBG[1] += vT, botBG[1] += vB, tbR[1] += vR;
          topBG[0] += vT, botBG[0] += vB, tbR[0] += vR;
        } else {
          topBG[0] += vT, botBG[0] += vB, tbR[0] += vR;
        }
      }
      const v2f ninth = splat2(1.0f / 9.0f);
      const v2f t00 = trunc2(topBG[0] * ninth), t01 = trunc2(topBG[1] * ninth), t10 = trunc2(botBG[0] * ninth),
                t11 = trunc2(botBG[1] * ninth);
      const v2f r0 = trunc2(tbR[0] * ninth), r1 = trunc2(tbR[1] * ninth);  // (top, bottom) of tap column 0 / 1
      const v2f sbBG = trunc2(splat2(w00) * t00 + splat2(w01) * t01 + splat2(w10) * t10 + splat2(w11) * t11);
      const v2f bBG = px.dstBiasBG - sbBG;
      bias[0] = bBG.x;
      bias[1] = bBG.y;
      bias[2] = px.dstBiasR - bilerp_u16(r0.x, r1.x, r0.y, r1.y, w00, w01, w10, w11);
    } else {
    float top[3][2], bot[3][2];  // [channel][tap column]: rows 0..2 and rows 1..3
#pragma unroll
    for (int k = 3; k >= 0; --k) {
      float c[3][4];  // [channel][row] of block column k
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned bg = word_bg(raw, r, k), rr = word_r(raw, r, k);
        c[0][r] = (float)(bg & 0xffff);
        c[1][r] = (float)(bg >> 16);
        c[2][r] = (float)(rr & 0xffff);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float m = c[ch][1] + c[ch][2];
        const float vT = m + c[ch][0], vB = m + c[ch][3];
        if (k == 3) {
          top[ch][1] = vT + 4.0f;  // the + 4 of round(s / 9) = (s + 4) / 9, once per tap
          bot[ch][1] = vB + 4.0f;
        } else if (k == 2) {
          top[ch][1] += vT;
          bot[ch][1] += vB;
          top[ch][0] = vT + 4.0f;
          bot[ch][0] = vB + 4.0f;
        } else if (k == 1) {
          top[ch][1] += vT;
          bot[ch][1] += vB;
          top[ch][0] += vT;
          bot[ch][0] += vB;
        } else {
          top[ch][0] += vT;
          bot[ch][0] += vB;
        }
      }
    }
    const float ninth = 1.0f / 9.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float t00 = __builtin_truncf(top[ch][0] * ninth), t01 = __builtin_truncf(top[ch][1] * ninth);
      const float t10 = __builtin_truncf(bot[ch][0] * ninth), t11 = __builtin_truncf(bot[ch][1] * ninth);
      bias[ch] = px.dstBias(ch) - bilerp_u16(t00, t01, t10, t11, w00, w01, w10, w11);
    }
    }
    if (__ballot(border) != 0ull) {  // taps on the image's rim: the table's values (loaded by ssd_issue_random)
      if (border) {
        const u4a8 a = T.ba, b = T.bb;
        bias[0] = px.dstBias(0) - bilerp_u16((float)(a.x & 0xffff), (float)(a.z & 0xffff), (float)(b.x & 0xffff), (float)(b.z & 0xffff), w00, w01, w10, w11);
        bias[1] = px.dstBias(1) - bilerp_u16((float)(a.x >> 16), (float)(a.z >> 16), (float)(b.x >> 16), (float)(b.z >> 16), w00, w01, w10, w11);
        bias[2] = px.dstBias(2) - bilerp_u16((float)(a.y & 0xffff), (float)(a.w & 0xffff), (float)(b.y & 0xffff), (float)(b.w & 0xffff), w00, w01, w10, w11);
      }
    }
  } else {
    const float xf = roundf(xDstSrc), yf = roundf(yDstSrc);
    const float xw = xDstSrc - xf + 0.5f, yw = yDstSrc - yf + 0.5f;
    const float w00 = (1 - xw) * (1 - yw), w01 = xw * (1 - yw), w10 = (1 - xw) * yw, w11 = xw * yw;
    const u4a8 a = T.ba, b = T.bb;
    // (B, G) as one packed pair, R alone — same per-lane operations as bilerp_u16
    const v2f sbBG = trunc2(splat2(w00) * bg_of(a.x) + splat2(w01) * bg_of(a.z) + splat2(w10) * bg_of(b.x) +
                            splat2(w11) * bg_of(b.z));
    const v2f bBG = px.dstBiasBG - sbBG;
    bias[0] = bBG.x;
    bias[1] = bBG.y;
    bias[2] = px.dstBiasR - bilerp_u16((float)(a.y & 0xffff), (float)(a.w & 0xffff), (float)(b.y & 0xffff),
                                       (float)(b.w & 0xffff), w00, w01, w10, w11);
  }
  // --- the 4x4 texel block arithmetic shared by the two block-shaped paths below. The block is streamed by COLUMNS
  // (round 6): offset column ix needs texel columns ix and ix + 1 only, and computeSSD's loop runs dx outer / dy inner
  // (DerpUtil.cpp:135-136), so the nine per-offset terms are added to the two sums the moment they exist, in the
  // reference's order — streamed by rows (rounds 2-5) the terms came out dy-major and 18 of them were parked in
  // registers until the end. Each operation is the same IEEE operation, in the same order, as the scalar expression it
  // replaces; the destination patch comes from the wave's LDS window (PixCtx).
  float first = 0.f, second = 0.f;
  // (block bias: columns 2 and 3 were converted once already for the box sums; an opaque copy of their words makes the
  // compiler convert them again here instead of keeping 24 more floats alive across the bias arithmetic)
  u4a8 rawB[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rawB[r][0] = raw[r][0];
    rawB[r][1] = raw[r][1];
    if constexpr (BLOCK_BIAS && DERP_RANDOM_RECONVERT) {
      unsigned a = raw[r][1].x, b = raw[r][1].y, c = raw[r][1].z, d = raw[r][1].w;
      asm("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
      rawB[r][1] = (u4a8){a, b, c, d};
    }
  }
  // Plain (unpacked) fp32: needs no register pairs. Random proposals run it (their gathers miss and the kernel wants
  // every wave it can get); at an even number of waves per SIMD a plain fp32 op costs 2.08 cycles, i.e. the same pipe
  // time as half a packed one (tools/valu_ubench.hip).
  auto block_scalar = [&](const float (&xw)[3], const float (&yw)[3]) {
    struct ColS {
      float c[3][4];  // [channel][texel row]
    };
    auto unpack = [&](int k, ColS& t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned bg = word_bg(rawB, r, k), rr = word_r(rawB, r, k);
        t.c[0][r] = (float)(bg & 0xffff);
        t.c[1][r] = (float)(bg >> 16);
        t.c[2][r] = (float)(rr & 0xffff);
      }
    };
    ColS lo, hi;
    unpack(0, lo);
#pragma unroll
    for (int ix = 0; ix < 3; ++ix) {
      unpack(ix + 1, hi);
      const float omx = 1 - xw[ix];
#pragma unroll
      for (int iy = 0; iy < 3; ++iy) {
        const float omy = 1 - yw[iy];
        const float w00 = omx * omy, w01 = xw[ix] * omy, w10 = omx * yw[iy], w11 = xw[ix] * yw[iy];
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float v = w00 * lo.c[c][iy] + w01 * hi.c[c][iy] + w10 * lo.c[c][iy + 1] + w11 * hi.c[c][iy + 1];
          const float db = px.patch(ix, iy, c) - __builtin_truncf(v);
          const float dn = db - bias[c];
          d0 = c == 0 ? db * db : d0 + db * db;
          d1 = c == 0 ? dn * dn : d1 + dn * dn;
        }
        first += d0;
        second += d1;
      }
      lo = hi;
    }
  };
  // Packed fp32 (v_pk_mul_f32 / v_pk_add_f32 work on register pairs): channels B and G of one offset share every
  // instruction, and so do the R channels of the offsets dy = -1 and dy = +1 (texel rows 0 / 2 and 1 / 3 of a column
  // are converted straight into such pairs). ywp = (yw[0], yw[2]), ywm = yw[1]; xw[ix] per offset column.
  auto block_packed = [&](const float (&xw)[3], v2f ywp, float ywm) {
    struct ColF {
      v2f bg[4];   // (B, G) of texel rows 0..3
      v2f rE, rO;  // R of rows (0, 2) and (1, 3)
    };
    auto unpack = [&](int k, ColF& t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t.bg[r] = bg_of(word_bg(rawB, r, k));
      }
      t.rE = (v2f){(float)(word_r(rawB, 0, k) & 0xffff), (float)(word_r(rawB, 2, k) & 0xffff)};
      t.rO = (v2f){(float)(word_r(rawB, 1, k) & 0xffff), (float)(word_r(rawB, 3, k) & 0xffff)};
    };
    const v2f biasBG = (v2f){bias[0], bias[1]};
    const v2f omyp = splat2(1.0f) - ywp;
    const float omym = 1 - ywm;
    ColF lo, hi;
    unpack(0, lo);
#pragma unroll
    for (int ix = 0; ix < 3; ++ix) {
      unpack(ix + 1, hi);
      const float omx = 1 - xw[ix];
      // weights of the offsets dy = -1 / +1 as pairs, of dy = 0 as scalars
      const v2f w00p = splat2(omx) * omyp, w01p = splat2(xw[ix]) * omyp, w10p = splat2(omx) * ywp, w11p = splat2(xw[ix]) * ywp;
      const float w00m = omx * omym, w01m = xw[ix] * omym, w10m = omx * ywm, w11m = xw[ix] * ywm;
      float bgd0[3], bgd1[3];
#pragma unroll
      for (int iy = 0; iy < 3; ++iy) {
        const float w00 = iy == 1 ? w00m : iy == 0 ? w00p.x : w00p.y;
        const float w01 = iy == 1 ? w01m : iy == 0 ? w01p.x : w01p.y;
        const float w10 = iy == 1 ? w10m : iy == 0 ? w10p.x : w10p.y;
        const float w11 = iy == 1 ? w11m : iy == 0 ? w11p.x : w11p.y;
        const v2f v = splat2(w00) * lo.bg[iy] + splat2(w01) * hi.bg[iy] + splat2(w10) * lo.bg[iy + 1] +
                      splat2(w11) * hi.bg[iy + 1];
        const v2f db = px.patchBG(ix, iy) - trunc2(v);
        const v2f dn = db - biasBG;
        const v2f s0 = db * db, s1 = dn * dn;
        // (0 + B) + G as two plain adds over the halves of the pairs; the empty asm keeps the SLP
        // vectoriser from re-pairing them across offsets, which costs three v_mov per v_pk_add
        float t0 = s0.x + s0.y, t1 = s1.x + s1.y;
        asm("" : "+v"(t0), "+v"(t1));
        bgd0[iy] = t0;
        bgd1[iy] = t1;
      }
      v2f e0, e1;  // the complete terms of dy = -1 / +1
      {  // R of dy = -1 / +1: texel rows (0, 1) and (2, 3)
        const v2f v = w00p * lo.rE + w01p * hi.rE + w10p * lo.rO + w11p * hi.rO;
        const v2f db = (v2f){px.patchR(ix, 0), px.patchR(ix, 2)} - trunc2(v);
        const v2f dn = db - splat2(bias[2]);
        e0 = (v2f){bgd0[0], bgd0[2]} + db * db;
        e1 = (v2f){bgd1[0], bgd1[2]} + dn * dn;
      }
      float m0, m1;
      {  // R of dy = 0: texel rows 1 and 2
        const float v = w00m * lo.rO.x + w01m * hi.rO.x + w10m * lo.rE.y + w11m * hi.rE.y;
        const float db = px.patchR(ix, 1) - __builtin_truncf(v);
        const float dn = db - bias[2];
        m0 = bgd0[1] + db * db;
        m1 = bgd1[1] + dn * dn;
      }
      first += e0.x;
      first += m0;
      first += e0.y;
      second += e1.x;
      second += m1;
      second += e1.y;
      lo = hi;
    }
  };
  // --- per-offset tap positions and weights. The reference evaluates round(x + dx) and the weight
  // x + dx - round(x + dx) + 0.5 for each dx in {-1, 0, 1}. When x >= 1 and fl(x + 1) is exact (checked
  // by subtracting the 1 again) all of x - 1, x, x + 1 are exact, so round(x + dx) = round(x) + dx and
  // the three weights are the same float (x + dx - round(x + dx) is exact by Sterbenz's lemma and equal
  // to x - round(x)): one rounding and one weight serve the three offsets. That holds except on the
  // first texel column / row and where x + 1 crosses into a binade that drops x's last bit.
  const float xfc = roundf(xDstSrc), yfc = roundf(yDstSrc);
  const bool exact = (xDstSrc >= 1.0f) && (yDstSrc >= 1.0f) && ((xDstSrc + 1.0f) - 1.0f == xDstSrc) &&
                     ((yDstSrc + 1.0f) - 1.0f == yDstSrc);
  int xi[3], yi[3];
  float xw[3], yw[3];
  bool regular = true;
  xi[1] = (int)xfc;
  yi[1] = (int)yfc;
  xw[0] = xw[1] = xw[2] = xDstSrc - xfc + 0.5f;
  yw[0] = yw[1] = yw[2] = yDstSrc - yfc + 0.5f;
  if (!exact) {
#pragma unroll
    for (int k = 0; k < 3; k += 2) {
      const float xs = xDstSrc + (float)(k - 1), ys = yDstSrc + (float)(k - 1);
      const float xf = roundf(xs), yf = roundf(ys);
      xi[k] = (int)xf;
      yi[k] = (int)yf;
      xw[k] = xs - xf + 0.5f;
      yw[k] = ys - yf + 0.5f;
    }
    regular = (xi[0] == xi[1] - 1) && (xi[2] == xi[1] + 1) && (yi[0] == yi[1] - 1) && (yi[2] == yi[1] + 1);
  }
  if (DERP_UNIFORM_WEIGHTS && (!SCALAR || DERP_UNIFORM_WEIGHTS_SCALAR) && __ballot(!exact) == 0ull) {
    // every lane of the wave is in the exact case (all but ~1 % of the waves: x or y inside [2^k - 1, 2^k) or below 1
    // breaks it): ONE weight per axis serves the nine offsets, and this copy of the block says so at compile time —
    // the four weight products are formed once, and no values of the general path have to be merged in (three dozen
    // register moves per call in the shared copy)
    const float xwu[3] = {xw[1], xw[1], xw[1]};
    if constexpr (SCALAR) {
      const float ywu[3] = {yw[1], yw[1], yw[1]};
      block_scalar(xwu, ywu);
    } else {
      block_packed(xwu, splat2(yw[1]), yw[1]);
    }
  } else if (regular) {
    if constexpr (SCALAR) {
      block_scalar(xw, yw);
    } else {
      block_packed(xw, (v2f){yw[0], yw[2]}, yw[1]);
    }
  } else if (DERP_MIX_HOT_ONLY) {
    first = second = 0.f;
  } else {
    // float rounding of x + dx crossed a .5 boundary: taps no longer form a 4x4 block
    for (int ix = 0; ix < 3; ++ix) {
      for (int iy = 0; iy < 3; ++iy) {
        const float w00 = (1 - xw[ix]) * (1 - yw[iy]), w01 = xw[ix] * (1 - yw[iy]);
        const float w10 = (1 - xw[ix]) * yw[iy], w11 = xw[ix] * yw[iy];
        const int x0 = min(max(xi[ix] - 1, -kPadC), V.W + kPadC - 1), x1 = min(max(xi[ix], -kPadC), V.W + kPadC - 1);
        const int y0 = min(max(yi[iy] - 1, -kPadC), V.H + kPadC - 1), y1 = min(max(yi[iy], -kPadC), V.H + kPadC - 1);
        const ushort4 q00 = col[(size_t)(y0 + kPadC) * pitch + x0 + kPadC];
        const ushort4 q01 = col[(size_t)(y0 + kPadC) * pitch + x1 + kPadC];
        const ushort4 q10 = col[(size_t)(y1 + kPadC) * pitch + x0 + kPadC];
        const ushort4 q11 = col[(size_t)(y1 + kPadC) * pitch + x1 + kPadC];
        const float p00[3] = {(float)q00.x, (float)q00.y, (float)q00.z};
        const float p01[3] = {(float)q01.x, (float)q01.y, (float)q01.z};
        const float p10[3] = {(float)q10.x, (float)q10.y, (float)q10.z};
        const float p11[3] = {(float)q11.x, (float)q11.y, (float)q11.z};
        // the patch texel of this offset (dynamic window index: the loop is not unrolled)
        const v2f pbg = px.winBG[iy * kWinW + ix];
        const float pch[3] = {pbg.x, pbg.y, px.winR[iy * kWinW + ix]};
        float d0 = 0.f, d1 = 0.f;
        for (int c = 0; c < 3; ++c) {
          const float cs = bilerp_u16(p00[c], p01[c], p10[c], p11[c], w00, w01, w10, w11);
          const float db = pch[c] - cs;
          const float dn = db - bias[c];
          d0 += db * db;
          d1 += dn * dn;
        }
        first += d0;
        second += d1;
      }
    }
  }
  const float scale = 1.0f / (65535.0f * 65535.0f);
  return {first * scale, second * scale};
}
// RANDOM: the random-proposal form (tiled block, bias from the block) when the context keeps the tiled copy `colT`
template <bool SCALAR, bool RANDOM = false>
__device__ __forceinline__ SsdPair compute_ssd(const LevelView& V, const PixCtx& px, const ushort4* __restrict__ col,
                                               const ushort4* __restrict__ bia, float xDstSrc, float yDstSrc,
                                               const ushort4* __restrict__ colT = nullptr) {
  SsdTexels T;
  if constexpr (RANDOM && (DERP_RANDOM_TILED || DERP_RANDOM_BLOCK_BIAS)) {
    bool border;
    ssd_issue_random<DERP_RANDOM_TILED != 0, DERP_RANDOM_BLOCK_BIAS != 0>(V, col, colT, bia, xDstSrc, yDstSrc, T, border);
    return ssd_arith<SCALAR, DERP_RANDOM_BLOCK_BIAS != 0>(V, px, col, T, xDstSrc, yDstSrc, border);
  } else {
    ssd_issue(V, col, bia, xDstSrc, yDstSrc, T);
    return ssd_arith<SCALAR>(V, px, col, T, xDstSrc, yDstSrc);
  }
}

// Sources that cannot see this wave's pixels at ANY candidate depth: the pixel's ray O + t D, pushed through a
// source's backward axis b (third row of its rotation), is linear in t — backward(t) = b.(O - pos) + t b.D. If that is
// positive with a margin at t = kCullMinDepth and grows with t (b.D > margin), the point is behind the source's
// image plane for every t >= kCullMinDepth, where Camera::sees (Camera.h:184-190, the isBehind / FOV-cone test)
// returns false for every camera whose FOV half-angle is <= 90 degrees (cos_fov >= 0). The margins (1e-6) are ten
// orders of magnitude above the rounding error of the fp64 evaluation in `sees`, so skipping these sources cannot
// change a result; a source is skipped only when EVERY active lane of the wave agrees (the mask is wave-uniform).
// On the 16-camera rig 8-9 of the 15 sources face away from any given tile: their per-candidate cone tests (and
// the scalar-cache round trip for their constants) were ~5 % of the ping-pong kernel.
static constexpr double kCullMinDepth = 0.05;  // m; candidates nearer than this (disparity > 20) take the full loop
__device__ __forceinline__ unsigned behind_mask(const LevelView& V, int own, const D3& rayO, const D3& rayD) {
  unsigned m = 0;
  for (int s = 0; s < V.S; ++s) {
    if (s == own) {
      continue;
    }
    const Cam& cs = V.camsSrc[s];
    const double a = sum3(cs.R[6] * (rayO.x - cs.pos[0]), cs.R[7] * (rayO.y - cs.pos[1]), cs.R[8] * (rayO.z - cs.pos[2]));
    const double b = sum3(cs.R[6] * rayD.x, cs.R[7] * rayD.y, cs.R[8] * rayD.z);
    if (cs.cos_fov >= 0 && b > 1e-6 && a + kCullMinDepth * b > 1e-6) {
      m |= 1u << slot(s, own);
    }
  }
  return m;
}
// the wave's cull mask: the sources every ACTIVE lane's pixel has behind it (per-pixel masks from k_pixel_rays)
__device__ __forceinline__ unsigned behind_sources(const LevelView& V, int d, size_t idx) {
  const unsigned m = V.behind[(size_t)d * ((size_t)V.W * V.H) + idx];
  unsigned cull = 0;
  for (int t = 0; t < V.S - 1; ++t) {
    if (__ballot(!((m >> t) & 1u)) == 0ull) {
      cull |= 1u << t;
    }
  }
  return __builtin_amdgcn_readfirstlane(cull);
}

// computeCost — Derp.cpp:104-226. `dl` = dst index inside the batch, `own` = dst2src. Returns (cost, confidence);
// (FLT_MAX, 0) when fewer than kMinOverlappingCams-1 sources see the point. The reference's loop over the sources
// (project, fetch the warp, computeSSD) runs as two phases — same arithmetic, same order of the SSD pairs:
//  (i)  every source is projected (fp64 Camera::sees) and its projWarp taps are fetched; the taps of source
//       s are consumed after the projection of source s + 1, so their latency hides behind that chain. What
//       survives (visible, not NaN) leaves (xDstSrc, yDstSrc) in the lane's LDS slot of that source and a bit
//       in `mask`; the wave keeps the union of the lanes' masks in a scalar.
//  (ii) the wave walks the union; lanes holding the bit run computeSSD. The entry slots alias the pair array:
//       the i-th pair of a lane is written after the entry of its i-th visible source (slot >= i) was read.
// The fp64 projection state and the 4x4 texel block are never live together.
// `cull` (wave-uniform, from behind_sources): slots of sources that no lane of the wave can see at any depth >=
// kCullMinDepth; they are skipped without their cone test. 0 = test every source.
// SCALAR: computeSSD's block arithmetic in plain instead of packed fp32 (ssd_arith).
// RELOAD_RAY (ping-pong): the pixel's ray direction is read from the rayDir table at every call (pixel index `pix`)
// instead of living in six registers across the candidate loop — where the allocator parked it in scratch (one store per
// pixel, one load per candidate: 2.4 GB of scratch writes per level-0 launch at config 2, round 4). The opaque copy of
// the index keeps the loads inside the loop.
// developer build (-DDERP_PHASE_TIMERS=1|2): wave cycles (s_memtime) spent in computeCost's phases, reported through the
// kernel's counter slots INSTEAD of the cost / pair counts: =1 -> [0] projection + taps, [1] SSD walk, [3] selection;
// =2 -> [0] the whole kernel body, [1] everything outside computeCost. tools/phase_timers.py prints them.
#ifndef DERP_PHASE_TIMERS
#define DERP_PHASE_TIMERS 0
#endif
struct PhaseTimers {
  unsigned proj = 0, ssd = 0, select = 0, inside = 0;  // (32 bits: a wave lives < 2^32 cycles)
};
__device__ __forceinline__ unsigned phase_clock() {
#if DERP_PHASE_TIMERS
  return (unsigned)__builtin_amdgcn_s_memtime();
#else
  return 0u;
#endif
}
template <bool SCALAR = false, bool RANDOM = false, bool RELOAD_RAY = false>
__device__ __forceinline__ float2 compute_cost(const LevelView& V, int dl, int own, const PixCtx& px, float disparity,
                                               LdsPairs& pairs, unsigned& nPair, unsigned cull = 0, unsigned pix = 0,
                                               unsigned* slots = nullptr, PhaseTimers* tm = nullptr) {
  const unsigned tc0 = phase_clock();
  const double depth = (double)(1.0f / disparity);
  D3 rayD = px.rayD;
  if constexpr (RELOAD_RAY) {
    // NOT volatile: clang gives a volatile asm no memory(none) attribute, MemorySSA then takes it for a store, and every
    // uniform load behind it (the source cameras' constants, once per projection) leaves the scalar cache for per-lane
    // vector loads — measured: +11 % on this kernel. A pure asm would be hoisted out of the candidate loop with the
    // loads it is meant to pin there; tying it to the candidate's disparity keeps it loop-variant.
    unsigned i = pix;
    asm("" : "+v"(i) : "v"(disparity));
    const double* rd = V.rayDir + (size_t)(V.dst0 + dl) * ((size_t)V.W * V.H);
    if constexpr (RANDOM) {  // (32-bit byte offsets from three scalar bases: no 64-bit per-lane index)
      rayD = {at32(rd, i), at32(rd + V.rayStride, i), at32(rd + 2 * V.rayStride, i)};
    } else {
      rayD = {rd[i], rd[V.rayStride + i], rd[2 * V.rayStride + i]};
    }
  }
  const D3 pWorld = {px.rayO.x + rayD.x * depth, px.rayO.y + rayD.y * depth, px.rayO.z + rayD.z * depth};
  const size_t wPlane = warp_plane(V), cPlane = color_plane(V);
  const int wPitch = V.W + 2 * kPadW;
  unsigned mask = 0, waveMask = 0;
  // the cull mask holds for depths >= kCullMinDepth only (NaN and negative depths fail the comparison too)
  cull = __builtin_amdgcn_readfirstlane(__ballot(!(depth >= kCullMinDepth)) != 0ull ? 0u : cull);
  const ushort4* colBase = V.projColor + (size_t)dl * (V.S - 1) * cPlane;
  const ushort4* biaBase = V.projBias + (size_t)dl * (V.S - 1) * cPlane;
  const size_t tPlane = tiled_plane(V.W, V.H);
  const ushort4* colTBase = RANDOM ? V.projColorT + (size_t)dl * (V.S - 1) * tPlane : nullptr;
  {
    bool pend = false;
    int pendSlot = 0;
    f4a8 pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
    float pxw = 0, pyw = 0;
    auto consume = [&]() {
      const float w00 = (1 - pxw) * (1 - pyw), w01 = pxw * (1 - pyw), w10 = (1 - pxw) * pyw, w11 = pxw * pyw;
      const float wx = bilerp_f(pa.x, pa.z, pb.x, pb.z, w00, w01, w10, w11);
      const float wy = bilerp_f(pa.y, pa.w, pb.y, pb.w, w00, w01, w10, w11);
      // the reference adds a double literal: (float)((double)wx + 0.5). A sum of two floats rounded to
      // double (53 >= 2 * 24 + 2 bits) and then to float equals the float sum, so one fp32 add does it
      const float xDstSrc = wx + 0.5f, yDstSrc = wy + 0.5f;
      const bool ok = pend && !(isnan(xDstSrc) || isnan(yDstSrc));
      if (ok) {
        pairs.set(pendSlot, SsdPair{xDstSrc, yDstSrc});
        mask |= 1u << pendSlot;
      }
      if (__ballot(ok) != 0ull) {
        waveMask |= 1u << pendSlot;
      }
    };
    for (int s = 0; s < V.S; ++s) {
      if (s == own || ((cull >> slot(s, own)) & 1u)) {
        continue;
      }
      const Cam& cs = V.camsSrc[s];
      D2 pn;
      // worldToSrcPoint (DerpUtil.cpp:56-73): Camera::sees on the normalised camera, then * (W, H)
#ifdef DERP_ABLATE_NO_PROJ
      const float fx = (float)pWorld.x * 0.07f + 0.013f * s, fy = (float)pWorld.y * 0.07f + 0.011f * s;
      pn.x = 0.5 + 0.45 * (double)(fx - floorf(fx) - 0.5f);
      pn.y = 0.5 + 0.45 * (double)(fy - floorf(fy) - 0.5f);
      const bool vis = (s & 1) != 0;
#else
      // (bit 2: square roots through sqrt_lean — random proposals, where its three registers decide between scratch and none)
      const bool vis = sees<(DERP_LEAN_PROJ != 0) * ((DERP_ATAN_LUT ? 2 : 1) + (RANDOM ? 4 : 0))>(cs, pWorld, cs.principal[0], cs.principal[1], cs.focal[0], cs.focal[1],
                                                 1.0, 1.0, pn, pairs.atanLut);
#endif
      if (__ballot(pend) != 0ull) {
        consume();
      }
      pend = vis;
      pendSlot = slot(s, own);
      if (__ballot(vis) != 0ull) {
        // (the level size as doubles from the kernel arguments: converted here, the pair is hoisted into vector registers
        // and is what the allocator spills first)
        const float sx = (float)(pn.x * V.Wd), sy = (float)(pn.y * V.Hd);
        // pDstSrc = getPixelBilinear(dstProjWarp, pSrc)
        const float xf = roundf(sx), yf = roundf(sy);
        const int xi = (int)xf, yi = (int)yf;
        pxw = sx - xf + 0.5f;
        pyw = sy - yf + 0.5f;
        const size_t tab = (size_t)dl * (V.S - 1) + pendSlot;
        // 32-bit byte offset from the wave-uniform plane base (scalar-base addressing form); lanes that do
        // not see the source read the table's first taps
        const char* wbase = reinterpret_cast<const char*>(V.projWarp + tab * wPlane);
        const unsigned woff = vis ? ((unsigned)(yi - 1 + kPadW) * (unsigned)wPitch + (unsigned)(xi - 1 + kPadW)) * 8u : 0u;
        pa = *reinterpret_cast<const f4a8*>(wbase + woff);
        pb = *reinterpret_cast<const f4a8*>(wbase + (woff + (unsigned)wPitch * 8u));
      }
    }
    if (__ballot(pend) != 0ull) {
      consume();
    }
  }
  const unsigned tc1 = phase_clock();
  const int ssdCount = __popc(mask);
  nPair += ssdCount;
#ifdef DERP_COUNT_UNION  // developer measurement: SSD iterations the WAVE walks (its lanes' union) per active lane
  if (slots) {
    *slots += __popc(__builtin_amdgcn_readfirstlane(waveMask));
  }
#endif
  int keep = 1;  // kMinOverlappingCams - 1
  if (ssdCount < keep) {
    return make_float2(3.402823466e+38f, 0.0f);
  }
  {
    int cnt = 0;
    // waveMask is uniform by construction; readfirstlane keeps the walk (and the table bases) scalar
    for (unsigned wm = __builtin_amdgcn_readfirstlane(waveMask); wm != 0; wm &= wm - 1) {
      const int t = __builtin_ctz(wm);
      if ((mask >> t) & 1) {
        const SsdPair e = pairs.get(t);
        const SsdPair ssd = compute_ssd<SCALAR, RANDOM>(V, px, colBase + (size_t)t * cPlane, biaBase + (size_t)t * cPlane, e.first,
                                                        e.second, RANDOM ? colTBase + (size_t)t * tPlane : nullptr);
        pairs.set(cnt, ssd);
        ++cnt;
      }
    }
  }
  const unsigned tc2 = phase_clock();
  keep = max(keep, ssdCount - 2);
  GccSelect<LdsPairs> sel(pairs);
#ifndef DERP_ABLATE_NO_SELECT
  sel.nth_element(keep, ssdCount);
#endif
  float cost = 0;
  for (int i = 0; i < keep; ++i) {
    cost += pairs.get(i).second;
  }
  cost /= (float)keep;
  const float trustCoef = 1.0f / (float)keep;
  float confidence = px.confidence;
  if constexpr (RELOAD_RAY) {
    // like the ray direction: read again where it is used (max(variance, kMinVar) of the pixel) instead of riding through
    // the source loops, where it was the first value the allocator put in scratch at 128 registers
    unsigned i = pix;
    asm("" : "+v"(i) : "v"(cost));
    if constexpr (RANDOM) {
      confidence = fmaxf(at32(V.srcVar + (size_t)own * ((size_t)V.W * V.H), i), kMinVar);
    } else {
      confidence = fmaxf((V.srcVar + (size_t)own * ((size_t)V.W * V.H))[i], kMinVar);
    }
  }
  const float costFinal = cost * trustCoef / confidence;
#if DERP_PHASE_TIMERS
  if (tm) {
    const unsigned tc3 = phase_clock();
    tm->proj += tc1 - tc0, tm->ssd += tc2 - tc1, tm->select += tc3 - tc2, tm->inside += tc3 - tc0;
  }
#endif
  return make_float2(costFinal, confidence);
}

// gather the per-pixel constants of computeCost: dst ray, 3x3 dst patch, dst bias, variance
template <bool WITH_RAY = true>
__device__ __forceinline__ void load_pixctx(const LevelView& V, int d, int own, int x, int y, const PatchWin* win, PixCtx& px) {
  const Cam& cd = V.camsDst[d];
  px.rayO = {cd.pos[0], cd.pos[1], cd.pos[2]};
  // the ray direction of the pixel centre (Camera::rig of p = ((x + .5) / W, (y + .5) / H): undistort's Newton
  // iteration, sin, cos) depends on the rig and the level size only: k_pixel_rays tabulates it with the warps
  const size_t n = (size_t)V.W * V.H;
  if constexpr (!WITH_RAY) {
    px.rayD = {0, 0, 0};  // compute_cost<.., RELOAD_RAY> reads it per call
  } else {
    const size_t i = (size_t)d * n + (size_t)y * V.W + x;
    px.rayD = {V.rayDir[i], V.rayDir[V.rayStride + i], V.rayDir[2 * V.rayStride + i]};
  }
  // the 3x3 patch: the lane's corner of the wave's window (patch_window_fill)
  const int corner = (int)((threadIdx.x & 63) >> 3) * kWinW + (int)(threadIdx.x & 7);
  px.winBG = win->bg + corner;
  px.winR = win->r + corner;
  const ushort4 b = V.ownBias[(size_t)own * n + (size_t)y * V.W + x];
  px.dstBiasBG = (v2f){(float)b.x, (float)b.y};
  px.dstBiasR = (float)b.z;
  const float var = V.srcVar[(size_t)own * n + (size_t)y * V.W + x];
  px.confidence = fmaxf(var, kMinVar);
}

__device__ __forceinline__ void flush_counters(const LevelView& V, unsigned nCost, unsigned nPair) {
  // wave-level reduction, one atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    nCost += __shfl_down(nCost, off);
    nPair += __shfl_down(nPair, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&V.counters[0], (unsigned long long)nCost);
    atomicAdd(&V.counters[1], (unsigned long long)nPair);
  }
}

// ----------------------------------------------------------------------------------------
// upload conversion: interleaved BGR u16 -> ushort4
// ----------------------------------------------------------------------------------------
__global__ void k_bgr_to_bgrx(const uint16_t* __restrict__ in, ushort4* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    out[i] = make_ushort4(in[3 * i], in[3 * i + 1], in[3 * i + 2], 0);
  }
}

__global__ void k_debug_atan2_ypos(const double* __restrict__ y, const double* __restrict__ x, double* __restrict__ out,
                                   size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
#if defined(__HIP_DEVICE_COMPILE__)  // the routine exists in the device pass only
  // what the cost kernels evaluate (the table variant when DERP_ATAN_LUT), checked against the literal variant on
  // the spot: a bitwise disagreement between the two is reported as NaN, which no caller's comparison survives
  __shared__ double atanLut[kAtanLutDoubles];
  atan_lut_fill(atanLut);
  __syncthreads();
  for (; i < n; i += step) {
    const double a = atan2_ypos(y[i], x[i]);
    const double b = atan2_ypos_lut(y[i], x[i], atanLut);
    out[i] = (__double_as_longlong(a) == __double_as_longlong(b)) ? (DERP_ATAN_LUT ? b : a) : __builtin_nan("");
  }
#endif
}

// Per destination pixel: the rig-space ray direction of its centre (dstToWorldPoint's Camera::rig, DerpUtil.cpp:38-52,
// Camera.h:131-138) and the slots of the sources that face away from that ray (behind_mask). Rig + level size only:
// built with the projection warps, read by every cost kernel instead of redoing ~800 fp64 instructions per pixel.
__global__ void k_pixel_rays(LevelView V, double* __restrict__ rays, unsigned* __restrict__ behind) {
  const int d = V.dst0 + (int)blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= V.W || y >= V.H) {
    return;
  }
  const Cam& cd = V.camsDst[d];
  const D3 dir = rig_direction(cd, (x + 0.5) / (double)V.W, (y + 0.5) / (double)V.H, cd.principal[0], cd.principal[1],
                               cd.focal[0], cd.focal[1]);
  const size_t i = (size_t)d * ((size_t)V.W * V.H) + (size_t)y * V.W + x;
  rays[i] = dir.x;
  rays[V.rayStride + i] = dir.y;
  rays[2 * V.rayStride + i] = dir.z;
  behind[i] = behind_mask(V, V.dst2src[d], D3{cd.pos[0], cd.pos[1], cd.pos[2]}, dir);
}

// generateFovMasks — DerpUtil.cpp:239-276 (normalised camera: p = (x+.5, y+.5) / (W, H))
__global__ void k_fov_mask(const Cam* __restrict__ cams, int W, int H, uint8_t* __restrict__ out) {
  const int d = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const Cam& c = cams[d];
  const double px = (x + 0.5) / (double)W, py = (y + 0.5) / (double)H;
  out[(size_t)d * W * H + (size_t)y * W + x] =
      !outside_image_circle(c, px, py, c.principal[0], c.principal[1], c.focal[0], c.focal[1]);
}

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) {
    return 0;
  }
  while (p < 0 || p >= len) {
    p = p < 0 ? -p : 2 * len - 2 - p;
  }
  return p;
}

// computeImageVariance — DerpUtil.cpp:214-237 (cv::blur 3x3 on CV_32FC3: double row sums,
// double column sums, * 1/9 -> float; BORDER_REFLECT_101), PyramidLevel.h:232-247
__global__ void k_variance(const ushort4* __restrict__ color, int W, int H, float* __restrict__ var) {
  const int s = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const ushort4* img = color + (size_t)s * W * H;
  const float scale = 1.0f / 65535.0f;
  double sum[3] = {0, 0, 0}, sumSq[3] = {0, 0, 0};
  const int xs[3] = {reflect101(x - 1, W), x, reflect101(x + 1, W)};
  for (int j = -1; j <= 1; ++j) {
    const int yy = reflect101(y + j, H);
    double rs[3], rq[3];
    {
      const ushort4 q0 = img[(size_t)yy * W + xs[0]], q1 = img[(size_t)yy * W + xs[1]], q2 = img[(size_t)yy * W + xs[2]];
      const float f0[3] = {q0.x * scale, q0.y * scale, q0.z * scale};
      const float f1[3] = {q1.x * scale, q1.y * scale, q1.z * scale};
      const float f2[3] = {q2.x * scale, q2.y * scale, q2.z * scale};
      for (int c = 0; c < 3; ++c) {
        rs[c] = (double)f0[c] + (double)f1[c] + (double)f2[c];
        rq[c] = (double)(f0[c] * f0[c]) + (double)(f1[c] * f1[c]) + (double)(f2[c] * f2[c]);
      }
    }
    for (int c = 0; c < 3; ++c) {
      sum[c] += rs[c];
      sumSq[c] += rq[c];
    }
  }
  const double k = 1. / 9;
  float v[3];
  for (int c = 0; c < 3; ++c) {
    const float mean = (float)(sum[c] * k);
    const float meanSq = (float)(sumSq[c] * k);
    v[c] = meanSq - mean * mean;
  }
  // varChannels[0]*w[2] + varChannels[1]*w[1] + varChannels[2]*w[0], kRgbWeights = {.3333,.3334,.3333}
  const float t = v[0] * 0.3333f + v[1] * 0.3334f;
  var[(size_t)s * W * H + (size_t)y * W + x] = t * 1.0f + v[2] * 0.3333f;
}

// colorBias = cv::blur 3x3 on CV_16UC3 (DerpUtil.cpp:208-210): exact integer sum, round(sum/9).
// Source and destination may carry replicated rings (padIn / padOut); the blur itself uses
// BORDER_REFLECT_101 over the image interior, and ring texels repeat the clamped interior value.
// One thread = one column of kBlurRows consecutive output rows. Away from the borders the three
// horizontal 3-sums of a column slide down the rows (3 loads per output instead of 9); threads whose
// strip touches a border, where BORDER_REFLECT_101 / the ring replication change the tap pattern, take
// the texel-by-texel form for their rows.
constexpr int kBlurRows = 8;
__device__ __forceinline__ ushort4 blur3_texel(const ushort4* __restrict__ img, int IW, int padIn, int W, int H, int x,
                                               int y) {
  unsigned s0 = 0, s1 = 0, s2 = 0;
  for (int j = -1; j <= 1; ++j) {
    const int yy = reflect101(y + j, H) + padIn;
    for (int i = -1; i <= 1; ++i) {
      const int xx = reflect101(x + i, W) + padIn;
      const ushort4 q = img[(size_t)yy * IW + xx];
      s0 += q.x;
      s1 += q.y;
      s2 += q.z;
    }
  }
  return make_ushort4((unsigned short)((s0 + 4) / 9), (unsigned short)((s1 + 4) / 9), (unsigned short)((s2 + 4) / 9), 0);
}
__global__ void k_blur3_u16(const ushort4* __restrict__ in, int padIn, ushort4* __restrict__ out, int padOut, int W,
                            int H, size_t planeIn, size_t planeOut) {
  const int p = blockIdx.z;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy0 = (blockIdx.y * blockDim.y + threadIdx.y) * kBlurRows;
  const int OW = W + 2 * padOut, OH = H + 2 * padOut;
  if (ox >= OW || oy0 >= OH) {
    return;
  }
  const ushort4* img = in + (size_t)p * planeIn;
  ushort4* dst = out + (size_t)p * planeOut;
  const int IW = W + 2 * padIn;
  const int x = ox - padOut, y0 = oy0 - padOut;
  if (x >= 1 && x <= W - 2 && y0 >= 1 && y0 + kBlurRows - 1 <= H - 2) {
    auto hsum = [&](int y, unsigned (&h)[3]) {
      const ushort4* r = img + (size_t)(y + padIn) * IW + (x + padIn);
      const ushort4 a = r[-1], b = r[0], c = r[1];
      h[0] = (unsigned)a.x + b.x + c.x;
      h[1] = (unsigned)a.y + b.y + c.y;
      h[2] = (unsigned)a.z + b.z + c.z;
    };
    unsigned hp[3], hc[3], hn[3];
    hsum(y0 - 1, hp);
    hsum(y0, hc);
#pragma unroll
    for (int r = 0; r < kBlurRows; ++r) {
      hsum(y0 + r + 1, hn);
      dst[(size_t)(oy0 + r) * OW + ox] =
          make_ushort4((unsigned short)((hp[0] + hc[0] + hn[0] + 4) / 9), (unsigned short)((hp[1] + hc[1] + hn[1] + 4) / 9),
                       (unsigned short)((hp[2] + hc[2] + hn[2] + 4) / 9), 0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        hp[c] = hc[c];
        hc[c] = hn[c];
      }
    }
    return;
  }
  const int xc = min(max(x, 0), W - 1);
  for (int r = 0; r < kBlurRows && oy0 + r < OH; ++r) {
    const int yc = min(max(y0 + r, 0), H - 1);
    dst[(size_t)(oy0 + r) * OW + ox] = blur3_texel(img, IW, padIn, W, H, xc, yc);
  }
}

// ----------------------------------------------------------------------------------------
// precomputeProjections (Derp.cpp:955-976) -> computeWarpDstToSrc (ImageUtil.cpp:142-167).
// projWarp(dst d, src s) lives on the SRC grid: for every src pixel, the dst pixel hit by the
// src ray at kNearInfinity. One thread per (src, padded pixel): the ray is shared by all dsts.
// Cameras are the level-size rescale of the normalised ones (Camera.cpp:217-223).
// ----------------------------------------------------------------------------------------
__global__ void k_proj_warp(LevelView V, float2* __restrict__ projWarp) {
  const int s = blockIdx.z;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  const int OW = V.W + 2 * kPadW, OH = V.H + 2 * kPadW;
  if (ox >= OW || oy >= OH) {
    return;
  }
  const int x = min(max(ox - kPadW, 0), V.W - 1), y = min(max(oy - kPadW, 0), V.H - 1);
  const Cam& cs = V.camsSrc[s];
  const double W = V.W, H = V.H;
  const double sprx = cs.principal[0] * W, spry = cs.principal[1] * H, sfx = cs.focal[0] * W, sfy = cs.focal[1] * H;
  const double px = x + 0.5, py = y + 0.5;
  const bool outside = outside_image_circle(cs, px, py, sprx, spry, sfx, sfy);
  D3 rig = {0, 0, 0};
  if (!outside) {
    const D3 dir = rig_direction(cs, px, py, sprx, spry, sfx, sfy);
    rig = {cs.pos[0] + dir.x * 1e4, cs.pos[1] + dir.y * 1e4, cs.pos[2] + dir.z * 1e4};
  }
  const size_t plane = (size_t)OW * OH;
  const float nan = __builtin_nanf("");
  for (int dl = 0; dl < V.D; ++dl) {
    const int d = V.dst0 + dl;
    const int own = V.dst2src[d];
    if (s == own) {
      continue;  // ids equal: all-NaN in the reference, never read (src == dst is skipped)
    }
    float2 val = make_float2(nan, nan);
    if (!outside) {
      const Cam& cd = V.camsDst[d];
      D2 p;
      if (sees(cd, rig, cd.principal[0] * W, cd.principal[1] * H, cd.focal[0] * W, cd.focal[1] * H, W, H, p)) {
        val = make_float2((float)(p.x - (double)0.5f), (float)(p.y - (double)0.5f));
      }
    }
    projWarp[((size_t)dl * (V.S - 1) + slot(s, own)) * plane + (size_t)oy * OW + ox] = val;
  }
}

// cubic coefficients: imgwarp.cpp interpolateCubic, A = -0.75, x = k/32
__device__ __forceinline__ void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
// cvRound(float): round-half-even, INT_MIN on NaN / overflow (cvtss2si)
__device__ __forceinline__ int cv_round(float v) {
  if (!(v > -2147483648.0f && v < 2147483648.0f)) {
    return (int)0x80000000;
  }
  return (int)rintf(v);
}

// cv::remap(src, map, INTER_CUBIC, BORDER_CONSTANT 0) at ONE map coordinate (mx, my) of a CV_16UC3 image
// (DerpUtil.cpp:199-205): 1/32-px fixed point, 4x4 taps, float weights, saturate_cast<ushort> = round-half-even.
__device__ __forceinline__ ushort4 remap_cubic_u16(const ushort4* __restrict__ img, int W, int H, float mx, float my) {
  const int fsx = cv_round(mx * 32.0f), fsy = cv_round(my * 32.0f);
  const int fx = fsx & 31, fy = fsy & 31;
  const int sx = min(max(fsx >> 5, -32768), 32767) - 1, sy = min(max(fsy >> 5, -32768), 32767) - 1;
  float cx[4], cy[4];
  cubic_coeffs((float)fx * (1.f / 32), cx);
  cubic_coeffs((float)fy * (1.f / 32), cy);
  float sum[3];
  if ((unsigned)sx < (unsigned)max(W - 3, 0) && (unsigned)sy < (unsigned)max(H - 3, 0)) {
    float rowsum[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // the four texels of a tap row as two 16-byte loads (8-byte aligned)
      const u4a8* r = reinterpret_cast<const u4a8*>(img + (size_t)(sy + i) * W + sx);
      const u4a8 a = r[0], b = r[1];
      const float w0 = cy[i] * cx[0], w1 = cy[i] * cx[1], w2 = cy[i] * cx[2], w3 = cy[i] * cx[3];
      rowsum[i][0] = (float)(a.x & 0xffff) * w0 + (float)(a.z & 0xffff) * w1 + (float)(b.x & 0xffff) * w2 + (float)(b.z & 0xffff) * w3;
      rowsum[i][1] = (float)(a.x >> 16) * w0 + (float)(a.z >> 16) * w1 + (float)(b.x >> 16) * w2 + (float)(b.z >> 16) * w3;
      rowsum[i][2] = (float)(a.y & 0xffff) * w0 + (float)(a.w & 0xffff) * w1 + (float)(b.y & 0xffff) * w2 + (float)(b.w & 0xffff) * w3;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = rowsum[0][c];
      acc += rowsum[1][c];
      acc += rowsum[2][c];
      acc += rowsum[3][c];
      sum[c] = acc;
    }
  } else if (sx >= W || sx + 4 <= 0 || sy >= H || sy + 4 <= 0) {
    sum[0] = sum[1] = sum[2] = 0.f;
  } else {
    sum[0] = sum[1] = sum[2] = 0.f;
    for (int i = 0; i < 4; ++i) {
      const int yy = sy + i;
      if ((unsigned)yy >= (unsigned)H) {
        continue;
      }
      for (int j = 0; j < 4; ++j) {
        const int xx = sx + j;
        if ((unsigned)xx >= (unsigned)W) {
          continue;
        }
        const ushort4 q = img[(size_t)yy * W + xx];
        const float w = cy[i] * cx[j];
        sum[0] += ((float)q.x - 0.f) * w;
        sum[1] += ((float)q.y - 0.f) * w;
        sum[2] += ((float)q.z - 0.f) * w;
      }
    }
  }
  const int r0 = min(max(cv_round(sum[0]), 0), 65535), r1 = min(max(cv_round(sum[1]), 0), 65535),
            r2 = min(max(cv_round(sum[2]), 0), 65535);
  return make_ushort4((unsigned short)r0, (unsigned short)r1, (unsigned short)r2, 0);
}

// projWarpInv(d, s) = computeWarpDstToSrc(camDst d, camSrc s) and projWarp(d', s') = computeWarpDstToSrc(camSrc s',
// camDst d') (Derp.cpp:969-970) are the SAME function of (camera the pixel grid belongs to, camera projected into):
// when source s is itself a destination ds of the batch, projWarpInv(d, s) is projWarp(ds, own(d)) — already built,
// by the same arithmetic (k_proj_warp and k_proj_warp_inv call the same rig_direction / sees with the same arguments).
// Returns ds - dst0, or -1 when s is not the own source of a destination of this batch (a --cameras subset, or a
// table-budget batch): only those inverse warps are computed and stored (k_proj_warp_inv).
__device__ __forceinline__ int batch_dst_of_source(const LevelView& V, int s) {
  for (int dl = 0; dl < V.D; ++dl) {
    if (V.dst2src[V.dst0 + dl] == s) {
      return dl;
    }
  }
  return -1;
}

// The rig-space point a dst pixel's ray reaches at kNearInfinity (computeWarpDstToSrc, ImageUtil.cpp:142-167);
// false outside the image circle.
__device__ __forceinline__ bool dst_far_point(const LevelView& V, const Cam& cd, int x, int y, D3& rig) {
  const double W = V.W, H = V.H;
  const double dprx = cd.principal[0] * W, dpry = cd.principal[1] * H, dfx = cd.focal[0] * W, dfy = cd.focal[1] * H;
  const double px = x + 0.5, py = y + 0.5;
  if (outside_image_circle(cd, px, py, dprx, dpry, dfx, dfy)) {
    return false;
  }
  const D3 dir = rig_direction(cd, px, py, dprx, dpry, dfx, dfy);
  rig = {cd.pos[0] + dir.x * 1e4, cd.pos[1] + dir.y * 1e4, cd.pos[2] + dir.z * 1e4};
  return true;
}
// projWarpInv(d, s)(x, y): where that point lands in src s, as cv::remap's map coordinate; NaN when it is not seen
__device__ __forceinline__ float2 warp_inv_of(const LevelView& V, const Cam& cs, bool inside, const D3& rig) {
  const double W = V.W, H = V.H;
  D2 p;
  if (inside && sees(cs, rig, cs.principal[0] * W, cs.principal[1] * H, cs.focal[0] * W, cs.focal[1] * H, W, H, p)) {
    return make_float2((float)(p.x - (double)0.5f), (float)(p.y - (double)0.5f));
  }
  const float nan = __builtin_nanf("");
  return make_float2(nan, nan);
}

// projWarpInv (PyramidLevel.h:46-51, the reference's second table) for the destinations of a batch, on the
// UNPADDED dst grid: [D][S-1][H][W] float2. Rig and level size only, like projWarp: built with it (precomputeProjections,
// Derp.cpp:955-976) and kept for as long as it is — every further frame of a sequence that runs this level skips
// these fp64 projections (15 sources x Newton undistort / atan2 per pixel), which were the whole cost of
// reprojectColors. The dst-pixel ray is computed once per thread and pushed through every src.
__global__ void k_proj_warp_inv(LevelView V, float2* __restrict__ warpInv) {
  const int dl = blockIdx.z;
  const int d = V.dst0 + dl;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= V.W || y >= V.H) {
    return;
  }
  const int own = V.dst2src[d];
  D3 rig = {0, 0, 0};
  const bool inside = dst_far_point(V, V.camsDst[d], x, y, rig);
  const size_t n = (size_t)V.W * V.H;
  for (int s = 0; s < V.S; ++s) {
    // sources that are destinations of this batch: k_reproject_bias reads projWarp(ds, own) instead
    if (s != own && (DERP_NO_WARP_IDENTITY || batch_dst_of_source(V, s) < 0)) {
      warpInv[((size_t)dl * (V.S - 1) + slot(s, own)) * n + (size_t)y * V.W + x] = warp_inv_of(V, V.camsSrc[s], inside, rig);
    }
  }
}

// reprojectColors (Derp.cpp:978-1003) + the colour bias (DerpUtil.cpp:208-210) in one pass:
//   projColor(d, s) = cv::remap(srcColor[s], projWarpInv(d, s), INTER_CUBIC, BORDER_CONSTANT 0)
//   projBias(d, s)  = cv::blur 3x3 of projColor(d, s) on CV_16UC3, BORDER_REFLECT_101: exact integer sum, round(s / 9)
// A block owns a 32 x 32 tile of dst pixels of ONE (dst, src) table (blockIdx.z). It remaps the tile and its
// 1-pixel halo (34 x 34 positions, +13 % — the remap, 16 taps and ~250 VALU instructions per position, is the
// expensive part; halo positions beyond the image are their BORDER_REFLECT_101 mirror pixels) into LDS, then every
// thread writes four pixels' colours and their 3x3 boxes over the LDS tile — the separate blur pass read back from
// HBM what the remap had just written (19.7 GB per level-0 launch, 68 % of its wave-cycles waiting). Both tables
// carry a 2-texel replicated ring: edge pixels write their ring texels too.
constexpr int kRbTile = 32, kRbPitch = kRbTile + 2, kRbCells = kRbPitch * kRbPitch;
// tileSeen [tables][tilesY][tilesX]: whether any position of the tile (halo included) has a valid map. The inverse warps
// depend on the rig and the level size only, so a tile that is all-NaN for one frame is all-NaN for every frame: its
// colour and bias texels are the constant border 0. skipBlank = 0: write everything and record the flags (the first
// frame that runs a level after its tables were (re)built); 1: blocks of blank tiles return at once — the zeros the
// first frame wrote are still there, nothing else writes these tables (on the 16-camera rig about half of the
// (tile, source) combinations: the sources that face away from that part of the destination image).
__global__ void __launch_bounds__(256)
    k_reproject_bias(LevelView V, const float2* __restrict__ warpInv, ushort4* __restrict__ projColor,
                     ushort4* __restrict__ projBias, ushort4* __restrict__ projColorT, uint8_t* __restrict__ tileSeen,
                     int skipBlank) {
  __shared__ ushort4 tile[kRbCells];
  const int tab = blockIdx.z;  // dl * (S - 1) + slot
  uint8_t* seen = tileSeen + ((size_t)tab * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (skipBlank && !*seen) {
    return;
  }
  const int dl = tab / (V.S - 1), sl = tab - dl * (V.S - 1);
  const int own = V.dst2src[V.dst0 + dl];
  const int s = sl < own ? sl : sl + 1;
  const int x0 = blockIdx.x * kRbTile, y0 = blockIdx.y * kRbTile;
  const int OW = V.W + 2 * kPadC, OH = V.H + 2 * kPadC;
  const size_t plane = (size_t)OW * OH, n = (size_t)V.W * V.H;
  const ushort4* img = V.srcColor + (size_t)s * n;
  // the inverse warp of this table: projWarp(ds, own) when source s is destination ds of the batch (the same
  // function, see batch_dst_of_source; that table carries a 1-texel ring), the stored projWarpInv otherwise
  const float2* map = warpInv + (size_t)tab * n;
  int mapPitch = V.W;
  const int ds = DERP_NO_WARP_IDENTITY ? -1 : batch_dst_of_source(V, s);
  if (ds >= 0) {
    mapPitch = V.W + 2 * kPadW;
    map = V.projWarp + ((size_t)ds * (V.S - 1) + slot(own, s)) * warp_plane(V) + (size_t)kPadW * mapPitch + kPadW;
  }
  int any = 0;
  for (int k = threadIdx.x; k < kRbCells; k += 256) {
    const int ty = k / kRbPitch, tx = k - ty * kRbPitch;
    const int qx = x0 - 1 + tx, qy = y0 - 1 + ty;
    if (qx >= -1 && qx <= V.W && qy >= -1 && qy <= V.H) {
      const float2 m = map[(size_t)reflect101(qy, V.H) * mapPitch + reflect101(qx, V.W)];
      // NaN map -> (-32768, -32768) in cv::remap's fixed point -> every tap outside -> constant border 0
      const bool valid = !(m.x != m.x);
      any |= valid;
      tile[k] = valid ? remap_cubic_u16(img, V.W, V.H, m.x, m.y) : make_ushort4(0, 0, 0, 0);
    }
  }
  any = __syncthreads_or(any);
  if (!skipBlank && threadIdx.x == 0) {
    *seen = (uint8_t)(any != 0);
  }
  ushort4* pc = projColor + (size_t)tab * plane;
  ushort4* pb = projBias + (size_t)tab * plane;
  // the random-proposal kernel's copy of projColor in 4x4-texel tiles (null when the context does not keep one)
  const int tilesX = tiled_tiles_x(V.W);
  ushort4* pt = projColorT ? projColorT + (size_t)tab * tiled_plane(V.W, V.H) : nullptr;
  const int lx = threadIdx.x & 31;
  const int x = x0 + lx;
  if (x >= V.W) {
    return;
  }
  const int oxa = x == 0 ? 0 : x + kPadC, oxb = x == V.W - 1 ? OW - 1 : x + kPadC;
#pragma unroll
  for (int j = 0; j < kRbTile / 8; ++j) {
    const int ly = (int)(threadIdx.x >> 5) + 8 * j;
    const int y = y0 + ly;
    if (y >= V.H) {
      break;
    }
    const ushort4* t = &tile[(ly + 1) * kRbPitch + lx + 1];
    unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int v = -1; v <= 1; ++v) {
#pragma unroll
      for (int u = -1; u <= 1; ++u) {
        const ushort4 q = t[v * kRbPitch + u];
        s0 += q.x;
        s1 += q.y;
        s2 += q.z;
      }
    }
    const ushort4 col = t[0];
    const ushort4 bia = make_ushort4((unsigned short)((s0 + 4) / 9), (unsigned short)((s1 + 4) / 9),
                                     (unsigned short)((s2 + 4) / 9), 0);
    // padded coordinates this pixel writes: its own texel plus the ring texels that replicate it
    const int oya = y == 0 ? 0 : y + kPadC, oyb = y == V.H - 1 ? OH - 1 : y + kPadC;
    for (int oy = oya; oy <= oyb; ++oy) {
      for (int ox = oxa; ox <= oxb; ++ox) {
        pc[(size_t)oy * OW + ox] = col;
        pb[(size_t)oy * OW + ox] = bia;
        if (pt) {
          pt[tiled_index(tilesX, ox, oy)] = col;
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// brute force — Derp.cpp:230-382. Stage 1: cost / confidence for (dst, disparity i, pixel);
// stage 2: strict-< argmin over i + fallbacks; stage 3: 1-px margin replicated from the interior.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float probe_disparity(int i, float minD, float maxD) {  // ImageUtil.cpp:100-107
  const double fraction = (double)i / (double)(kNumDepths - 1);
  return (float)(fraction * (double)minD + (1 - fraction) * (double)maxD);
}

__global__ void __launch_bounds__(DERP_COST_BLOCK, DERP_COST_MIN_WAVES)
    k_brute_costs(LevelView V, float* __restrict__ costs, float* __restrict__ confs, int tilesX, int tilesPerDst) {
  extern __shared__ SsdPair ldsPairs[];
  const int i = blockIdx.y;   // disparity index
  const int dl = blockIdx.z;
  const int d = V.dst0 + dl;
  // the coarsest level is small (50 x 50): interior pixels are numbered densely in 8-wide strips of
  // 8 rows so that no lane of the grid is padding (tilesX = strips per row of strips here)
  const int iw = V.W - 2, ih = V.H - 2;
  const int lane = threadIdx.x & 63;
  const int sx = (int)(blockIdx.x % (unsigned)tilesX), sy = (int)(blockIdx.x / (unsigned)tilesX);
  const int x = 1 + sx * 8 + (lane & 7), y = 1 + sy * 8 + (lane >> 3);
  // the wave's 10x10 window of destination colours (PatchWin): filled by all 64 lanes before anything diverges
  __shared__ PatchWin patchWin[DERP_COST_BLOCK / 64];
  PatchWin* win = &patchWin[threadIdx.x >> 6];
  patch_window_fill(V, V.dst2src[d], x - (int)(threadIdx.x & 7), y - (int)((threadIdx.x & 63) >> 3), win);
#if DERP_ATAN_LUT && DERP_LEAN_PROJ && defined(__HIP_DEVICE_COMPILE__)
  __shared__ double atanLut[kAtanLutDoubles];
  atan_lut_fill(atanLut);
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, atanLut};
#else
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, nullptr};
#endif
  __syncthreads();
  unsigned nCost = 0, nPair = 0;
  if (x <= iw && y <= ih) {
    const int own = V.dst2src[d];
    const size_t n = (size_t)V.W * V.H, idx = (size_t)y * V.W + x;
    const float minDisparity = 1.0f / V.maxDepthM, maxDisparity = 1.0f / V.minDepthM;
    const float disparity = probe_disparity(i, minDisparity, maxDisparity);
    const bool fov = V.fovMask[(size_t)d * n + idx], fg = V.srcFg[(size_t)own * n + idx];
    const bool closer = V.hasFg ? (V.bgDisp[(size_t)d * n + idx] < disparity) : true;
    float2 r = make_float2(__builtin_nanf(""), __builtin_nanf(""));
    if (fov && fg && closer) {
      PixCtx px;
      load_pixctx(V, d, own, x, y, win, px);
      r = compute_cost<DERP_COST_SSD_SCALAR != 0>(V, dl, own, px, disparity, pairs, nPair);
      ++nCost;
    }
    const size_t o = ((size_t)dl * kNumDepths + i) * n + idx;
    costs[o] = r.x;
    confs[o] = r.y;
  }
  flush_counters(V, nCost, nPair);
}

__global__ void k_brute_select(LevelView V, const float* __restrict__ costs, const float* __restrict__ confs) {
  const int dl = blockIdx.z;
  const int d = V.dst0 + dl;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x < 1 || y < 1 || x >= V.W - 1 || y >= V.H - 1) {
    return;
  }
  const int own = V.dst2src[d];
  const size_t n = (size_t)V.W * V.H, idx = (size_t)y * V.W + x;
  float* disp = V.disparity + (size_t)d * n;
  if (!V.fovMask[(size_t)d * n + idx]) {
    disp[idx] = __builtin_nanf("");
    return;
  }
  if (!V.srcFg[(size_t)own * n + idx]) {
    disp[idx] = V.bgDisp[(size_t)d * n + idx];
    return;
  }
  float minCost = 3.402823466e+38f, minConf = 0;
  int best = -1;
  for (int i = 0; i < kNumDepths; ++i) {
    const size_t o = ((size_t)dl * kNumDepths + i) * n + idx;
    const float c = costs[o];
    if (c < minCost) {
      minCost = c;
      minConf = confs[o];
      best = i;
    }
  }
  const float minDisparity = 1.0f / V.maxDepthM, maxDisparity = 1.0f / V.minDepthM;
  if (best == -1) {
    atomicAdd(&V.counters[2], 1ull);  // "Insufficient coverage" warning / CHECK in the reference
    disp[idx] = minDisparity;
  } else {
    disp[idx] = probe_disparity(best, minDisparity, maxDisparity);
  }
  V.cost[(size_t)d * n + idx] = minCost;
  V.confidence[(size_t)d * n + idx] = minConf;
}

__global__ void k_brute_margin(LevelView V) {
  const int d = V.dst0 + blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= V.W || y >= V.H) {
    return;
  }
  if (!(x < 1 || x >= V.W - 1 || y < 1 || y >= V.H - 1)) {
    return;
  }
  const int own = V.dst2src[d];
  const size_t n = (size_t)V.W * V.H, idx = (size_t)y * V.W + x;
  if (!V.srcFg[(size_t)own * n + idx]) {
    V.disparity[(size_t)d * n + idx] = V.bgDisp[(size_t)d * n + idx];
    return;
  }
  const int yy = min(max(y, 1), V.H - 2), xx = min(max(x, 1), V.W - 2);
  const size_t src = (size_t)yy * V.W + xx;
  V.disparity[(size_t)d * n + idx] = V.disparity[(size_t)d * n + src];
  V.cost[(size_t)d * n + idx] = V.cost[(size_t)d * n + src];
  V.confidence[(size_t)d * n + idx] = V.confidence[(size_t)d * n + src];
}

// ----------------------------------------------------------------------------------------
// random proposals — Derp.cpp:750-873. The reference walks each row with one engine seeded
// y * level; a pixel's draws start at numProposals * (#gated-in pixels to its left).
// k_row_rank computes that count and leaves the engine's state at that position for every pixel.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ bool random_gate(const LevelView& V, int d, int own, size_t idx) {
  const size_t n = (size_t)V.W * V.H;
  if (!V.fovMask[(size_t)d * n + idx] || !V.srcFg[(size_t)own * n + idx]) {
    return false;
  }
  const float varHighDev = 0.1f * V.varHighThresh;  // kRandomPropHighVarDeviation
  const float thresh = fmaxf(varHighDev, V.varNoiseFloor);
  return !(V.srcVar[(size_t)own * n + idx] < thresh);
}

__global__ void __launch_bounds__(256) k_row_rank(LevelView V, int* __restrict__ rank) {
  __shared__ int partial[256];
  const int d = V.dst0 + blockIdx.y, y = blockIdx.x + 1;
  const int own = V.dst2src[d];
  const size_t n = (size_t)V.W * V.H;
  const int inner = V.W - 2;
  const int per = (inner + 255) / 256;
  const int x0 = 1 + threadIdx.x * per, x1 = min(x0 + per, V.W - 1);
  int cnt = 0;
  for (int x = x0; x < x1; ++x) {
    cnt += random_gate(V, d, own, (size_t)y * V.W + x);
  }
  partial[threadIdx.x] = cnt;
  __syncthreads();
  // exclusive scan of 256 partials (Hillis-Steele)
  for (int off = 1; off < 256; off <<= 1) {
    const int v = ((int)threadIdx.x >= off) ? partial[threadIdx.x - off] : 0;
    __syncthreads();
    partial[threadIdx.x] += v;
    __syncthreads();
  }
  // the engine's state in front of every pixel's draws: one O(log n) jump to the thread's first pixel, then one
  // multiplication by a^P per gated-in pixel (the random-proposal kernel used to jump per pixel: ~60 dependent
  // 64-bit multiply-reduce steps at the head of every pixel's chain)
  const int run = partial[threadIdx.x] - cnt;
  uint32_t state = minstd_jump(minstd_seed(y * V.level), (uint64_t)V.randomProposals * (uint64_t)run);
  const uint32_t aP = minstd_jump(1u, (uint64_t)V.randomProposals);
  for (int x = x0; x < x1; ++x) {
    rank[(size_t)d * n + (size_t)y * V.W + x] = (int)state;
    if (random_gate(V, d, own, (size_t)y * V.W + x)) {
      state = minstd_mulmod(state, aP);
    }
  }
}

// (the body of k_random_proposals / k_random_proposals_w3: the same code under two register budgets, see below)
__device__ __forceinline__ void random_proposals_body(const LevelView& V, const int* __restrict__ rank, int tilesX, int tilesPerDst) {
  extern __shared__ SsdPair ldsPairs[];
  const int dl = blockIdx.y;
  const int d = V.dst0 + dl;
  int x, y;
  tile_pixel(DERP_RANDOM_SWIZZLE ? xcd_swizzle(blockIdx.x, gridDim.x, V.xcdRotate ? d : 0) : (int)blockIdx.x, tilesX, x, y);
  // the wave's 10x10 window of destination colours (PatchWin): filled by all 64 lanes before anything diverges
  __shared__ PatchWin patchWin[DERP_COST_BLOCK / 64];
  PatchWin* win = &patchWin[threadIdx.x >> 6];
  patch_window_fill(V, V.dst2src[d], x - (int)(threadIdx.x & 7), y - (int)((threadIdx.x & 63) >> 3), win);
#if DERP_ATAN_LUT && DERP_LEAN_PROJ && defined(__HIP_DEVICE_COMPILE__)
  __shared__ double atanLut[kAtanLutDoubles];
  atan_lut_fill(atanLut);
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, atanLut};
#else
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, nullptr};
#endif
  __syncthreads();
  unsigned nCost = 0, nPair = 0, nSlots = 0, nSlotsFirst = 0;
  PhaseTimers tm;
  const unsigned tk0 = phase_clock();
  if (x >= 1 && y >= 1 && x < V.W - 1 && y < V.H - 1) {
    const int own = V.dst2src[d];
    // per-destination planes (wave-uniform bases) and a 32-bit pixel index, as in k_ping_pong
    const size_t n = (size_t)V.W * V.H;
    const unsigned idx = (unsigned)y * (unsigned)V.W + (unsigned)x;
    float* disp = V.disparity + (size_t)d * n;
    if ((V.fovMask + (size_t)d * n)[idx]) {
      if (!(V.srcFg + (size_t)own * n)[idx]) {
        disp[idx] = (V.bgDisp + (size_t)d * n)[idx];
      } else if (random_gate(V, d, own, idx)) {
        PixCtx px;
        load_pixctx<!DERP_RANDOM_RELOAD_RAY>(V, d, own, x, y, win, px);
        const unsigned cull = DERP_SOURCE_CULL ? behind_sources(V, d, idx) : 0u;
        // What the proposal loop carries per lane: the pixel index, the current disparity / cost / confidence / pair count,
        // the acceptance threshold, the amplitude, the engine state, the pair counter; the lower bound of the range is
        // read again per proposal.
        // One copy of computeCost serves the evaluation of the current disparity (i = -1) and the proposals: the kernel's
        // code is half the size it was with two inlined copies (45 KB against a 64 KB instruction cache).
        float currDisp = at32(disp, idx);
        float currCost = 0.f, costThresh = 0.f, amplitude = 0.f;
        // the confidence and pair count of the current best ride in the lane's spare pair slot (S slots are allocated, S - 1
        // sources use them): results, read back once after the loop
        const int spare = V.S - 1;
        const float maxDisp = 1.0f / V.minDepthM;
        uint32_t state = (uint32_t)at32(rank + (size_t)d * n, idx);  // minstd_rand0 positioned by k_row_rank
        for (int i = -1; i < V.randomProposals; ++i) {
          unsigned pi = idx;
          asm("" : "+v"(pi) : "s"(i));  // (opaque per proposal: the loads below stay inside the loop)
          const float minDisp = V.hasFg ? at32(V.bgDisp + (size_t)d * n, pi) : (1.0f / V.maxDepthM);
          float propDisp = currDisp;
          if (i >= 0) {
            const float lo = fmaxf(minDisp, currDisp - amplitude), hi = fminf(maxDisp, currDisp + amplitude);
            propDisp = minstd_uniform(state, lo, hi);
          }
          unsigned np = 0;
          const float2 pr = compute_cost<DERP_RANDOM_SSD_SCALAR != 0, true, DERP_RANDOM_RELOAD_RAY != 0>(V, dl, own, px, propDisp, pairs, np, cull, pi, i < 0 ? &nSlotsFirst : &nSlots, &tm);
          nPair += np;
          bool take;
          if (i < 0) {
            costThresh = fminf(0.5f * pr.x, 5.0f);  // kRandomPropMaxCost
            amplitude = (maxDisp - minDisp) / 2.0f;
            take = true;
          } else {
            take = pr.x < currCost && pr.x < costThresh;
            if (take) {
              amplitude /= 2.0f;
            }
          }
          if (take) {
            currCost = pr.x;
            currDisp = propDisp;
            pairs.set(spare, SsdPair{pr.y, __uint_as_float(np)});
          }
        }
        {
          unsigned pi = idx;
          asm("" : "+v"(pi) : "v"(currCost));
          const SsdPair best = pairs.get(spare);
          at32(V.confidence + (size_t)d * n, pi) = best.first;
          at32(V.pairCount + (size_t)d * n, pi) = (uint8_t)__float_as_uint(best.second);
          at32(disp, pi) = currDisp;
          at32(V.cost + (size_t)d * n, pi) = currCost;
        }
        nCost = 1u + (unsigned)max(V.randomProposals, 0);
      }
    }
  }
#if DERP_PHASE_TIMERS
  if ((threadIdx.x & 63) == 0) {
    const unsigned body = phase_clock() - tk0;
    atomicAdd(&V.counters[0], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.proj : body));
    atomicAdd(&V.counters[1], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.ssd : body - tm.inside));
    atomicAdd(&V.counters[3], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.select : 0u));
  }
  return;
#endif
  flush_counters(V, nCost, nPair);
#ifdef DERP_COUNT_UNION
  atomicAdd(&V.counters[3], (unsigned long long)nSlots);       // random candidates: lane-slots the waves walked
  atomicAdd(&V.counters[2], (unsigned long long)nSlotsFirst);  // the current disparity's evaluation
#endif
}

// Two register budgets for the same body. Up to 16 cameras a wave's LDS (8 B x 64 lanes x S pair slots + the patch window)
// lets sixteen one-wave blocks share a CU: the kernels are held to 128 VGPRs = FOUR waves per SIMD. With more cameras the
// pair slots fill the LDS first (24 cameras: twelve blocks = three waves) and the 128-register build would pay its spills
// and tighter schedule for nothing: the _w3 kernels keep the 168 registers of three waves (config 4, same box: 261 -> 271
// Mpix/s, profiles/r06_kernel_variants.txt run 9). The launcher picks by camera count (cost_four_waves, derp_capi.hip).
__global__ void __launch_bounds__(DERP_COST_BLOCK, DERP_RANDOM_MIN_WAVES)
    k_random_proposals(LevelView V, const int* __restrict__ rank, int tilesX, int tilesPerDst) {
  random_proposals_body(V, rank, tilesX, tilesPerDst);
}
__global__ void __launch_bounds__(DERP_COST_BLOCK, 3)
    k_random_proposals_w3(LevelView V, const int* __restrict__ rank, int tilesX, int tilesPerDst) {
  random_proposals_body(V, rank, tilesX, tilesPerDst);
}

// ----------------------------------------------------------------------------------------
// ping-pong propagation — Derp.cpp:403-538. Jacobi: reads disparity, writes dispRes / costRes.
// ----------------------------------------------------------------------------------------
__constant__ int kCandidates[9][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-2, -2}, {2, -2}, {-2, 2}, {2, 2}};

__device__ __forceinline__ void ping_pong_body(const LevelView& V, const uint8_t* __restrict__ changed, float* __restrict__ dispRes,
                                               float* __restrict__ costRes, int tilesX, int useMemo) {
  extern __shared__ SsdPair ldsPairs[];
  const int dl = blockIdx.y;
  const int d = V.dst0 + dl;
  int x, y;
  tile_pixel(xcd_swizzle(blockIdx.x, gridDim.x, V.xcdRotate ? d : 0), tilesX, x, y);
  // the wave's 10x10 window of destination colours (PatchWin): filled by all 64 lanes before anything diverges
  __shared__ PatchWin patchWin[DERP_COST_BLOCK / 64];
  PatchWin* win = &patchWin[threadIdx.x >> 6];
  patch_window_fill(V, V.dst2src[d], x - (int)(threadIdx.x & 7), y - (int)((threadIdx.x & 63) >> 3), win);
#if DERP_ATAN_LUT && DERP_LEAN_PROJ && defined(__HIP_DEVICE_COMPILE__)
  __shared__ double atanLut[kAtanLutDoubles];
  atan_lut_fill(atanLut);
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, atanLut};
#else
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, nullptr};
#endif
  __syncthreads();
  // per-lane counters in one register: pairs (bits 0..15: <= 9 * 31), cost evaluations (16..23: <= 9), memoised (24..)
  unsigned counts = 0;
  PhaseTimers tm;
  const unsigned tk0 = phase_clock();
  if (x < V.W && y < V.H) {
    const int own = V.dst2src[d];
    // per-destination planes (wave-uniform bases) and a 32-bit pixel index: the loads take the scalar-base + 32-bit
    // offset form, and nothing 64-bit per lane has to survive the candidate loop
    const size_t n = (size_t)V.W * V.H;
    const unsigned idx = (unsigned)y * (unsigned)V.W + (unsigned)x;
    const float* disp = V.disparity + (size_t)d * n;
    const uint8_t* fov = V.fovMask + (size_t)d * n;
    const uint8_t* chg = changed + (size_t)d * n;
    dispRes += (size_t)d * n;
    costRes += (size_t)d * n;
    float outDisp = disp[idx];
    float outCost = __builtin_inff();
    const bool interior = x >= 1 && y >= 1 && x < V.W - 1 && y < V.H - 1;
    if (interior && fov[idx]) {
      if (!V.srcFg[(size_t)own * n + idx]) {
        outDisp = V.bgDisp[(size_t)d * n + idx];
      } else if (!(V.srcVar[(size_t)own * n + idx] < V.varNoiseFloor)) {
        PixCtx px;
        load_pixctx<!DERP_PP_RELOAD_RAY>(V, d, own, x, y, win, px);
        const unsigned cull = DERP_SOURCE_CULL ? behind_sources(V, d, idx) : 0u;
        float bestCost = __builtin_inff();
        float bestDisp = outDisp;
        // what the candidate loop carries per lane: the pixel (x | y << 16), the best candidate so far, the counters.
        // Everything else is derived again per candidate from an opaque copy of `xy` (four plain instructions) instead
        // of riding through computeCost's registers: x, y, the pixel index, the background disparity.
        const unsigned xy = (unsigned)x | ((unsigned)y << 16);
        for (int k = 0; k < 9; ++k) {
          unsigned q = xy;
          asm("" : "+v"(q) : "s"(k));
          const int px0 = (int)(q & 0xffffu), py0 = (int)(q >> 16);
          const unsigned pidx = (unsigned)py0 * (unsigned)V.W + (unsigned)px0;
          const int xx = min(max(px0 + kCandidates[k][0], 0), V.W - 1);
          const int yy = min(max(py0 + kCandidates[k][1], 0), V.H - 1);
          const unsigned j = (unsigned)yy * (unsigned)V.W + (unsigned)xx;
          if (fov[j]) {
            const float cand = disp[j];
            const float bg = V.hasFg ? (V.bgDisp + (size_t)d * n)[pidx] : 0.f;
            if (cand >= bg && chg[j]) {
              float2 r;
              // Candidate (0,0) is the pixel's own disparity. In the first iteration, where random
              // proposals evaluated this pixel, computeCost(own disparity) is exactly the value they
              // left in cost / confidence (a pure function of the same arguments): reuse it.
              const float memoConf = (k == 0 && useMemo) ? (V.confidence + (size_t)d * n)[pidx] : 0.0f;
              if (memoConf != 0.0f) {
                r = make_float2((V.cost + (size_t)d * n)[pidx], memoConf);
                counts += (unsigned)(V.pairCount + (size_t)d * n)[pidx] + (1u << 24);
              } else {
                unsigned np = 0;
                r = compute_cost<DERP_COST_SSD_SCALAR != 0, false, DERP_PP_RELOAD_RAY != 0>(V, dl, own, px, cand, pairs, np, cull, pidx, nullptr, &tm);
                counts += np;
              }
              counts += 1u << 16;
              if (r.x < bestCost) {
                bestCost = r.x;
                bestDisp = cand;
              }
            }
          }
        }
        outDisp = bestDisp;
        outCost = bestCost;
      }
    }
    dispRes[idx] = outDisp;
    costRes[idx] = outCost;
  }
#if DERP_PHASE_TIMERS
  if ((threadIdx.x & 63) == 0) {
    const unsigned body = phase_clock() - tk0;
    atomicAdd(&V.counters[0], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.proj : body));
    atomicAdd(&V.counters[1], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.ssd : body - tm.inside));
    atomicAdd(&V.counters[3], (unsigned long long)(DERP_PHASE_TIMERS == 1 ? tm.select : 0u));
  }
  return;
#endif
  unsigned nMemo = counts >> 24;
  flush_counters(V, (counts >> 16) & 0xffu, counts & 0xffffu);
  for (int off = 32; off > 0; off >>= 1) {
    nMemo += __shfl_down(nMemo, off);
  }
  if ((threadIdx.x & 63) == 0 && nMemo) {
    atomicAdd(&V.counters[3], (unsigned long long)nMemo);
  }
}

__global__ void __launch_bounds__(DERP_COST_BLOCK, DERP_COST_MIN_WAVES)
    k_ping_pong(LevelView V, const uint8_t* __restrict__ changed, float* __restrict__ dispRes,
                float* __restrict__ costRes, int tilesX, int useMemo) {
  ping_pong_body(V, changed, dispRes, costRes, tilesX, useMemo);
}
__global__ void __launch_bounds__(DERP_COST_BLOCK, 3)  // more than 16 cameras: see k_random_proposals_w3
    k_ping_pong_w3(LevelView V, const uint8_t* __restrict__ changed, float* __restrict__ dispRes,
                   float* __restrict__ costRes, int tilesX, int useMemo) {
  ping_pong_body(V, changed, dispRes, costRes, tilesX, useMemo);
}

// changed = disp != dispRes; dispRes -> disp; costRes -> cost (Derp.cpp:527-529)
__global__ void k_ping_pong_commit(float* __restrict__ disp, float* __restrict__ cost, const float* __restrict__ dispRes,
                                   const float* __restrict__ costRes, uint8_t* __restrict__ changed, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float a = disp[i], b = dispRes[i];
    changed[i] = (a != b);
    disp[i] = b;
    cost[i] = costRes[i];
  }
}

// ----------------------------------------------------------------------------------------
// handleDisparityMismatches — Derp.cpp:553-748 (off unless level <= --mismatches_start_level).
// Jacobi over destinations: reads every camera's disparity, writes newDisp + the mismatch mask.
// The reference indexes dstDisparity(srcIdx), i.e. requires dst i == src i (checked on the host).
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp_fetch(const float* img, int x, int y, int W, int H) {
  return img[(size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)];
}

__global__ void __launch_bounds__(256)
    k_mismatch(LevelView V, float* __restrict__ newDisp, uint8_t* __restrict__ mismatchMask) {
  extern __shared__ float ldsMis[];
  const int d = blockIdx.z;
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= V.W || y >= V.H) {
    return;
  }
  const size_t n = (size_t)V.W * V.H, idx = (size_t)y * V.W + x;
  if (!V.fovMask[(size_t)d * n + idx]) {
    newDisp[(size_t)d * n + idx] = __builtin_nanf("");  // dstDispNew starts as NaN and is never written here
    return;
  }
  float* mis = ldsMis + threadIdx.x;  // mis[i * 256]
  const int own = V.dst2src[d];
  const float curr = V.disparity[(size_t)d * n + idx];
  int nMatch = 0, nMis = 0;
  if (V.srcFg[(size_t)own * n + idx]) {  // getSrcMismatches
    const Cam& cd = V.camsDst[d];
    const D3 dir = rig_direction(cd, (x + 0.5) / (double)V.W, (y + 0.5) / (double)V.H, cd.principal[0], cd.principal[1],
                                 cd.focal[0], cd.focal[1]);
    const double depth = (double)(1.0f / curr);
    const D3 pWorld = {cd.pos[0] + dir.x * depth, cd.pos[1] + dir.y * depth, cd.pos[2] + dir.z * depth};
    const float dMin = (1.0f - 0.1f) * curr, dMax = (1.0f + 0.1f) * curr;
    for (int s = 0; s < V.S; ++s) {
      if (s == own) {
        continue;
      }
      const Cam& cs = V.camsSrc[s];
      D2 pn;
      if (!sees(cs, pWorld, cs.principal[0], cs.principal[1], cs.focal[0], cs.focal[1], 1.0, 1.0, pn)) {
        continue;
      }
      const float sx = (float)(pn.x * (double)V.W), sy = (float)(pn.y * (double)V.H);
      const float xf = roundf(sx), yf = roundf(sy);
      const int xi = (int)xf, yi = (int)yf;
      const float xw = sx - xf + 0.5f, yw = sy - yf + 0.5f;
      const float* img = V.disparity + (size_t)s * n;  // dstDisparity(srcIdx)
      const float dSrc = bilerp_f(clamp_fetch(img, xi - 1, yi - 1, V.W, V.H), clamp_fetch(img, xi, yi - 1, V.W, V.H),
                                  clamp_fetch(img, xi - 1, yi, V.W, V.H), clamp_fetch(img, xi, yi, V.W, V.H),
                                  (1 - xw) * (1 - yw), xw * (1 - yw), (1 - xw) * yw, xw * yw);
      if (dMin <= dSrc && dSrc <= dMax) {
        ++nMatch;
      } else {
        mis[(nMis++) * 256] = dSrc;
      }
    }
  }
  // updateDstDisparityAndMismatchMask
  bool mask = false;
  float dispNew = curr;
  if (nMatch + nMis != 0) {
    const float var = V.srcVar[(size_t)own * n + idx];
    if (!(nMatch >= 1 || V.varHighThresh < var || var < V.varNoiseFloor)) {
      mask = true;
      for (int i = 1; i < nMis; ++i) {  // std::sort: ascending values
        const float v = mis[i * 256];
        int k = i;
        while (k > 0 && mis[(k - 1) * 256] > v) {
          mis[k * 256] = mis[(k - 1) * 256];
          --k;
        }
        mis[k * 256] = v;
      }
      int closer = 0;
      for (; closer < nMis; ++closer) {
        if (mis[closer * 256] >= curr) {
          break;
        }
      }
      const float m = mis[(closer / 2) * 256];
      dispNew = (m < curr) ? m : curr;  // std::min(dispCurr, dispMismatches[median])
    }
  }
  mismatchMask[(size_t)d * n + idx] = mask;
  newDisp[(size_t)d * n + idx] = dispNew;
}

// LayerDisparities.cpp:45-55: mask = fg > 0; layer = fg*mask + bg*(1-mask); cv::imwrite(layer * 255) -> CV_8U
__global__ void k_layer_disparities(const float* __restrict__ fg, const float* __restrict__ bg, size_t n,
                                    uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float m = fg[i] > 0.0f ? 1.0f : 0.0f;
    const float layer = fg[i] * m + bg[i] * (1 - m);
    const int r = cv_round(layer * 255.0f);
    out[i] = (uint8_t)min(max(r, 0), 255);
  }
}

// cost map of a caller-supplied disparity image (test hook over compute_cost)
__global__ void __launch_bounds__(DERP_COST_BLOCK, DERP_COST_MIN_WAVES)
    k_cost_map(LevelView V, int d, const float* __restrict__ dispIn, float* __restrict__ costOut,
               float* __restrict__ confOut, int tilesX) {
  extern __shared__ SsdPair ldsPairs[];
  const int dl = d - V.dst0;
  int x, y;
  tile_pixel(blockIdx.x, tilesX, x, y);
  // the wave's 10x10 window of destination colours (PatchWin): filled by all 64 lanes before anything diverges
  __shared__ PatchWin patchWin[DERP_COST_BLOCK / 64];
  PatchWin* win = &patchWin[threadIdx.x >> 6];
  patch_window_fill(V, V.dst2src[d], x - (int)(threadIdx.x & 7), y - (int)((threadIdx.x & 63) >> 3), win);
#if DERP_ATAN_LUT && DERP_LEAN_PROJ && defined(__HIP_DEVICE_COMPILE__)
  __shared__ double atanLut[kAtanLutDoubles];
  atan_lut_fill(atanLut);
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, atanLut};
#else
  LdsPairs pairs{ldsPairs + threadIdx.x, (int)blockDim.x, nullptr};
#endif
  __syncthreads();
  unsigned nCost = 0, nPair = 0;
  if (x >= 1 && y >= 1 && x < V.W - 1 && y < V.H - 1) {
    const int own = V.dst2src[d];
    PixCtx px;
    load_pixctx(V, d, own, x, y, win, px);
    const float2 r = compute_cost<DERP_COST_SSD_SCALAR != 0>(V, dl, own, px, dispIn[(size_t)y * V.W + x], pairs, nPair);
    ++nCost;
    costOut[(size_t)y * V.W + x] = r.x;
    confOut[(size_t)y * V.W + x] = r.y;
  }
  flush_counters(V, nCost, nPair);
}

// ----------------------------------------------------------------------------------------
// filters
// ----------------------------------------------------------------------------------------
// expf exactly as the reference's libm computes it. glibc >= 2.27 (sysdeps/ieee754/flt-32/e_expf.c,
// from ARM's optimized-routines): exp(x) = 2^(k/32) * p(r) in fp64 with a 32-entry table and a cubic,
// rounded once to fp32. On x86-64 hosts with FMA the ifunc picks the -mfma build, in which GCC
// contracts BOTH uses of z = InvLn2N * x (z + SHIFT and z - kd) and the three polynomial steps; that
// variant is restated here with explicit fma (checked bit-for-bit against this image's libm on 7.6e7
// inputs, 0 mismatches; the non-FMA build differs from it on ~3 inputs in 1e8).
__constant__ unsigned long long kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// Correctly rounded fp32 quotient by a constant without the fp32 divide sequence: for floats y, d the
// product RN64(y * RN64(1/d)) carries a relative error <= 2^-52, while y/d can come no closer than
// 2^-49 (relative) to a rounding boundary of fp32 unless it sits exactly on a representable value
// (y - m*d is a non-zero multiple of the 49-bit product's last bit for every 25-bit midpoint m), so
// rounding the double product to float gives exactly the IEEE result of y / d. `rcp` = 1.0 / (double)d.
__device__ __forceinline__ float div_by_const(float y, double rcp) {
  return (float)((double)y * rcp);
}

// `tab` = kExp2fTab staged in LDS by the caller (a divergent __constant__ index would be a global load per call)
__device__ __forceinline__ float expf_glibc(float x, const unsigned long long* tab) {
  const double InvLn2N = 0x1.71547652b82fep+0 * 32;
  const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32,
               C2 = 0x1.62e42ff0c52d6p-1 / 32;
  const double SHIFT = 0x1.8p+52;
  const unsigned abstop = (__float_as_uint(x) >> 20) & 0x7ff;
  if (abstop >= (0x42b00000u >> 20)) {  // |x| >= 88 or NaN
    if (__float_as_uint(x) == 0xff800000u) {
      return 0.0f;
    }
    if (abstop >= (0x7f800000u >> 20)) {
      return x + x;
    }
    if (x > 0x1.62e42ep6f) {
      return __builtin_inff();
    }
    if (x < -0x1.9fe368p6f) {
      return 0.0f;
    }
  }
  const double xd = (double)x;
  double kd = __builtin_fma(InvLn2N, xd, SHIFT);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= SHIFT;
  const double r = __builtin_fma(InvLn2N, xd, -kd);
  const unsigned long long t = tab[ki & 31] + (ki << 47);
  const double sc = __longlong_as_double((long long)t);
  const double z = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * sc;
  return (float)y;
}

// generalizedJointBilateralFilter — TemporalBilateralFilter.h:39-124. One 16x16 pixel tile per
// block; the (16+2r)^2 neighbourhood (guide colour, mask, image value, already clamped to the image
// edge like the reference's sample coordinates) is staged once in LDS, so the (2r+1)^2 taps of each
// pixel are LDS reads instead of 3 global loads each. Tap order (v outer, u inner), the masked-tap
// skip and the float accumulation are the reference's.
//   GUIDE_U16 = true : guide is BGRX u16 (TGuide = Vec3w, factor 1/65535)   — Derp.cpp:875-902
//   GUIDE_U16 = false: guide is 3 x f32 (TGuide = Vec3f, factor 1)          — UpsampleDisparity.cpp:109-128
template <bool GUIDE_U16>
__global__ void __launch_bounds__(256)
    k_joint_bilateral(const float* __restrict__ image, const void* __restrict__ guideV,
                      const uint8_t* __restrict__ mask, int W, int H, int radius, float sigma, float weight0,
                      float weight1, float weight2, float* __restrict__ out, size_t planeStride, size_t guideStride,
                      const int* __restrict__ guideIndex) {
  extern __shared__ float ldsTile[];
  __shared__ unsigned long long expTab[32];
  if (threadIdx.x < 32) {
    expTab[threadIdx.x] = kExp2fTab[threadIdx.x];
  }
  const int p = blockIdx.z;
  const int T = 16 + 2 * radius;
  // row pitch: the next multiple of 16 texels. A wave reads four rows of 16 lanes; ds_read_b128 serves 16 lanes per
  // pass (e.g. lanes 0-3, 12-15 of one row with lanes 4-11 of the next), and with 16-byte records those lanes cover
  // all 64 banks exactly once iff consecutive rows start 0 mod 256 bytes apart (round 3 measured 32 % of the
  // LDS-active cycles as bank conflicts with pitch T)
  const int P = (T + 15) & ~15;
  const int n = P * T;
  // per tile texel one 16-byte record (guide x 3 already scaled by the factor, image): one LDS read per tap
  float4* tTex = reinterpret_cast<float4*>(ldsTile);  // [T rows][P]
  uint8_t* tMask = reinterpret_cast<uint8_t*>(ldsTile + 4 * n);
  const float* img = image + (size_t)p * planeStride;
  const uint8_t* m = mask + (size_t)p * planeStride;
  const size_t gplane = (size_t)(guideIndex ? guideIndex[p] : p) * guideStride;
  const int x0 = blockIdx.x * 16 - radius, y0 = blockIdx.y * 16 - radius;
  for (int k = threadIdx.x; k < T * T; k += 256) {
    const int ty = k / T, tx = k - ty * T;
    const int i = ty * P + tx;
    const int sx = min(max(x0 + tx, 0), W - 1), sy = min(max(y0 + ty, 0), H - 1);
    const size_t j = (size_t)sy * W + sx;
    tMask[i] = m[j];
    if (GUIDE_U16) {
      const ushort4 g = reinterpret_cast<const ushort4*>(guideV)[gplane + j];
      const float factor = 1 / 65535.0f;
      tTex[i] = make_float4(g.x * factor, g.y * factor, g.z * factor, img[j]);
    } else {
      const float* g = reinterpret_cast<const float*>(guideV) + (gplane + j) * 3;
      const float factor = 1 / 1.0f;
      tTex[i] = make_float4(g[0] * factor, g[1] * factor, g[2] * factor, img[j]);
    }
  }
  __syncthreads();
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = blockIdx.x * 16 + lx, y = blockIdx.y * 16 + ly;
  if (x >= W || y >= H) {
    return;
  }
  const int c = (ly + radius) * P + lx + radius;
  const float4 centre = tTex[c];
  float result = centre.w;
  if (tMask[c]) {
    const float g0 = centre.x, g1 = centre.y, g2 = centre.z;
    const float denom = 2.0f * (sigma * sigma);
    const double rcpDenom = 1.0 / (double)denom, rcp3 = 1.0 / 3.0;
    float sumWeight = 0.f, weightedAvg = 0.f;
    for (int v = -radius; v <= radius; ++v) {
      const int row = c + v * P;
      for (int u = -radius; u <= radius; ++u) {
        const int j = row + u;
        if (!tMask[j]) {
          continue;
        }
        const float4 t = tTex[j];
        const float d0 = g0 - t.x, d1 = g1 - t.y, d2 = g2 - t.z;
        const float colorDiffSq = weight0 * (d0 * d0) + weight1 * (d1 * d1) + weight2 * (d2 * d2);
        const float weight = expf_glibc(div_by_const(div_by_const(-colorDiffSq, rcp3), rcpDenom), expTab);  // (-c / 3.0f) / denom
        sumWeight += weight;
        weightedAvg += weight * t.w;
      }
    }
    if (sumWeight != 0.0f) {
      result = weightedAvg / sumWeight;
    }
  }
  out[(size_t)p * planeStride + (size_t)y * W + x] = result;
}

// maskedMedianBlur — CvUtil.h:336-385; optional fused maskFov (Derp.cpp:940-951).
// Median of the taps that are in bounds, inside the mask, not NaN and not 0; even count -> mean of
// the two middle values (in double, as the reference's "/ 2.0"). Radius 1 (the value DerpCLI uses,
// Derp.h:36) keeps the 9 candidates in registers and orders them with a 25-exchange sorting network,
// invalid taps as +inf; the result depends only on the multiset of values, not on the sort used.
__device__ __forceinline__ void cswap(float& a, float& b) {
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  a = lo;
  b = hi;
}

__global__ void k_masked_median(const float* __restrict__ image, const float* __restrict__ background,
                                const uint8_t* __restrict__ mask, int W, int H, int radius, float* __restrict__ out,
                                size_t planeStride, const uint8_t* __restrict__ fovForNan) {
  const int p = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const size_t idx = (size_t)y * W + x;
  const float* img = image + (size_t)p * planeStride;
  const uint8_t* m = mask + (size_t)p * planeStride;
  float result = 0.0f;
  if (!m[idx]) {
    if (background) {
      result = background[(size_t)p * planeStride + idx];
    }
  } else if (radius == 1) {
    const float inf = __builtin_inff();
    float v[9];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int yy = y + j - 1, xx = x + i - 1;
        float t = inf;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const size_t q = (size_t)yy * W + xx;
          const float val = img[q];
          if (m[q] && !isnan(val) && val != 0) {
            t = val;
            ++cnt;
          }
        }
        v[j * 3 + i] = t;
      }
    }
    // 9-input sorting network (25 compare-exchanges)
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[1], v[2]); cswap(v[4], v[5]); cswap(v[7], v[8]);
    cswap(v[0], v[1]); cswap(v[3], v[4]); cswap(v[6], v[7]);
    cswap(v[0], v[3]); cswap(v[3], v[6]); cswap(v[0], v[3]);
    cswap(v[1], v[4]); cswap(v[4], v[7]); cswap(v[1], v[4]);
    cswap(v[2], v[5]); cswap(v[5], v[8]); cswap(v[2], v[5]);
    cswap(v[1], v[3]); cswap(v[5], v[7]); cswap(v[2], v[6]);
    cswap(v[4], v[6]); cswap(v[2], v[4]); cswap(v[2], v[3]);
    cswap(v[5], v[6]);
    if (cnt > 0) {
      const int h = cnt / 2;
      float lo = v[0], hi = v[0];
#pragma unroll
      for (int k = 0; k < 9; ++k) {  // v[h - 1], v[h] without dynamic register indexing
        lo = (k == h - 1) ? v[k] : lo;
        hi = (k == h) ? v[k] : hi;
      }
      result = (cnt & 1) ? hi : (float)(((double)(lo + hi)) / 2.0);
    }
  } else {
    float vals[25];
    int cnt = 0;
    for (int yy = y - radius; yy <= y + radius; ++yy) {
      for (int xx = x - radius; xx <= x + radius; ++xx) {
        if (0 > yy || yy >= H || 0 > xx || xx >= W) {
          continue;
        }
        const size_t j = (size_t)yy * W + xx;
        if (!m[j]) {
          continue;
        }
        const float v = img[j];
        if (isnan(v) || v == 0) {
          continue;
        }
        int k = cnt++;
        while (k > 0 && vals[k - 1] > v) {
          vals[k] = vals[k - 1];
          --k;
        }
        vals[k] = v;
      }
    }
    if (cnt > 0) {
      const int h = cnt / 2;
      if (cnt % 2 == 1) {
        result = vals[h];
      } else {
        result = (float)(((double)(vals[h - 1] + vals[h])) / 2.0);
      }
    }
  }
  if (fovForNan && !fovForNan[(size_t)p * planeStride + idx]) {
    result = __builtin_nanf("");
  }
  out[(size_t)p * planeStride + idx] = result;
}

__global__ void k_mask_fov(float* __restrict__ disp, const uint8_t* __restrict__ fov, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    if (!fov[i]) {
      disp[i] = __builtin_nanf("");
    }
  }
}

__global__ void k_and_masks(const uint8_t* fov, const uint8_t* srcFg, const int* __restrict__ dst2src, int dst0,
                            size_t n, uint8_t* out) {
  const int d = dst0 + blockIdx.y;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    out[(size_t)d * n + i] = fov[(size_t)d * n + i] & srcFg[(size_t)dst2src[d] * n + i];
  }
}

// ----------------------------------------------------------------------------------------
// upsample — UpsampleDisparityLib.cpp:98-147
// ----------------------------------------------------------------------------------------
// Lanczos4 coefficients are computed on the host (resize.cpp interpolateLanczos4 uses libm
// sin/cos in fp64): xofs/alpha per output column, yofs/beta per output row.
__global__ void k_lanczos_h(const float* __restrict__ in, int SW, int SH, int DW, const int* __restrict__ xofs,
                            const float* __restrict__ alpha, float* __restrict__ tmp, size_t inStride,
                            size_t tmpStride) {
  const int p = blockIdx.z;
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= DW || y >= SH) {
    return;
  }
  const float* S = in + (size_t)p * inStride + (size_t)y * SW;
  const float* a = alpha + (size_t)dx * 8;
  const int sx = xofs[dx] - 3;
  float v = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sxj = min(max(sx + j, 0), SW - 1);
    float sv = S[sxj];
    if (sv != sv) {
      sv = 1e-4f;  // OpenCV doesn't handle NaNs: NaN -> minDisp (UpsampleDisparityLib.cpp:141-144)
    }
    v += sv * a[j];
  }
  tmp[(size_t)p * tmpStride + (size_t)y * DW + dx] = v;
}
__global__ void k_lanczos_v(const float* __restrict__ tmp, int SH, int DW, int DH, const int* __restrict__ yofs,
                            const float* __restrict__ beta, float* __restrict__ out, size_t tmpStride,
                            size_t outStride) {
  const int p = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= DW || dy >= DH) {
    return;
  }
  const float* T = tmp + (size_t)p * tmpStride;
  const float* b = beta + (size_t)dy * 8;
  const int sy = yofs[dy] - 3;
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int syk = min(max(sy + k, 0), SH - 1);
    r[k] = T[(size_t)syk * DW + x];
  }
  out[(size_t)p * outStride + (size_t)dy * DW + x] =
      r[0] * b[0] + r[1] * b[1] + r[2] * b[2] + r[3] * b[3] + r[4] * b[4] + r[5] * b[5] + r[6] * b[6] + r[7] * b[7];
}

// masked path, steps 1-3: coarse value outside (fov & fg) -> NaN, INTER_NEAREST, NaN outside maskUp
// blockIdx.z = plane (destination camera): coarse planes are SW * SH apart, fine planes DW * DH
__global__ void k_upsample_nearest_masked(const float* __restrict__ in, const uint8_t* __restrict__ mask, int SW,
                                          int SH, const uint8_t* __restrict__ maskUp, int DW, int DH,
                                          float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= DW || y >= DH) {
    return;
  }
  {
    const size_t ps = (size_t)blockIdx.z * SW * SH, pd = (size_t)blockIdx.z * DW * DH;
    in += ps;
    mask += ps;
    maskUp += pd;
    out += pd;
  }
  const double ifx = (double)SW / DW, ify = (double)SH / DH;
  const int sx = min((int)floor(x * ifx), SW - 1), sy = min((int)floor(y * ify), SH - 1);
  float v = in[(size_t)sy * SW + sx];
  if (!mask[(size_t)sy * SW + sx] || !maskUp[(size_t)y * DW + x]) {
    v = __builtin_nanf("");
  }
  out[(size_t)y * DW + x] = v;
}
// step 4: replaceNans — first value > 0 along the clockwise spiral (host-generated offsets)
__global__ void k_spiral_fill(const float* __restrict__ dispUp, const float* __restrict__ bgUp,
                              const uint8_t* __restrict__ maskUp, int W, int H, const int2* __restrict__ spiral,
                              int nSpiral, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  {
    const size_t pd = (size_t)blockIdx.z * W * H;  // blockIdx.z = plane (destination camera)
    dispUp += pd;
    bgUp += pd;
    maskUp += pd;
    out += pd;
  }
  const size_t idx = (size_t)y * W + x;
  float v = dispUp[idx];
  if (maskUp[idx] && !(v > 0)) {
    for (int k = 0; k < nSpiral; ++k) {
      const int xx = min(max(x + spiral[k].x, 0), W - 1), yy = min(max(y + spiral[k].y, 0), H - 1);
      const float c = dispUp[(size_t)yy * W + xx];
      if (c > 0) {
        v = c;
        break;
      }
    }
  }
  if (isnan(v) || v == 0) {
    v = bgUp[idx];
  }
  out[idx] = v;
}

// ----------------------------------------------------------------------------------------
// pyramid builder — scripts/render/resize.py:51-85: cv2.resize(full frame, (w, h), INTER_AREA) per
// level (+ threshold 127 for masks). cv::resize's three INTER_AREA code paths when shrinking:
// 2x2 integer average (a+b+c+d+2)>>2 on 8U/16U, float block sum * (1/area) for other integer
// scales, and the fractional-scale tables of computeResizeAreaTab (built on the host).
//   KIND 0: BGR u16 interleaved in -> BGRX ushort4 out (the pyramid's colour layout)
//   KIND 1: u8 in -> u8 {0,1} out = (resized > threshold)        (fg masks: threshold 127)
//   KIND 2: f32 in -> f32 out                                     (background disparity)
//   KIND 3: f32 x3 interleaved in -> f32 x3 interleaved out        (UpsampleDisparity's Vec3f colour guide,
//           cv_util::resizeImage CvUtil.h:139-147; channels are independent in cv::resize)
// ----------------------------------------------------------------------------------------
struct AreaAxis {
  const int* start;    // [dsize + 1] first table entry of each output index
  const int* si;       // source index per entry
  const float* alpha;  // weight per entry
  int iscale;          // integer scale factor, or 0 when fractional
};

template <int KIND>
__device__ __forceinline__ float area_src(const void* src, size_t pix, int c) {
  if (KIND == 0) {
    return (float)reinterpret_cast<const uint16_t*>(src)[pix * 3 + c];
  } else if (KIND == 1) {
    return (float)reinterpret_cast<const uint8_t*>(src)[pix];
  } else if (KIND == 3) {
    return reinterpret_cast<const float*>(src)[pix * 3 + c];
  } else {
    return reinterpret_cast<const float*>(src)[pix];
  }
}

// cv::resize(INTER_AREA) when the image is ENLARGED along an axis: OpenCV emulates it with its bilinear machinery and
// area-mode taps (resize.cpp; oracle_cv.h resizeLinearAreaF32 spells the rule out): per axis
//   s = floor(d * scale), f = (d + 1) - (s + 1) / scale, f = f <= 0 ? 0 : f - floor(f); weights (1 - f, f);
// a second tap beyond the last column is dropped (first tap x 1); rows are clamped. Horizontal pass, then vertical.
// CN = 1 (float) or 3 (Vec3f: cv_util::resizeImage of UpsampleDisparity's colour guide, CvUtil.h:139-147).
__device__ __forceinline__ void linear_area_tap(int d, int ssize, int dsize, int& s0, float& f, bool& one) {
  const double inv_scale = (double)dsize / (double)ssize, scale = 1. / inv_scale;
  int sx = (int)floor((double)d * scale);
  float fx = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
  fx = fx <= 0 ? 0.f : fx - floorf(fx);
  if (sx < 0) {
    fx = 0.f;
    sx = 0;
  }
  one = sx + 1 >= ssize;
  if (sx >= ssize - 1) {
    fx = 0.f;
    sx = ssize - 1;
  }
  s0 = sx;
  f = fx;
}
template <int CN>
__global__ void k_resize_linear_area_f32(const float* __restrict__ src, int SW, int SH, float* __restrict__ dst, int DW,
                                         int DH) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= DW || dy >= DH) {
    return;
  }
  int sx, sy;
  float fx, fy;
  bool onex, oney;
  linear_area_tap(dx, SW, DW, sx, fx, onex);
  linear_area_tap(dy, SH, DH, sy, fy, oney);
  const int y0 = min(max(sy, 0), SH - 1), y1 = min(max(sy + 1, 0), SH - 1);
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    const float* r0 = src + ((size_t)y0 * SW + sx) * CN + c;
    const float* r1 = src + ((size_t)y1 * SW + sx) * CN + c;
    const float h0 = onex ? r0[0] * 1.f : r0[0] * a0 + r0[CN] * a1;
    const float h1 = onex ? r1[0] * 1.f : r1[0] * a0 + r1[CN] * a1;
    dst[((size_t)dy * DW + dx) * CN + c] = h0 * b0 + h1 * b1;
  }
}

template <int KIND>
__global__ void k_resize_area(const void* __restrict__ src, int SW, int SH, void* __restrict__ dst, int DW, int DH,
                              AreaAxis ax, AreaAxis ay, int threshold) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= DW || dy >= DH) {
    return;
  }
  constexpr int CN = (KIND == 0 || KIND == 3) ? 3 : 1;
  float res[CN];
  if (SW == DW && SH == DH) {
    for (int c = 0; c < CN; ++c) {
      res[c] = area_src<KIND>(src, (size_t)dy * SW + dx, c);
    }
  } else if (ax.iscale > 0 && ay.iscale > 0) {
    if (ax.iscale == 2 && ay.iscale == 2 && KIND != 2 && KIND != 3) {
      for (int c = 0; c < CN; ++c) {
        const size_t p = (size_t)(2 * dy) * SW + 2 * dx;
        const int v = (int)area_src<KIND>(src, p, c) + (int)area_src<KIND>(src, p + 1, c) +
            (int)area_src<KIND>(src, p + SW, c) + (int)area_src<KIND>(src, p + SW + 1, c);
        res[c] = (float)((v + 2) >> 2);
      }
    } else {
      const float scale = 1.f / (float)(ax.iscale * ay.iscale);
      for (int c = 0; c < CN; ++c) {
        float sum = 0;
        for (int sy = 0; sy < ay.iscale; ++sy) {
          for (int sx = 0; sx < ax.iscale; ++sx) {
            sum += area_src<KIND>(src, (size_t)(dy * ay.iscale + sy) * SW + dx * ax.iscale + sx, c);
          }
        }
        res[c] = sum * scale;
      }
    }
  } else {
    float sum[CN];
    const int y0 = ay.start[dy], y1 = ay.start[dy + 1], x0 = ax.start[dx], x1 = ax.start[dx + 1];
    for (int j = y0; j < y1; ++j) {
      const float beta = ay.alpha[j];
      const size_t row = (size_t)ay.si[j] * SW;
      float buf[CN];
      for (int c = 0; c < CN; ++c) {
        buf[c] = 0.f;
      }
      for (int k = x0; k < x1; ++k) {
        const float a = ax.alpha[k];
        for (int c = 0; c < CN; ++c) {
          buf[c] += area_src<KIND>(src, row + ax.si[k], c) * a;
        }
      }
      for (int c = 0; c < CN; ++c) {
        sum[c] = (j == y0) ? beta * buf[c] : sum[c] + beta * buf[c];
      }
    }
    for (int c = 0; c < CN; ++c) {
      res[c] = sum[c];
    }
  }
  const size_t o = (size_t)dy * DW + dx;
  if (KIND == 0) {
    reinterpret_cast<ushort4*>(dst)[o] =
        make_ushort4((unsigned short)min(max(cv_round(res[0]), 0), 65535), (unsigned short)min(max(cv_round(res[1]), 0), 65535),
                     (unsigned short)min(max(cv_round(res[2]), 0), 65535), 0);
  } else if (KIND == 1) {
    const int v = min(max(cv_round(res[0]), 0), 255);
    reinterpret_cast<uint8_t*>(dst)[o] = threshold >= 0 ? (uint8_t)(v > threshold) : (uint8_t)v;
  } else if (KIND == 3) {
    float* q = reinterpret_cast<float*>(dst) + o * 3;
    for (int c = 0; c < CN; ++c) {
      q[c] = res[c];
    }
  } else {
    reinterpret_cast<float*>(dst)[o] = res[0];
  }
}

// ----------------------------------------------------------------------------------------
// GenerateForegroundMasks — source/render/BackgroundSubtractionUtil.h:20-60.
// ----------------------------------------------------------------------------------------
// cv::GaussianBlur(ksize 2r+1, sigma 0) on CV_16UC3, r in 1..3: fixed small kernels, exact integer
// products, one round-half-up at the end (OpenCV 4's ufixedpoint32 path); BORDER_REFLECT_101.
__global__ void k_gauss_u16(const ushort4* __restrict__ in, ushort4* __restrict__ out, int W, int H, int radius) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const int k3[7] = {1, 2, 1, 0, 0, 0, 0}, k5[7] = {1, 4, 6, 4, 1, 0, 0}, k7[7] = {2, 7, 14, 18, 14, 7, 2};
  const int* k = radius == 1 ? k3 : radius == 2 ? k5 : k7;
  const int sh = 2 * (radius == 1 ? 2 : radius == 2 ? 4 : 6);
  unsigned long long acc[3] = {0, 0, 0};
  for (int j = -radius; j <= radius; ++j) {
    const int yy = reflect101(y + j, H);
    unsigned long long row[3] = {0, 0, 0};
    for (int i = -radius; i <= radius; ++i) {
      const ushort4 q = in[(size_t)yy * W + reflect101(x + i, W)];
      const unsigned long long kw = (unsigned long long)k[i + radius];
      row[0] += kw * q.x;
      row[1] += kw * q.y;
      row[2] += kw * q.z;
    }
    const unsigned long long kw = (unsigned long long)k[j + radius];
    acc[0] += kw * row[0];
    acc[1] += kw * row[1];
    acc[2] += kw * row[2];
  }
  const unsigned long long half = 1ull << (sh - 1);
  out[(size_t)y * W + x] = make_ushort4((unsigned short)min((acc[0] + half) >> sh, 65535ull),
                                        (unsigned short)min((acc[1] + half) >> sh, 65535ull),
                                        (unsigned short)min((acc[2] + half) >> sh, 65535ull), 0);
}
// mask = ||template - frame||_2 > threshold on [0,1] floats; cv::norm(Vec3f) sums squares in double
__global__ void k_fg_threshold(const ushort4* __restrict__ templ, const ushort4* __restrict__ frame, size_t n,
                               float threshold, uint8_t* __restrict__ mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const float s = 1.0f / 65535.0f;
  for (; i < n; i += step) {
    const ushort4 a = templ[i], b = frame[i];
    const float d0 = fabsf(a.x * s - b.x * s), d1 = fabsf(a.y * s - b.y * s), d2 = fabsf(a.z * s - b.z * s);
    double acc = 0;
    acc += (double)d0 * d0;
    acc += (double)d1 * d1;
    acc += (double)d2 * d2;
    mask[i] = sqrt(acc) > (double)threshold;
  }
}
// one pass of cv::dilate / cv::erode with a k x k rectangle anchored at (k/2, k/2); taps outside the image ignored
__global__ void k_morph_rect(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int H, int k, int dilate) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const int a = k / 2;
  int v = dilate ? 0 : 1;
  for (int j = 0; j < k; ++j) {
    const int yy = y + j - a;
    if (yy < 0 || yy >= H) {
      continue;
    }
    for (int i = 0; i < k; ++i) {
      const int xx = x + i - a;
      if (xx < 0 || xx >= W) {
        continue;
      }
      const int t = in[(size_t)yy * W + xx];
      v = dilate ? max(v, t) : min(v, t);
    }
  }
  out[(size_t)y * W + x] = (uint8_t)v;
}

__global__ void k_bgrx_to_bgr(const ushort4* __restrict__ in, uint16_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const ushort4 q = in[i];
    out[3 * i] = q.x;
    out[3 * i + 1] = q.y;
    out[3 * i + 2] = q.z;
  }
}

// ----------------------------------------------------------------------------------------
// temporal joint bilateral — TemporalBilateralFilter.h:126-172 (quirks kept: accumulates the
// CENTRE pixel of frame t, int colour difference / 65535.f, no sumWeight == 0 guard)
// ----------------------------------------------------------------------------------------
constexpr int kMaxTemporalFrames = 31;  // frames one launch walks; longer windows run as several launches (carry)
struct TemporalFrames {
  const ushort4* guides[kMaxTemporalFrames];  // per frame: colour planes [S or 1][H*W]
  const float* images[kMaxTemporalFrames];    // per frame: disparity planes [D][H*W]
  const uint8_t* masks[kMaxTemporalFrames];   // per frame: fov & fg planes [D][H*W]
  int n;
  // the frame being filtered (TemporalBilateralFilter.h:140-147: its guide is the reference colour, its mask gates
  // the pixel, its disparity passes through where the mask is 0)
  const ushort4* refGuide;
  const float* refImage;
  const uint8_t* refMask;
  // a window longer than kMaxTemporalFrames (--time_radius has no limit in the reference,
  // TemporalBilateralFilter.cpp:55,108-109) is walked by consecutive launches over consecutive chunks of frames; the
  // two float accumulators travel through `carry` [planes][H*W], so the sums see the same additions in the same order
  float2* carry;
  int first, last;  // chunk flags: first = start the sums at 0, last = divide and write `out`
};
// blockIdx.z = destination camera d: disparity / mask plane d, colour plane dst2src[d] (plane 0 when
// dst2src is null: the single-camera entry point)
__global__ void k_temporal(TemporalFrames F, int W, int H, float sigma, int radius, float weight0,
                           float weight1, float weight2, float* __restrict__ out, const int* __restrict__ dst2src) {
  __shared__ unsigned long long expTab[32];
  {
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < 32) {
      expTab[t] = kExp2fTab[t];
    }
  }
  __syncthreads();
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const size_t n = (size_t)W * H;
  const size_t pd = (size_t)blockIdx.z * n, pg = dst2src ? (size_t)dst2src[blockIdx.z] * n : 0;
  const size_t idx = (size_t)y * W + x;
  if (!F.refMask[pd + idx]) {
    if (F.last) {
      out[pd + idx] = F.refImage[pd + idx];
    }
    return;
  }
  const ushort4 ref = F.refGuide[pg + idx];
  const float sig2 = sigma * sigma;
  const double rcpSig2 = 1.0 / (double)sig2;
  float weightedSumPix = 0.f, sumWeight = 0.f;
  if (!F.first) {
    const float2 acc = F.carry[pd + idx];
    weightedSumPix = acc.x;
    sumWeight = acc.y;
  }
  for (int t = 0; t < F.n; ++t) {
    const float centre = F.images[t][pd + idx];
    const uint8_t* __restrict__ mask = F.masks[t] + pd;
    const ushort4* __restrict__ guide = F.guides[t] + pg;
    for (int u = -radius; u <= radius; ++u) {
      const int sx = min(max(x + u, 0), W - 1);
      for (int v = -radius; v <= radius; ++v) {
        const int sy = min(max(y + v, 0), H - 1);
        const size_t j = (size_t)sy * W + sx;
        if (!mask[j]) {
          continue;
        }
        const ushort4 sc = guide[j];
        const float e0 = (float)((int)ref.x - (int)sc.x) / 65535.0f;
        const float e1 = (float)((int)ref.y - (int)sc.y) / 65535.0f;
        const float e2 = (float)((int)ref.z - (int)sc.z) / 65535.0f;
        const float weightedDiff = weight0 * (e0 * e0) + weight1 * (e1 * e1) + weight2 * (e2 * e2);
        const float weight = expf_glibc(div_by_const(-weightedDiff, rcpSig2), expTab);  // -weightedDiff / sig2
        weightedSumPix += centre * weight;
        sumWeight += weight;
      }
    }
  }
  if (F.last) {
    out[pd + idx] = weightedSumPix / sumWeight;
  } else {
    F.carry[pd + idx] = make_float2(weightedSumPix, sumWeight);
  }
}

// The same filter with the window's taps staged through LDS: a 32 x 8 block loads, per window frame, the guide texels
// and mask bytes of its tile + `radius` halo once (clamped coordinates, like the taps) and every pixel reads its
// (2 radius + 1)^2 taps from there — the direct form issued 19 global loads per pixel per frame for radius 1 and sat at
// a third of the VALU peak waiting for them. Same operations in the same order; two LDS buffers, one barrier per frame.
// Dynamic LDS: 2 x (32 + 2 radius) x (8 + 2 radius) x 9 bytes (guide 8, mask 1), rounded up to 16.
__global__ void __launch_bounds__(256)
    k_temporal_tiled(TemporalFrames F, int W, int H, float sigma, int radius, float weight0, float weight1, float weight2,
                     float* __restrict__ out, const int* __restrict__ dst2src) {
  extern __shared__ unsigned char ldsTemporal[];
  __shared__ unsigned long long expTab[32];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if (tid < 32) {
    expTab[tid] = kExp2fTab[tid];
  }
  const int TW = 32 + 2 * radius, TH = 8 + 2 * radius, cells = TW * TH;
  const size_t bufBytes = ((size_t)cells * 9 + 15) & ~(size_t)15;
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
  const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
  const bool inside = x < W && y < H;
  const size_t n = (size_t)W * H;
  const size_t pd = (size_t)blockIdx.z * n, pg = dst2src ? (size_t)dst2src[blockIdx.z] * n : 0;
  const size_t idx = inside ? (size_t)y * W + x : 0;
  const bool active = inside && F.refMask[pd + idx] != 0;
  ushort4 ref = make_ushort4(0, 0, 0, 0);
  float weightedSumPix = 0.f, sumWeight = 0.f;
  if (active) {
    ref = F.refGuide[pg + idx];
    if (!F.first) {
      const float2 acc = F.carry[pd + idx];
      weightedSumPix = acc.x;
      sumWeight = acc.y;
    }
  }
  const float sig2 = sigma * sigma;
  const double rcpSig2 = 1.0 / (double)sig2;
  for (int t = 0; t < F.n; ++t) {
    unsigned char* buf = ldsTemporal + (size_t)(t & 1) * bufBytes;
    ushort4* tg = reinterpret_cast<ushort4*>(buf);
    unsigned char* tm = buf + (size_t)cells * 8;
    const uint8_t* __restrict__ mask = F.masks[t] + pd;
    const ushort4* __restrict__ guide = F.guides[t] + pg;
    for (int k = tid; k < cells; k += 256) {
      const int ty = k / TW, tx = k - ty * TW;
      const int sx = min(max(x0 - radius + tx, 0), W - 1), sy = min(max(y0 - radius + ty, 0), H - 1);
      const size_t j = (size_t)sy * W + sx;
      const unsigned char m = mask[j];
      tm[k] = m;
      if (m) {
        tg[k] = guide[j];
      }
    }
    __syncthreads();
    if (active) {
      const float centre = F.images[t][pd + idx];
      for (int u = -radius; u <= radius; ++u) {
        for (int v = -radius; v <= radius; ++v) {
          const int k = ((int)threadIdx.y + radius + v) * TW + (int)threadIdx.x + radius + u;
          if (!tm[k]) {
            continue;
          }
          const ushort4 sc = tg[k];
          const float e0 = (float)((int)ref.x - (int)sc.x) / 65535.0f;
          const float e1 = (float)((int)ref.y - (int)sc.y) / 65535.0f;
          const float e2 = (float)((int)ref.z - (int)sc.z) / 65535.0f;
          const float weightedDiff = weight0 * (e0 * e0) + weight1 * (e1 * e1) + weight2 * (e2 * e2);
          const float weight = expf_glibc(div_by_const(-weightedDiff, rcpSig2), expTab);  // -weightedDiff / sig2
          weightedSumPix += centre * weight;
          sumWeight += weight;
        }
      }
    }
    // the buffer written next (t + 1) is the one read at t - 1: everyone passed this frame's barrier after reading it
  }
  if (!inside) {
    return;
  }
  if (!active) {
    if (F.last) {
      out[pd + idx] = F.refImage[pd + idx];
    }
    return;
  }
  if (F.last) {
    out[pd + idx] = weightedSumPix / sumWeight;
  } else {
    F.carry[pd + idx] = make_float2(weightedSumPix, sumWeight);
  }
}

// ----------------------------------------------------------------------------------------
// rephotography score — RephotographyUtil.h:38-116, ComputeRephotographyErrors.cpp:69-189.
// Camera-space stand-in for the reference's OpenGL cubemaps: pass 1 z-buffers the
// other cameras' points into the target image with a 64-bit atomicMin on (distance, camera, pixel),
// pass 2 walks each covered target ray to that distance and fetches the winner's colour.
// ----------------------------------------------------------------------------------------
__global__ void k_rephoto_splat(const Cam* __restrict__ cams, int target, const float* __restrict__ disps, int W, int H,
                                unsigned long long* __restrict__ key) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  const int j = blockIdx.z;
  if (x >= W || y >= H || j == target) {
    return;
  }
  const size_t n = (size_t)W * H, idx = (size_t)y * W + x;
  const float d = disps[(size_t)j * n + idx];
  if (!(d > 0) || isinf(d)) {
    return;
  }
  const Cam& cj = cams[j];
  const Cam& ct = cams[target];
  const D3 dir = rig_direction(cj, (x + 0.5) / (double)W, (y + 0.5) / (double)H, cj.principal[0], cj.principal[1],
                               cj.focal[0], cj.focal[1]);
  const double depth = (double)(1.0f / d);
  const D3 p = {cj.pos[0] + dir.x * depth, cj.pos[1] + dir.y * depth, cj.pos[2] + dir.z * depth};
  D2 pn;
  if (!sees(ct, p, ct.principal[0], ct.principal[1], ct.focal[0], ct.focal[1], 1.0, 1.0, pn)) {
    return;
  }
  const double px = pn.x * (double)W, py = pn.y * (double)H;
  const double dx = p.x - ct.pos[0], dy = p.y - ct.pos[1], dz = p.z - ct.pos[2];
  const float dist = (float)sqrt(sum3(dx * dx, dy * dy, dz * dz));
  const unsigned long long k =
      ((unsigned long long)__float_as_uint(dist) << 32) | ((unsigned long long)j << 24) | (unsigned long long)idx;
  const int x0 = (int)floor(px - 0.5), y0 = (int)floor(py - 0.5);
  for (int yy = y0; yy <= y0 + 1; ++yy) {
    for (int xx = x0; xx <= x0 + 1; ++xx) {
      if (xx >= 0 && yy >= 0 && xx < W && yy < H) {
        atomicMin(&key[(size_t)yy * W + xx], k);
      }
    }
  }
}

// colours: S planes of interleaved BGR u16; out: BGRA float, alpha = covered
__global__ void k_rephoto_resolve(const Cam* __restrict__ cams, int target, const uint16_t* __restrict__ colors,
                                  const unsigned long long* __restrict__ key, int W, int H, float4* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const size_t n = (size_t)W * H, idx = (size_t)y * W + x;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long k = key[idx];
  if (k != ~0ull) {
    const int j = (int)((k >> 24) & 0xff);
    const float dist = __uint_as_float((unsigned)(k >> 32));
    const Cam& ct = cams[target];
    const Cam& cj = cams[j];
    const D3 dir = rig_direction(ct, (x + 0.5) / (double)W, (y + 0.5) / (double)H, ct.principal[0], ct.principal[1],
                                 ct.focal[0], ct.focal[1]);
    const double depth = (double)dist;
    const D3 p = {ct.pos[0] + dir.x * depth, ct.pos[1] + dir.y * depth, ct.pos[2] + dir.z * depth};
    D2 pn;
    if (sees(cj, p, cj.principal[0], cj.principal[1], cj.focal[0], cj.focal[1], 1.0, 1.0, pn)) {
      // getPixelBilinear on Vec3w (CvUtil.h:78-120): clamp-to-edge taps, truncating u16 blend
      const float sx = (float)(pn.x * (double)W), sy = (float)(pn.y * (double)H);
      const float xf = roundf(sx), yf = roundf(sy);
      const int xi = (int)xf, yi = (int)yf;
      const float xw = sx - xf + 0.5f, yw = sy - yf + 0.5f;
      const float w00 = (1 - xw) * (1 - yw), w01 = xw * (1 - yw), w10 = (1 - xw) * yw, w11 = xw * yw;
      const int xa = min(max(xi - 1, 0), W - 1), xb = min(max(xi, 0), W - 1);
      const int ya = min(max(yi - 1, 0), H - 1), yb = min(max(yi, 0), H - 1);
      const uint16_t* c = colors + (size_t)j * n * 3;
      const uint16_t* p00 = c + ((size_t)ya * W + xa) * 3;
      const uint16_t* p01 = c + ((size_t)ya * W + xb) * 3;
      const uint16_t* p10 = c + ((size_t)yb * W + xa) * 3;
      const uint16_t* p11 = c + ((size_t)yb * W + xb) * 3;
      const float s = 1.0f / 65535.0f;
      o.x = bilerp_u16((float)p00[0], (float)p01[0], (float)p10[0], (float)p11[0], w00, w01, w10, w11) * s;
      o.y = bilerp_u16((float)p00[1], (float)p01[1], (float)p10[1], (float)p11[1], w00, w01, w10, w11) * s;
      o.z = bilerp_u16((float)p00[2], (float)p01[2], (float)p10[2], (float)p11[2], w00, w01, w10, w11) * s;
      o.w = 1.0f;
    }
  }
  out[idx] = o;
}

// rephoto_util::blur = cv::GaussianBlur(ksize 2r+1, sigma 1.5) on CV_32FC3, one separable pass:
// k0 * x0 + k1 * (x-1 + x1) + ... in float, BORDER_REFLECT_101. `coef` = centre .. outermost tap.
struct GaussCoef {
  float k[16];
};
__device__ __forceinline__ int reflect101_dev(int p, int n) {
  if (n == 1) {
    return 0;
  }
  while (p < 0 || p >= n) {
    p = p < 0 ? -p : 2 * (n - 1) - p;
  }
  return p;
}
__global__ void k_gauss_f32c3(const float* __restrict__ in, float* __restrict__ out, int W, int H, int radius,
                              GaussCoef coef, int vertical) {
  const int xc = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (xc >= W * 3 || y >= H) {
    return;
  }
  const int x = xc / 3, c = xc - 3 * x;
  float s = in[((size_t)y * W + x) * 3 + c] * coef.k[0];
  for (int i = 1; i <= radius; ++i) {
    float a, b;
    if (vertical) {
      a = in[((size_t)reflect101_dev(y - i, H) * W + x) * 3 + c];
      b = in[((size_t)reflect101_dev(y + i, H) * W + x) * 3 + c];
    } else {
      a = in[((size_t)y * W + reflect101_dev(x - i, W)) * 3 + c];
      b = in[((size_t)y * W + reflect101_dev(x + i, W)) * 3 + c];
    }
    s += (a + b) * coef.k[i];
  }
  out[((size_t)y * W + x) * 3 + c] = s;
}

// computeSSIM (RephotographyUtil.h:38-86), element-wise parts
__global__ void k_ssim_moments(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ muX,
                               const float* __restrict__ muY, float* __restrict__ a, float* __restrict__ b,
                               float* __restrict__ c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const float dx = x[i] - muX[i], dy = y[i] - muY[i];
    a[i] = dx * dx;
    b[i] = dy * dy;
    c[i] = dx * dy;
  }
}
__global__ void k_ssim_score(const float* __restrict__ muX, const float* __restrict__ muY,
                             const float* __restrict__ sig2X, const float* __restrict__ sig2Y,
                             const float* __restrict__ sigXY, int useLuminance, int useContrast, int useStructure,
                             float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  const float c1 = 0.0001f, c2 = 0.0009f, c3 = (float)((double)0.0009f / 2.0);
  for (; i < n; i += step) {
    const float mu2X = muX[i] * muX[i], mu2Y = muY[i] * muY[i], muXY = muX[i] * muY[i];
    const float sigX = sqrtf(sig2X[i]), sigY = sqrtf(sig2Y[i]);
    const float sxy = sigX * sigY;
    const float luminance = useLuminance ? (2 * muXY + c1) * (1.0f / (mu2X + mu2Y + c1)) : 1.0f;
    const float contrast = useContrast ? (2 * sxy + c2) * (1.0f / (sig2X[i] + sig2Y[i] + c2)) : 1.0f;
    const float structure = useStructure ? (sigXY[i] + c3) * (1.0f / (sxy + c3)) : 1.0f;
    out[i] = contrast * luminance * structure;
  }
}

// ----------------------------------------------------------------------------------------
// Rephotography renderer — CanopyScene::cubemap (source/render/CanopyScene.cpp:36-69,72-196,198-283,285-374,
// 447-476) as ComputeRephotographyErrors.cpp:77-95 drives it, without OpenGL: per camera a vertex per
// disparity pixel (camera.rig(pixel centre, 1 / disparity)), stripify()'s two triangles per pixel quad, depth
// test GL_LEQUAL, fragment = bilinear colour sample (alpha = inside the image circle, discarded at 0) weighted
// by the minor axis of the screen -> texture Jacobian and the cone; weight = exp(30 alpha) - 1, premultiplied
// accumulation over the cameras, un-premultiply, NaN -> 0; six 90-degree faces stacked top to bottom.
// Choices where OpenGL is implementation-defined: see DESIGN.md §8. fp32 like the shaders.
//   k_canopy_mesh     vertex + RGBA per (camera, pixel)
//   k_canopy_raster   one thread per triangle: 64-bit atomicMax of (1/z bits, triangle id) per covered pixel
//   k_canopy_resolve  one thread per face pixel: winner's interpolants, derivatives, weight -> accumulate
//   k_canopy_finish   un-premultiply into the stacked cubemap
// ----------------------------------------------------------------------------------------
struct CanopyTri {
  float sx[3], sy[3], invd[3], tu[3], tv[3];
  float area;
};
__constant__ int kCubeAxes[6][3][2] = {  // EXT_texture_cube_map: {major axis, sc, tc} x {axis index, sign}
    {{0, +1}, {2, -1}, {1, -1}}, {{0, -1}, {2, +1}, {1, -1}}, {{1, +1}, {0, +1}, {2, +1}},
    {{1, -1}, {0, +1}, {2, -1}}, {{2, +1}, {0, +1}, {1, -1}}, {{2, -1}, {0, -1}, {1, -1}}};

__global__ void k_canopy_mesh(const Cam* __restrict__ cams, int s, const uint16_t* __restrict__ bgr,
                              const float* __restrict__ disp, int W, int H, float4* __restrict__ vert,
                              float4* __restrict__ rgba) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) {
    return;
  }
  const Cam& c = cams[s];
  const size_t i = (size_t)y * W + x;
  const double px = (x + 0.5) / (double)W, py = (y + 0.5) / (double)H;
  const float distance = 1.0f / disp[i];  // disparityMesh, CanopyScene.cpp:453
  const D3 dir = rig_direction(c, px, py, c.principal[0], c.principal[1], c.focal[0], c.focal[1]);
  const double depth = (double)distance;
  vert[i] = make_float4((float)(c.pos[0] + dir.x * depth), (float)(c.pos[1] + dir.y * depth),
                        (float)(c.pos[2] + dir.z * depth), 0.0f);
  const float a = outside_image_circle(c, px, py, c.principal[0], c.principal[1], c.focal[0], c.focal[1]) ? 0.0f : 1.0f;
  rgba[i] = make_float4((float)bgr[3 * i] / 65535.0f, (float)bgr[3 * i + 1] / 65535.0f, (float)bgr[3 * i + 2] / 65535.0f, a);
}

__device__ __forceinline__ bool canopy_setup(const float4* __restrict__ vert, int W, int H, int qx, int qy, int t,
                                             float cxp, float cyp, float czp, int face, int E, CanopyTri& T) {
  const float scaleX = (float)(1.0 / (double)W), scaleY = (float)(1.0 / (double)H);  // Canopy::scale
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // t = 0: A B C, t = 1: B C D with A = (x, y), B = (x, y+1), C = (x+1, y), D = (x+1, y+1)
    const int ox = t == 0 ? (k == 2) : (k >= 1), oy = t == 0 ? (k == 1) : (k != 1);
    const int vx = qx + ox, vy = qy + oy;
    const float4 p = vert[(size_t)vy * W + vx];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) {
      return false;
    }
    const float q[3] = {p.x - cxp, p.y - cyp, p.z - czp};
    const float d = (float)kCubeAxes[face][0][1] * q[kCubeAxes[face][0][0]];
    const float cx = (float)kCubeAxes[face][1][1] * q[kCubeAxes[face][1][0]];
    const float cy = (float)kCubeAxes[face][2][1] * q[kCubeAxes[face][2][0]];
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
  return T.area != 0.0f && isfinite(T.area);
}
__device__ __forceinline__ bool canopy_bary(const CanopyTri& T, float px, float py, float (&l)[3], bool test) {
  const float e0 = (T.sx[2] - T.sx[1]) * (py - T.sy[1]) - (T.sy[2] - T.sy[1]) * (px - T.sx[1]);
  const float e1 = (T.sx[0] - T.sx[2]) * (py - T.sy[2]) - (T.sy[0] - T.sy[2]) * (px - T.sx[2]);
  const float e2 = (T.sx[1] - T.sx[0]) * (py - T.sy[0]) - (T.sy[1] - T.sy[0]) * (px - T.sx[0]);
  l[0] = e0 / T.area;
  l[1] = e1 / T.area;
  l[2] = e2 / T.area;
  return !test || (l[0] >= 0.0f && l[1] >= 0.0f && l[2] >= 0.0f);
}
__device__ __forceinline__ float canopy_invz(const CanopyTri& T, const float (&l)[3]) {
  return l[0] * T.invd[0] + l[1] * T.invd[1] + l[2] * T.invd[2];
}
__device__ __forceinline__ void canopy_tex(const CanopyTri& T, float px, float py, float& u, float& v) {
  float l[3];
  canopy_bary(T, px, py, l, false);
  const float iz = canopy_invz(T, l);
  u = (l[0] * (T.tu[0] * T.invd[0]) + l[1] * (T.tu[1] * T.invd[1]) + l[2] * (T.tu[2] * T.invd[2])) / iz;
  v = (l[0] * (T.tv[0] * T.invd[0]) + l[1] * (T.tv[1] * T.invd[1]) + l[2] * (T.tv[2] * T.invd[2])) / iz;
}
// mip chain of one camera's RGBA texture (glGenerateMipmap): level k at texel offset off[k], size w[k] x h[k]
constexpr int kCanopyMaxLevels = 16;
constexpr int kCanopyMaxAniso = 16;  // GL_MAX_TEXTURE_MAX_ANISOTROPY of current hardware
struct CanopyMips {
  int n;
  int w[kCanopyMaxLevels], h[kCanopyMaxLevels];
  unsigned off[kCanopyMaxLevels];
};
// level k = 2x2 box of level k - 1 (an odd last row / column repeats its edge texel)
__global__ void k_canopy_mip(const float4* __restrict__ src, int sw, int sh, float4* __restrict__ dst, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) {
    return;
  }
  const int x0 = min(2 * x, sw - 1), x1 = min(2 * x + 1, sw - 1), y0 = min(2 * y, sh - 1), y1 = min(2 * y + 1, sh - 1);
  const float4 a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1];
  const float4 e = src[(size_t)y1 * sw + x0], f = src[(size_t)y1 * sw + x1];
  dst[(size_t)y * dw + x] = make_float4(((a.x + b.x) + (e.x + f.x)) * 0.25f, ((a.y + b.y) + (e.y + f.y)) * 0.25f,
                                        ((a.z + b.z) + (e.z + f.z)) * 0.25f, ((a.w + b.w) + (e.w + f.w)) * 0.25f);
}
// GL_LINEAR inside mip level k, clamp to edge
__device__ __forceinline__ float4 canopy_bilinear(const float4* __restrict__ rgba, const CanopyMips& M, int k, float u,
                                                  float v) {
  const int W = M.w[k], H = M.h[k];
  const float4* img = rgba + M.off[k];
  const float fx = u * (float)W - 0.5f, fy = v * (float)H - 0.5f;
  const float x0f = floorf(fx), y0f = floorf(fy);
  const float ax = fx - x0f, ay = fy - y0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
  const int ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
  const float4 c00 = img[(size_t)ya * W + xa], c10 = img[(size_t)ya * W + xb];
  const float4 c01 = img[(size_t)yb * W + xa], c11 = img[(size_t)yb * W + xb];
  auto lerp2 = [&](float p00, float p10, float p01, float p11) {
    const float top = p00 * (1.0f - ax) + p10 * ax, bot = p01 * (1.0f - ax) + p11 * ax;
    return top * (1.0f - ay) + bot * ay;
  };
  return make_float4(lerp2(c00.x, c10.x, c01.x, c11.x), lerp2(c00.y, c10.y, c01.y, c11.y),
                     lerp2(c00.z, c10.z, c01.z, c11.z), lerp2(c00.w, c10.w, c01.w, c11.w));
}
// texture(sampler, texVar) as the reference asks OpenGL for it (source/gpu/GlUtil.h:316-338: mipmaps,
// GL_LINEAR_MIPMAP_LINEAR, maximum anisotropy), following EXT_texture_filter_anisotropic: N = min(ceil(Pmax / Pmin),
// 16) trilinear taps along the major axis of the pixel footprint, LOD from Pmax / N. The LOD fraction is linear in
// the footprint inside an octave (frexp, no log2), which keeps it exact in fp32. (ax, ay) = dFdx(texVar),
// (bx, by) = dFdy(texVar), normalised texture coordinates.
__device__ __forceinline__ float4 canopy_sample(const float4* __restrict__ rgba, const CanopyMips& M, float u, float v,
                                                float ax, float ay, float bx, float by) {
  const float axT = ax * (float)M.w[0], ayT = ay * (float)M.h[0], bxT = bx * (float)M.w[0], byT = by * (float)M.h[0];
  const float px2 = axT * axT + ayT * ayT, py2 = bxT * bxT + byT * byT;
  const bool xMajor = px2 >= py2;
  const float pMax = sqrtf(xMajor ? px2 : py2), pMin = sqrtf(xMajor ? py2 : px2);
  if (!(pMax > 0.0f) || !isfinite(pMax)) {
    return canopy_bilinear(rgba, M, 0, u, v);
  }
  int n = kCanopyMaxAniso;
  if (pMin > 0.0f) {
    const float r = ceilf(pMax / pMin);
    n = r < (float)kCanopyMaxAniso ? (int)r : kCanopyMaxAniso;
  }
  const float rho = pMax / (float)n;
  const int top = M.n - 1;
  int level = 0;
  float frac = 0.0f;
  if (rho > 1.0f) {
    int e;
    const float mant = frexpf(rho, &e);  // rho = mant * 2^e, mant in [0.5, 1)
    level = e - 1;
    frac = 2.0f * mant - 1.0f;
    if (level >= top) {
      level = top;
      frac = 0.0f;
    }
  }
  const float du = xMajor ? ax : bx, dv = xMajor ? ay : by;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 1; i <= n; ++i) {
    const float t = (float)i / (float)(n + 1) - 0.5f;
    const float uu = u + du * t, vv = v + dv * t;
    float4 a = canopy_bilinear(rgba, M, level, uu, vv);
    if (frac > 0.0f) {
      const float4 b = canopy_bilinear(rgba, M, level + 1, uu, vv);
      a = make_float4(a.x * (1.0f - frac) + b.x * frac, a.y * (1.0f - frac) + b.y * frac, a.z * (1.0f - frac) + b.z * frac,
                      a.w * (1.0f - frac) + b.w * frac);
    }
    sum = make_float4(sum.x + a.x, sum.y + a.y, sum.z + a.z, sum.w + a.w);
  }
  return make_float4(sum.x / (float)n, sum.y / (float)n, sum.z / (float)n, sum.w / (float)n);
}
// texVar and its fine 2x2-quad derivatives at pixel (i, j) of triangle T
__device__ __forceinline__ void canopy_grad(const CanopyTri& T, int i, int j, float& u, float& v, float& ax, float& ay,
                                            float& bx, float& by) {
  canopy_tex(T, i + 0.5f, j + 0.5f, u, v);
  const int ib = i & ~1, jb = j & ~1;
  float ua, va, ub, vb;
  canopy_tex(T, ib + 0.5f, j + 0.5f, ua, va);
  canopy_tex(T, ib + 1.5f, j + 0.5f, ub, vb);
  ax = ub - ua;  // dFdx(texVar): fine derivative inside the 2x2 quad
  ay = vb - va;
  canopy_tex(T, i + 0.5f, jb + 0.5f, ua, va);
  canopy_tex(T, i + 0.5f, jb + 1.5f, ub, vb);
  bx = ub - ua;  // dFdy(texVar)
  by = vb - va;
}

// depth + discard test of triangle T at face pixel (i, j): the fragment shader's `discard` needs the filtered alpha
__device__ __forceinline__ void canopy_fragment(const CanopyTri& T, const float4* __restrict__ rgba, const CanopyMips& M,
                                                int i, int j, int E, unsigned triId, unsigned long long* __restrict__ zbuf) {
  float l[3];
  if (!canopy_bary(T, i + 0.5f, j + 0.5f, l, true)) {
    return;
  }
  const float iz = canopy_invz(T, l);
  if (!(iz > 0.0f)) {
    return;
  }
  float u, v, ax, ay, bx, by;
  canopy_grad(T, i, j, u, v, ax, ay, bx, by);
  if (canopy_sample(rgba, M, u, v, ax, ay, bx, by).w == 0.0f) {
    return;  // discard: no colour, no depth
  }
  // nearer = larger 1/z; equal depth: the later triangle wins (GL_LEQUAL)
  atomicMax(&zbuf[(size_t)j * E + i], ((unsigned long long)__float_as_uint(iz) << 32) | triId);
}

// One thread per mesh triangle. Triangles whose bounding box holds more than kCanopyBigTri face pixels (the
// stretched triangles across depth discontinuities cover thousands) are appended to `big` and rasterised by
// k_canopy_raster_big, one workgroup per triangle, instead of serialising one thread.
constexpr int kCanopyBigTri = 64;
__global__ void k_canopy_raster(const float4* __restrict__ vert, const float4* __restrict__ rgba, CanopyMips M, int W, int H,
                                float cx, float cy, float cz, int face, int E, unsigned long long* __restrict__ zbuf,
                                unsigned* __restrict__ big, unsigned* __restrict__ nBig) {
  const int qx = blockIdx.x * blockDim.x + threadIdx.x, qy = blockIdx.y * blockDim.y + threadIdx.y;
  const int t = blockIdx.z;
  if (qx + 1 >= W || qy + 1 >= H) {
    return;
  }
  CanopyTri T;
  if (!canopy_setup(vert, W, H, qx, qy, t, cx, cy, cz, face, E, T)) {
    return;
  }
  const float minx = fminf(T.sx[0], fminf(T.sx[1], T.sx[2])), maxx = fmaxf(T.sx[0], fmaxf(T.sx[1], T.sx[2]));
  const float miny = fminf(T.sy[0], fminf(T.sy[1], T.sy[2])), maxy = fmaxf(T.sy[0], fmaxf(T.sy[1], T.sy[2]));
  if (!(maxx >= 0.0f && maxy >= 0.0f && minx <= (float)E && miny <= (float)E)) {
    return;
  }
  const int i0 = max(0, (int)ceilf(minx - 0.5f)), i1 = min(E - 1, (int)floorf(maxx - 0.5f));
  const int j0 = max(0, (int)ceilf(miny - 0.5f)), j1 = min(E - 1, (int)floorf(maxy - 0.5f));
  if (i1 < i0 || j1 < j0) {
    return;
  }
  const unsigned triId = (unsigned)(((size_t)qy * W + qx) * 2 + t);
  if ((long long)(i1 - i0 + 1) * (j1 - j0 + 1) > kCanopyBigTri) {
    big[atomicAdd(nBig, 1u)] = triId;
    return;
  }
  for (int j = j0; j <= j1; ++j) {
    for (int i = i0; i <= i1; ++i) {
      canopy_fragment(T, rgba, M, i, j, E, triId, zbuf);
    }
  }
}
__global__ void __launch_bounds__(256)
    k_canopy_raster_big(const float4* __restrict__ vert, const float4* __restrict__ rgba, CanopyMips M, int W, int H, float cx,
                        float cy, float cz, int face, int E, unsigned long long* __restrict__ zbuf,
                        const unsigned* __restrict__ big, const unsigned* __restrict__ nBig) {
  const unsigned count = *nBig;
  for (unsigned b = blockIdx.x; b < count; b += gridDim.x) {
    const unsigned triId = big[b];
    const int t = triId & 1, q = triId >> 1, qx = q % W, qy = q / W;
    CanopyTri T;
    if (!canopy_setup(vert, W, H, qx, qy, t, cx, cy, cz, face, E, T)) {
      continue;
    }
    const float minx = fminf(T.sx[0], fminf(T.sx[1], T.sx[2])), maxx = fmaxf(T.sx[0], fmaxf(T.sx[1], T.sx[2]));
    const float miny = fminf(T.sy[0], fminf(T.sy[1], T.sy[2])), maxy = fmaxf(T.sy[0], fmaxf(T.sy[1], T.sy[2]));
    const int i0 = max(0, (int)ceilf(minx - 0.5f)), i1 = min(E - 1, (int)floorf(maxx - 0.5f));
    const int j0 = max(0, (int)ceilf(miny - 0.5f)), j1 = min(E - 1, (int)floorf(maxy - 0.5f));
    const int bw = i1 - i0 + 1;
    const long long area = (long long)bw * (j1 - j0 + 1);
    for (long long p = threadIdx.x; p < area; p += blockDim.x) {
      canopy_fragment(T, rgba, M, i0 + (int)(p % bw), j0 + (int)(p / bw), E, triId, zbuf);
    }
  }
}

__global__ void k_canopy_resolve(const float4* __restrict__ vert, const float4* __restrict__ rgba, CanopyMips M, int W, int H,
                                 float cx, float cy, float cz, int face, int E,
                                 const unsigned long long* __restrict__ zbuf, float4* __restrict__ acc) {
  __shared__ unsigned long long expTab[32];
  {
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < 32) {
      expTab[t] = kExp2fTab[t];
    }
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= E || j >= E) {
    return;
  }
  const unsigned long long key = zbuf[(size_t)j * E + i];
  if (!key) {
    return;
  }
  const unsigned triId = (unsigned)key;
  const int t = triId & 1, q = triId >> 1, qx = q % W, qy = q / W;
  CanopyTri T;
  canopy_setup(vert, W, H, qx, qy, t, cx, cy, cz, face, E, T);
  float u, v, ax, ay, bx, by;
  canopy_grad(T, i, j, u, v, ax, ay, bx, by);
  const float4 c = canopy_sample(rgba, M, u, v, ax, ay, bx, by);
  const float aa = ax * ax + ay * ay, bb = bx * bx + by * by, ab = ax * bx + ay * by;
  const float hx = (aa - bb) / 2.0f;
  const float minor = (aa + bb) / 2.0f - sqrtf(hx * hx + ab * ab);
  float alpha = c.w * minor;
  const float du = u - 0.5f, dv = v - 0.5f;
  const float cone = fmaxf(1.0f / 255.0f, 1.0f - 2.0f * sqrtf(du * du + dv * dv));
  alpha *= cone;
  const float weight = expf_glibc(30.0f * alpha, expTab) - 1.0f;  // accumulateFS, kLogK = 30
  float4 a = acc[(size_t)j * E + i];
  a.x = weight * c.x + a.x;  // glBlendFuncSeparate(GL_SRC_ALPHA, GL_ONE, GL_ONE, GL_ONE)
  a.y = weight * c.y + a.y;
  a.z = weight * c.z + a.z;
  a.w = weight + a.w;
  acc[(size_t)j * E + i] = a;
}

// unpremulFS + zeroOutNans; GL row j (bottom-up) of face -> row face * E + (E - 1 - j) of the stacked cubemap
__global__ void k_canopy_finish(const float4* __restrict__ acc, int face, int E, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= E || j >= E) {
    return;
  }
  const float4 a = acc[(size_t)j * E + i];
  float4 o = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, a.w / a.w);
  o.x = o.x != o.x ? 0.0f : o.x;
  o.y = o.y != o.y ? 0.0f : o.y;
  o.z = o.z != o.z ? 0.0f : o.z;
  o.w = o.w != o.w ? 0.0f : o.w;
  out[((size_t)face * E + (E - 1 - j)) * E + i] = o;
}

}  // namespace derp
