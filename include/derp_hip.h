/*
 * derp_hip.h — C-ABI of the MI355X-native depth_estimation hot path.
 *
 * The reference (facebook360_dep) has no plugin/FFI seam for this path: its boundary is the
 * DerpCLI / TemporalBilateralFilter / UpsampleDisparity process CLIs and, in-process, the free
 * functions of source/depth_estimation/Derp.h. This header is the seam a maintainer would bind
 * those mains to (INTEGRATION.md shows the binding). Every entry point cites the reference
 * function it replaces (paths relative to the reference tree).
 *
 * Conventions: POD only, plain pointers and sizes; images row-major; colour = interleaved BGR
 * uint16 (cv::Vec3w, DerpUtil.h:19); masks uint8 {0,1}; disparity float32 with NaN = invalid.
 * Host pointers unless a name says `_dev`. One context per GPU, one calling thread per context.
 * Every function returns 0 on success, non-zero on error (never throws / aborts across the ABI);
 * derp_last_error() holds the glog-style message the CLI prints before exiting 1.
 * There is NO CPU fallback: derp_create fails when no gfx950-class HIP device is present.
 */
#ifndef DERP_HIP_H
#define DERP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct derp_ctx derp_ctx;

/* Camera::Type, source/util/Camera.h:43 */
enum { DERP_FTHETA = 0, DERP_RECTILINEAR = 1, DERP_EQUISOLID = 2, DERP_ORTHOGRAPHIC = 3 };

/* One camera exactly as the rig JSON holds it (Camera::Camera(const folly::dynamic&),
 * source/util/Camera.cpp:30-75). The library performs the constructor's work itself:
 * right-handedness check + AngleAxis re-unitarisation (Camera.cpp:77-87), distortionMax
 * (Camera.cpp:119-154), cosFov (Camera.cpp:204-207), then Camera::normalize (Camera.cpp:225-229). */
typedef struct {
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
} derp_camera_desc;

/* DerpCLI flags that shape the computation (source/depth_estimation/DerpCLI.cpp:40-67);
 * derp_options_default() fills the reference defaults. */
typedef struct {
  float min_depth_m;            /* --min_depth_m            0.5   */
  float max_depth_m;            /* --max_depth_m            1e4   */
  float var_noise_floor;        /* --var_noise_floor        4e-5  */
  float var_high_thresh;        /* --var_high_thresh        1e-3  */
  int32_t random_proposals;     /* --random_proposals       2     */
  int32_t ping_pong_iterations; /* --ping_pong_iterations   1     */
  int32_t mismatches_start_level; /* --mismatches_start_level -1 (-1 = no mismatch handling) */
  int32_t do_bilateral_filter;  /* --do_bilateral_filter    1     */
  int32_t do_median_filter;     /* --do_median_filter       1     */
  int32_t use_foreground_masks; /* --use_foreground_masks   0     */
  int32_t partial_coverage;     /* --partial_coverage       0     */
  int32_t rebuild_warp_tables;  /* 1 = recompute projection warps every process_level like the
                                   reference (DerpCLI.cpp:274); 0 = cache per level across frames */
} derp_options;

void derp_options_default(derp_options* o);

/* ---- lifetime --------------------------------------------------------------------------- */
/* rigSrc / rigDst (DerpCLI.cpp:185-192). dst cameras are matched to src by id
 * (mapSrcToDstIndexes, DerpUtil.cpp:75-89). */
int derp_create(derp_ctx** out, int device, const derp_camera_desc* src, int n_src,
                const derp_camera_desc* dst, int n_dst);
void derp_destroy(derp_ctx* ctx);
const char* derp_last_error(const derp_ctx* ctx);
int derp_set_options(derp_ctx* ctx, const derp_options* o);

/* Pyramid geometry (getPyramidLevelSizes, Derp.cpp:72-99; widthFullSize/heightFullSize,
 * DerpCLI.cpp:212-214). Allocates the HBM-resident colour / disparity pyramid. */
int derp_set_pyramid(derp_ctx* ctx, int num_levels, const int* widths, const int* heights,
                     int width_full, int height_full);

/* ---- frame slots: several frames of one sequence resident on this GPU ------------------------
 * The reference keeps a sequence's frames on disk and loads one PyramidLevel at a time
 * (DerpCLI.cpp:220-323 frame loop); here each frame's colour / mask / background / result pyramid can
 * stay in HBM. derp_set_frame_slots (after derp_set_pyramid) allocates n_slots pyramids; uploads,
 * derp_process_*, derp_download_* and derp_dev_* act on the slot chosen by derp_select_frame (slot 0
 * initially). Working buffers and projection tables are shared by all slots. */
int derp_set_frame_slots(derp_ctx* ctx, int n_slots);
int derp_select_frame(derp_ctx* ctx, int slot);
int derp_frame_slots(const derp_ctx* ctx, int* n_slots, int* selected);

/* ---- host staging memory ----------------------------------------------------------------------
 * Page-locked host memory for the buffers handed to derp_upload_* / derp_download_*: the copies then run
 * at PCIe rate instead of through the runtime's pageable bounce buffers. Optional: any host pointer works.
 * derp_host_alloc returns NULL on failure (callers fall back to malloc). */
void* derp_host_alloc(size_t bytes);
void derp_host_free(void* p);
/* make the context's device current on the CALLING thread: a worker thread that allocates page-locked memory
 * (derp_host_alloc) or registers buffers for a context created on another thread calls this first */
int derp_bind_thread(derp_ctx* ctx);
/* page-lock / release memory the caller allocated itself (image buffers decoded before the runtime was up):
 * uploads from it then run at the PCIe rate. derp_host_register returns non-zero when the runtime refuses; the
 * memory stays usable, only slower. */
int derp_host_register(void* p, size_t bytes);
void derp_host_unregister(void* p);

/* ---- inputs (loadLevelImages, ImageUtil.h:79-94; DerpCLI.cpp:235-248,276-303) ------------ */
int derp_upload_color(derp_ctx* ctx, int level, int src, const uint16_t* bgr);
int derp_upload_foreground_mask(derp_ctx* ctx, int level, int src, const uint8_t* mask);
int derp_upload_background_disparity(derp_ctx* ctx, int level, int dst, const float* disp);
/* disparity of an already-finished level (resume from disk: DerpCLI.cpp:153-155,276-303) */
int derp_upload_disparity(derp_ctx* ctx, int level, int dst, const float* disp);

/* ---- pyramid builder: the step before the path (scripts/render/resize.py:51-85 resize_camera) --
 * cv2.resize(full-size frame, (w_L, h_L), INTER_AREA) into every level declared in derp_set_pyramid;
 * the frame crosses PCIe once and all its levels are produced in HBM. Masks: 8-bit image, resized,
 * then `> threshold` (resize.py passes 127; DerpCLI re-thresholds at 127, CvUtil.h:235-239). */
int derp_build_pyramid_color(derp_ctx* ctx, int src, const uint16_t* bgr, int w, int h);
int derp_build_pyramid_foreground_mask(derp_ctx* ctx, int src, const uint8_t* mask, int w, int h,
                                       int threshold);
int derp_build_pyramid_background_disparity(derp_ctx* ctx, int dst, const float* disp, int w, int h);
/* read a level back (what resize.py writes to level_<L>/<cam>/<frame>) */
int derp_download_level_color(derp_ctx* ctx, int level, int src, uint16_t* bgr);
int derp_download_level_mask(derp_ctx* ctx, int level, int src, uint8_t* mask01);
int derp_download_level_background(derp_ctx* ctx, int level, int dst, float* disp);
/* ---- raster inputs (host only; facebook360_dep_amd/csrc/derp_images.cpp over cli/image_codecs.h) ------------------
 * What `cv::imread(path, cv::IMREAD_UNCHANGED)` returns (CvUtil.cpp:23-29; scripts/render/resize.py:66-70) for the bytes
 * of a PNG / JPEG / TIFF / BMP / PNM file, the decoder chosen by signature: derp_image_info gives the geometry
 * (channels 1 / 3 / 4; bitdepth 8, 16 or 32 = float), derp_image_decode fills `out` with w * h * channels samples in
 * OpenCV's order (B, G, R [, A]) — uint16 for bitdepth 8 (widened, not scaled) and 16, float for 32. 0 / non-zero;
 * derp_image_last_error() holds the reason ("unsupported JPEG colour space (CMYK / YCCK)", ...), per thread. */
int derp_image_info(const void* bytes, size_t n, int* w, int* h, int* channels, int* bitdepth);
int derp_image_decode(const void* bytes, size_t n, void* out, size_t out_bytes);
const char* derp_image_last_error(void);
/* What cv::imwrite(".jpg", 8-bit image) writes with its defaults (scripts/render/resize.py:82-85 on a JPEG source
 * directory): baseline JPEG, 4:2:0, the standard tables, `quality` as libjpeg scales it (OpenCV's default: 95) — libjpeg's
 * compressor restated, byte-identical to libjpeg-turbo's output. pixels: w * h * channels bytes, gray or B, G, R.
 * out == NULL: only *size is set (the bytes needed); otherwise 0 and *size bytes written, non-zero if cap is too small. */
int derp_jpeg_encode(const void* pixels, int w, int h, int channels, int quality, void* out, size_t cap, size_t* size);

/* one image: kind 0 = BGR u16 x3, 1 = u8 x1, 2 = f32 x1, 3 = BGR f32 x3 (cv_util::resizeImage<Vec3f>,
 * CvUtil.h:139-147 — the colour guide of UpsampleDisparity.cpp:117). Shrinking only. */
int derp_resize_area(derp_ctx* ctx, int kind, const void* src, int w, int h, void* dst, int dw, int dh);

/* background_subtraction::generateForegroundMask<Vec3w, Vec3f> for one camera
 * (source/render/BackgroundSubtractionUtil.h:20-60; GenerateForegroundMasks.cpp:65-130): Gaussian blur
 * (radius 0..3) of template and frame, ||template - frame||_2 > threshold on [0,1] colours, k x k
 * morphological close. Both images BGR u16 at the same size; mask01 gets {0,1}. */
int derp_generate_foreground_mask(derp_ctx* ctx, const uint16_t* template_bgr, const uint16_t* frame_bgr,
                                  int w, int h, int blur_radius, float threshold,
                                  int morph_closing_size, uint8_t* mask01);

/* ---- the hot path ------------------------------------------------------------------------ */
/* One (frame, level): generateFovMasks (DerpUtil.cpp:259-276) + PyramidLevel ctor's
 * computeVariances (PyramidLevel.h:232-247) + precomputeProjections (Derp.cpp:955-976) +
 * upsampleDisparities from level+1 when level < num_levels-1 (UpsampleDisparityLib.cpp:149-182)
 * + processLevel (Derp.cpp:1005-1034) minus file output. */
int derp_process_level(derp_ctx* ctx, int level);
/* DerpCLI main's level loop for one frame (DerpCLI.cpp:220-323), level_start >= level_end. */
int derp_process_pyramid(derp_ctx* ctx, int level_start, int level_end);
int derp_synchronize(derp_ctx* ctx);

/* ---- outputs (PyramidLevel::saveResults, PyramidLevel.h:487-529) -------------------------- */
int derp_download_disparity(derp_ctx* ctx, int level, int dst, float* disparity);
/* cost / confidence of the level processed last (debug images, PyramidLevel.h:418-485) */
int derp_download_cost(derp_ctx* ctx, int dst, float* cost, float* confidence);

/* ---- stage-level entry points (Derp.h:64-133), used by the parity tests ------------------ */
int derp_level_begin(derp_ctx* ctx, int level);     /* masks, variances, warp tables, upsample hand-off */
int derp_stage_reproject_colors(derp_ctx* ctx);     /* reprojectColors,        Derp.cpp:978-1003 */
int derp_stage_brute_force(derp_ctx* ctx);          /* preprocessLevel,        Derp.cpp:826-842  */
int derp_stage_random_proposals(derp_ctx* ctx);     /* randomProposals,        Derp.cpp:844-873  */
int derp_stage_ping_pong(derp_ctx* ctx);            /* pingPongPropagation,    Derp.cpp:540-551  */
int derp_stage_mismatches(derp_ctx* ctx);           /* handleDisparityMismatches, Derp.cpp:722-748 */
int derp_stage_bilateral_filter(derp_ctx* ctx);     /* bilateralFilter,        Derp.cpp:875-902  */
int derp_stage_median_filter(derp_ctx* ctx);        /* medianFilter,           Derp.cpp:904-920  */
int derp_stage_mask_fov(derp_ctx* ctx);             /* maskFov,                Derp.cpp:940-951  */
int derp_level_end(derp_ctx* ctx);                  /* publish the level's disparity to the pyramid */
/* set / read the working disparity of the current level (between stages) */
int derp_set_level_disparity(derp_ctx* ctx, int dst, const float* disp);
int derp_get_level_disparity(derp_ctx* ctx, int dst, float* disp);
/* computeCost (Derp.cpp:104-226) of a caller-supplied disparity map at every interior pixel */
int derp_cost_map(derp_ctx* ctx, int dst, const float* disp, float* cost, float* confidence);
/* tables of the current level: which = 0 projWarp (float2), 2 projColor (u16x3), 3 projColorBias
 * (u16x3), 4 variance of src (float), 5 fov mask of dst (u8; src ignored) */
int derp_debug_download(derp_ctx* ctx, int dst, int src, int which, void* out);
/* the cost kernels' own fp64 atan2 (FTHETA branch of Camera::cameraToSensor, Camera.h:306-309): out[i] =
 * atan2(y[i], x[i]) for y >= 0, as the device routine computes it — so that a test can hold it against libm */
int derp_debug_atan2_ypos(derp_ctx* ctx, const double* y, const double* x, double* out, size_t n);

/* mismatch mask of the level processed last (dstMismatchedDisparityMask, PyramidLevel.h:332-338) */
int derp_download_mismatch_mask(derp_ctx* ctx, int dst, uint8_t* out);

/* ---- the sibling binaries' kernels ------------------------------------------------------- */
/* layerDisparities (LayerDisparities.cpp:45-55): fg over bg where fg > 0, x255, saturate to 8 bit */
int derp_layer_disparities(derp_ctx* ctx, const float* foreground, const float* background, size_t n,
                           uint8_t* out);
/* ---- rephotography score ---------------------------------------------------- */
/* rephoto_util::computeSSIM (RephotographyUtil.h:38-86) on two interleaved BGR float images in
 * [0, 1]; alpha / beta / gamma must each be 0 or 1 (computeScoreMap, :110-123: MSSIM = 1,1,1 and
 * NCC = 0,0,1). score = [h][w][3] float. */
int derp_ssim(derp_ctx* ctx, const float* x_bgr, const float* y_bgr, int w, int h, int blur_radius, float alpha,
              float beta, float gamma, float* score_bgr);
/* rephoto_util::averageScore (RephotographyUtil.h:88-108): per-channel mean of the score over
 * mask != 0 and not-NaN pixels, accumulated in double (host side, row-major order). */
int derp_average_score(const float* score_bgr, const uint8_t* mask, int w, int h, double* avg_bgr3);
/* Camera-space stand-in for ComputeRephotographyErrors.cpp:69-189 `generateCubemaps(removeOne(i))`:
 * what the other source cameras' colour + disparity say camera `target` sees (point z-buffer + one
 * bilinear fetch; no OpenGL). colors[s] = BGR u16 [h][w][3], disparities[s] = f32 [h][w] for every
 * source camera s. out = BGRA float [h][w][4], alpha = covered. */
int derp_rephotograph(derp_ctx* ctx, int target, const uint16_t* const* colors, const float* const* disparities,
                      int w, int h, float* out_bgra);
/* the same in two steps, for rendering several targets from one upload (every entry must be non-NULL) */
int derp_rephotograph_upload(derp_ctx* ctx, const uint16_t* const* colors, const float* const* disparities, int w,
                             int h);
int derp_rephotograph_render(derp_ctx* ctx, int target, float* out_bgra);
/* The reference's renderer for that score, without OpenGL: CanopyScene::cubemap
 * (source/render/CanopyScene.cpp:198-374) of the cameras include[s] != 0 of the last
 * derp_rephotograph_upload, seen from `centre` (rig space, 3 doubles) — disparity meshes, depth test, stretch /
 * cone weights, soft-max accumulation, un-premultiply — as six edge x edge faces (+X -X +Y -Y +Z -Z) stacked
 * top to bottom: out = BGRA float [6 * edge][edge][4], alpha in {0, 1}. generateCubemaps(removeOne(i)) =
 * include everything but i; the reference side = include only i (ComputeRephotographyErrors.cpp:140-145). */
int derp_canopy_cubemap(derp_ctx* ctx, const uint8_t* include, const double* centre, int edge, float* out_bgra);
/* generateFovMasks for one destination camera at an arbitrary size (DerpUtil.cpp:259-276) */
int derp_fov_mask(derp_ctx* ctx, int dst, int w, int h, uint8_t* out);
/* upsampleDisparities for one camera (UpsampleDisparityLib.cpp:98-182). fg_mask / fg_mask_up /
 * bg_disp_up may be NULL when use_foreground_masks == 0. `dst` selects the FOV mask camera. */
int derp_upsample_disparity(derp_ctx* ctx, int dst, const float* disp, int w, int h,
                            const float* bg_disp_up, const uint8_t* fg_mask,
                            const uint8_t* fg_mask_up, int w_up, int h_up,
                            int use_foreground_masks, float* out);
/* generalizedJointBilateralFilter<float, Vec3w / Vec3f> (TemporalBilateralFilter.h:39-124) */
int derp_joint_bilateral_u16(derp_ctx* ctx, const float* image, const uint16_t* guide_bgr,
                             const uint8_t* mask, int w, int h, int radius, float sigma,
                             float weight0, float weight1, float weight2, float* out);
int derp_joint_bilateral_f32(derp_ctx* ctx, const float* image, const float* guide_bgr,
                             const uint8_t* mask, int w, int h, int radius, float sigma,
                             float weight0, float weight1, float weight2, float* out);
/* cv_util::maskedMedianBlur (CvUtil.h:336-385); background may be NULL */
int derp_masked_median(derp_ctx* ctx, const float* image, const float* background,
                       const uint8_t* mask, int w, int h, int radius, float* out);
/* temporalJointBilateralFilter (TemporalBilateralFilter.h:126-215) over n frames of one camera;
 * masks[t] = fg & fov as filterFrame builds them (TemporalBilateralFilter.cpp:139-160). */
int derp_temporal_filter(derp_ctx* ctx, const uint16_t* const* guides_bgr,
                         const float* const* disparities, const uint8_t* const* masks, int n_frames,
                         int w, int h, int frame_offset, float sigma, int space_radius,
                         float weight0, float weight1, float weight2, float* out);
/* same, on disparities that already live in HBM (multi-GPU path: received over RCCL) */
int derp_temporal_filter_dev(derp_ctx* ctx, const void* const* guides_bgrx_dev,
                             const float* const* disparities_dev, const uint8_t* const* masks_dev,
                             int n_frames, int w, int h, int frame_offset, float sigma,
                             int space_radius, float weight0, float weight1, float weight2,
                             float* out_dev);
/* device views of the selected frame's resident pyramid (not owned by the caller). Pointer lifetime:
 * derp_dev_disparity / derp_dev_color stay valid until derp_set_pyramid / derp_set_frame_slots /
 * derp_destroy; their CONTENT is written by kernels on the context's stream, so call derp_synchronize
 * before reading it from another stream. derp_dev_mask computes fov & fg of every destination at
 * `level` into a buffer of its own and has finished when it returns; that buffer is reused by the next
 * derp_dev_mask call (copy it if you need two levels at once). */
int derp_dev_disparity(derp_ctx* ctx, int level, int dst, float** ptr, size_t* bytes);
int derp_dev_color(derp_ctx* ctx, int level, int src, void** ptr_bgrx_u16, size_t* bytes);
int derp_dev_mask(derp_ctx* ctx, int level, int dst, uint8_t** ptr_fov_and_fg, size_t* bytes);

/* ---- sequence driver: frames of a sequence sharded over GPUs ----------------------
 * Replaces the depth_estimation stage of scripts/render/pipeline.py:364-408 for the frames one process
 * (= one GPU) owns: per level L, coarse to fine, DerpCLI(L) on every frame -> TemporalBilateralFilter(L)
 * over the window [t - R, t + R] clamped to the sequence (TemporalBilateralFilter.cpp:96-119) ->
 * "Transfer" (the filtered level overwrites disparity_levels/level_L) -> level L-1. Frames are
 * partitioned over `world` ranks in contiguous chunks (render.py:169-175) or cyclically; the raw level
 * disparity of the frames a neighbour rank's window reaches into is the only data exchanged inside the
 * level loop. Transports: RCCL send/recv on the context's stream (derp_seq_attach_rccl), same-process
 * loopback (tests: several ranks emulated on one GPU), or external (the caller moves the buffers that
 * derp_seq_buffer names). */
typedef struct derp_seq derp_seq;
enum { DERP_SEQ_BLOCK = 0, DERP_SEQ_CYCLIC = 1 };
typedef struct {
  int32_t time_radius;          /* --time_radius  2     TemporalBilateralFilter.cpp:54 */
  float sigma;                  /* --sigma        0.01  :51 */
  float weight_b;               /* --weight_b     0.5   :57 */
  float weight_g;               /* --weight_g     1.0   :58 */
  float weight_r;               /* --weight_r     1.0   :59; never read by the reference (:176-178 passes b, g, b) */
  int32_t space_radius;         /* --space_radius -1 = max(ceil(0.9^level), 1)  :52,164-168 */
  int32_t use_foreground_masks; /* temporal masking (pipeline.py:386 do_temporal_masking) */
  int32_t partition;            /* DERP_SEQ_BLOCK | DERP_SEQ_CYCLIC */
  int32_t do_temporal_filter;   /* 0 = replicas: no window, no exchange (pipeline.py:378) */
  int32_t resident_frames;      /* 0 = every owned frame lives in HBM; N < owned frames = out of core: N frame slots,
                                 * inputs streamed per level from host memory (derp_seq_host_inputs), results kept in
                                 * page-locked host buffers; N >= 2 * time_radius + 1. The reference is independent of
                                 * the sequence length because every level round-trips through the file system
                                 * (render.py:169-175, pipeline.py:120-171) */
} derp_seq_options;
typedef struct {
  int32_t frame, from_rank, to_rank;
} derp_seq_transfer;
void derp_seq_options_default(derp_seq_options* o);
/* host-only plan (no GPU needed): window of a frame, owner of a frame, and every (frame, from, to)
 * transfer of one exchange, in the global order all ranks post them. derp_seq_plan returns the count. */
void derp_seq_window(int frame, int first, int last, int time_radius, int* lo, int* hi);
int derp_seq_owner(int first, int last, int world, int partition, int frame);
int derp_seq_plan(int first, int last, int world, int time_radius, int partition, derp_seq_transfer* out, int cap);
/* Allocates one frame slot per owned frame (derp_set_frame_slots) and the halo buffers; call after
 * derp_set_pyramid and before uploading frames. Upload frame t with derp_select_frame(ctx,
 * derp_seq_frame_slot(seq, t)) + the derp_upload_* / derp_build_pyramid_* functions. */
int derp_seq_create(derp_seq** out, derp_ctx* ctx, int first, int last, int rank, int world,
                    const derp_seq_options* opt);
void derp_seq_destroy(derp_seq* seq);
int derp_seq_counts(const derp_seq* seq, int* n_owned, int* n_halo);
int derp_seq_frames(const derp_seq* seq, int halo, int* frames, int cap);
int derp_seq_frame_slot(const derp_seq* seq, int frame);   /* -1 when not owned by this rank */
/* Inputs of one level of an owned frame from host memory — loadLevelImages (ImageUtil.h:79-94) for the frame:
 * colour [S][h*w][3] interleaved BGR u16, optional foreground masks [S][h*w] u8 and background disparity
 * [D][h*w] f32. Resident mode: uploaded into the frame's slot now. Out of core: the pointers are kept and must
 * stay valid until the sequence is destroyed (or replaced by another call). */
int derp_seq_host_inputs(derp_seq* seq, int frame, int level, const uint16_t* color_bgr, const uint8_t* fg_masks,
                         const float* background_disparity);
/* Resident mode: one camera's colour image [h*w][3] BGR u16 into the frame's slot on the library's copy stream —
 * it overlaps whatever another frame is computing (the frame itself must not be in flight). */
int derp_seq_upload_color_plane(derp_seq* seq, int frame, int level, int src, const uint16_t* bgr);
/* level disparity of an owned frame, whichever mode: upload = the previous level read back from disk when a run
 * resumes (DerpCLI.cpp:287-288); download = the level's result (filtered when the temporal filter is on) */
int derp_seq_upload_disparity(derp_seq* seq, int frame, int level, int dst, const float* disparity);
int derp_seq_download_disparity(derp_seq* seq, int frame, int level, int dst, float* disparity);  /* dst = -1: all, [D][h*w] */
/* buffer of an owned or halo frame: kind 0 colour [S][h*w] BGRX u16, 1 fg mask [S][h*w] u8,
 * 2 level disparity [D][h*w] f32 */
int derp_seq_buffer(derp_seq* seq, int frame, int level, int kind, void** ptr, size_t* bytes);
/* external transports that stage through host memory: copy such a buffer to (to_device = 0) or from (1) `host` */
int derp_seq_buffer_copy(derp_seq* seq, int frame, int level, int kind, void* host, size_t bytes, int to_device);
/* transports */
int derp_rccl_unique_id(void* out128, size_t cap);         /* ncclGetUniqueId; rank 0 calls, caller distributes */
int derp_seq_attach_rccl(derp_seq* seq, const void* unique_id, size_t bytes);
int derp_seq_attach_loopback(derp_seq* seq, derp_seq* const* peers, int n_peers);
int derp_seq_attach_external(derp_seq* seq);
int derp_seq_selftest(derp_seq* seq, int words);           /* one ring step over RCCL, verified */
/* schedule */
int derp_seq_exchange_inputs(derp_seq* seq);               /* colour (+ fg) pyramids of the halo frames, once */
int derp_seq_exchange_inputs_level(derp_seq* seq, int level);  /* ... one level of them (inputs that arrive level by level) */
int derp_seq_level_compute(derp_seq* seq, int level);      /* processLevel(level) of every owned frame */
int derp_seq_level_compute_frame(derp_seq* seq, int level, int frame);  /* ... of one owned frame (each once per level) */
/* ... or: the frame's raw level was computed by another process and uploaded with derp_seq_upload_disparity
 * (TemporalBilateralFilter's inputs: DerpCLI's files of the level, TemporalBilateralFilter.cpp:139-160) */
int derp_seq_level_provided_frame(derp_seq* seq, int level, int frame);
int derp_seq_level_exchange(derp_seq* seq, int level);     /* raw level disparity of the halo frames */
int derp_seq_mark_exchanged(derp_seq* seq, int level);     /* external transport: the halo frames' level has arrived */
int derp_seq_level_filter(derp_seq* seq, int level);       /* temporal filter of every owned frame + Transfer; fails
                                                            * unless compute (and, with halo frames, the exchange) of
                                                            * this level completed first */
/* One owned frame's filter AHEAD of derp_seq_level_filter: possible as soon as every frame of its window holds the
 * level's raw result (owned frames computed, halo frames exchanged) — for the frames in the middle of a chunk long
 * before the level's last frame is computed, so that their files can be written behind the remaining compute
 * (TemporalBilateralFilter.cpp:139-184 per frame). The result stays in the frame's scratch; the Transfer is still
 * derp_seq_level_filter's. Returns 0 = filtered, 2 = not possible yet / out of core / filter off (no error), 1 = error. */
int derp_seq_level_filter_frame(derp_seq* seq, int level, int frame);
/* ... and its download from that scratch, on the copy stream (the compute stream keeps running); dst = -1: every
 * destination's plane in one copy, [D][h*w] */
int derp_seq_download_filtered(derp_seq* seq, int frame, int level, int dst, float* disparity);
int derp_seq_run(derp_seq* seq, int level_start, int level_end);
int derp_seq_stats(derp_seq* seq, uint64_t* bytes_sent, uint64_t* bytes_received, double* exchange_ms);
int derp_seq_stats_reset(derp_seq* seq);
/* the part of derp_seq_stats' exchange_ms this rank's compute stream stood still for: a level's exchange runs on its own
 * stream beside the filter of the owned frames whose windows hold no halo frame (reset by derp_seq_stats_reset) */
int derp_seq_exchange_exposed_ms(derp_seq* seq, double* exposed_ms);

/* ---- measurement -------------------------------------------------------------------------- */
/* computeCost evaluations and (evaluation, src) pairs reaching computeSSD since the last reset:
 * the N_cost / N_pair of BASELINE.md's B_alg = 64*N_cost + 272*N_pair. */
int derp_get_counters(derp_ctx* ctx, uint64_t* n_cost, uint64_t* n_pair, uint64_t* insufficient);
int derp_reset_counters(derp_ctx* ctx);
/* per-stage HIP-event timing on the context's own stream, plus per-stage computeCost counters.
 * stage names: "fov_mask", "variance", "own_bias", "upsample", "proj_warp", "reproject",
 * "proj_bias", "brute_force", "random_proposals", "ping_pong", "mismatches", "bilateral", "median", "mask_fov", "temporal" (the
 * sequence driver's filter + Transfer), "lanes_wall" (a coarse level whose frames derp_seq_level_compute ran on overlapping
 * work lanes: the level's wall on the context's stream — the per-stage spans of such a level overlap in time).
 * level = -1 aggregates all levels. ms / launches / n_cost / n_pair may be NULL. */
int derp_profile_enable(derp_ctx* ctx, int on);
int derp_profile_reset(derp_ctx* ctx);
int derp_profile_query(derp_ctx* ctx, const char* stage, int level, double* ms, int* launches,
                       uint64_t* n_cost, uint64_t* n_pair);
/* computeCost evaluations of `stage` / `level` that were served from an earlier identical evaluation
 * instead of being recomputed (first ping-pong iteration, candidate (0,0)); they ARE included in the
 * logical n_cost / n_pair above, which count what the reference algorithm issues. */
int derp_profile_memoised(derp_ctx* ctx, const char* stage, int level, uint64_t* n_memoised);
int derp_device_name(derp_ctx* ctx, char* buf, int n);
/* free / total HBM of the context's device right now (hipMemGetInfo): what a host uses to decide how many frames of a
 * sequence to keep resident (the reference has no counterpart: its working set lives in host memory) */
int derp_device_memory(derp_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes);

/* host-only self checks (no GPU needed): restated libstdc++ algorithms the device code uses */
int derp_host_nth_element_pairs(float* pairs, int n, int nth);
float derp_host_minstd_uniform(int seed, uint64_t draw_index, float a, float b);

#ifdef __cplusplus
}
#endif
#endif /* DERP_HIP_H */
