"""ctypes binding of the C-ABI in include/derp_hip.h (libderp_hip.so, built in-tree).

Plumbing only: every computation happens in the HIP library. There is no CPU fallback —
`Derp(...)` raises when the library or a gfx950 device is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DERP_LIB") or os.path.join(_HERE, "libderp_hip.so")
CAM_TYPES = {"FTHETA": 0, "RECTILINEAR": 1, "EQUISOLID": 2, "ORTHOGRAPHIC": 3}

STAGES = ["fov_mask", "variance", "own_bias", "upsample", "proj_warp", "reproject", "proj_bias", "brute_force",
          "random_proposals", "ping_pong", "mismatches", "bilateral", "median", "mask_fov", "temporal"]


class SeqOptions(C.Structure):
    _fields_ = [
        ("time_radius", C.c_int32), ("sigma", C.c_float), ("weight_b", C.c_float), ("weight_g", C.c_float),
        ("weight_r", C.c_float), ("space_radius", C.c_int32), ("use_foreground_masks", C.c_int32),
        ("partition", C.c_int32), ("do_temporal_filter", C.c_int32), ("resident_frames", C.c_int32),
    ]


class SeqTransfer(C.Structure):
    _fields_ = [("frame", C.c_int32), ("from_rank", C.c_int32), ("to_rank", C.c_int32)]


class CameraDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("has_principal", C.c_int32), ("has_distortion", C.c_int32), ("has_fov", C.c_int32),
        ("origin", C.c_double * 3), ("forward", C.c_double * 3), ("up", C.c_double * 3), ("right", C.c_double * 3),
        ("resolution", C.c_double * 2), ("focal", C.c_double * 2), ("principal", C.c_double * 2),
        ("distortion", C.c_double * 3), ("fov", C.c_double), ("id", C.c_char * 64),
    ]


class Options(C.Structure):
    _fields_ = [
        ("min_depth_m", C.c_float), ("max_depth_m", C.c_float), ("var_noise_floor", C.c_float),
        ("var_high_thresh", C.c_float), ("random_proposals", C.c_int32), ("ping_pong_iterations", C.c_int32),
        ("mismatches_start_level", C.c_int32), ("do_bilateral_filter", C.c_int32), ("do_median_filter", C.c_int32),
        ("use_foreground_masks", C.c_int32), ("partial_coverage", C.c_int32), ("rebuild_warp_tables", C.c_int32),
    ]


# every symbol include/derp_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "derp_options_default", "derp_create", "derp_destroy", "derp_last_error", "derp_set_options", "derp_set_pyramid",
    "derp_build_pyramid_color", "derp_build_pyramid_foreground_mask", "derp_build_pyramid_background_disparity",
    "derp_download_level_color", "derp_download_level_mask", "derp_download_level_background", "derp_resize_area",
    "derp_generate_foreground_mask", "derp_image_info", "derp_image_decode", "derp_image_last_error", "derp_jpeg_encode",
    "derp_upload_color", "derp_upload_foreground_mask", "derp_upload_background_disparity", "derp_upload_disparity",
    "derp_process_level", "derp_process_pyramid", "derp_synchronize", "derp_download_disparity", "derp_download_cost",
    "derp_level_begin", "derp_stage_reproject_colors", "derp_stage_brute_force", "derp_stage_random_proposals",
    "derp_stage_ping_pong", "derp_stage_mismatches", "derp_stage_bilateral_filter", "derp_stage_median_filter", "derp_stage_mask_fov",
    "derp_level_end", "derp_set_level_disparity", "derp_get_level_disparity", "derp_cost_map", "derp_debug_download",
    "derp_debug_atan2_ypos",
    "derp_ssim", "derp_average_score", "derp_rephotograph", "derp_rephotograph_upload", "derp_rephotograph_render", "derp_canopy_cubemap",
    "derp_fov_mask", "derp_layer_disparities", "derp_download_mismatch_mask", "derp_upsample_disparity", "derp_joint_bilateral_u16", "derp_joint_bilateral_f32", "derp_masked_median",
    "derp_temporal_filter", "derp_temporal_filter_dev", "derp_dev_disparity", "derp_dev_color", "derp_dev_mask",
    "derp_get_counters", "derp_reset_counters", "derp_profile_enable", "derp_profile_reset", "derp_profile_query", "derp_profile_memoised",
    "derp_device_name", "derp_device_memory", "derp_host_nth_element_pairs", "derp_host_minstd_uniform",
    "derp_set_frame_slots", "derp_select_frame", "derp_frame_slots", "derp_host_alloc", "derp_host_free", "derp_bind_thread", "derp_host_register", "derp_host_unregister",
    "derp_seq_options_default", "derp_seq_window", "derp_seq_owner", "derp_seq_plan", "derp_seq_create", "derp_seq_destroy",
    "derp_seq_counts", "derp_seq_frames", "derp_seq_frame_slot", "derp_seq_buffer", "derp_rccl_unique_id",
    "derp_seq_attach_rccl", "derp_seq_attach_loopback", "derp_seq_attach_external", "derp_seq_selftest",
    "derp_seq_exchange_inputs", "derp_seq_level_compute", "derp_seq_level_exchange", "derp_seq_level_filter",
    "derp_seq_host_inputs", "derp_seq_buffer_copy", "derp_seq_upload_color_plane", "derp_seq_upload_disparity", "derp_seq_download_disparity", "derp_seq_exchange_inputs_level",
    "derp_seq_level_compute_frame", "derp_seq_level_provided_frame", "derp_seq_mark_exchanged", "derp_seq_level_filter_frame", "derp_seq_download_filtered",
    "derp_seq_run", "derp_seq_stats", "derp_seq_stats_reset", "derp_seq_exchange_exposed_ms",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libderp_hip.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the depth path."
            )
        # One HIP runtime per process: PyTorch-ROCm preloads its bundled libamdhip64 by path, and a
        # second copy (the system one this library would otherwise pull in) cannot open the device.
        # Importing torch first makes libderp_hip.so bind to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.derp_last_error.restype = C.c_char_p
        _lib.derp_last_error.argtypes = [C.c_void_p]
        _lib.derp_host_minstd_uniform.restype = C.c_float
        _lib.derp_host_minstd_uniform.argtypes = [C.c_int, C.c_uint64, C.c_float, C.c_float]
    return _lib


def camera_desc(cam):
    j = CameraDesc()
    j.type = CAM_TYPES[cam["type"]]
    for k in ("origin", "forward", "up", "right", "resolution", "focal"):
        for i, v in enumerate(cam[k]):
            getattr(j, k)[i] = float(v)
    j.has_principal = int("principal" in cam)
    if "principal" in cam:
        j.principal[0], j.principal[1] = map(float, cam["principal"])
    j.has_distortion = int("distortion" in cam)
    if "distortion" in cam:
        d = list(cam["distortion"]) + [0.0] * (3 - len(cam["distortion"]))
        for i in range(3):
            j.distortion[i] = float(d[i])
    j.has_fov = int("fov" in cam)
    if "fov" in cam:
        j.fov = float(cam["fov"])
    j.id = cam["id"].encode()
    return j


def filter_destinations(cams, destinations):
    """image_util::filterDestinations (source/util/ImageUtil.cpp:110-125)."""
    if not destinations:
        return list(cams)
    out = []
    for dest in destinations.split(","):
        for cam in cams:
            if cam["id"] == dest:
                out.append(cam)
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DerpError(RuntimeError):
    pass


class Derp:
    """One context per GPU (derp_create .. derp_destroy)."""

    def __init__(self, cams_src, cams_dst=None, device=0, **options):
        cams_dst = cams_src if cams_dst is None else cams_dst
        self.cams_src, self.cams_dst = list(cams_src), list(cams_dst)
        self.S, self.D = len(self.cams_src), len(self.cams_dst)
        a = (CameraDesc * self.S)(*[camera_desc(c) for c in self.cams_src])
        b = (CameraDesc * self.D)(*[camera_desc(c) for c in self.cams_dst])
        h = C.c_void_p()
        if lib().derp_create(C.byref(h), device, a, self.S, b, self.D):
            raise DerpError(lib().derp_last_error(None).decode())
        self.h = h
        self._seqs = []  # derp_seq objects living on this context (destroyed first)
        self.sizes = None
        self.opt = Options()
        lib().derp_options_default(C.byref(self.opt))
        self.set_options(**options)

    def close(self):
        if getattr(self, "h", None):
            for ref in list(getattr(self, "_seqs", [])):
                seq = ref()
                if seq is not None:
                    seq.close()
            lib().derp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise DerpError(lib().derp_last_error(self.h).decode())

    def set_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.opt, k):
                raise KeyError(k)
            setattr(self.opt, k, type(getattr(self.opt, k))(v))
        self._ck(lib().derp_set_options(self.h, C.byref(self.opt)))

    def set_pyramid(self, sizes, width_full, height_full):
        self.sizes = list(sizes)
        w = (C.c_int * len(sizes))(*[s[0] for s in sizes])
        h = (C.c_int * len(sizes))(*[s[1] for s in sizes])
        self._ck(lib().derp_set_pyramid(self.h, len(sizes), w, h, width_full, height_full))

    def _shape(self, level):
        w, h = self.sizes[level]
        return h, w

    def upload_color(self, level, s, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint16)
        assert bgr.shape == self._shape(level) + (3,)
        self._ck(lib().derp_upload_color(self.h, level, s, _p(bgr)))

    def upload_foreground_mask(self, level, s, mask):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        assert mask.shape == self._shape(level)
        self._ck(lib().derp_upload_foreground_mask(self.h, level, s, _p(mask)))

    def upload_background_disparity(self, level, d, disp):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        assert disp.shape == self._shape(level)
        self._ck(lib().derp_upload_background_disparity(self.h, level, d, _p(disp)))

    def upload_disparity(self, level, d, disp):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        assert disp.shape == self._shape(level)
        self._ck(lib().derp_upload_disparity(self.h, level, d, _p(disp)))

    # ---- pyramid builder (scripts/render/resize.py)
    def build_pyramid_color(self, s, bgr):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint16)
        self._ck(lib().derp_build_pyramid_color(self.h, s, _p(bgr), bgr.shape[1], bgr.shape[0]))

    def build_pyramid_foreground_mask(self, s, mask_u8, threshold=127):
        mask_u8 = np.ascontiguousarray(mask_u8, dtype=np.uint8)
        self._ck(lib().derp_build_pyramid_foreground_mask(self.h, s, _p(mask_u8), mask_u8.shape[1], mask_u8.shape[0],
                                                          threshold))

    def build_pyramid_background_disparity(self, d, disp):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        self._ck(lib().derp_build_pyramid_background_disparity(self.h, d, _p(disp), disp.shape[1], disp.shape[0]))

    def download_level_color(self, level, s):
        out = np.zeros(self._shape(level) + (3,), dtype=np.uint16)
        self._ck(lib().derp_download_level_color(self.h, level, s, _p(out)))
        return out

    def download_level_mask(self, level, s):
        out = np.zeros(self._shape(level), dtype=np.uint8)
        self._ck(lib().derp_download_level_mask(self.h, level, s, _p(out)))
        return out

    def download_level_background(self, level, d):
        out = np.zeros(self._shape(level), dtype=np.float32)
        self._ck(lib().derp_download_level_background(self.h, level, d, _p(out)))
        return out

    def resize_area(self, src, dw, dh):
        src = np.ascontiguousarray(src)
        kind = {(np.dtype(np.uint16), 3): 0, (np.dtype(np.uint8), 2): 1, (np.dtype(np.float32), 2): 2,
                (np.dtype(np.float32), 3): 3}[(src.dtype, src.ndim)]
        out = np.zeros((dh, dw, 3) if kind in (0, 3) else (dh, dw), dtype=src.dtype)
        self._ck(lib().derp_resize_area(self.h, kind, _p(src), src.shape[1], src.shape[0], _p(out), dw, dh))
        return out

    def generate_foreground_mask(self, template, frame, blur_radius=1, threshold=0.04, morph_closing_size=4):
        template = np.ascontiguousarray(template, dtype=np.uint16)
        frame = np.ascontiguousarray(frame, dtype=np.uint16)
        h, w = frame.shape[:2]
        out = np.zeros((h, w), dtype=np.uint8)
        self._ck(lib().derp_generate_foreground_mask(self.h, _p(template), _p(frame), w, h, blur_radius,
                                                     C.c_float(threshold), morph_closing_size, _p(out)))
        return out

    # ---- frame slots
    def set_frame_slots(self, n):
        self._ck(lib().derp_set_frame_slots(self.h, n))

    def select_frame(self, slot):
        self._ck(lib().derp_select_frame(self.h, slot))

    def frame_slots(self):
        n, cur = C.c_int(), C.c_int()
        self._ck(lib().derp_frame_slots(self.h, C.byref(n), C.byref(cur)))
        return n.value, cur.value

    def upload_frame(self, frame):
        """frame: dict from synth.make_frame (color[level][cam], optional masks / bg_disp)."""
        for level in range(len(self.sizes)):
            for s in range(self.S):
                self.upload_color(level, s, frame["color"][level][s])
                if frame.get("masks"):
                    self.upload_foreground_mask(level, s, frame["masks"][level][s])
            if frame.get("bg_disp"):
                for d in range(self.D):
                    self.upload_background_disparity(level, d, frame["bg_disp"][level][self._dst_src(d)])

    def _dst_src(self, d):
        ids = [c["id"] for c in self.cams_src]
        return ids.index(self.cams_dst[d]["id"])

    def process_level(self, level):
        self._ck(lib().derp_process_level(self.h, level))

    def process_pyramid(self, level_start=None, level_end=0):
        level_start = len(self.sizes) - 1 if level_start is None else level_start
        self._ck(lib().derp_process_pyramid(self.h, level_start, level_end))

    def synchronize(self):
        self._ck(lib().derp_synchronize(self.h))

    def download_disparity(self, level, d):
        out = np.zeros(self._shape(level), dtype=np.float32)
        self._ck(lib().derp_download_disparity(self.h, level, d, _p(out)))
        return out

    def download_cost(self, level, d):
        cost = np.zeros(self._shape(level), dtype=np.float32)
        conf = np.zeros(self._shape(level), dtype=np.float32)
        self._ck(lib().derp_download_cost(self.h, d, _p(cost), _p(conf)))
        return cost, conf

    # ---- stage-level
    def level_begin(self, level):
        self._cur = level
        self._ck(lib().derp_level_begin(self.h, level))

    def stage(self, name):
        fn = {
            "reproject_colors": "derp_stage_reproject_colors", "brute_force": "derp_stage_brute_force",
            "random_proposals": "derp_stage_random_proposals", "ping_pong": "derp_stage_ping_pong",
            "mismatches": "derp_stage_mismatches", "bilateral": "derp_stage_bilateral_filter", "median": "derp_stage_median_filter",
            "mask_fov": "derp_stage_mask_fov", "end": "derp_level_end",
        }[name]
        self._ck(getattr(lib(), fn)(self.h))

    def set_level_disparity(self, d, disp):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        self._ck(lib().derp_set_level_disparity(self.h, d, _p(disp)))

    def get_level_disparity(self, d):
        out = np.zeros(self._shape(self._cur), dtype=np.float32)
        self._ck(lib().derp_get_level_disparity(self.h, d, _p(out)))
        return out

    def cost_map(self, d, disp):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        cost = np.zeros(self._shape(self._cur), dtype=np.float32)
        conf = np.zeros(self._shape(self._cur), dtype=np.float32)
        self._ck(lib().derp_cost_map(self.h, d, _p(disp), _p(cost), _p(conf)))
        return cost, conf

    def debug_atan2_ypos(self, y, x):
        """The cost kernels' own fp64 atan2(y >= 0, x) evaluated on the device."""
        y = np.ascontiguousarray(y, np.float64)
        x = np.ascontiguousarray(x, np.float64)
        out = np.zeros_like(y)
        self._ck(lib().derp_debug_atan2_ypos(self.h, _p(y), _p(x), _p(out), C.c_size_t(y.size)))
        return out

    def debug(self, d, s, which):
        h, w = self._shape(self._cur)
        spec = {"warp": (0, (h, w, 2), np.float32), "color": (2, (h, w, 3), np.uint16),
                "bias": (3, (h, w, 3), np.uint16), "variance": (4, (h, w), np.float32),
                "fov": (5, (h, w), np.uint8)}[which]
        out = np.zeros(spec[1], dtype=spec[2])
        self._ck(lib().derp_debug_download(self.h, d, s, spec[0], _p(out)))
        return out

    def mismatch_mask(self, d):
        out = np.zeros(self._shape(self._cur), dtype=np.uint8)
        self._ck(lib().derp_download_mismatch_mask(self.h, d, _p(out)))
        return out

    # ---- sibling kernels
    def layer_disparities(self, fg, bg):
        fg = np.ascontiguousarray(fg, dtype=np.float32)
        bg = np.ascontiguousarray(bg, dtype=np.float32)
        out = np.zeros(fg.shape, dtype=np.uint8)
        self._ck(lib().derp_layer_disparities(self.h, _p(fg), _p(bg), C.c_size_t(fg.size), _p(out)))
        return out

    # ---- rephotography score (RephotographyUtil.h, ComputeRephotographyErrors.cpp)
    def ssim(self, x, y, blur_radius=1, alpha=1.0, beta=1.0, gamma=1.0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        h, w = x.shape[:2]
        out = np.zeros((h, w, 3), dtype=np.float32)
        self._ck(lib().derp_ssim(self.h, _p(x), _p(y), w, h, blur_radius, C.c_float(alpha), C.c_float(beta),
                                 C.c_float(gamma), _p(out)))
        return out

    def rephotograph(self, target, colors, disps):
        """colors[s] u16 [h, w, 3], disps[s] f32 [h, w] for every source camera -> BGRA f32 [h, w, 4]."""
        colors = [np.ascontiguousarray(c, dtype=np.uint16) for c in colors]
        disps = [np.ascontiguousarray(d, dtype=np.float32) for d in disps]
        h, w = disps[0].shape
        cp = (C.c_void_p * len(colors))(*[c.ctypes.data for c in colors])
        dp = (C.c_void_p * len(disps))(*[d.ctypes.data for d in disps])
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._ck(lib().derp_rephotograph(self.h, target, cp, dp, w, h, _p(out)))
        return out

    def rephotograph_upload(self, colors, disps):
        colors = [np.ascontiguousarray(c, dtype=np.uint16) for c in colors]
        disps = [np.ascontiguousarray(d, dtype=np.float32) for d in disps]
        h, w = disps[0].shape
        cp = (C.c_void_p * len(colors))(*[c.ctypes.data for c in colors])
        dp = (C.c_void_p * len(disps))(*[d.ctypes.data for d in disps])
        self._ck(lib().derp_rephotograph_upload(self.h, cp, dp, w, h))

    def canopy_cubemap(self, include, centre, edge):
        """CanopyScene::cubemap of the uploaded cameras with include[s] != 0, seen from `centre`
        -> BGRA f32 [6 * edge, edge, 4]."""
        inc = np.ascontiguousarray(include, dtype=np.uint8)
        ctr = np.ascontiguousarray(centre, dtype=np.float64)
        out = np.zeros((6 * edge, edge, 4), dtype=np.float32)
        self._ck(lib().derp_canopy_cubemap(self.h, _p(inc), _p(ctr), edge, _p(out)))
        return out

    def fov_mask(self, d, w, h):
        out = np.zeros((h, w), dtype=np.uint8)
        self._ck(lib().derp_fov_mask(self.h, d, w, h, _p(out)))
        return out

    def upsample_disparity(self, d, disp, w_up, h_up, bg_up=None, fg=None, fg_up=None):
        disp = np.ascontiguousarray(disp, dtype=np.float32)
        h, w = disp.shape
        use = fg is not None
        out = np.zeros((h_up, w_up), dtype=np.float32)
        if use:
            bg_up = np.ascontiguousarray(bg_up, dtype=np.float32)
            fg = np.ascontiguousarray(fg, dtype=np.uint8)
            fg_up = np.ascontiguousarray(fg_up, dtype=np.uint8)
        self._ck(lib().derp_upsample_disparity(self.h, d, _p(disp), w, h, _p(bg_up) if use else None,
                                               _p(fg) if use else None, _p(fg_up) if use else None, w_up, h_up,
                                               int(use), _p(out)))
        return out

    def joint_bilateral_u16(self, image, guide, mask, radius, sigma, w0, w1, w2):
        image = np.ascontiguousarray(image, dtype=np.float32)
        guide = np.ascontiguousarray(guide, dtype=np.uint16)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = image.shape
        out = np.zeros_like(image)
        self._ck(lib().derp_joint_bilateral_u16(self.h, _p(image), _p(guide), _p(mask), w, h, radius, C.c_float(sigma),
                                                C.c_float(w0), C.c_float(w1), C.c_float(w2), _p(out)))
        return out

    def joint_bilateral_f32(self, image, guide, mask, radius, sigma, w0, w1, w2):
        image = np.ascontiguousarray(image, dtype=np.float32)
        guide = np.ascontiguousarray(guide, dtype=np.float32)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = image.shape
        out = np.zeros_like(image)
        self._ck(lib().derp_joint_bilateral_f32(self.h, _p(image), _p(guide), _p(mask), w, h, radius, C.c_float(sigma),
                                                C.c_float(w0), C.c_float(w1), C.c_float(w2), _p(out)))
        return out

    def masked_median(self, image, background, mask, radius=1):
        image = np.ascontiguousarray(image, dtype=np.float32)
        background = None if background is None else np.ascontiguousarray(background, dtype=np.float32)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = image.shape
        out = np.zeros_like(image)
        self._ck(lib().derp_masked_median(self.h, _p(image), _p(background), _p(mask), w, h, radius, _p(out)))
        return out

    def temporal_filter(self, guides, images, masks, frame_offset, sigma, radius, w0, w1, w2):
        n = len(guides)
        guides = [np.ascontiguousarray(g, dtype=np.uint16) for g in guides]
        images = [np.ascontiguousarray(g, dtype=np.float32) for g in images]
        masks = [np.ascontiguousarray(g, dtype=np.uint8) for g in masks]
        h, w = images[0].shape
        gp = (C.c_void_p * n)(*[g.ctypes.data for g in guides])
        ip = (C.c_void_p * n)(*[g.ctypes.data for g in images])
        mp = (C.c_void_p * n)(*[g.ctypes.data for g in masks])
        out = np.zeros((h, w), dtype=np.float32)
        self._ck(lib().derp_temporal_filter(self.h, gp, ip, mp, n, w, h, frame_offset, C.c_float(sigma), radius,
                                            C.c_float(w0), C.c_float(w1), C.c_float(w2), _p(out)))
        return out

    def temporal_filter_dev(self, guide_ptrs, disp_ptrs, mask_ptrs, w, h, frame_offset, sigma, radius, w0, w1, w2,
                            out_ptr):
        n = len(guide_ptrs)
        gp = (C.c_void_p * n)(*guide_ptrs)
        ip = (C.c_void_p * n)(*disp_ptrs)
        mp = (C.c_void_p * n)(*mask_ptrs)
        self._ck(lib().derp_temporal_filter_dev(self.h, gp, ip, mp, n, w, h, frame_offset, C.c_float(sigma), radius,
                                                C.c_float(w0), C.c_float(w1), C.c_float(w2), C.c_void_p(out_ptr)))

    def dev_disparity(self, level, d):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(lib().derp_dev_disparity(self.h, level, d, C.byref(p), C.byref(n)))
        return p.value, n.value

    def dev_color(self, level, s):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(lib().derp_dev_color(self.h, level, s, C.byref(p), C.byref(n)))
        return p.value, n.value

    def dev_mask(self, level, d):
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(lib().derp_dev_mask(self.h, level, d, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- measurement
    def counters(self):
        a, b, i = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(lib().derp_get_counters(self.h, C.byref(a), C.byref(b), C.byref(i)))
        return dict(n_cost=a.value, n_pair=b.value, insufficient=i.value)

    def reset_counters(self):
        self._ck(lib().derp_reset_counters(self.h))

    def profile_enable(self, on=True):
        self._ck(lib().derp_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._ck(lib().derp_profile_reset(self.h))

    def profile_query(self, stage, level=-1):
        ms, n = C.c_double(), C.c_int()
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(lib().derp_profile_query(self.h, stage.encode(), level, C.byref(ms), C.byref(n), C.byref(a), C.byref(b)))
        return dict(ms=ms.value, launches=n.value, n_cost=a.value, n_pair=b.value)

    def profile_memoised(self, stage, level=-1):
        m = C.c_uint64()
        self._ck(lib().derp_profile_memoised(self.h, stage.encode(), level, C.byref(m)))
        return m.value

    def device_memory(self):
        """-> (free, total) bytes of HBM right now"""
        f, t = C.c_uint64(), C.c_uint64()
        self._ck(lib().derp_device_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._ck(lib().derp_device_name(self.h, buf, 256))
        return buf.value.decode()


def host_nth_element_pairs(pairs, nth):
    p = np.ascontiguousarray(pairs, dtype=np.float32).copy()
    lib().derp_host_nth_element_pairs(_p(p), len(p), nth)
    return p


def host_minstd_uniform(seed, draw_index, a, b):
    return lib().derp_host_minstd_uniform(seed, draw_index, C.c_float(a), C.c_float(b))


def average_score(score, mask):
    """rephoto_util::averageScore -> [B, G, R] means over mask != 0 and not-NaN."""
    score = np.ascontiguousarray(score, dtype=np.float32)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    out = (C.c_double * 3)()
    if lib().derp_average_score(_p(score), _p(mask), w, h, out):
        raise DerpError("derp_average_score: bad arguments")
    return [out[0], out[1], out[2]]


def format_results(avg):
    """rephoto_util::formatResults (RephotographyUtil.h:110-116)."""
    return "R %.2f%%, G %.2f%%, B %.2f%%" % (100 * avg[2], 100 * avg[1], 100 * avg[0])
