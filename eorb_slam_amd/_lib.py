"""ctypes binding of libeorb_fe.so (include/eorb_fe.h).  There is no CPU fallback: if the HIP library
is missing or fails to load, importing/using the front end raises."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EORB_FE_LIB") or os.path.join(_HERE, "csrc", "libeorb_fe.so")   # override: kernel experiments

EORB_OK, EORB_E_EMPTY, EORB_E_CONFIG, EORB_E_CAPACITY, EORB_E_ARG, EORB_E_HIP, EORB_E_NOTCONF = 0, -1, -2, -3, -4, -5, -6

# every symbol include/eorb_fe.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "eorb_create", "eorb_destroy", "eorb_sync", "eorb_debug_option", "eorb_debug_counter", "eorb_last_error", "eorb_version",
    "eorb_prof_enable", "eorb_prof_reset", "eorb_prof_only", "eorb_prof_count", "eorb_prof_get",
    "eorb_ev2im", "eorb_ev2im_gauss", "eorb_set_undistort_maps", "eorb_undistort_events", "eorb_parse_events_text", "eorb_ev2im_gauss_raw", "eorb_ev2im_raw", "eorb_fe_run_batch_raw_dev", "eorb_fe_run_batch_raw4_dev", "eorb_fe_run_batch_raw2_dev", "eorb_fe_run_batch_images_dev", "eorb_ev2mci_se3", "eorb_ev2mci_se2", "eorb_ev2mci_se3_cam", "eorb_ev2mci_se2_cam", "eorb_measure_image_focus", "eorb_measure_image_focus_n", "eorb_normalize_minmax_u8",
    "eorb_orb_configure", "eorb_orb_max_keypoints", "eorb_orb_get_tables", "eorb_orb_extract",
    "eorb_search_for_initialization", "eorb_search_by_projection_last", "eorb_search_by_projection_map", "eorb_search_by_projection_kf", "eorb_search_by_projection_last_stereo", "eorb_search_by_projection_map_stereo", "eorb_frame_stereo",
    "eorb_hamming_bf_knn2", "eorb_search_by_bow", "eorb_search_by_bow_kf", "eorb_distinctive_descriptors", "eorb_hamming_window_match", "eorb_calc_optical_flow_pyr_lk", "eorb_bow_set_vocabulary", "eorb_bow_transform", "eorb_search_for_triangulation", "eorb_kf_radius_match", "eorb_kf_radius_match_stereo", "eorb_sort_by_response", "eorb_resolve_num_mixed",
    "eorb_orb_tracked_descriptors", "eorb_orb_assign_level_by_best_desc",
    "eorb_fe_configure", "eorb_fe_run_batch_dev", "eorb_fe_last_f32_dev",
    "eorb_ev_slice_extract", "eorb_ev_slice_track", "eorb_ev_slice_image", "eorb_ev_mc_contest",
    "eorb_selfcheck_division", "eorb_selfcheck_math",
    "eorb_pack_events", "eorb_dev_alloc", "eorb_dev_free", "eorb_dev_upload", "eorb_dev_download",
]


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_float), ("nlevels", C.c_int),
                ("iniThFAST", C.c_int), ("minThFAST", C.c_int), ("edgeTh", C.c_int), ("imWidth", C.c_int)]


class Pinhole(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("model", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("k", C.c_float * 4), ("precision", C.c_float)]


def camera(cam):
    """(fx, fy, cx, cy) -> Pinhole; (fx, fy, cx, cy, k1, k2, k3, k4[, precision]) -> KannalaBrandt8 (KB8_DEF_PRECISION 1e-6)"""
    c = Camera()
    c.fx, c.fy, c.cx, c.cy = [float(v) for v in cam[:4]]
    if len(cam) > 4:
        c.model = 1
        for i in range(4):
            c.k[i] = float(cam[4 + i])
        c.precision = float(cam[8]) if len(cam) > 8 else 1e-6
    return c


class KltParams(C.Structure):
    _fields_ = [("win", C.c_int), ("maxLevel", C.c_int), ("maxCount", C.c_int), ("epsilon", C.c_double), ("minEigThreshold", C.c_float)]


class Se3Motion(C.Structure):
    _fields_ = [("angle", C.c_double), ("axis", C.c_double * 3), ("t", C.c_double * 3), ("medDepth", C.c_float)]


def se3_motion(m):
    """dict(angle, axis, t, medDepth) -> eorb_se3_motion (None -> None)"""
    if m is None:
        return None
    o = Se3Motion()
    o.angle = float(m["angle"]); o.medDepth = float(m["medDepth"])
    for i in range(3):
        o.axis[i] = float(m["axis"][i]); o.t[i] = float(m["t"][i])
    return o


class GridBounds(C.Structure):
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float),
                ("invW", C.c_float), ("invH", C.c_float)]


class FeConfig(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("sigma", C.c_float), ("pol", C.c_int),
                ("orb", OrbParams), ("lap0", C.c_int), ("lap1", C.c_int), ("want_desc", C.c_int),
                ("max_batch", C.c_int), ("max_events", C.c_int), ("match", C.c_int),
                ("windowSize", C.c_int), ("nnratio", C.c_float), ("checkOri", C.c_int)]


def build(force=False):
    """Compile libeorb_fe.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))
            if f.endswith((".hip", ".h"))] + [os.path.join(_HERE, "..", "include", "eorb_fe.h")]
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libeorb_fe.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C eorb_slam_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    pi = C.POINTER(C.c_int)
    L.eorb_create.restype = ci; L.eorb_create.argtypes = [ci, vp, C.POINTER(vp)]
    L.eorb_destroy.restype = None; L.eorb_destroy.argtypes = [vp]
    L.eorb_sync.restype = ci; L.eorb_sync.argtypes = [vp]
    L.eorb_debug_option.restype = ci; L.eorb_debug_option.argtypes = [vp, C.c_char_p, ci]
    L.eorb_debug_counter.restype = C.c_longlong; L.eorb_debug_counter.argtypes = [vp, C.c_char_p]
    L.eorb_last_error.restype = C.c_char_p; L.eorb_last_error.argtypes = [vp]
    L.eorb_version.restype = C.c_char_p; L.eorb_version.argtypes = []
    L.eorb_prof_enable.restype = ci; L.eorb_prof_enable.argtypes = [vp, ci]
    L.eorb_prof_reset.restype = ci; L.eorb_prof_reset.argtypes = [vp]
    L.eorb_prof_only.restype = ci; L.eorb_prof_only.argtypes = [vp, C.c_char_p]
    L.eorb_prof_count.restype = ci; L.eorb_prof_count.argtypes = [vp]
    L.eorb_prof_get.restype = ci
    L.eorb_prof_get.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.eorb_ev2im.restype = ci; L.eorb_ev2im.argtypes = [vp, vp, C.c_size_t, ci, ci, ci, ci, vp, vp, vp, pi]
    L.eorb_set_undistort_maps.restype = ci; L.eorb_set_undistort_maps.argtypes = [vp, vp, vp, ci, ci, ci]
    L.eorb_undistort_events.restype = ci
    L.eorb_undistort_events.argtypes = [vp, vp, C.c_size_t, ci, ci, C.c_double, vp, C.POINTER(C.c_size_t)]
    L.eorb_parse_events_text.restype = ci
    L.eorb_parse_events_text.argtypes = [vp, C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]
    L.eorb_ev2im_gauss_raw.restype = ci; L.eorb_ev2im_gauss_raw.argtypes = [vp, vp, C.c_size_t, ci, ci, cf, ci, ci, vp, vp, vp]
    L.eorb_ev2im_raw.restype = ci; L.eorb_ev2im_raw.argtypes = [vp, vp, C.c_size_t, ci, ci, ci, ci, vp, vp, vp, pi]
    L.eorb_fe_run_batch_raw_dev.restype = ci
    L.eorb_fe_run_batch_raw_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    L.eorb_fe_run_batch_raw4_dev.restype = ci
    L.eorb_fe_run_batch_raw4_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    L.eorb_fe_run_batch_raw2_dev.restype = ci
    L.eorb_fe_run_batch_raw2_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    L.eorb_fe_run_batch_images_dev.restype = ci
    L.eorb_fe_run_batch_images_dev.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp]
    L.eorb_ev2im_gauss.restype = ci; L.eorb_ev2im_gauss.argtypes = [vp, vp, C.c_size_t, ci, ci, cf, ci, ci, vp, vp, vp]
    cd = C.c_double
    L.eorb_ev2mci_se3.restype = ci
    L.eorb_ev2mci_se3.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pinhole), cd, vp, vp, cf, vp, ci, ci, cf, ci, ci, vp, vp, vp]
    L.eorb_ev2mci_se2.restype = ci
    L.eorb_ev2mci_se2.argtypes = [vp, vp, C.c_size_t, C.POINTER(Pinhole), vp, ci, ci, ci, cf, ci, ci, vp, vp, vp]
    L.eorb_ev2mci_se3_cam.restype = ci
    L.eorb_ev2mci_se3_cam.argtypes = [vp, vp, C.c_size_t, C.POINTER(Camera), cd, vp, vp, cf, vp, ci, ci, cf, ci, ci, vp, vp, vp]
    L.eorb_ev2mci_se2_cam.restype = ci
    L.eorb_ev2mci_se2_cam.argtypes = [vp, vp, C.c_size_t, C.POINTER(Camera), vp, ci, ci, ci, cf, ci, ci, vp, vp, vp]
    L.eorb_measure_image_focus.restype = ci; L.eorb_measure_image_focus.argtypes = [vp, vp, ci, ci, C.POINTER(cf)]
    L.eorb_measure_image_focus_n.restype = ci; L.eorb_measure_image_focus_n.argtypes = [vp, vp, ci, ci, ci, vp]
    L.eorb_normalize_minmax_u8.restype = ci; L.eorb_normalize_minmax_u8.argtypes = [vp, vp, ci, ci, vp]
    L.eorb_orb_configure.restype = ci; L.eorb_orb_configure.argtypes = [vp, C.POINTER(OrbParams), ci, ci]
    L.eorb_orb_max_keypoints.restype = ci; L.eorb_orb_max_keypoints.argtypes = [vp]
    L.eorb_orb_get_tables.restype = ci; L.eorb_orb_get_tables.argtypes = [vp, vp, vp, vp, pi]
    L.eorb_orb_extract.restype = ci
    L.eorb_orb_extract.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, pi, pi]
    L.eorb_search_for_initialization.restype = ci
    L.eorb_search_for_initialization.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci, vp, ci, vp,
                                                 C.POINTER(GridBounds), vp, vp, ci, cf, ci, pi]
    L.eorb_search_by_projection_last.restype = ci
    L.eorb_search_by_projection_last.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp,
                                                 C.POINTER(GridBounds), vp, cf, ci, ci, pi]
    L.eorb_search_by_projection_kf.restype = ci
    L.eorb_search_by_projection_kf.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, pi]
    L.eorb_search_by_projection_map.restype = ci
    L.eorb_search_by_projection_map.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp,
                                                C.POINTER(GridBounds), vp, cf, cf, pi]
    L.eorb_search_by_projection_last_stereo.restype = ci
    L.eorb_search_by_projection_last_stereo.argtypes = [vp, vp, ci, vp, ci, vp, vp, ci, vp, vp, vp, vp, vp, vp,
                                                        C.POINTER(GridBounds), vp, cf, ci, ci, vp, vp, pi]
    L.eorb_frame_stereo.restype = ci
    L.eorb_frame_stereo.argtypes = [vp, vp, vp, ci, ci, ci, cf, cf, vp, vp, pi, vp, vp, pi, ci, vp, vp, pi]
    L.eorb_search_by_projection_map_stereo.restype = ci
    L.eorb_search_by_projection_map_stereo.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp,
                                                       C.POINTER(GridBounds), vp, cf, cf, vp, vp, pi]
    L.eorb_orb_tracked_descriptors.restype = ci; L.eorb_orb_tracked_descriptors.argtypes = [vp, vp, ci, ci, ci, vp, ci, vp, vp]
    L.eorb_orb_assign_level_by_best_desc.restype = ci
    L.eorb_orb_assign_level_by_best_desc.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci]
    L.eorb_search_by_bow.restype = ci
    L.eorb_search_by_bow.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, cf, ci, pi]
    L.eorb_search_by_bow_kf.restype = ci
    L.eorb_search_by_bow_kf.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, cf, ci, pi]
    L.eorb_search_for_triangulation.restype = ci
    L.eorb_search_for_triangulation.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, ci,
                                                vp, vp, vp, vp, ci, ci, ci, vp, pi]
    L.eorb_kf_radius_match.restype = ci
    L.eorb_kf_radius_match.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, cf, vp, vp]
    L.eorb_kf_radius_match_stereo.restype = ci
    L.eorb_kf_radius_match_stereo.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp]
    L.eorb_calc_optical_flow_pyr_lk.restype = ci
    L.eorb_calc_optical_flow_pyr_lk.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, ci, ci, ci, ci, C.c_double, ci, cf, vp, vp]
    L.eorb_bow_set_vocabulary.restype = ci; L.eorb_bow_set_vocabulary.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp]
    L.eorb_bow_transform.restype = ci
    L.eorb_bow_transform.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, pi, vp, vp, vp, pi, vp, vp]
    L.eorb_hamming_window_match.restype = ci
    L.eorb_hamming_window_match.argtypes = [vp, vp, ci, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp]
    L.eorb_distinctive_descriptors.restype = ci; L.eorb_distinctive_descriptors.argtypes = [vp, vp, vp, ci, vp]
    L.eorb_sort_by_response.restype = ci; L.eorb_sort_by_response.argtypes = [vp, vp, ci, vp]
    L.eorb_resolve_num_mixed.restype = None; L.eorb_resolve_num_mixed.argtypes = [ci, ci, ci, ci, pi, pi]
    L.eorb_hamming_bf_knn2.restype = ci; L.eorb_hamming_bf_knn2.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    L.eorb_fe_configure.restype = ci; L.eorb_fe_configure.argtypes = [vp, C.POINTER(FeConfig)]
    L.eorb_fe_run_batch_dev.restype = ci
    L.eorb_fe_run_batch_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp]
    L.eorb_fe_last_f32_dev.restype = ci; L.eorb_fe_last_f32_dev.argtypes = [vp, C.POINTER(vp), vp, ci]
    L.eorb_ev_slice_extract.restype = ci
    L.eorb_ev_slice_extract.argtypes = [vp, vp, vp, C.c_size_t, cf, ci, ci, ci, vp, vp, vp, ci, pi, pi, vp]
    L.eorb_ev_slice_track.restype = ci
    L.eorb_ev_slice_track.argtypes = [vp, vp, vp, C.c_size_t, cf, C.POINTER(KltParams), vp, vp, vp, ci, vp]
    L.eorb_ev_slice_image.restype = ci; L.eorb_ev_slice_image.argtypes = [vp, vp]
    L.eorb_ev_mc_contest.restype = ci
    L.eorb_ev_mc_contest.argtypes = [vp, vp, C.c_size_t, C.POINTER(Camera), C.POINTER(Se3Motion), C.POINTER(Se3Motion), vp, ci, ci, ci, cf,
                                     vp, pi, vp, vp, ci, ci, vp, ci, pi]
    L.eorb_selfcheck_division.restype = ci; L.eorb_selfcheck_division.argtypes = [vp, cf, cf, cf, C.POINTER(C.c_uint64)]
    L.eorb_selfcheck_math.restype = ci; L.eorb_selfcheck_math.argtypes = [vp, ci, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.eorb_pack_events.restype = None; L.eorb_pack_events.argtypes = [vp, C.c_size_t, vp]
    L.eorb_dev_alloc.restype = vp; L.eorb_dev_alloc.argtypes = [vp, C.c_size_t]
    L.eorb_dev_free.restype = ci; L.eorb_dev_free.argtypes = [vp, vp]
    L.eorb_dev_upload.restype = ci; L.eorb_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.eorb_dev_download.restype = ci; L.eorb_dev_download.argtypes = [vp, vp, vp, C.c_size_t]
    _lib = L
    return L
