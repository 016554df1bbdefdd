"""ctypes binding of the CPU ORACLE (oracle/eorb_oracle.h).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (eorb_slam_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

EVENT_DTYPE = np.dtype([("ts", "<f8"), ("x", "<f4"), ("y", "<f4"), ("p", "u1"), ("pad", "u1", (7,))])
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
RAW_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("p", "<u4"), ("t", "<f8")])
assert EVENT_DTYPE.itemsize == 24 and KP_DTYPE.itemsize == 28 and RAW_DTYPE.itemsize == 16


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scaleFactor", C.c_float), ("nlevels", C.c_int),
                ("iniThFAST", C.c_int), ("minThFAST", C.c_int), ("edgeTh", C.c_int), ("imWidth", C.c_int)]


class Pinhole(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("model", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("k", C.c_float * 4), ("precision", C.c_float)]


def _camera(cam):
    """(fx, fy, cx, cy) -> Pinhole; (fx, fy, cx, cy, k1, k2, k3, k4[, precision]) -> KannalaBrandt8"""
    c = Camera()
    c.fx, c.fy, c.cx, c.cy = [float(v) for v in cam[:4]]
    if len(cam) > 4:
        c.model = 1
        for i in range(4):
            c.k[i] = float(cam[4 + i])
        c.precision = float(cam[8]) if len(cam) > 8 else 1e-6
    return c


class GridBounds(C.Structure):
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("maxX", C.c_float), ("maxY", C.c_float),
                ("invW", C.c_float), ("invH", C.c_float)]


def _host_signature():
    """What -march=native means on this machine (CPU model + ISA flags)."""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        keep = [l for l in txt.splitlines() if l.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = []
    import platform
    return hashlib.sha1(("|".join(keep) + platform.machine()).encode()).hexdigest()


def build(force=False):
    """Compile the oracle: the parity build and the timing build (make is incremental).  The timing build is -march=native: when the
    library on disk was built on another machine (the .so travels with the tree to the GPU box) it is rebuilt here, so that
    bench.py's cpu_baseline is native to the host it is timed on."""
    out = os.path.join(_HERE, "_build", "liboracle.so")
    fast = os.path.join(_HERE, "_build", "liboracle_fast.so")
    stamp = os.path.join(_HERE, "_build", "fast.host")
    sig = _host_signature()
    try:
        native_here = open(stamp).read().strip() == sig
    except OSError:
        native_here = False
    if force or not native_here:
        try:
            os.remove(fast)
        except OSError:
            pass
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    if force or not native_here:
        with open(stamp, "w") as f:
            f.write(sig)
    return out


_libs = {}


def lib(fast=False):
    key = "fast" if fast else "parity"
    if key in _libs:
        return _libs[key]
    build()
    path = os.path.join(_HERE, "_build", "liboracle_fast.so" if fast else "liboracle.so")
    if os.environ.get("EORB_ORACLE_VARIANT") == "asan":       # tests/test_oracle_hygiene.py: both roles served by the sanitizer build
        subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
        path = os.path.join(_HERE, "_build", "liboracle_asan.so")
    L = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.orc_expf.restype = cf; L.orc_expf.argtypes = [cf]
    L.orc_sinf.restype = cf; L.orc_sinf.argtypes = [cf]
    L.orc_cosf.restype = cf; L.orc_cosf.argtypes = [cf]
    L.orc_fast_atan2.restype = cf; L.orc_fast_atan2.argtypes = [cf, cf]
    L.orc_cvround.restype = ci; L.orc_cvround.argtypes = [C.c_double]
    L.orc_math_hash.restype = C.c_uint64; L.orc_math_hash.argtypes = [ci, C.c_uint32, C.c_uint32]
    L.orc_atan2_hash.restype = C.c_uint64; L.orc_atan2_hash.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_tanf.restype = cf; L.orc_tanf.argtypes = [cf]
    L.orc_atanf.restype = cf; L.orc_atanf.argtypes = [cf]
    L.orc_atan2f.restype = cf; L.orc_atan2f.argtypes = [cf, cf]
    L.orc_ev2im.restype = ci
    L.orc_ev2im.argtypes = [vp, C.c_size_t, ci, ci, ci, ci, vp, vp, vp]
    L.orc_ev2im_gauss.restype = ci
    L.orc_ev2im_gauss.argtypes = [vp, C.c_size_t, ci, ci, cf, ci, ci, vp, vp, vp]
    cd = C.c_double
    L.orc_dsin.restype = cd; L.orc_dsin.argtypes = [cd]
    L.orc_dcos.restype = cd; L.orc_dcos.argtypes = [cd]
    L.orc_mci_warp_se3.restype = None
    L.orc_mci_warp_se3.argtypes = [vp, C.c_size_t, C.POINTER(Pinhole), cd, vp, vp, cf, vp, vp]
    L.orc_mci_warp_se2.restype = None; L.orc_mci_warp_se2.argtypes = [vp, C.c_size_t, C.POINTER(Pinhole), vp, ci, vp]
    L.orc_ev2mci_se3.restype = ci
    L.orc_ev2mci_se3.argtypes = [vp, C.c_size_t, C.POINTER(Pinhole), cd, vp, vp, cf, vp, ci, ci, cf, ci, ci, vp, vp, vp]
    L.orc_ev2mci_se2.restype = ci
    L.orc_ev2mci_se2.argtypes = [vp, C.c_size_t, C.POINTER(Pinhole), vp, ci, ci, ci, cf, ci, ci, vp, vp, vp]
    L.orc_ev2mci_se3_cam.restype = ci
    L.orc_ev2mci_se3_cam.argtypes = [vp, C.c_size_t, C.POINTER(Camera), cd, vp, vp, cf, vp, ci, ci, cf, ci, ci, vp, vp, vp]
    L.orc_ev2mci_se2_cam.restype = ci
    L.orc_ev2mci_se2_cam.argtypes = [vp, C.c_size_t, C.POINTER(Camera), vp, ci, ci, ci, cf, ci, ci, vp, vp, vp]
    L.orc_measure_image_focus.restype = cf; L.orc_measure_image_focus.argtypes = [vp, ci, ci]
    L.orc_cv_normalize_minmax_u8.restype = None; L.orc_cv_normalize_minmax_u8.argtypes = [vp, C.c_size_t, vp]
    L.orc_normalize_u8.restype = None
    L.orc_normalize_u8.argtypes = [vp, C.c_size_t, cf, cf, vp]
    L.orc_orb_create.restype = vp; L.orc_orb_create.argtypes = [C.POINTER(OrbParams)]
    L.orc_orb_destroy.restype = None; L.orc_orb_destroy.argtypes = [vp]
    L.orc_orb_edge_threshold.restype = ci; L.orc_orb_edge_threshold.argtypes = [vp]
    for f in ("orc_orb_scale_factors", "orc_orb_inv_scale_factors"):
        getattr(L, f).restype = C.POINTER(cf); getattr(L, f).argtypes = [vp]
    for f in ("orc_orb_features_per_level", "orc_orb_umax"):
        getattr(L, f).restype = C.POINTER(ci); getattr(L, f).argtypes = [vp]
    L.orc_orb_max_keypoints.restype = ci; L.orc_orb_max_keypoints.argtypes = [vp]
    L.orc_orb_extract.restype = ci
    L.orc_orb_extract.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci, C.POINTER(ci)]
    L.orc_orb_level_size.restype = ci; L.orc_orb_level_size.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
    L.orc_orb_level_buffer.restype = vp; L.orc_orb_level_buffer.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
    L.orc_orb_level_blur.restype = vp; L.orc_orb_level_blur.argtypes = [vp, ci]
    L.orc_orb_level_candidates.restype = ci; L.orc_orb_level_candidates.argtypes = [vp, ci, C.POINTER(vp)]
    L.orc_orb_level_keypoints.restype = ci; L.orc_orb_level_keypoints.argtypes = [vp, ci, C.POINTER(vp)]
    L.orc_resize_linear_u8.restype = None; L.orc_resize_linear_u8.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    L.orc_gaussian_blur5_u8.restype = None; L.orc_gaussian_blur5_u8.argtypes = [vp, ci, ci, ci, vp, ci]
    L.orc_gauss_kernel_q8.restype = None; L.orc_gauss_kernel_q8.argtypes = [ci, C.c_double, vp]
    L.orc_fast9_16.restype = ci; L.orc_fast9_16.argtypes = [vp, ci, ci, ci, ci, vp, ci]
    L.orc_distribute_octree.restype = ci
    L.orc_distribute_octree.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, ci]
    L.orc_ic_angle.restype = cf; L.orc_ic_angle.argtypes = [vp, ci, vp]
    L.orc_orb_descriptor.restype = ci; L.orc_orb_descriptor.argtypes = [vp, ci, ci, ci, cf, cf, cf, vp]
    L.orc_descriptor_distance.restype = ci; L.orc_descriptor_distance.argtypes = [vp, vp]
    L.orc_three_maxima.restype = None
    L.orc_three_maxima.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.orc_frame_create.restype = vp; L.orc_frame_create.argtypes = [vp, ci, vp, ci, vp, C.POINTER(GridBounds)]
    L.orc_frame_destroy.restype = None; L.orc_frame_destroy.argtypes = [vp]
    L.orc_grid_bounds_for_image.restype = None; L.orc_grid_bounds_for_image.argtypes = [ci, ci, C.POINTER(GridBounds)]
    L.orc_get_features_in_area.restype = ci; L.orc_get_features_in_area.argtypes = [vp, cf, cf, cf, ci, ci, vp, ci]
    L.orc_search_for_initialization.restype = ci
    L.orc_search_for_initialization.argtypes = [vp, vp, vp, vp, ci, cf, ci]
    L.orc_search_by_projection_last.restype = ci
    L.orc_search_by_projection_last.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, vp]
    L.orc_search_by_projection_map.restype = ci
    L.orc_search_by_projection_map.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp]
    L.orc_search_by_projection_last_stereo.restype = ci
    L.orc_search_by_projection_last_stereo.argtypes = [vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, vp, vp, vp]
    L.orc_search_by_projection_map_stereo.restype = ci
    L.orc_search_by_projection_map_stereo.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp, vp, vp]
    L.orc_orb_tracked_descriptors.restype = ci; L.orc_orb_tracked_descriptors.argtypes = [vp, vp, ci, ci, ci, vp, ci, vp, vp]
    L.orc_orb_assign_level_by_best_desc.restype = ci
    L.orc_orb_assign_level_by_best_desc.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci]
    L.orc_search_by_bow.restype = ci
    L.orc_search_by_bow.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, cf, ci]
    L.orc_search_by_bow_kf.restype = ci
    L.orc_search_by_bow_kf.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, cf, ci]
    L.orc_search_for_triangulation.restype = ci
    L.orc_search_for_triangulation.argtypes = [vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, ci,
                                               vp, vp, vp, vp, ci, ci, vp]
    L.orc_search_by_projection_kf.restype = ci
    L.orc_search_by_projection_kf.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci]
    L.orc_kf_radius_match.restype = None
    L.orc_kf_radius_match.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp]
    L.orc_kf_radius_match_stereo.restype = None
    L.orc_kf_radius_match_stereo.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_distinctive_descriptors.restype = None; L.orc_distinctive_descriptors.argtypes = [vp, vp, ci, vp]
    L.orc_sort_by_response.restype = None; L.orc_sort_by_response.argtypes = [vp, ci, vp]
    L.orc_resolve_num_mixed.restype = None
    L.orc_resolve_num_mixed.argtypes = [ci, ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.orc_bf_knn2.restype = None; L.orc_bf_knn2.argtypes = [vp, ci, vp, ci, vp, vp]
    _libs[key] = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- events -----------------------------------------------------------------------------------
def make_events(x, y, ts=None, p=None):
    n = len(x)
    ev = np.zeros(n, dtype=EVENT_DTYPE)
    ev["x"] = x; ev["y"] = y
    ev["ts"] = np.arange(n, dtype=np.float64) * 1e-6 if ts is None else ts
    ev["p"] = 1 if p is None else p
    return ev


def ev2im(ev, W, H, pol=False, normalized=True, fast=False):
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    f32 = np.empty((H, W), np.float32); u8 = np.zeros((H, W), np.uint8); mm = np.zeros(2, np.float32)
    is_u8 = lib(fast).orc_ev2im(_p(ev), len(ev), W, H, int(pol), int(normalized), _p(f32), _p(u8), _p(mm))
    return f32, (u8 if is_u8 else None), mm


def ev2im_gauss(ev, W, H, sigma=1.0, pol=False, normalized=True, fast=False):
    ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
    f32 = np.empty((H, W), np.float32); u8 = np.zeros((H, W), np.uint8); mm = np.zeros(2, np.float32)
    is_u8 = lib(fast).orc_ev2im_gauss(_p(ev), len(ev), W, H, float(sigma), int(pol), int(normalized),
                                      _p(f32), _p(u8), _p(mm))
    return f32, (u8 if is_u8 else None), mm


# ---- extractor ----------------------------------------------------------------------------------
class OrbExtractor:
    """Oracle mirror of ORB_SLAM3::ORBextractor (src/ORBextractor.cc)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0,
                 edgeTh=19, imWidth=240, fast=False):
        self.L = lib(fast)
        self.params = OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, edgeTh, imWidth)
        self.h = self.L.orc_orb_create(C.byref(self.params))
        if not self.h:
            raise ValueError("bad ORB params")
        self.nlevels = nlevels
        self.cap = self.L.orc_orb_max_keypoints(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_orb_destroy(self.h); self.h = None

    @property
    def edge(self):
        return self.L.orc_orb_edge_threshold(self.h)

    @property
    def scale_factors(self):
        return np.array(self.L.orc_orb_scale_factors(self.h)[:self.nlevels], np.float32)

    @property
    def inv_scale_factors(self):
        return np.array(self.L.orc_orb_inv_scale_factors(self.h)[:self.nlevels], np.float32)

    @property
    def features_per_level(self):
        return list(self.L.orc_orb_features_per_level(self.h)[:self.nlevels])

    @property
    def umax(self):
        return list(self.L.orc_orb_umax(self.h)[:16])

    def extract(self, img, lap=(0, 1000), want_desc=True):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        oob = np.zeros(self.cap, np.uint8); n = C.c_int(0)
        mono = self.L.orc_orb_extract(self.h, _p(img), W, H, W, lap[0], lap[1], int(want_desc),
                                      _p(kps), _p(desc), _p(oob), self.cap, C.byref(n))
        if mono < 0:
            return mono, None, None, None
        return mono, kps[:n.value].copy(), (desc[:n.value].copy() if want_desc else None), oob[:n.value].copy()

    def tracked_descriptors(self, img, kps):
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape
        kps = np.ascontiguousarray(kps, KP_DTYPE); n = len(kps)
        desc = np.zeros((n, 32), np.uint8); oob = np.zeros(n, np.uint8)
        rc = self.L.orc_orb_tracked_descriptors(self.h, _p(img), W, H, W, _p(kps), n, _p(desc), _p(oob))
        assert rc == 0, rc
        return desc, oob

    def assign_level_by_best_desc(self, img, ref_desc, kps):
        img = np.ascontiguousarray(img, np.uint8); H, W = img.shape
        kps = np.ascontiguousarray(kps, KP_DTYPE).copy(); ref = np.ascontiguousarray(ref_desc, np.uint8)
        rc = self.L.orc_orb_assign_level_by_best_desc(self.h, _p(img), W, H, W, _p(ref), _p(kps), len(kps))
        assert rc == 0, rc
        return kps

    def compute_stereo_matches(self, right, kL, dL, kR, dR, mb, mbf):
        """Frame::ComputeStereoMatches (src/Frame.cc:869-1048): self / right = the extractors of the left / right image (after their
        extract() calls); returns (mvuRight, mvDepth, matches before the median cut)."""
        kL = np.ascontiguousarray(kL, KP_DTYPE); kR = np.ascontiguousarray(kR, KP_DTYPE)
        dL = np.ascontiguousarray(dL, np.uint8); dR = np.ascontiguousarray(dR, np.uint8)
        ur = np.zeros(len(kL), np.float32); dp = np.zeros(len(kL), np.float32)
        self.L.orc_compute_stereo_matches.restype = C.c_int
        self.L.orc_compute_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        n = self.L.orc_compute_stereo_matches(self.h, right.h, _p(kL), len(kL), _p(dL), _p(kR), len(kR), _p(dR), mb, mbf, _p(ur), _p(dp))
        return ur, dp, n

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_size(self.h, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_buffer(self, l):
        bw, bh = C.c_int(), C.c_int()
        ptr = self.L.orc_orb_level_buffer(self.h, l, C.byref(bw), C.byref(bh))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (bh.value, bw.value)).copy()

    def level_blur(self, l):
        w, h = self.level_size(l)
        ptr = self.L.orc_orb_level_blur(self.h, l)
        if not ptr:
            return None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (h, w)).copy()

    def _kparr(self, fn, l):
        ptr = C.c_void_p()
        n = fn(self.h, l, C.byref(ptr))
        if n == 0:
            return np.zeros(0, KP_DTYPE)
        buf = (C.c_char * (n * 28)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=KP_DTYPE, count=n).copy()

    def level_candidates(self, l):
        return self._kparr(self.L.orc_orb_level_candidates, l)

    def level_keypoints(self, l):
        return self._kparr(self.L.orc_orb_level_keypoints, l)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8); sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    return dst


def gaussian_blur5(src):
    src = np.ascontiguousarray(src, np.uint8); h, w = src.shape
    dst = np.empty((h, w), np.uint8)
    lib().orc_gaussian_blur5_u8(_p(src), w, h, w, _p(dst), w)
    return dst


def fast9_16(img, threshold):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    cap = w * h
    out = np.zeros((cap, 3), np.int32)
    n = lib().orc_fast9_16(_p(img), w, h, w, threshold, _p(out), cap)
    return out[:n].copy()


def distribute_octree(cands, minX, maxX, minY, maxY, N):
    cands = np.ascontiguousarray(cands, KP_DTYPE)
    cap = max(N + 64, 64)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().orc_distribute_octree(_p(cands), len(cands), minX, maxX, minY, maxY, N, _p(out), cap)
    assert 0 <= n <= cap, n
    return out[:n].copy()


# ---- matchers ---------------------------------------------------------------------------------------
def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def three_maxima(sizes):
    s = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orc_three_maxima(_p(s), len(s), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def grid_bounds(W, H):
    gb = GridBounds()
    lib().orc_grid_bounds_for_image(W, H, C.byref(gb))
    return gb


class Frame:
    """Oracle mirror of the Frame grid + descriptors (src/Frame.cc:431-460,710-793)."""

    def __init__(self, kps, desc, W, H, is_orb=None, fast=False):
        self.L = lib(fast)
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        self.is_orb = None if is_orb is None else np.ascontiguousarray(is_orb, np.uint8)
        self.gb = grid_bounds(W, H)
        self.N = len(self.kps)
        self.h = self.L.orc_frame_create(_p(self.kps), self.N, _p(self.desc), self.desc.shape[1],
                                         _p(self.is_orb), C.byref(self.gb))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_frame_destroy(self.h); self.h = None

    def features_in_area(self, x, y, r, minLevel=-1, maxLevel=-1):
        out = np.zeros(max(self.N, 1), np.int32)
        n = self.L.orc_get_features_in_area(self.h, x, y, r, minLevel, maxLevel, _p(out), self.N)
        return out[:n].copy()


def search_for_initialization(F1, F2, prev_matched, windowSize=100, nnratio=0.9, checkOri=True):
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.full(F1.N, -1, np.int32)
    n = F1.L.orc_search_for_initialization(F1.h, F2.h, _p(pm), _p(m12), windowSize, nnratio, int(checkOri))
    return n, m12, pm


def search_by_projection_last(cur, last, valid, uv, mp_desc, mp_obs, cur_mp, th, level_scale,
                              mode=0, checkOri=True, uright=None, proj_ur=None):
    valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.uint8)
    cm = np.ascontiguousarray(cur_mp, np.int32).copy()
    ls = np.ascontiguousarray(level_scale, np.float32)
    if uright is not None:
        ur = np.ascontiguousarray(uright, np.float32); pu = np.ascontiguousarray(proj_ur, np.float32)
        n = cur.L.orc_search_by_projection_last_stereo(cur.h, last.h, _p(valid), _p(uv), _p(mp_desc), _p(mp_obs),
                                                       _p(cm), th, mode, int(checkOri), _p(ls), _p(ur), _p(pu))
        return n, cm
    n = cur.L.orc_search_by_projection_last(cur.h, last.h, _p(valid), _p(uv), _p(mp_desc), _p(mp_obs),
                                            _p(cm), th, mode, int(checkOri), _p(ls))
    return n, cm


def search_by_projection_map(F, in_view, proj_xy, level, view_cos, mp_desc, mp_obs, frame_mp, th,
                             nnratio, level_scale, mp_is_orb=None, uright=None, proj_xr=None):
    in_view = np.ascontiguousarray(in_view, np.uint8); proj_xy = np.ascontiguousarray(proj_xy, np.float32)
    level = np.ascontiguousarray(level, np.int32); view_cos = np.ascontiguousarray(view_cos, np.float32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.uint8)
    level_scale = np.ascontiguousarray(level_scale, np.float32)
    mio = None if mp_is_orb is None else np.ascontiguousarray(mp_is_orb, np.uint8)
    fm = np.ascontiguousarray(frame_mp, np.int32).copy()
    if uright is not None:
        ur = np.ascontiguousarray(uright, np.float32); px = np.ascontiguousarray(proj_xr, np.float32)
        n = F.L.orc_search_by_projection_map_stereo(F.h, len(in_view), _p(in_view), _p(proj_xy), _p(level),
                                                    _p(view_cos), _p(mp_desc), _p(mp_obs), _p(mio), _p(fm),
                                                    th, nnratio, _p(level_scale), _p(ur), _p(px))
        return n, fm
    n = F.L.orc_search_by_projection_map(F.h, len(in_view), _p(in_view), _p(proj_xy), _p(level),
                                         _p(view_cos), _p(mp_desc), _p(mp_obs), _p(mio), _p(fm),
                                         th, nnratio, _p(level_scale))
    return n, fm


def bf_knn2(q, t, fast=False):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
    lib(fast).orc_bf_knn2(_p(q), len(q), _p(t), len(t), _p(idx), _p(dist))
    return idx, dist


def search_by_bow(kf_kps, kf_desc, kf_has_mp, kf_fv, f_kps, f_desc, f_fv, nnratio=0.7, checkOri=True):
    """kf_fv / f_fv: (nodes uint32[nn] ascending, node_off int32[nn+1], idx int32[])."""
    kf_kps = np.ascontiguousarray(kf_kps, KP_DTYPE); f_kps = np.ascontiguousarray(f_kps, KP_DTYPE)
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8); f_desc = np.ascontiguousarray(f_desc, np.uint8)
    hm = np.ascontiguousarray(kf_has_mp, np.uint8)
    kn, ko, ki = [np.ascontiguousarray(a, t) for a, t in zip(kf_fv, (np.uint32, np.int32, np.int32))]
    fn, fo, fi = [np.ascontiguousarray(a, t) for a, t in zip(f_fv, (np.uint32, np.int32, np.int32))]
    m = np.full(len(f_kps), -1, np.int32)
    n = lib().orc_search_by_bow(_p(kf_kps), len(kf_kps), _p(kf_desc), _p(hm), _p(kn), _p(ko), _p(ki), len(kn),
                                _p(f_kps), len(f_kps), _p(f_desc), _p(fn), _p(fo), _p(fi), len(fn), _p(m), nnratio, int(checkOri))
    return n, m


def sort_by_response(kps):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    perm = np.zeros(len(kps), np.int32)
    lib().orc_sort_by_response(_p(kps), len(kps), _p(perm))
    return perm


def resolve_num_mixed(nDetORB, nDetAK, nDesired, nDesiredAK):
    a, b = C.c_int(-1), C.c_int(-1)
    lib().orc_resolve_num_mixed(nDetORB, nDetAK, nDesired, nDesiredAK, C.byref(a), C.byref(b))
    return a.value, b.value


# ---- motion-compensated accumulation (f1) --------------------------------------------------------------------------------
def ev2mci_se3(ev, cam, angle, axis, tt, medDepth, W, H, sigma=1.0, pol=False, normalized=False, depth=None):
    ev = np.ascontiguousarray(ev, EVENT_DTYPE)
    ax = np.ascontiguousarray(axis, np.float64); t = np.ascontiguousarray(tt, np.float64)
    dp = None if depth is None else np.ascontiguousarray(depth, np.float32)
    f32 = np.empty((H, W), np.float32); u8 = np.zeros((H, W), np.uint8); mm = np.zeros(2, np.float32)
    pc = _camera(cam)
    r = lib().orc_ev2mci_se3_cam(_p(ev), len(ev), C.byref(pc), float(angle), _p(ax), _p(t), float(medDepth), _p(dp), W, H,
                             float(sigma), int(pol), int(normalized), _p(f32), _p(u8), _p(mm))
    return f32, (u8 if r else None), mm


def ev2mci_se2(ev, cam, params2D, W, H, sigma=1.0, pol=False, normalized=False):
    ev = np.ascontiguousarray(ev, EVENT_DTYPE)
    pr = np.ascontiguousarray(params2D, np.float32)
    f32 = np.empty((H, W), np.float32); u8 = np.zeros((H, W), np.uint8); mm = np.zeros(2, np.float32)
    pc = _camera(cam)
    r = lib().orc_ev2mci_se2_cam(_p(ev), len(ev), C.byref(pc), _p(pr), len(pr), W, H, float(sigma), int(pol), int(normalized),
                             _p(f32), _p(u8), _p(mm))
    return f32, (u8 if r else None), mm


def measure_image_focus(img):
    img = np.ascontiguousarray(img, np.float32); H, W = img.shape
    return lib().orc_measure_image_focus(_p(img), W, H)


def cv_normalize_minmax_u8(img):
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros(img.shape, np.uint8)
    lib().orc_cv_normalize_minmax_u8(_p(img), img.size, _p(out))
    return out


def search_by_bow_kf(kps1, desc1, has_mp1, fv1, kps2, desc2, has_mp2, fv2, nnratio=0.8, checkOri=True):
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE); kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    n1, o1, i1 = [np.ascontiguousarray(a, t) for a, t in zip(fv1, (np.uint32, np.int32, np.int32))]
    n2, o2, i2 = [np.ascontiguousarray(a, t) for a, t in zip(fv2, (np.uint32, np.int32, np.int32))]
    m = np.full(len(kps1), -1, np.int32)
    n = lib().orc_search_by_bow_kf(_p(kps1), len(kps1), _p(desc1), _p(h1), _p(n1), _p(o1), _p(i1), len(n1),
                                   _p(kps2), len(kps2), _p(desc2), _p(h2), _p(n2), _p(o2), _p(i2), len(n2), _p(m), nnratio, int(checkOri))
    return n, m


def distinctive_descriptors(desc, offsets):
    desc = np.ascontiguousarray(desc, np.uint8); offsets = np.ascontiguousarray(offsets, np.int32)
    best = np.zeros(len(offsets) - 1, np.int32)
    lib().orc_distinctive_descriptors(_p(desc), _p(offsets), len(offsets) - 1, _p(best))
    return best


def search_for_triangulation(kps1, desc1, elig1, fv1, kps2, desc2, elig2, fv2, ep, F12, scale2, sigma2_2, coarse=False, checkOri=True):
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE); kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    e1 = np.ascontiguousarray(elig1, np.uint8); e2 = np.ascontiguousarray(elig2, np.uint8)
    n1, o1, i1 = [np.ascontiguousarray(a, t) for a, t in zip(fv1, (np.uint32, np.int32, np.int32))]
    n2, o2, i2 = [np.ascontiguousarray(a, t) for a, t in zip(fv2, (np.uint32, np.int32, np.int32))]
    ep = np.ascontiguousarray(ep, np.float32); F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
    m = np.full(len(kps1), -1, np.int32)
    n = lib().orc_search_for_triangulation(_p(kps1), len(kps1), _p(desc1), desc1.shape[1], _p(e1), _p(n1), _p(o1), _p(i1), len(n1),
                                           _p(kps2), len(kps2), _p(desc2), desc2.shape[1], _p(e2), _p(n2), _p(o2), _p(i2), len(n2),
                                           _p(ep), _p(F), _p(sc), _p(sg), int(coarse), int(checkOri), _p(m))
    return n, m


def kf_radius_match(frame, valid, uv, radius, level, q_desc, inv_sigma2=None, taken=None, accept_thr=0.0, uright=None, q_ur=None):
    """frame: oracle Frame (KeyFrame grid).  Returns (best_idx, best_dist[, taken]).  uright / q_ur: Fuse's stereo gate."""
    valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
    radius = np.ascontiguousarray(radius, np.float32); level = np.ascontiguousarray(level, np.int32)
    q_desc = np.ascontiguousarray(q_desc, np.uint8)
    M = len(valid)
    bi = np.zeros(M, np.int32); bd = np.zeros(M, np.int32)
    isg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    tk = None if taken is None else np.array(taken, np.uint8)
    if uright is not None:
        ur = np.ascontiguousarray(uright, np.float32); qr = np.ascontiguousarray(q_ur, np.float32)
        lib().orc_kf_radius_match_stereo(frame.h, M, _p(valid), _p(uv), _p(radius), _p(level), _p(q_desc), _p(isg), _p(ur), _p(qr), _p(bi), _p(bd))
        return bi, bd
    lib().orc_kf_radius_match(frame.h, M, _p(valid), _p(uv), _p(radius), _p(level), _p(q_desc),
                              None if isg is None else _p(isg), None if tk is None else _p(tk), float(accept_thr), _p(bi), _p(bd))
    return (bi, bd) if tk is None else (bi, bd, tk)


def search_by_projection_kf(cur, kf_kps, kf_is_orb, valid, uv, pred_level, level_scale, mp_desc, cur_mp, th, ORBdist, checkOri=True):
    kf_kps = np.ascontiguousarray(kf_kps, KP_DTYPE)
    kio = None if kf_is_orb is None else np.ascontiguousarray(kf_is_orb, np.uint8)
    valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
    pl = np.ascontiguousarray(pred_level, np.int32); ls = np.ascontiguousarray(level_scale, np.float32)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    cm = np.ascontiguousarray(cur_mp, np.int32).copy()
    n = lib().orc_search_by_projection_kf(cur.h, _p(kf_kps), len(kf_kps), _p(kio), _p(valid), _p(uv), _p(pl), _p(ls), _p(mp_desc),
                                          _p(cm), float(th), int(ORBdist), int(checkOri))
    return n, cm


def undistort_events(raw, mapX, mapY, W, H, checkInImage=True, tsFactor=1.0):
    raw = np.ascontiguousarray(raw, RAW_DTYPE)
    mapX = np.ascontiguousarray(mapX, np.float32); mapY = np.ascontiguousarray(mapY, np.float32)
    out = np.zeros(len(raw), EVENT_DTYPE)
    L = lib()
    L.orc_undistort_events.restype = C.c_size_t
    L.orc_undistort_events.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_void_p]
    k = L.orc_undistort_events(_p(raw), len(raw), _p(mapX), _p(mapY), mapX.shape[1], mapX.shape[0], W, H, int(checkInImage),
                               float(tsFactor), _p(out))
    return out[:k].copy()


class Vocabulary(C.Structure):
    _fields_ = [("nnodes", C.c_int), ("L", C.c_int), ("child_off", C.c_void_p), ("child_ids", C.c_void_p),
                ("node_desc", C.c_void_p), ("word_id", C.c_void_p), ("weight", C.c_void_p)]


def bow_transform(voc, desc, levelsup=4, weighting=0, norm=1):
    """voc: dict(L, child_off, child_ids, node_desc, word_id, weight). Returns (bow_word, bow_val, (fv_node, fv_off, fv_idx), word_of, node_of)."""
    co = np.ascontiguousarray(voc["child_off"], np.int32); ci = np.ascontiguousarray(voc["child_ids"], np.int32)
    nd = np.ascontiguousarray(voc["node_desc"], np.uint8); wi = np.ascontiguousarray(voc["word_id"], np.int32)
    ww = np.ascontiguousarray(voc["weight"], np.float64)
    v = Vocabulary(len(co) - 1, int(voc["L"]), co.ctypes.data, ci.ctypes.data, nd.ctypes.data, wi.ctypes.data, ww.ctypes.data)
    desc = np.ascontiguousarray(desc, np.uint8); n = len(desc)
    bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); nw = C.c_int(0)
    fn = np.zeros(max(n, 1), np.uint32); fo = np.zeros(n + 1, np.int32); fi = np.zeros(max(n, 1), np.int32); nn = C.c_int(0)
    wo = np.zeros(max(n, 1), np.int32); no = np.zeros(max(n, 1), np.int32)
    L = lib(); L.orc_bow_transform.restype = None
    L.orc_bow_transform(C.byref(v), _p(desc), C.c_int(n), C.c_int(desc.shape[1] if n else 32), C.c_int(levelsup), C.c_int(weighting),
                        C.c_int(norm), _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fo), _p(fi), C.byref(nn), _p(wo), _p(no))
    return bw[:nw.value].copy(), bv[:nw.value].copy(), (fn[:nn.value].copy(), fo[:nn.value + 1].copy(), fi[:fo[nn.value]].copy()), wo[:n], no[:n]


def parse_events_text(text):
    """text: bytes. Returns RAW_DTYPE events, or raises ValueError(line) for a line outside the accepted grammar."""
    L = lib(); L.orc_parse_events_text.restype = C.c_long
    L.orc_parse_events_text.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    cap = text.count(b"\n") + 1
    out = np.zeros(cap, RAW_DTYPE)
    r = L.orc_parse_events_text(text, len(text), _p(out), cap)
    if r < 0:
        raise ValueError(-r - 1)
    return out[:r].copy()


def calc_optical_flow_pyr_lk(prev, nxt, prev_pts, next_pts=None, win=23, maxLevel=1, maxCount=10, epsilon=0.03, flags=0, minEig=1e-4):
    prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
    pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    npts = np.zeros_like(pp) if next_pts is None else np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
    n = len(pp)
    st = np.zeros(max(n, 1), np.uint8); er = np.zeros(max(n, 1), np.float32)
    L = lib(); L.orc_calc_optical_flow_pyr_lk.restype = None
    L.orc_calc_optical_flow_pyr_lk(_p(prev), _p(nxt), C.c_int(prev.shape[1]), C.c_int(prev.shape[0]), C.c_int(prev.shape[1]), _p(pp), _p(npts),
                                   C.c_int(n), C.c_int(win), C.c_int(maxLevel), C.c_int(maxCount), C.c_double(epsilon), C.c_int(flags),
                                   C.c_float(minEig), _p(st), _p(er))
    return npts, st[:n], er[:n]


def hamming_window_match(q_desc, t_desc, cand_offsets, cand_idx):
    q = np.ascontiguousarray(q_desc, np.uint8); t = np.ascontiguousarray(t_desc, np.uint8)
    co = np.ascontiguousarray(cand_offsets, np.int32); ci = np.ascontiguousarray(cand_idx, np.int32)
    nq = len(q)
    out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
    L = lib(); L.orc_hamming_window_match.restype = None
    L.orc_hamming_window_match(_p(q), C.c_int(nq), C.c_int(q.shape[1] if nq else 32), _p(t), C.c_int(t.shape[1] if len(t) else 32), _p(co), _p(ci),
                               _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]))
    return tuple(o[:nq] for o in out)
