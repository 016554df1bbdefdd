"""Python host side of the front end: thin mirrors of the reference's three C++ seams over the C ABI
(include/eorb_fe.h).  Names and argument meaning follow the reference:

  EvImConverter.ev2im / ev2im_gauss   include/Event/EventConversion.h:52-56
  ORBextractor.__call__               include/ORBextractor.h:75-81
  ORBmatcher.SearchForInitialization / SearchByProjection / DescriptorDistance   include/ORBmatcher.h:46-94
  BFMatcher.knnMatch                  cv::BFMatcher use at src/Frame.cc:1228

Used by tests/ and bench.py; the C++ mirror for the reference's own build is in eorb_slam_amd/host/.
Every call goes to the HIP library; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from .synth import EVENT_DTYPE, KP_DTYPE, RAW_DTYPE

EV16_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("t", "<f8")])


class EorbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("eorb_fe error %d: %s" % (code, msg))
        self.code = code


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """eorb_ctx: device workspaces + one HIP stream.  Single-threaded; create one per thread."""

    def __init__(self, device=0, stream=None):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.eorb_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise EorbError(rc, "eorb_create failed (no usable HIP device?)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.eorb_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise EorbError(rc, self.L.eorb_last_error(self.h).decode())

    def sync(self):
        """Waits for the stream; raises EorbError(EORB_E_CAPACITY) once if an earlier batched call overflowed internally."""
        self.check(self.L.eorb_sync(self.h))

    def debug_option(self, name, value):
        self.check(self.L.eorb_debug_option(self.h, name.encode(), int(value)))

    def debug_counter(self, name):
        return int(self.L.eorb_debug_counter(self.h, name.encode()))

    # profiling ------------------------------------------------------------------------------------
    def prof_enable(self, on=True):
        self.check(self.L.eorb_prof_enable(self.h, int(on)))

    def prof_only(self, names=()):
        """Time only these scopes (empty: all)."""
        self.check(self.L.eorb_prof_only(self.h, ",".join(names).encode()))

    def prof_reset(self):
        self.check(self.L.eorb_prof_reset(self.h))

    def prof_results(self):
        out = {}
        n = self.L.eorb_prof_count(self.h)
        for i in range(n):
            name = C.c_char_p(); ms = C.c_double(); cnt = C.c_int64()
            self.L.eorb_prof_get(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out

    # raw device memory (for the HBM-resident batch path without torch) ---------------------------
    def dev_alloc(self, nbytes):
        p = self.L.eorb_dev_alloc(self.h, nbytes)
        if not p:
            raise EorbError(_lib.EORB_E_HIP, self.L.eorb_last_error(self.h).decode())
        return p

    def dev_free(self, p):
        self.check(self.L.eorb_dev_free(self.h, C.c_void_p(p)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.L.eorb_dev_upload(self.h, C.c_void_p(dptr), _p(arr), arr.nbytes))

    def download(self, arr, dptr):
        self.check(self.L.eorb_dev_download(self.h, _p(arr), C.c_void_p(dptr), arr.nbytes))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def pack_events(ev):
    """EventData[] (24 B AoS) -> HBM record (16 B: x, y, t with the polarity in t's sign bit)."""
    ev = np.ascontiguousarray(ev, EVENT_DTYPE)
    out = np.zeros(len(ev), EV16_DTYPE)
    _lib.lib().eorb_pack_events(_p(ev), len(ev), _p(out))
    return out


def pack_raw_events4(raw):
    """eorb_raw_event[] (16 B) -> eorb_raw_event4[] (x | p << 15 | y << 16): the wire record of the event images."""
    raw = np.ascontiguousarray(raw, RAW_DTYPE)
    return (raw["x"].astype(np.uint32) & 0x7fff) | ((raw["p"] != 0).astype(np.uint32) << 15) | (raw["y"].astype(np.uint32) << 16)


def pack_raw_events2(raw, LW):
    """eorb_raw_event[] (16 B) -> eorb_raw_event2[] (u16: y * LW + x): the wire record of polarity-free images on sensors of <= 65 535 pixels"""
    raw = np.ascontiguousarray(raw, RAW_DTYPE)
    return (raw["y"].astype(np.uint32) * np.uint32(LW) + raw["x"].astype(np.uint32)).astype(np.uint16)


class EvImConverter:
    """EORB_SLAM::EvImConverter (include/Event/EventConversion.h:45-75), static-method style."""

    @staticmethod
    def ev2im(vEvData, imWidth, imHeight, pol=False, normalized=True, ctx=None, return_all=False):
        ctx = ctx or default_context()
        ev = np.ascontiguousarray(vEvData, EVENT_DTYPE)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8)
        mm = np.zeros(2, np.float32); is_u8 = C.c_int(0)
        ctx.check(ctx.L.eorb_ev2im(ctx.h, _p(ev), len(ev), imWidth, imHeight, int(pol), int(normalized),
                                   _p(f32), _p(u8), _p(mm), C.byref(is_u8)))
        if return_all:
            return f32, (u8 if is_u8.value else None), mm
        return u8 if is_u8.value else f32

    @staticmethod
    def ev2im_gauss(vEvData, imWidth, imHeight, sigma=1.0, pol=False, normalized=True, ctx=None, return_all=False):
        ctx = ctx or default_context()
        ev = np.ascontiguousarray(vEvData, EVENT_DTYPE)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8)
        mm = np.zeros(2, np.float32)
        ctx.check(ctx.L.eorb_ev2im_gauss(ctx.h, _p(ev), len(ev), imWidth, imHeight, float(sigma), int(pol),
                                         int(normalized), _p(f32), _p(u8), _p(mm)))
        if return_all:
            return f32, (u8 if normalized else None), mm
        return u8 if normalized else f32


    # ---- raw sensor events + MyCalibrator undistortion maps (src/Event/EventLoader.cpp:264-305, Utils/MyCalibrator.cpp:164-180) ----
    @staticmethod
    def set_undistort_maps(mapX, mapY, checkInImage=True, ctx=None):
        ctx = ctx or default_context()
        mx = np.ascontiguousarray(mapX, np.float32); my = np.ascontiguousarray(mapY, np.float32)
        ctx.check(ctx.L.eorb_set_undistort_maps(ctx.h, _p(mx), _p(my), mx.shape[1], mx.shape[0], int(checkInImage)))

    @staticmethod
    def undistort_events(raw, imWidth, imHeight, tsFactor=1.0, ctx=None):
        """The rectification of EventDataStore::getEventChunkRectified: raw (RAW_DTYPE) -> EVENT_DTYPE events kept by checkInImage."""
        ctx = ctx or default_context()
        raw = np.ascontiguousarray(raw, RAW_DTYPE)
        out = np.zeros(len(raw), EVENT_DTYPE); k = C.c_size_t(0)
        ctx.check(ctx.L.eorb_undistort_events(ctx.h, _p(raw), len(raw), imWidth, imHeight, float(tsFactor), _p(out), C.byref(k)))
        return out[:k.value].copy()

    @staticmethod
    def parse_events_text(text, ctx=None):
        """The text half of the loader (EventLoader.cpp:80-92): bytes of "ts x y p" lines -> RAW_DTYPE events."""
        ctx = ctx or default_context()
        cap = text.count(b"\n") + 1
        out = np.zeros(cap, RAW_DTYPE); k = C.c_size_t(0); bad = C.c_int64(-1)
        ctx.check(ctx.L.eorb_parse_events_text(ctx.h, text, len(text), _p(out), cap, C.byref(k), C.byref(bad)))
        return out[:k.value].copy()

    @staticmethod
    def ev2im_gauss_raw(raw, imWidth, imHeight, sigma=1.0, pol=False, normalized=True, ctx=None, return_all=False):
        ctx = ctx or default_context()
        raw = np.ascontiguousarray(raw, RAW_DTYPE)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8)
        mm = np.zeros(2, np.float32)
        ctx.check(ctx.L.eorb_ev2im_gauss_raw(ctx.h, _p(raw), len(raw), imWidth, imHeight, float(sigma), int(pol),
                                             int(normalized), _p(f32), _p(u8), _p(mm)))
        if return_all:
            return f32, (u8 if normalized else None), mm
        return u8 if normalized else f32

    @staticmethod
    def ev2im_raw(raw, imWidth, imHeight, pol=False, normalized=True, ctx=None, return_all=False):
        ctx = ctx or default_context()
        raw = np.ascontiguousarray(raw, RAW_DTYPE)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8)
        mm = np.zeros(2, np.float32); is_u8 = C.c_int(0)
        ctx.check(ctx.L.eorb_ev2im_raw(ctx.h, _p(raw), len(raw), imWidth, imHeight, int(pol), int(normalized),
                                       _p(f32), _p(u8), _p(mm), C.byref(is_u8)))
        if return_all:
            return f32, (u8 if is_u8.value else None), mm
        return u8 if is_u8.value else f32

    @staticmethod
    def ev2mci_gg_f_se3(vEvData, cam, angle, axis, t, medDepth, imWidth, imHeight, sigma=1.0, pol=False, normalized=False,
                        depth_per_event=None, ctx=None):
        """ev2mci_gg_f(evs, pCamera, Tcw, medDepth | depth map, ...) (src/Event/EventConversion.cc:280-360, 451-531); cam =
        (fx, fy, cx, cy) for a Pinhole, (fx, fy, cx, cy, k1, k2, k3, k4) for a KannalaBrandt8 camera; angle/axis = AngleAxisd(R(Tcw)), t = translation.  Returns (f32 image, u8 image|None, minmax)."""
        ctx = ctx or default_context()
        ev = np.ascontiguousarray(vEvData, EVENT_DTYPE)
        ax = np.ascontiguousarray(axis, np.float64); tt = np.ascontiguousarray(t, np.float64)
        dp = None if depth_per_event is None else np.ascontiguousarray(depth_per_event, np.float32)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8); mm = np.zeros(2, np.float32)
        pc = _lib.camera(cam)
        ctx.check(ctx.L.eorb_ev2mci_se3_cam(ctx.h, _p(ev), len(ev), C.byref(pc), float(angle), _p(ax), _p(tt), float(medDepth), _p(dp),
                                        imWidth, imHeight, float(sigma), int(pol), int(normalized), _p(f32), _p(u8), _p(mm)))
        return f32, (u8 if (normalized and len(ev)) else None), mm

    @staticmethod
    def ev2mci_gg_f_se2(vEvData, cam, params2D, imWidth, imHeight, sigma=1.0, pol=False, normalized=False, ctx=None):
        """ev2mci_gg_f(evs, pCamera, params2D, ...) (src/Event/EventConversion.cc:363-448)"""
        ctx = ctx or default_context()
        ev = np.ascontiguousarray(vEvData, EVENT_DTYPE)
        pr = np.ascontiguousarray(params2D, np.float32)
        f32 = np.empty((imHeight, imWidth), np.float32); u8 = np.zeros((imHeight, imWidth), np.uint8); mm = np.zeros(2, np.float32)
        pc = _lib.camera(cam)
        ctx.check(ctx.L.eorb_ev2mci_se2_cam(ctx.h, _p(ev), len(ev), C.byref(pc), _p(pr), len(pr), imWidth, imHeight, float(sigma),
                                        int(pol), int(normalized), _p(f32), _p(u8), _p(mm)))
        return f32, (u8 if (normalized and len(ev)) else None), mm

    @staticmethod
    def measureImageFocus(image, ctx=None):
        """src/Event/EventConversion.cc:74-111"""
        ctx = ctx or default_context()
        img = np.ascontiguousarray(image, np.float32); H, W = img.shape
        f = C.c_float(0)
        ctx.check(ctx.L.eorb_measure_image_focus(ctx.h, _p(img), W, H, C.byref(f)))
        return f.value

    @staticmethod
    def measureImageFocusN(images, ctx=None):
        """the focus scores of n images (n, H, W) in one call: the four reconstructions of the motion-compensation contest
        (src/Event/EvImBuilder.cpp:1165-1203)"""
        ctx = ctx or default_context()
        imgs = np.ascontiguousarray(images, np.float32); n, H, W = imgs.shape
        f = np.zeros(n, np.float32)
        ctx.check(ctx.L.eorb_measure_image_focus_n(ctx.h, _p(imgs), n, W, H, _p(f)))
        return f


def cv_normalize_minmax_u8(image, ctx=None):
    """cv::normalize(img, img, 255, 0, NORM_MINMAX, CV_8UC1) as called at src/Event/EvImBuilder.cpp:1076"""
    ctx = ctx or default_context()
    img = np.ascontiguousarray(image, np.float32); H, W = img.shape
    out = np.zeros((H, W), np.uint8)
    ctx.check(ctx.L.eorb_normalize_minmax_u8(ctx.h, _p(img), W, H, _p(out)))
    return out


class ORBextractor:
    """ORB_SLAM3::ORBextractor (include/ORBextractor.h:49-139).  One instance per thread, like the reference."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10, minThFAST=0, edgeTh=19,
                 imSize=(240, 180), ctx=None):
        self.ctx = ctx or Context()
        self.params = _lib.OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, edgeTh, imSize[0])
        self.W, self.H = imSize
        self.nlevels = nlevels
        self.ctx.check(self.ctx.L.eorb_orb_configure(self.ctx.h, C.byref(self.params), self.W, self.H))
        self.cap = self.ctx.L.eorb_orb_max_keypoints(self.ctx.h)
        sf = np.zeros(nlevels, np.float32); inv = np.zeros(nlevels, np.float32); nf = np.zeros(nlevels, np.int32)
        e = C.c_int()
        self.ctx.check(self.ctx.L.eorb_orb_get_tables(self.ctx.h, _p(sf), _p(inv), _p(nf), C.byref(e)))
        self.mvScaleFactor, self.mvInvScaleFactor, self.mnFeaturesPerLevel, self.edge = sf, inv, nf, e.value

    def stereo(self, imLeft, imRight, mb, mbf):
        """Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:97-152): both extractions and Frame::ComputeStereoMatches (:869-1048) in one
        call.  Returns dict(kpsL, descL, kpsR, descR, uRight (mvuRight), depth (mvDepth), nmatches (before the median cut))."""
        imL = np.ascontiguousarray(imLeft, np.uint8); imR = np.ascontiguousarray(imRight, np.uint8)
        H, W = imL.shape
        if imR.shape != imL.shape:
            raise ValueError("left and right image differ in size")
        cap = self.cap
        kL = np.zeros(cap, KP_DTYPE); kR = np.zeros(cap, KP_DTYPE); dL = np.zeros((cap, 32), np.uint8); dR = np.zeros((cap, 32), np.uint8)
        ur = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
        nL, nR, nm = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.ctx.L.eorb_frame_stereo(self.ctx.h, _p(imL), _p(imR), W, H, imL.strides[0], float(mb), float(mbf), _p(kL), _p(dL), C.byref(nL),
                                                    _p(kR), _p(dR), C.byref(nR), cap, _p(ur), _p(dp), C.byref(nm)))
        a, b = nL.value, nR.value
        return dict(kpsL=kL[:a].copy(), descL=dL[:a].copy(), kpsR=kR[:b].copy(), descR=dR[:b].copy(), uRight=ur[:a].copy(), depth=dp[:a].copy(), nmatches=nm.value)

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactors(self):
        return self.mvScaleFactor

    def __call__(self, image, vLappingArea=(0, 1000), want_desc=True):
        """Returns (monoIndex, keypoints, descriptors|None, oob_flags).  monoIndex = -1 for an empty image."""
        if image is None or image.size == 0:
            return -1, None, None, None
        image = np.ascontiguousarray(image, np.uint8)
        H, W = image.shape
        kps = np.zeros(self.cap, KP_DTYPE); desc = np.zeros((self.cap, 32), np.uint8); oob = np.zeros(self.cap, np.uint8)
        n = C.c_int(); mono = C.c_int()
        self.ctx.check(self.ctx.L.eorb_orb_extract(self.ctx.h, _p(image), W, H, image.strides[0], vLappingArea[0],
                                                   vLappingArea[1], int(want_desc), _p(kps), _p(desc), _p(oob),
                                                   self.cap, C.byref(n), C.byref(mono)))
        k = n.value
        return mono.value, kps[:k].copy(), (desc[:k].copy() if want_desc else None), oob[:k].copy()


    def ComputeTrackedKPtsDesc(self, trackedImage, trackedKPts):
        """src/ORBextractor.cc:1316-1363 -> (refDescs n x 32, oob flags)"""
        img = np.ascontiguousarray(trackedImage, np.uint8); H, W = img.shape
        kps = np.ascontiguousarray(trackedKPts, KP_DTYPE); n = len(kps)
        desc = np.zeros((n, 32), np.uint8); oob = np.zeros(n, np.uint8)
        self.ctx.check(self.ctx.L.eorb_orb_tracked_descriptors(self.ctx.h, _p(img), W, H, img.strides[0], _p(kps), n, _p(desc), _p(oob)))
        return desc, oob

    def AssignKPtLevelByBestDesc(self, refDescs, trackedImage, trackedKPts):
        """src/ORBextractor.cc:1267-1314 -> keypoints with the octave field reassigned"""
        img = np.ascontiguousarray(trackedImage, np.uint8); H, W = img.shape
        kps = np.ascontiguousarray(trackedKPts, KP_DTYPE).copy(); ref = np.ascontiguousarray(refDescs, np.uint8)
        self.ctx.check(self.ctx.L.eorb_orb_assign_level_by_best_desc(self.ctx.h, _p(img), W, H, img.strides[0], _p(ref), _p(kps), len(kps)))
        return kps


def grid_bounds(W, H):
    """Frame::ComputeImageBounds for an undistorted image + grid pitch (src/Frame.cc:862-866, 362-363)."""
    gb = _lib.GridBounds(0.0, 0.0, float(W), float(H), 0.0, 0.0)
    gb.invW = np.float32(64.0) / (np.float32(gb.maxX) - np.float32(gb.minX))
    gb.invH = np.float32(48.0) / (np.float32(gb.maxY) - np.float32(gb.minY))
    return gb


class FrameView:
    """What the matchers read from a Frame: undistorted keypoints, descriptors, (Mixed) type flags, bounds."""

    def __init__(self, kps, desc, W, H, is_orb=None):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        self.is_orb = None if is_orb is None else np.ascontiguousarray(is_orb, np.uint8)
        self.gb = grid_bounds(W, H)
        self.N = len(self.kps)


class ORBmatcher:
    """ORB_SLAM3::ORBmatcher / EORB_SLAM::MixedMatcher (include/ORBmatcher.h:40-116, include/MixedMatcher.h)."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, checkOri=True, ctx=None):
        self.mfNNratio = float(nnratio); self.mbCheckOrientation = bool(checkOri)
        self.ctx = ctx or default_context()

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.full(F1.N, -1, np.int32); nm = C.c_int(0)
        c = self.ctx
        c.check(c.L.eorb_search_for_initialization(c.h, _p(F1.kps), F1.N, _p(F1.desc), F1.desc.shape[1], _p(F1.is_orb),
                                                   _p(F2.kps), F2.N, _p(F2.desc), F2.desc.shape[1], _p(F2.is_orb),
                                                   C.byref(F2.gb), _p(pm), _p(m12), int(windowSize), self.mfNNratio,
                                                   int(self.mbCheckOrientation), C.byref(nm)))
        return nm.value, m12, pm

    def SearchByProjectionLast(self, Cur, Last, valid, uv, mp_desc, mp_obs, cur_mp, th, level_scale, mode=0,
                               uright=None, proj_ur=None):
        """SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) with the projection done
        by the caller (valid/uv), src/ORBmatcher.cc:1969-2187.  uright = CurrentFrame.mvuRight and proj_ur = u - mbf*invzc
        per last-frame point switch the rectified-stereo gate (:2056-2062) on."""
        valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.uint8)
        ls = np.ascontiguousarray(level_scale, np.float32)
        cm = np.ascontiguousarray(cur_mp, np.int32).copy(); nm = C.c_int(0)
        c = self.ctx
        if uright is not None:
            ur = np.ascontiguousarray(uright, np.float32); pu = np.ascontiguousarray(proj_ur, np.float32)
            if len(ur) != Cur.N or len(pu) != Last.N:
                raise ValueError("uright is per current keypoint, proj_ur per last-frame keypoint")
            c.check(c.L.eorb_search_by_projection_last_stereo(c.h, _p(Cur.kps), Cur.N, _p(Cur.desc), Cur.desc.shape[1], _p(Cur.is_orb),
                                                              _p(Last.kps), Last.N, _p(Last.is_orb), _p(valid), _p(uv), _p(mp_desc),
                                                              _p(mp_obs), _p(ls), C.byref(Cur.gb), _p(cm), float(th), int(mode),
                                                              int(self.mbCheckOrientation), _p(ur), _p(pu), C.byref(nm)))
            return nm.value, cm
        c.check(c.L.eorb_search_by_projection_last(c.h, _p(Cur.kps), Cur.N, _p(Cur.desc), Cur.desc.shape[1], _p(Cur.is_orb),
                                                   _p(Last.kps), Last.N, _p(Last.is_orb), _p(valid), _p(uv), _p(mp_desc),
                                                   _p(mp_obs), _p(ls), C.byref(Cur.gb), _p(cm), float(th), int(mode),
                                                   int(self.mbCheckOrientation), C.byref(nm)))
        return nm.value, cm

    def SearchByProjectionKF(self, Cur, kf_kps, kf_is_orb, valid, uv, pred_level, level_scale, mp_desc, cur_mp, th, ORBdist):
        """SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) after projection,
        src/ORBmatcher.cc:2189-2312 (relocalisation)."""
        kf_kps = np.ascontiguousarray(kf_kps, KP_DTYPE)
        kio = None if kf_is_orb is None else np.ascontiguousarray(kf_is_orb, np.uint8)
        valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
        pl = np.ascontiguousarray(pred_level, np.int32); ls = np.ascontiguousarray(level_scale, np.float32)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        cm = np.ascontiguousarray(cur_mp, np.int32).copy(); nm = C.c_int(0)
        c = self.ctx
        c.check(c.L.eorb_search_by_projection_kf(c.h, _p(Cur.kps), Cur.N, _p(Cur.desc), Cur.desc.shape[1], _p(Cur.is_orb),
                                                 _p(kf_kps), len(kf_kps), _p(kio), _p(valid), _p(uv), _p(pl), _p(ls), _p(mp_desc),
                                                 C.byref(Cur.gb), _p(cm), float(th), int(ORBdist), int(self.mbCheckOrientation),
                                                 C.byref(nm)))
        return nm.value, cm

    def SearchByProjectionMap(self, F, in_view, proj_xy, level, view_cos, mp_desc, mp_obs, frame_mp, th, level_scale,
                              mp_is_orb=None, uright=None, proj_xr=None):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th) with Frame::isInFrustum's outputs as inputs,
        src/ORBmatcher.cc:44-219.  uright = F.mvuRight and proj_xr = mTrackProjXR per map point switch the
        rectified-stereo gate (:96-104) on."""
        in_view = np.ascontiguousarray(in_view, np.uint8); proj_xy = np.ascontiguousarray(proj_xy, np.float32)
        level = np.ascontiguousarray(level, np.int32); view_cos = np.ascontiguousarray(view_cos, np.float32)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8); mp_obs = np.ascontiguousarray(mp_obs, np.uint8)
        ls = np.ascontiguousarray(level_scale, np.float32)
        mio = None if mp_is_orb is None else np.ascontiguousarray(mp_is_orb, np.uint8)
        fm = np.ascontiguousarray(frame_mp, np.int32).copy(); nm = C.c_int(0)
        c = self.ctx
        if uright is not None:
            ur = np.ascontiguousarray(uright, np.float32); px = np.ascontiguousarray(proj_xr, np.float32)
            if len(ur) != F.N or len(px) != len(in_view):
                raise ValueError("uright is per frame keypoint, proj_xr per map point")
            c.check(c.L.eorb_search_by_projection_map_stereo(c.h, _p(F.kps), F.N, _p(F.desc), F.desc.shape[1], _p(F.is_orb),
                                                             len(in_view), _p(in_view), _p(proj_xy), _p(level), _p(view_cos),
                                                             _p(mp_desc), _p(mp_obs), _p(mio), _p(ls), C.byref(F.gb), _p(fm),
                                                             float(th), self.mfNNratio, _p(ur), _p(px), C.byref(nm)))
            return nm.value, fm
        c.check(c.L.eorb_search_by_projection_map(c.h, _p(F.kps), F.N, _p(F.desc), F.desc.shape[1], _p(F.is_orb),
                                                  len(in_view), _p(in_view), _p(proj_xy), _p(level), _p(view_cos),
                                                  _p(mp_desc), _p(mp_obs), _p(mio), _p(ls), C.byref(F.gb), _p(fm),
                                                  float(th), self.mfNNratio, C.byref(nm)))
        return nm.value, fm


def SearchByBoW(kf_kps, kf_desc, kf_has_mp, kf_fv, f_kps, f_desc, f_fv, nnratio=0.7, checkOri=True, ctx=None):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:276-478); feature vectors as CSR triples
    (nodes uint32 ascending, node_off int32[nn+1], idx int32[]).  Returns (nmatches, match_f)."""
    c = ctx or default_context()
    kf_kps = np.ascontiguousarray(kf_kps, KP_DTYPE); f_kps = np.ascontiguousarray(f_kps, KP_DTYPE)
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8); f_desc = np.ascontiguousarray(f_desc, np.uint8)
    hm = np.ascontiguousarray(kf_has_mp, np.uint8)
    kn, ko, ki = [np.ascontiguousarray(a, t) for a, t in zip(kf_fv, (np.uint32, np.int32, np.int32))]
    fn, fo, fi = [np.ascontiguousarray(a, t) for a, t in zip(f_fv, (np.uint32, np.int32, np.int32))]
    m = np.full(len(f_kps), -1, np.int32); nm = C.c_int(0)
    c.check(c.L.eorb_search_by_bow(c.h, _p(kf_kps), len(kf_kps), _p(kf_desc), _p(hm), _p(kn), _p(ko), _p(ki), len(kn),
                                   _p(f_kps), len(f_kps), _p(f_desc), _p(fn), _p(fo), _p(fi), len(fn), _p(m), float(nnratio),
                                   int(checkOri), C.byref(nm)))
    return nm.value, m


def SearchByBoW_KF(kps1, desc1, has_mp1, fv1, kps2, desc2, has_mp2, fv2, nnratio=0.8, checkOri=True, ctx=None):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:833-973). Returns (nmatches, match12)."""
    c = ctx or default_context()
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE); kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    n1, o1, i1 = [np.ascontiguousarray(a, t) for a, t in zip(fv1, (np.uint32, np.int32, np.int32))]
    n2, o2, i2 = [np.ascontiguousarray(a, t) for a, t in zip(fv2, (np.uint32, np.int32, np.int32))]
    m = np.full(len(kps1), -1, np.int32); nm = C.c_int(0)
    c.check(c.L.eorb_search_by_bow_kf(c.h, _p(kps1), len(kps1), _p(desc1), _p(h1), _p(n1), _p(o1), _p(i1), len(n1),
                                      _p(kps2), len(kps2), _p(desc2), _p(h2), _p(n2), _p(o2), _p(i2), len(n2), _p(m),
                                      float(nnratio), int(checkOri), C.byref(nm)))
    return nm.value, m


def SearchForTriangulation(kps1, desc1, elig1, fv1, kps2, desc2, elig2, fv2, ep, F12, scale2, sigma2_2,
                           bCoarse=False, checkOri=True, ctx=None):
    """Mono ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:975-1214). Returns (nmatches, vMatchedPairs as (k,2) int32)."""
    c = ctx or default_context()
    kps1 = np.ascontiguousarray(kps1, KP_DTYPE); kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    e1 = np.ascontiguousarray(elig1, np.uint8); e2 = np.ascontiguousarray(elig2, np.uint8)
    n1, o1, i1 = [np.ascontiguousarray(a, t) for a, t in zip(fv1, (np.uint32, np.int32, np.int32))]
    n2, o2, i2 = [np.ascontiguousarray(a, t) for a, t in zip(fv2, (np.uint32, np.int32, np.int32))]
    ep = np.ascontiguousarray(ep, np.float32); F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
    m = np.full(len(kps1), -1, np.int32); nm = C.c_int(0)
    c.check(c.L.eorb_search_for_triangulation(c.h, _p(kps1), len(kps1), _p(desc1), desc1.shape[1], _p(e1), _p(n1), _p(o1), _p(i1), len(n1),
                                              _p(kps2), len(kps2), _p(desc2), desc2.shape[1], _p(e2), _p(n2), _p(o2), _p(i2), len(n2),
                                              _p(ep), _p(F), _p(sc), _p(sg), len(sc), int(bCoarse), int(checkOri), _p(m), C.byref(nm)))
    k = np.nonzero(m >= 0)[0]
    return nm.value, np.stack([k, m[k]], axis=1).astype(np.int32)


def KeyFrameRadiusMatch(kps, desc, gb, valid, uv, radius, level, q_desc, inv_sigma2=None, taken=None, accept_thr=0.0, ctx=None,
                        uright=None, q_ur=None):
    """Search core of Fuse / SearchBySim3 / SearchByProjection(KeyFrame*, Scw, ...) (src/ORBmatcher.cc:1512-1578, :1829-1860,
    :548-588). Returns (best_idx, best_dist) or (best_idx, best_dist, taken) when `taken` is given."""
    c = ctx or default_context()
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    valid = np.ascontiguousarray(valid, np.uint8); uv = np.ascontiguousarray(uv, np.float32)
    radius = np.ascontiguousarray(radius, np.float32); level = np.ascontiguousarray(level, np.int32)
    q_desc = np.ascontiguousarray(q_desc, np.uint8)
    M = len(valid)
    bi = np.zeros(M, np.int32); bd = np.zeros(M, np.int32)
    isg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    tk = None if taken is None else np.array(taken, np.uint8)
    if uright is not None:
        if isg is None or q_ur is None or tk is not None:
            raise ValueError("the stereo gate takes uright, q_ur and inv_sigma2, and no taken flags")
        ur = np.ascontiguousarray(uright, np.float32); qr = np.ascontiguousarray(q_ur, np.float32)
        if len(ur) != len(kps) or len(qr) != M:
            raise ValueError("uright is per keypoint, q_ur per map point")
        c.check(c.L.eorb_kf_radius_match_stereo(c.h, _p(kps), len(kps), _p(desc), desc.shape[1], C.byref(gb), M, _p(valid), _p(uv), _p(radius),
                                                _p(level), _p(q_desc), _p(isg), len(isg), _p(ur), _p(qr), _p(bi), _p(bd)))
        return bi, bd
    c.check(c.L.eorb_kf_radius_match(c.h, _p(kps), len(kps), _p(desc), desc.shape[1], C.byref(gb), M, _p(valid), _p(uv), _p(radius),
                                     _p(level), _p(q_desc), None if isg is None else _p(isg), 0 if isg is None else len(isg),
                                     None if tk is None else _p(tk), float(accept_thr), _p(bi), _p(bd)))
    return (bi, bd) if tk is None else (bi, bd, tk)


def Fuse(kps, desc, gb, valid, uv, level, scale_factors, inv_sigma2, q_desc, th=3.0, ctx=None, uright=None, q_ur=None):
    """Search part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:1407-1617): returns best_idx per map point
    with bestDist <= TH_LOW (-1 otherwise); the caller performs Replace / AddObservation (:1581-1600) in order.
    uright (pKF->mvuRight) and q_ur (u - bf*invz per map point) select the stereo reprojection gate (:1541-1553)."""
    sf = np.asarray(scale_factors, np.float32)
    lv = np.ascontiguousarray(level, np.int32)
    radius = (np.float32(th) * sf[np.clip(lv, 0, len(sf) - 1)]).astype(np.float32)
    bi, bd = KeyFrameRadiusMatch(kps, desc, gb, valid, uv, radius, lv, q_desc, inv_sigma2=inv_sigma2, ctx=ctx, uright=uright, q_ur=q_ur)
    return np.where(bd <= 50, bi, -1).astype(np.int32)


def SearchBySim3(kf1, kf2, q1, q2, th=7.5, ctx=None):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1743-1967) after projection. kf = (kps, desc, gb, scale_factors);
    q1 = (valid, uv in KF2, level, mp_desc) for the map points of KF1 (valid already excludes vbAlreadyMatched1, bad points and
    the depth / image / distance gates), q2 likewise into KF1.  Returns (nFound, match12) with match12[i1] = i2 or -1."""
    def side(kf, q):
        kps, desc, gb, sf = kf
        valid, uv, level, qd = q
        sf = np.asarray(sf, np.float32); lv = np.ascontiguousarray(level, np.int32)
        radius = (np.float32(th) * sf[np.clip(lv, 0, len(sf) - 1)]).astype(np.float32)
        bi, bd = KeyFrameRadiusMatch(kps, desc, gb, valid, uv, radius, lv, qd, ctx=ctx)
        return np.where(bd <= 100, bi, -1)                                   # TH_HIGH (:1862, :1942)
    vn1 = side(kf2, q1); vn2 = side(kf1, q2)
    i1 = np.arange(len(vn1))
    ok = (vn1 >= 0) & (vn2[np.clip(vn1, 0, max(len(vn2) - 1, 0))] == i1) if len(vn2) else np.zeros(len(vn1), bool)
    return int(ok.sum()), np.where(ok, vn1, -1).astype(np.int32)


class ELK_Tracker:
    """EORB_SLAM::ELK_Tracker (src/Event/KLT_Tracker.cpp): pyramidal LK on the device + the reference's match bookkeeping."""

    def __init__(self, kltWinSize=23, maxLevel=1, kltMaxItr=10, kltEps=0.03, ctx=None):
        self.ctx = ctx or default_context()
        self.mPatchSz, self.mMaxLevel, self.maxItr, self.eps = kltWinSize, maxLevel, kltMaxItr, kltEps
        self.mRefFrame = None

    def setRefImage(self, image, refPts):                       # :22-46
        self.mRefFrame = np.ascontiguousarray(image, np.uint8).copy()
        self.mRefKPoints = np.ascontiguousarray(refPts, KP_DTYPE).copy()
        self.mLastTrackedKPts = self.mRefKPoints.copy()
        self.mRefPoints = np.stack([self.mRefKPoints["x"], self.mRefKPoints["y"]], axis=1).astype(np.float32)
        self.mLastTrackedPts = self.mRefPoints.copy()

    def calcOpticalFlowPyrLK(self, prev, nxt, prev_pts, next_pts=None, flags=0, minEig=1e-4):
        c = self.ctx
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        pp = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
        npts = np.zeros_like(pp) if next_pts is None else np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
        n = len(pp)
        st = np.zeros(max(n, 1), np.uint8); er = np.zeros(max(n, 1), np.float32)
        c.check(c.L.eorb_calc_optical_flow_pyr_lk(c.h, _p(prev), _p(nxt), prev.shape[1], prev.shape[0], prev.shape[1], _p(pp), _p(npts), n,
                                                  self.mPatchSz, self.mMaxLevel, self.maxItr, float(self.eps), int(flags), float(minEig),
                                                  _p(st), _p(er)))
        return npts, st[:n], er[:n]

    def trackCurrImage(self, currImage, kpts=None):             # :49-98: the initial-flow form when a guess of matching size is given
        if kpts is not None and len(kpts) == len(self.mRefPoints):
            return self.calcOpticalFlowPyrLK(self.mRefFrame, currImage, self.mRefPoints, kpts, flags=4)
        return self.calcOpticalFlowPyrLK(self.mRefFrame, currImage, self.mRefPoints)

    def refineTrackedPts(self, currPts, status, vCntMatches=None):           # :104-151
        n = len(self.mRefPoints)
        cnt = np.ones(n, np.int32) if vCntMatches is None or len(vCntMatches) == 0 else np.asarray(vCntMatches, np.int32).copy()
        p1 = self.mRefKPoints.copy(); p1["x"] = currPts[:, 0]; p1["y"] = currPts[:, 1]
        H, W = self.mRefFrame.shape
        ok = (status == 1) & (currPts[:, 0] >= 0) & (currPts[:, 0] < np.float32(W)) & (currPts[:, 1] >= 0) & (currPts[:, 1] < np.float32(H))
        cnt[ok] += 1
        m12 = np.where(ok, np.arange(n), -1).astype(np.int32)
        d = currPts[ok] - self.mRefPoints[ok]
        disp = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32))
        return int(ok.sum()), p1, m12, cnt, disp

    def trackAndMatchCurrImage(self, image, vCntMatches=None):               # :215-234
        pts, st, err = self.trackCurrImage(image, self.mLastTrackedPts)
        self.mLastTrackedPts = pts
        nm, p1, m12, cnt, disp = self.refineTrackedPts(pts, st, vCntMatches)
        self.mLastTrackedKPts = p1
        return nm, p1, m12, cnt, disp

    def refineFirstOctaveLevel(self, vMatches12, vCntMatches, nMatches):     # :153-181
        m = np.asarray(vMatches12, np.int32).copy(); cnt = np.asarray(vCntMatches, np.int32).copy()
        bad = (m >= 0) & (self.mRefKPoints["octave"] > 0)
        m[bad] = -1; cnt[bad] -= 1
        return nMatches - int(bad.sum()), m, cnt


class EvImBuilder:
    """The chunk loop of EORB_SLAM::EvImBuilder::Track (src/Event/EvImBuilder.cpp:1300-1515) over the one-call seams
    eorb_ev_slice_extract / eorb_ev_slice_track / eorb_ev_mc_contest: per chunk of l1ChunkSize events the event image and its frame
    (INIT: detect-only ORBextractor + ELK_Tracker::setRefImage; TRACKING: ELK_Tracker::trackAndMatchCurrImage), the window-size rule
    (resolveEvWinSize :209-232) and, on a dispatch, the four-way reconstruction contest with the L2 detection of the winner
    (generateMCImage :1146-1247, isMcImageGood :260-267).  What stays with the caller, as in the reference: the optimisers that
    produce the poses of the motion-compensated reconstructions (`mci_poses`), the two-view RANSAC refinement of step() (:600-668,
    `reconstruct`), the IMU and the L2 tracker's queue."""
    IDLE, INIT, TRACKING = 0, 1, 2
    DEF_TH_MIN_KPTS, DEF_TH_MIN_MATCHES = 100, 50                     # include/Event/EventData.h:24-26

    def __init__(self, W=240, H=180, l1ChunkSize=2000, l1NumLoop=3, l1FixedWinSz=False, continTracking=True, maxPixelDisp=3.0,
                 l1WinOverlap=0.5, minEvGenRate=1.0, l1ImSigma=1.0, maxNumPts=400, fastTh=0, imMargin=9, klt=(23, 1, 10, 0.03),
                 cam=None, raw_events=False, keep_images=True):
        self.W, self.H, self.sigma = W, H, float(l1ImSigma)
        self.mbContTracking, self.mbFixedWinSize = bool(continTracking), bool(l1FixedWinSz)
        self.mL1EvWinSize = self.mInitL1EvWinSize = int(l1ChunkSize)
        self.mL1NumLoop, self.mL1MaxPxDisp, self.l1WinOverlap, self.minEvGenRate = int(l1NumLoop), float(maxPixelDisp), float(l1WinOverlap), float(minEvGenRate)
        self.mnWinOverlap = int(self.l1WinOverlap * float(self.mL1NumLoop * self.mL1EvWinSize))          # :40
        self.cam, self.raw, self.keep_images = cam, bool(raw_events), keep_images
        # the L1 builder's own extractor (FAST = ORB with one level, :49-54 + EvBaseTracker.cpp:150-164) and the L2 tracker's
        # (2 x maxNumPts, src/Event/EvAsynchTracker.cpp:51,59-62), one context each as the reference has one extractor object each
        self.ctx = Context(); self.ctx_l2 = Context()
        self.l1 = ORBextractor(maxNumPts, 1.0, 1, fastTh, 0, imMargin, imSize=(W, H), ctx=self.ctx)
        self.l2 = ORBextractor(2 * maxNumPts, 1.0, 1, fastTh, 0, imMargin, imSize=(W, H), ctx=self.ctx_l2)
        self.klt = _lib.KltParams(int(klt[0]), int(klt[1]), int(klt[2]), float(klt[3]), 1e-4)
        self.mStat = self.IDLE
        self.reset()

    def close(self):
        self.ctx.close(); self.ctx_l2.close()

    def set_undistort_maps(self, mapX, mapY, checkInImage=True):
        EvImConverter.set_undistort_maps(mapX, mapY, checkInImage, ctx=self.ctx)

    def resetAll(self):                                              # :70-93
        self.mStat = self.IDLE
        self.updateL1ChunkSize(self.mInitL1EvWinSize)
        self.reset()

    def reset(self):                                                 # :95-139
        self.mCurrIdx = 0
        self.mCntLowEvGenRate = 0
        self.mvSharedL2Evs = []
        self.mvMatchesCnt = None
        self.nref = -1

    def updateL1ChunkSize(self, newSz):                              # :188-195
        self.mL1EvWinSize = int(newSz)

    def _events(self, evs):
        if self.raw:
            return None, np.ascontiguousarray(evs, RAW_DTYPE)
        return np.ascontiguousarray(evs, EVENT_DTYPE), None

    def _slice_extract(self, evs):
        c = self.ctx
        ev, raw = self._events(evs)
        cap = c.L.eorb_orb_max_keypoints(c.h)
        kps = np.zeros(cap, KP_DTYPE); n = C.c_int(0); mono = C.c_int(0)
        img = np.zeros((self.H, self.W), np.uint8) if self.keep_images else None
        c.check(c.L.eorb_ev_slice_extract(c.h, _p(ev), _p(raw), len(evs), self.sigma, 0, 1000, 0, _p(kps), None, None, cap, C.byref(n), C.byref(mono), _p(img)))
        return kps[:n.value].copy(), img

    def _slice_track(self, evs, pts):
        c = self.ctx
        ev, raw = self._events(evs)
        pts = np.ascontiguousarray(pts, np.float32).copy()
        n = len(pts)
        st = np.zeros(max(n, 1), np.uint8); er = np.zeros(max(n, 1), np.float32)
        img = np.zeros((self.H, self.W), np.uint8) if self.keep_images else None
        c.check(c.L.eorb_ev_slice_track(c.h, _p(ev), _p(raw), len(evs), self.sigma, C.byref(self.klt), _p(pts), _p(st), _p(er), n, _p(img)))
        return pts, st[:n], er[:n], img

    def generateMCImage(self, evs, poses):                           # :1146-1247 + isMcImageGood :260-267
        c = self.ctx
        ev = np.ascontiguousarray(evs, EVENT_DTYPE)
        poses = poses or {}
        dp, ba = _lib.se3_motion(poses.get("dp")), _lib.se3_motion(poses.get("ba"))
        se2 = None if poses.get("se2") is None else np.ascontiguousarray(poses["se2"], np.float32)
        cam = None if self.cam is None else _lib.camera(self.cam)
        focus = np.zeros(5, np.float32); win = C.c_int(-1)
        img = np.zeros((self.H, self.W), np.uint8)
        cap = self.ctx_l2.L.eorb_orb_max_keypoints(self.ctx_l2.h)
        kps = np.zeros(cap, KP_DTYPE); n = C.c_int(0)
        c.check(c.L.eorb_ev_mc_contest(c.h, _p(ev), len(ev), C.byref(cam) if cam is not None else None, C.byref(dp) if dp is not None else None,
                                       C.byref(ba) if ba is not None else None, _p(se2), 0 if se2 is None else len(se2), self.W, self.H, self.sigma,
                                       _p(focus), C.byref(win), _p(img), self.ctx_l2.h, 0, 1000, _p(kps), cap, C.byref(n)))
        return dict(focus=focus, winner=win.value, image=img, l2_kps=kps[:n.value].copy())

    def Track(self, l1Evs, mci_poses=None, reconstruct=None):
        """One pass of the loop body for the chunk `l1Evs` (float EventData, or raw sensor records when raw_events): returns what the
        chunk produced.  mci_poses(window events) -> dict(dp=..., ba=..., se2=...) stands where resolveLastDPose / resolveLastPoseMap /
        resolveLastAtt2Params deliver the optimisers' results; reconstruct(p0, p1, matches12) -> (ok, inlier flags) is the two-view
        refinement of step(), skipped when None (step returns 1, :609-611)."""
        out = dict(state=None, dispatched=False)
        if self.mStat == self.IDLE:
            self.mStat = self.INIT
        if self.mStat == self.INIT:
            self.reset()
        if len(l1Evs) == 0:
            return out
        ts = l1Evs["ts"] if not self.raw else l1Evs["t"]
        evTspan = float(ts[-1]) - float(ts[0])                        # calcEventGenRate, src/Event/EventData.cpp:14-19
        with np.errstate(divide="ignore", invalid="ignore"):
            evGenRate = np.float64(len(l1Evs)) / (np.float64(evTspan) * self.W * self.H)
        rate = self.checkEvGenRate(evGenRate)
        out["state"] = self.mStat
        if rate != 0:
            if rate == -1:
                self.mStat = self.INIT
            elif rate == 1:
                self.mvSharedL2Evs.append(l1Evs)
            out["skipped"] = rate
            return out
        sendMCF = False
        if self.mStat == self.INIT:
            kps, img = self._slice_extract(l1Evs)                    # :1345 + :1348 (INIT branch of makeFrame :507-523)
            out.update(kps=kps, image=img)
            nKpts = len(kps)                                         # init() :568-592
            self.mvMatchesCnt = np.ones(nKpts, np.int32)
            self.mRefKPoints = kps
            self.mRefPoints = np.stack([kps["x"], kps["y"]], axis=1).astype(np.float32)
            self.mLastTrackedPts = self.mRefPoints.copy()
            if nKpts > self.DEF_TH_MIN_KPTS or (self.mbContTracking and self.mbFixedWinSize):
                self.updateState(l1Evs)
                self.mStat = self.TRACKING
            else:
                self.updateL1ChunkSize(self.mInitL1EvWinSize)
        elif self.mStat == self.TRACKING:
            nref = len(self.mRefPoints)
            pts, st, err, img = self._slice_track(l1Evs, self.mLastTrackedPts)      # :1348 (makeFrame :528-548) -> trackAndMatchCurrImage
            self.mLastTrackedPts = pts
            ok = (st == 1) & (pts[:, 0] >= 0) & (pts[:, 0] < np.float32(self.W)) & (pts[:, 1] >= 0) & (pts[:, 1] < np.float32(self.H))    # refineTrackedPts :104-151
            self.mvMatchesCnt = self.mvMatchesCnt + ok.astype(np.int32)
            vMatches12 = np.where(ok, np.arange(nref), -1).astype(np.int32)
            d = pts[ok] - self.mRefPoints[ok]
            vPxDisp = np.sort(np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)))
            nMatches = int(ok.sum())
            medPxDisp = float(vPxDisp[len(vPxDisp) // 2]) if len(vPxDisp) else 0.0      # (:540-541; the reference indexes an empty vector here)
            if reconstruct is not None and nMatches >= self.DEF_TH_MIN_MATCHES:          # step() :600-668
                res, inl = reconstruct(self.mRefKPoints, pts, vMatches12)
                drop = (vMatches12 >= 0) & ~np.asarray(inl, bool)
                vMatches12[drop] = -1; self.mvMatchesCnt[drop] -= 1; nMatches -= int(drop.sum())
            out.update(pts=pts, status=st, err=err, image=img, matches12=vMatches12, nMatches=nMatches, medPxDisp=medPxDisp)
            if nMatches < self.DEF_TH_MIN_MATCHES and not self.mbContTracking and self.mCurrIdx < 3:       # :1389-1402
                self.updateL1ChunkSize(self.mInitL1EvWinSize)
                self.mStat = self.INIT
                return out
            self.updateState(l1Evs)
            dispatchMCI = self.resolveEvWinSize(medPxDisp)
            if dispatchMCI or nMatches < self.DEF_TH_MIN_MATCHES:
                self.mStat = self.INIT
                sendMCF = True
        out["chunk_size"] = self.mL1EvWinSize
        if sendMCF:                                                  # :1433-1470
            vAccEvs = np.concatenate(self.mvSharedL2Evs)
            if self.raw:
                raise EorbError(_lib.EORB_E_ARG, "EvImBuilder: the reconstruction contest warps float EventData (time stamps); run the builder on float events")
            mc = self.generateMCImage(vAccEvs, mci_poses(vAccEvs) if mci_poses else None)
            good = len(mc["l2_kps"]) > self.DEF_TH_MIN_KPTS or self.mbContTracking
            out.update(dispatched=True, mci=mc, mci_good=good, window=len(vAccEvs))
            if self.mbContTracking:                                  # the overlap goes back to the front of the event queue (:1465-1469)
                nWinOverlap = self.mnWinOverlap if self.mbFixedWinSize else int(len(vAccEvs) * self.l1WinOverlap)
                out["overlap"] = vAccEvs[len(vAccEvs) - nWinOverlap:]
        return out

    def checkEvGenRate(self, eventRate):                             # :284-328
        if eventRate > self.minEvGenRate:
            self.mCntLowEvGenRate = 0
            return 0
        if self.mbContTracking:
            return -1 if (not self.mbFixedWinSize and self.mStat == self.INIT) else 0
        if self.mStat == self.INIT:
            return -1
        self.mCntLowEvGenRate += 1
        return -1 if self.mCntLowEvGenRate > 3 else 1                # DEF_L1_MAX_TRACK_LOST

    def updateState(self, l1Evs):                                    # :427-435
        self.mCurrIdx += 1
        self.mvSharedL2Evs.append(l1Evs)

    def resolveEvWinSize(self, medPxDisp):                           # :209-232
        if not self.mbFixedWinSize and np.float32(medPxDisp) > np.float32(self.mL1MaxPxDisp):
            new = int(np.floor(np.float32(np.float32(self.mCurrIdx + 1) / np.float32(medPxDisp)) * np.float32(self.mL1EvWinSize)))     # calcNewL1ChunkSize :197-201
            self.updateL1ChunkSize(new)
            return True
        return self.mbFixedWinSize and self.mCurrIdx >= self.mL1NumLoop


def feed_chunks(builder, events, mci_poses=None, max_chunks=10 ** 9):
    """The track manager's feed of the L1 builder (src/Event/EvTrackManager.cpp:272-286 consumeEventsBegin(l1ChunkSize), and
    injectEventsBegin for the overlap a dispatch hands back, src/Event/EvImBuilder.cpp:1465-1469): chunks of the builder's CURRENT
    chunk size off the front of the queue.  Returns the chunks' results."""
    queue = events
    res = []
    size = builder.mL1EvWinSize
    while len(queue) >= max(size, 1) and len(res) < max_chunks:
        chunk, queue = queue[:size], queue[size:]
        r = builder.Track(chunk, mci_poses)
        res.append(r)
        if r.get("overlap") is not None and len(r["overlap"]):
            queue = np.concatenate([r["overlap"], queue])
        size = builder.mL1EvWinSize
        if size < 1:
            break
    return res


class ORBVocabulary:
    """DBoW2 vocabulary resident on the device (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h); voc = dict(L, child_off, child_ids,
    node_desc, word_id, weight) as TemplatedVocabulary::loadFromTextFile leaves m_nodes (node 0 = root)."""

    def __init__(self, voc, weighting=0, norm=1, ctx=None):
        self.ctx = ctx or default_context()
        self.weighting, self.norm = weighting, norm
        co = np.ascontiguousarray(voc["child_off"], np.int32); ci = np.ascontiguousarray(voc["child_ids"], np.int32)
        nd = np.ascontiguousarray(voc["node_desc"], np.uint8); wi = np.ascontiguousarray(voc["word_id"], np.int32)
        ww = np.ascontiguousarray(voc["weight"], np.float64)
        c = self.ctx
        c.check(c.L.eorb_bow_set_vocabulary(c.h, len(co) - 1, int(voc["L"]), _p(co), _p(ci), _p(nd), _p(wi), _p(ww)))

    def transform(self, desc, levelsup=4, return_assignments=False):
        """ORBVocabulary::transform(vCurrentDesc, mBowVec, mFeatVec, levelsup) (Frame::ComputeBoW).
        Returns (bow_word, bow_val, (fv_node, fv_off, fv_idx)) [+ word_of, node_of]."""
        c = self.ctx
        desc = np.ascontiguousarray(desc, np.uint8); n = len(desc)
        bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); nw = C.c_int(0)
        fn = np.zeros(max(n, 1), np.uint32); fo = np.zeros(n + 1, np.int32); fi = np.zeros(max(n, 1), np.int32); nn = C.c_int(0)
        wo = np.zeros(max(n, 1), np.int32); no = np.zeros(max(n, 1), np.int32)
        c.check(c.L.eorb_bow_transform(c.h, _p(desc), n, desc.shape[1] if n else 32, int(levelsup), self.weighting, self.norm,
                                       _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fo), _p(fi), C.byref(nn), _p(wo), _p(no)))
        fv = (fn[:nn.value].copy(), fo[:nn.value + 1].copy(), fi[:fo[nn.value]].copy())
        if return_assignments:
            return bw[:nw.value].copy(), bv[:nw.value].copy(), fv, wo[:n], no[:n]
        return bw[:nw.value].copy(), bv[:nw.value].copy(), fv


def HammingWindowMatch(q_desc, t_desc, cand_offsets, cand_idx, ctx=None):
    """Best / second-best candidate per query in candidate order (the loop body of the windowed matchers, ORBmatcher.cc:754-774).
    Returns (best_idx, best_dist, second_idx, second_dist)."""
    c = ctx or default_context()
    q = np.ascontiguousarray(q_desc, np.uint8); t = np.ascontiguousarray(t_desc, np.uint8)
    co = np.ascontiguousarray(cand_offsets, np.int32); ci = np.ascontiguousarray(cand_idx, np.int32)
    nq = len(q)
    out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
    c.check(c.L.eorb_hamming_window_match(c.h, _p(q), nq, q.shape[1] if nq else 32, _p(t), len(t), t.shape[1] if len(t) else 32, _p(co), _p(ci),
                                          _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3])))
    return tuple(o[:nq] for o in out)


def ComputeDistinctiveDescriptors(desc, offsets, ctx=None):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:349-423) for a batch of map points (CSR offsets)."""
    c = ctx or default_context()
    desc = np.ascontiguousarray(desc, np.uint8); offsets = np.ascontiguousarray(offsets, np.int32)
    best = np.zeros(len(offsets) - 1, np.int32)
    c.check(c.L.eorb_distinctive_descriptors(c.h, _p(desc), _p(offsets), len(offsets) - 1, _p(best)))
    return best


def sortFeaturesResponse(kps, ctx=None):
    """MixedFrame::sortFeaturesResponse (src/MixedFrame.cpp:211-225): permutation (descending response, stable)."""
    c = ctx or default_context()
    kps = np.ascontiguousarray(kps, KP_DTYPE); perm = np.zeros(len(kps), np.int32)
    c.check(c.L.eorb_sort_by_response(c.h, _p(kps), len(kps), _p(perm)))
    return perm


def resolveNumMixedPts(nDetectedORB, nDetectedAK, nDesired, nDesiredAK):
    a, b = C.c_int(-1), C.c_int(-1)
    _lib.lib().eorb_resolve_num_mixed(nDetectedORB, nDetectedAK, nDesired, nDesiredAK, C.byref(a), C.byref(b))
    return a.value, b.value


class BFMatcher:
    """cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2) as used at src/Frame.cc:1228."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()

    def knnMatch2(self, query, train):
        q = np.ascontiguousarray(query, np.uint8); t = np.ascontiguousarray(train, np.uint8)
        assert q.shape[1] == 32 and t.shape[1] == 32
        idx = np.zeros((len(q), 2), np.int32); dist = np.zeros((len(q), 2), np.int32)
        c = self.ctx
        c.check(c.L.eorb_hamming_bf_knn2(c.h, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist)))
        return idx, dist


class FrontEndBatch:
    """HBM-resident batched pipeline: accumulate -> extract -> match against the previous slice."""

    def __init__(self, W=240, H=180, sigma=1.0, pol=False, nfeatures=1000, scaleFactor=1.2, nlevels=4, iniThFAST=10,
                 minThFAST=0, edgeTh=19, lap=(0, 1000), want_desc=True, max_batch=16, max_events=1000000, match=True,
                 windowSize=100, nnratio=0.9, checkOri=True, ctx=None):
        self.ctx = ctx or Context()
        cfg = _lib.FeConfig()
        cfg.W, cfg.H, cfg.sigma, cfg.pol = W, H, sigma, int(pol)
        cfg.orb = _lib.OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, edgeTh, W)
        cfg.lap0, cfg.lap1, cfg.want_desc = lap[0], lap[1], int(want_desc)
        cfg.max_batch, cfg.max_events, cfg.match = max_batch, max_events, int(match)
        cfg.windowSize, cfg.nnratio, cfg.checkOri = windowSize, nnratio, int(checkOri)
        self.cfg = cfg
        self.ctx.check(self.ctx.L.eorb_fe_configure(self.ctx.h, C.byref(cfg)))
        self.cap = self.ctx.L.eorb_orb_max_keypoints(self.ctx.h)
        self.W, self.H = W, H

    def run_dev(self, d_events, offsets, d_images=None, d_kps=None, d_desc=None, d_nkps=None, d_m12=None, d_nm=None, raw=False):
        """d_* are raw device pointers (ints), e.g. torch tensors' data_ptr(); offsets: int64[B+1] on the host.
        raw=True: d_events holds eorb_raw_event records (sensor pixels; set_undistort_maps first); raw=4: eorb_raw_event4 records
        (pack_raw_events4); raw=2: eorb_raw_event2 records (pack_raw_events2)."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        B = len(offsets) - 1
        vp = lambda x: C.c_void_p(x) if x else None
        fn = (self.ctx.L.eorb_fe_run_batch_raw4_dev if raw == 4 else (self.ctx.L.eorb_fe_run_batch_raw2_dev if raw == 2 else self.ctx.L.eorb_fe_run_batch_raw_dev)) if raw else self.ctx.L.eorb_fe_run_batch_dev
        self.ctx.check(fn(self.ctx.h, vp(d_events), _p(offsets), B, vp(d_images), vp(d_kps),
                          vp(d_desc), vp(d_nkps), vp(d_m12), vp(d_nm)))

    def last_f32(self, B, want_minmax=True):
        """(device pointer of the last batch's B x H x W float images, (min, max) per slice) -- eorb_fe_last_f32_dev"""
        p = C.c_void_p()
        mm = np.zeros((B, 2), np.float32)
        self.ctx.check(self.ctx.L.eorb_fe_last_f32_dev(self.ctx.h, C.byref(p), _p(mm) if want_minmax else None, int(B)))
        return p.value, mm

    def run_images_dev(self, d_images, B, d_kps=None, d_desc=None, d_nkps=None, d_m12=None, d_nm=None):
        """B camera frames (u8, W x H each, back to back in HBM): extraction + SearchForInitialization of frame b against frame b-1."""
        vp = lambda x: C.c_void_p(x) if x else None
        self.ctx.check(self.ctx.L.eorb_fe_run_batch_images_dev(self.ctx.h, vp(d_images), int(B), vp(d_kps), vp(d_desc), vp(d_nkps),
                                                               vp(d_m12), vp(d_nm)))
